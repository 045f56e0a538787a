// Options (JSON) reader + ABI odds and ends.
#include "common.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <thread>

namespace bfh {

thread_local std::string g_create_error;

bool host_range_mapped(const void* p, size_t bytes) {
    if (!p || !bytes) return p != nullptr;
    const uintptr_t page = static_cast<uintptr_t>(sysconf(_SC_PAGESIZE));
    const uintptr_t a = reinterpret_cast<uintptr_t>(p), first = a & ~(page - 1), last = (a + bytes - 1) & ~(page - 1);
    unsigned char vec = 0;
    for (uintptr_t q : {first, last})
        if (mincore(reinterpret_cast<void*>(q), page, &vec) != 0 && errno == ENOMEM) return false;
    return true;
}

HostStager::~HostStager() {
    for (auto& e : ev_)
        if (e) (void)hipEventDestroy(e);
    if (ring_) (void)hipHostFree(ring_);
}

void HostStager::d2h(void* dst, const void* src_dev, size_t bytes, hipStream_t s, int device) {
    if (!bytes) return;
    if (bytes < kChunk / 4) {   // small: the runtime's own staged copy of pageable memory
        BFH_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, s));
        BFH_HIP(hipStreamSynchronize(s));
        return;
    }
    if (!ring_) {
        BFH_HIP(hipHostMalloc(reinterpret_cast<void**>(&ring_), kSlots * kChunk, hipHostMallocDefault));
        for (auto& e : ev_) BFH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int64_t n_chunks = static_cast<int64_t>((bytes + kChunk - 1) / kChunk);
    const int workers = static_cast<int>(std::min<int64_t>({4, n_chunks, std::max(1u, std::thread::hardware_concurrency())}));
    std::atomic<int64_t> issued{0};            // chunks whose DMA and event are in the stream
    std::atomic<int64_t> drained[kSlots];      // per ring slot: 1 + the last chunk copied out of it
    for (auto& d : drained) d.store(0);
    std::atomic<int> failed{0};
    char* const out = static_cast<char*>(dst);
    auto work = [&](int t) {
        (void)hipSetDevice(device);
        for (int64_t k = t; k < n_chunks; k += workers) {
            while (issued.load(std::memory_order_acquire) <= k && !failed.load()) std::this_thread::yield();
            if (failed.load()) return;
            const int slot = static_cast<int>(k % kSlots);
            if (hipEventSynchronize(ev_[slot]) != hipSuccess) { failed.store(1); return; }
            const size_t off = static_cast<size_t>(k) * kChunk, n = std::min(kChunk, bytes - off);
            std::memcpy(out + off, ring_ + static_cast<size_t>(slot) * kChunk, n);
            drained[slot].store(k + 1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    th.reserve(static_cast<size_t>(workers));
    try {
        for (int t = 0; t < workers; ++t) th.emplace_back(work, t);
    } catch (...) {   // a thread could not be started: the ones that run are told to stop and JOINED (a joinable std::thread's destructor terminates)
        failed.store(1);
        for (auto& x : th) x.join();
        throw Error(BFH_ERR_HIP, "device -> host staging copy: could not start a drain thread");
    }
    hipError_t err = hipSuccess;
    for (int64_t k = 0; k < n_chunks && err == hipSuccess && !failed.load(); ++k) {
        const int slot = static_cast<int>(k % kSlots);
        while (k >= kSlots && drained[slot].load(std::memory_order_acquire) < k - kSlots + 1 && !failed.load()) std::this_thread::yield();
        const size_t off = static_cast<size_t>(k) * kChunk, n = std::min(kChunk, bytes - off);
        err = hipMemcpyAsync(ring_ + static_cast<size_t>(slot) * kChunk, static_cast<const char*>(src_dev) + off, n, hipMemcpyDeviceToHost, s);
        if (err == hipSuccess) err = hipEventRecord(ev_[slot], s);
        if (err != hipSuccess) failed.store(1);
        else issued.store(k + 1, std::memory_order_release);
    }
    for (auto& x : th) x.join();
    if (err != hipSuccess) BFH_HIP(err);
    BFH_REQUIRE(!failed.load(), "device -> host staging copy failed");
    BFH_HIP(hipStreamSynchronize(s));
}

uint64_t content_signature(const int32_t* keys, int64_t n) {
    const int64_t per_thread = int64_t(1) << 20;   // below ~4 MB a second thread costs more than it hashes
    unsigned hw = std::thread::hardware_concurrency();
    const int parts = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({8, hw ? hw : 1, n / per_thread})));
    uint64_t h[8] = {0};
    if (parts == 1) {
        h[0] = content_signature_range(keys, n);
    } else {
        const int64_t step = ((n / parts) + 7) / 8 * 8;
        std::vector<std::thread> th;
        for (int t = 0; t < parts; ++t) {
            const int64_t b = std::min<int64_t>(n, t * step), e = t + 1 == parts ? n : std::min<int64_t>(n, (t + 1) * step);
            th.emplace_back([&h, keys, b, e, t] { h[t] = content_signature_range(keys + b, e - b); });
        }
        for (auto& x : th) x.join();
    }
    uint64_t r = 0x9E3779B97F4A7C15ull ^ static_cast<uint64_t>(n);
    for (int t = 0; t < parts; ++t) r = (r ^ (h[t] + (r << 6) + (r >> 2))) * 0xff51afd7ed558ccdull;
    return r ^ (r >> 33);
}

namespace {
struct JsonReader {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit JsonReader(const std::string& str) : s(str) {}
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
    }
    bool fail(const char* m) {
        if (err.empty()) err = std::string(m) + " at offset " + std::to_string(i);
        return false;
    }
    bool string(std::string* out) {
        if (i >= s.size() || s[i] != '"') return fail("expected string");
        ++i;
        out->clear();
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c == '\\') {
                if (i >= s.size()) return fail("bad escape");
                char e = s[i++];
                switch (e) {
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'r': out->push_back('\r'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'u':
                        if (i + 4 > s.size()) return fail("bad \\u escape");
                        out->push_back('?');
                        i += 4;
                        break;
                    default: out->push_back(e); break;
                }
            } else {
                out->push_back(c);
            }
        }
        if (i >= s.size()) return fail("unterminated string");
        ++i;
        return true;
    }
    // kind: 0 number, 1 string, 2 bool, 3 other (null / object / array)
    bool value(int* kind, double* num, std::string* str, bool* boo) {
        ws();
        if (i >= s.size()) return fail("unexpected end");
        char c = s[i];
        if (c == '"') {
            *kind = 1;
            return string(str);
        }
        if (c == '{') {
            *kind = 3;
            ++i;
            ws();
            if (i < s.size() && s[i] == '}') { ++i; return true; }
            while (true) {
                ws();
                std::string k;
                if (!string(&k)) return false;
                ws();
                if (i >= s.size() || s[i] != ':') return fail("expected ':'");
                ++i;
                int kk; double nn; std::string ss; bool bb;
                if (!value(&kk, &nn, &ss, &bb)) return false;
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            *kind = 3;
            ++i;
            ws();
            if (i < s.size() && s[i] == ']') { ++i; return true; }
            while (true) {
                int kk; double nn; std::string ss; bool bb;
                if (!value(&kk, &nn, &ss, &bb)) return false;
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (s.compare(i, 4, "true") == 0) { *kind = 2; *boo = true; i += 4; return true; }
        if (s.compare(i, 5, "false") == 0) { *kind = 2; *boo = false; i += 5; return true; }
        if (s.compare(i, 4, "null") == 0) { *kind = 3; i += 4; return true; }
        // python's json.dumps may emit these for float('inf') / nan
        if (s.compare(i, 8, "Infinity") == 0) { *kind = 0; *num = INFINITY; i += 8; return true; }
        if (s.compare(i, 9, "-Infinity") == 0) { *kind = 0; *num = -INFINITY; i += 9; return true; }
        if (s.compare(i, 3, "NaN") == 0) { *kind = 0; *num = NAN; i += 3; return true; }
        size_t j = i;
        while (j < s.size() && (isdigit(static_cast<unsigned char>(s[j])) || s[j] == '-' || s[j] == '+' || s[j] == '.' ||
                                s[j] == 'e' || s[j] == 'E'))
            ++j;
        if (j == i) return fail("unexpected character");
        try {
            *num = std::stod(s.substr(i, j - i));
        } catch (...) {
            return fail("bad number");
        }
        *kind = 0;
        i = j;
        return true;
    }
};
}  // namespace

bool Options::load(const std::string& path, std::string* err) {
    std::ifstream in(path.c_str());
    if (!in.is_open()) {
        *err = "File not exists: " + path;
        return false;
    }
    std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    JsonReader r(text);
    r.ws();
    if (r.i >= text.size() || text[r.i] != '{') {
        *err = "Failed to parse: top-level value is not an object";
        return false;
    }
    ++r.i;
    r.ws();
    num_.clear(); str_.clear(); boo_.clear();
    if (r.i < text.size() && text[r.i] == '}') return true;
    while (true) {
        r.ws();
        std::string k;
        if (!r.string(&k)) break;
        r.ws();
        if (r.i >= text.size() || text[r.i] != ':') { r.fail("expected ':'"); break; }
        ++r.i;
        int kind = 3; double num = 0; std::string str; bool boo = false;
        if (!r.value(&kind, &num, &str, &boo)) break;
        if (kind == 0) num_[k] = num;
        else if (kind == 1) str_[k] = str;
        else if (kind == 2) boo_[k] = boo;
        r.ws();
        if (r.i < text.size() && text[r.i] == ',') { ++r.i; continue; }
        if (r.i < text.size() && text[r.i] == '}') {
            ++r.i;
            r.ws();
            if (r.i != text.size()) r.fail("trailing characters");
            break;
        }
        r.fail("expected ',' or '}'");
        break;
    }
    if (!r.err.empty()) {
        *err = "Failed to parse: " + r.err;
        return false;
    }
    return true;
}

double Options::num(const std::string& k) const {
    auto it = num_.find(k);
    if (it == num_.end()) throw Error(BFH_ERR_INVALID, "option '" + k + "' missing or not a number");
    return it->second;
}
double Options::num_or(const std::string& k, double dflt) const {
    auto it = num_.find(k);
    return it == num_.end() ? dflt : it->second;
}
bool Options::boolean(const std::string& k) const {
    auto it = boo_.find(k);
    if (it == boo_.end()) throw Error(BFH_ERR_INVALID, "option '" + k + "' missing or not a bool");
    return it->second;
}
bool Options::boolean_or(const std::string& k, bool dflt) const {
    auto it = boo_.find(k);
    return it == boo_.end() ? dflt : it->second;
}
std::string Options::str(const std::string& k) const {
    auto it = str_.find(k);
    if (it == str_.end()) throw Error(BFH_ERR_INVALID, "option '" + k + "' missing or not a string");
    return it->second;
}

}  // namespace bfh

extern "C" {

const char* bfh_version(void) { return "buffalo_hip 0.2.0 (gfx950)"; }
size_t bfh_stats_size(void) { return sizeof(bfh_stats); }

const char* bfh_last_error(const void* handle) {
    if (!handle) return bfh::g_create_error.c_str();
    return static_cast<const bfh::HandleBase*>(handle)->last_error.c_str();
}

int bfh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
