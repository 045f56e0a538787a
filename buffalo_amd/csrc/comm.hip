// bfh_comm_*: RCCL communicator objects for the data-parallel paths (users sharded, item factors replicated;
// ALS rows sharded) -- SURVEY.md section 8(e).  The reference has no multi-device code.
#include "comm.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>

namespace bfh {

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) throw Error(BFH_ERR_HIP, std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?"));
        auto sym = [&](const char* name) {
            void* p = dlsym(x.lib, name);
            if (!p) throw Error(BFH_ERR_HIP, std::string("librccl: missing symbol ") + name);
            return p;
        };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        return x;
    }();
    return r;
}

void check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Error(BFH_ERR_HIP, std::string("RCCL ") + what + ": " + rccl().GetErrorString(r));
}

// ---- the shared-memory test transport (comm.hpp) ---------------------------------------------------------------
constexpr char kShmMagic[8] = {'B', 'F', 'H', 'S', 'H', 'M', '1', 0};
constexpr size_t kShmSlot = size_t(4) << 20;      // bytes a rank stages per round
constexpr size_t kShmHeader = 4096;
constexpr int kShmMaxRanks = 16;
constexpr double kShmTimeoutS = 120.0;

struct ShmHeader {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> attached;
};

bool shm_requested() {
    const char* t = std::getenv("BFH_COMM_TRANSPORT");
    return t && std::string(t) == "shm";
}

}  // namespace

struct Comm::Shm {
    std::string name;
    int fd = -1;
    size_t bytes = 0;
    char* base = nullptr;
    ShmHeader* hdr = nullptr;
    std::vector<char> host;
    char* slot(int r) { return base + kShmHeader + static_cast<size_t>(r) * kShmSlot; }
    // central-counter barrier; the last rank to arrive opens the next generation
    void barrier(int n) {
        const uint32_t gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(n)) {
            hdr->arrived.store(0, std::memory_order_relaxed);
            hdr->generation.fetch_add(1, std::memory_order_release);
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (hdr->generation.load(std::memory_order_acquire) == gen) {
            if (++spins < 2000) continue;
            std::this_thread::yield();
            if ((spins & 0xfff) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutS)
                throw Error(BFH_ERR_HIP, "shm transport: a rank did not reach the collective within " + std::to_string(int(kShmTimeoutS)) +
                                             " s (ranks must issue the same collectives in the same order)");
        }
    }
    ~Shm() {
        bool last = false;
        if (hdr) last = hdr->attached.fetch_sub(1, std::memory_order_acq_rel) == 1;
        if (base) munmap(base, bytes);
        if (fd >= 0) close(fd);
        if (last) shm_unlink(name.c_str());
    }
};

void Comm::unique_id(char* out128) {
    if (shm_requested()) {   // the id names the segment: magic + 16 random bytes
        std::memset(out128, 0, 128);
        std::memcpy(out128, kShmMagic, sizeof(kShmMagic));
        int fd = open("/dev/urandom", O_RDONLY);
        if (fd < 0 || read(fd, out128 + 8, 16) != 16) {
            if (fd >= 0) close(fd);
            throw Error(BFH_ERR_HIP, "shm transport: /dev/urandom is not readable");
        }
        close(fd);
        return;
    }
    ncclUniqueId id;
    check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof(id));
}

Comm::Comm(int n_ranks, int rank, const char* id128, int dev) {
    BFH_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks && id128, "comm_create: bad rank / size / id");
    device = dev;
    rank_ = rank;
    size_ = n_ranks;
    BFH_HIP(hipSetDevice(dev));
    BFH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (std::memcmp(id128, kShmMagic, sizeof(kShmMagic)) == 0) {
        BFH_REQUIRE(n_ranks <= kShmMaxRanks, "shm transport: at most 16 ranks");
        struct Guard {   // a throw below must not leak the mapping, the segment's name or the stream (the destructor of a half-built Comm never runs)
            Shm*& p; hipStream_t& st; bool armed = true, created = false;
            ~Guard() {
                if (!armed) return;
                if (p && created && !p->name.empty()) shm_unlink(p->name.c_str());   // never counted in `attached`: ~Shm would not unlink it
                delete p;
                p = nullptr;
                if (st) { (void)hipStreamDestroy(st); st = nullptr; }
            }
        } guard{shm_, stream};
        shm_ = new Shm();
        static const char* hex = "0123456789abcdef";
        shm_->name = "/bfh_";
        for (int i = 0; i < 16; ++i) {
            const unsigned char b = static_cast<unsigned char>(id128[8 + i]);
            shm_->name += hex[b >> 4];
            shm_->name += hex[b & 15];
        }
        shm_->bytes = kShmHeader + static_cast<size_t>(n_ranks) * kShmSlot;
        shm_->fd = shm_open(shm_->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (shm_->fd >= 0) guard.created = true;
        else shm_->fd = shm_open(shm_->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (shm_->fd < 0 || ftruncate(shm_->fd, static_cast<off_t>(shm_->bytes)) != 0)   // a fresh segment reads as zeros: counters start at 0
            throw Error(BFH_ERR_HIP, "shm transport: cannot create " + shm_->name);
        void* m = mmap(nullptr, shm_->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, shm_->fd, 0);
        if (m == MAP_FAILED) throw Error(BFH_ERR_HIP, "shm transport: mmap failed");
        shm_->base = static_cast<char*>(m);
        shm_->hdr = reinterpret_cast<ShmHeader*>(m);
        shm_->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
        guard.created = false;   // counted: from here ~Shm unlinks when the last rank leaves
        shm_->host.resize(kShmSlot);
        shm_->barrier(n_ranks);   // every rank is attached before the first collective
        guard.armed = false;
        return;
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    check(rccl().CommInitRank(&c, n_ranks, id, rank), "ncclCommInitRank");
    comm_ = c;
    int count = 0;
    check(rccl().CommCount(c, &count), "ncclCommCount");
    BFH_REQUIRE(count == n_ranks, "comm_create: RCCL reports " + std::to_string(count) + " ranks, expected " + std::to_string(n_ranks));
}

Comm::~Comm() {
    delete shm_;
    if (comm_) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
    if (stream) (void)hipStreamDestroy(stream);
}

// rounds of at most one slot per rank: D2H into the rank's slot | barrier | sum the slots | barrier | H2D.
// The sum runs in rank order, or -- BFH_COMM_SHM_ORDER=ring -- in the order a ring all-reduce produces: the round's elements are cut
// into N segments, and segment c is accumulated starting at rank c + 1 and ending at rank c (reduce-scatter), then handed to everybody
// (all-gather).  Different segments see different summation orders; every rank still receives the SAME bits, which is the property
// the exchange protocol relies on (tests/test_comm_ranks_gpu.py holds the replicas to bit-identity under both orders).
template <typename T>
void Comm::shm_all_reduce(const T* send, T* recv, size_t count, hipStream_t s) {
    BFH_HIP(hipStreamSynchronize(s));
    const size_t per = kShmSlot / sizeof(T);
    T* out = reinterpret_cast<T*>(shm_->host.data());
    static const bool ring = [] { const char* o = std::getenv("BFH_COMM_SHM_ORDER"); return o && std::string(o) == "ring"; }();
    for (size_t off = 0; off < count; off += per) {
        const size_t n = std::min(per, count - off);
        BFH_HIP(hipMemcpy(shm_->slot(rank_), send + off, n * sizeof(T), hipMemcpyDeviceToHost));
        shm_->barrier(size_);
        if (ring) {
            const size_t seg = (n + size_ - 1) / size_;
            for (int c = 0; c < size_; ++c) {
                const size_t lo = std::min(n, c * seg), hi = std::min(n, (c + 1) * seg);
                const T* a = reinterpret_cast<const T*>(shm_->slot((c + 1) % size_));
                for (size_t i = lo; i < hi; ++i) out[i] = a[i];
                for (int k = 2; k <= size_; ++k) {
                    const T* b = reinterpret_cast<const T*>(shm_->slot((c + k) % size_));
                    for (size_t i = lo; i < hi; ++i) out[i] += b[i];
                }
            }
        } else {
            const T* a = reinterpret_cast<const T*>(shm_->slot(0));
            for (size_t i = 0; i < n; ++i) out[i] = a[i];
            for (int r = 1; r < size_; ++r) {
                const T* b = reinterpret_cast<const T*>(shm_->slot(r));
                for (size_t i = 0; i < n; ++i) out[i] += b[i];
            }
        }
        shm_->barrier(size_);
        BFH_HIP(hipMemcpy(recv + off, out, n * sizeof(T), hipMemcpyHostToDevice));
    }
}

void Comm::all_reduce_f32(const float* send, float* recv, size_t count, hipStream_t s) {
    if (shm_) { shm_all_reduce(send, recv, count, s); return; }
    if (count) check(rccl().AllReduce(send, recv, count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm_), s), "ncclAllReduce(f32)");
}
void Comm::all_reduce_i32(const int* send, int* recv, size_t count, hipStream_t s) {
    if (shm_) { shm_all_reduce(send, recv, count, s); return; }
    if (count) check(rccl().AllReduce(send, recv, count, ncclInt32, ncclSum, static_cast<ncclComm_t>(comm_), s), "ncclAllReduce(i32)");
}
void Comm::all_reduce_f64(const double* send, double* recv, size_t count, hipStream_t s) {
    if (shm_) { shm_all_reduce(send, recv, count, s); return; }
    if (count) check(rccl().AllReduce(send, recv, count, ncclFloat64, ncclSum, static_cast<ncclComm_t>(comm_), s), "ncclAllReduce(f64)");
}
void Comm::broadcast_bytes(void* buf, size_t bytes, int root, hipStream_t s) {
    if (shm_) {
        BFH_REQUIRE(root >= 0 && root < size_, "broadcast: bad root");
        BFH_HIP(hipStreamSynchronize(s));
        char* p = static_cast<char*>(buf);
        for (size_t off = 0; off < bytes; off += kShmSlot) {
            const size_t n = std::min(kShmSlot, bytes - off);
            if (rank_ == root) BFH_HIP(hipMemcpy(shm_->slot(0), p + off, n, hipMemcpyDeviceToHost));
            shm_->barrier(size_);
            if (rank_ != root) BFH_HIP(hipMemcpy(p + off, shm_->slot(0), n, hipMemcpyHostToDevice));
            shm_->barrier(size_);
        }
        return;
    }
    if (bytes) check(rccl().Broadcast(buf, buf, bytes, ncclInt8, root, static_cast<ncclComm_t>(comm_), s), "ncclBroadcast");
}
void Comm::group_start() { if (shm_) return; check(rccl().GroupStart(), "ncclGroupStart"); }
void Comm::group_end() { if (shm_) return; check(rccl().GroupEnd(), "ncclGroupEnd"); }

}  // namespace bfh

using bfh::Comm;
using bfh::guarded;

extern "C" {

int bfh_comm_unique_id(char* out, size_t bytes) {
    try {
        if (!out || bytes < 128) throw bfh::Error(BFH_ERR_INVALID, "comm_unique_id: need a 128-byte buffer");
        Comm::unique_id(out);
        return BFH_OK;
    } catch (const bfh::Error& e) {
        bfh::g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return BFH_ERR_HIP;
    }
}

void* bfh_comm_create(int n_ranks, int rank, const char* unique_id, int device) {
    try {
        return new Comm(n_ranks, rank, unique_id, device);
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return nullptr;
    }
}

void bfh_comm_destroy(void* c) { delete static_cast<Comm*>(c); }
int bfh_comm_rank(void* c) { return c ? static_cast<Comm*>(c)->rank() : BFH_ERR_INVALID; }
int bfh_comm_size(void* c) { return c ? static_cast<Comm*>(c)->size() : BFH_ERR_INVALID; }

// every rank contributes rank + 1 per element; all must read n (n + 1) / 2 back
int bfh_comm_self_test(void* c) {
    return guarded(c, [&] {
        Comm* cm = static_cast<Comm*>(c);
        const int n = 1024;
        bfh::DevBuf<float> buf;
        buf.resize(n);
        std::vector<float> host(n, static_cast<float>(cm->rank() + 1));
        BFH_HIP(hipMemcpyAsync(buf.get(), host.data(), n * sizeof(float), hipMemcpyHostToDevice, cm->comm_stream()));
        cm->all_reduce_f32(buf.get(), buf.get(), n, cm->comm_stream());
        BFH_HIP(hipMemcpyAsync(host.data(), buf.get(), n * sizeof(float), hipMemcpyDeviceToHost, cm->comm_stream()));
        BFH_HIP(hipStreamSynchronize(cm->comm_stream()));
        const float want = 0.5f * cm->size() * (cm->size() + 1);
        for (float v : host)
            if (v != want) throw bfh::Error(BFH_ERR_HIP, "comm self test: all-reduce returned " + std::to_string(v) + ", expected " + std::to_string(want));
        return BFH_OK;
    });
}

// sum all-reduce of a host double array (loss sums, counters) through a staging buffer
int bfh_comm_all_reduce_f64(void* c, double* values, int n) {
    return guarded(c, [&] {
        Comm* cm = static_cast<Comm*>(c);
        BFH_REQUIRE(values && n >= 0, "comm_all_reduce_f64: bad arguments");
        if (n == 0) return BFH_OK;
        bfh::DevBuf<double> buf;
        buf.resize(static_cast<size_t>(n));
        BFH_HIP(hipMemcpyAsync(buf.get(), values, n * sizeof(double), hipMemcpyHostToDevice, cm->comm_stream()));
        cm->all_reduce_f64(buf.get(), buf.get(), static_cast<size_t>(n), cm->comm_stream());
        BFH_HIP(hipMemcpyAsync(values, buf.get(), n * sizeof(double), hipMemcpyDeviceToHost, cm->comm_stream()));
        BFH_HIP(hipStreamSynchronize(cm->comm_stream()));
        return BFH_OK;
    });
}

}  // extern "C"
