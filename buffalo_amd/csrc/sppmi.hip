// SPPMI matrix of a stream on the device (the context input of CoFactor / CFR) -- kernels + C ABI.
//
// Reference semantics, three steps of buffalo's data creation:
//   * pair lines, /root/reference/buffalo/data/stream.py:257-267: for every user's sequence, each item and the `windows`
//     items after it are written as the two text lines "w c" and "c w"; sppmi_total_lines (D) counts the lines;
//   * _parallel_build_sppmi, /root/reference/buffalo/data/fileio.hpp:109-254, over the lines sorted by their first id:
//     appearances[id] = lines starting with id; for every distinct pair (probe, c), c <= probe, with cnt lines:
//     pmi = log(cnt) + log(D) - log(app[probe]) - log(app[c]) (double, left to right), sppmi = pmi - log(k); when
//     sppmi > 0 the lines "probe c sppmi" and "c probe sppmi" are written -- as text with six significant digits
//     (`fout << double`), which the next step parses with "%f" (fileio.hpp:84);
//   * the output is sorted by (row, col) and compressed into (indptr, key, val) like any matrix (stream.py:181-195).
//
// Device formulation -- no text, no files, one pass each:
//   1. pairs per user in closed form, exclusive scan -> where each user's lines start;
//   2. one thread per event writes its <= 2 * windows lines as keys  first * num_items + second  (32-bit while num_items^2 fits,
//      which halves the bytes of steps 2-3; 64-bit beyond 65,536 items);
//   3. radix sort of the keys over the bits they use (rocprim::radix_sort_keys), run-length encode (distinct pairs + cnt);
//   4. appearances by two binary searches per item in the sorted keys (no atomics on popular items);
//   5. one thread per distinct pair: the reference's double arithmetic, the six-digit decimal rounding of the text
//      round trip restated in exact double steps (powers of ten up to 1e22 are exact), emit count 0 / 1 / 2;
//   6. exclusive scan of the emit counts, scatter.  The distinct pairs are already in (row, col) order and the lines
//      are symmetric (cnt(a, b) == cnt(b, a)), so entry (a, b) is produced from its own run with probe = max(a, b) --
//      the output needs no second sort; indptr comes from the row changes.
// HBM-bound integer work: 8 (16) B per line and radix pass; the log / rounding step touches only the distinct pairs.
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.hpp"

namespace bfh {

// lines of a sequence of L events: sum_i min(windows, L - 1 - i), times two
__host__ __device__ __forceinline__ int64_t sppmi_pairs_of(int64_t L, int64_t w) {
    if (L <= 1) return 0;
    return L <= w + 1 ? L * (L - 1) / 2 : (L - w) * w + w * (w - 1) / 2;
}
// pairs written by the events before position i of a sequence of L events
__host__ __device__ __forceinline__ int64_t sppmi_pairs_before(int64_t i, int64_t L, int64_t w) {
    // event t writes min(w, L-1-t) pairs: w of them while t <= L-1-w
    const int64_t full = L - w > 0 ? (i < L - w ? i : L - w) : 0;   // events before i that write w pairs
    const int64_t rest = i - full;                                    // they write L-1-t = (L-1-full), (L-2-full), ...
    const int64_t first = L - 1 - full;
    return full * w + rest * first - rest * (rest - 1) / 2;
}

__global__ __launch_bounds__(256) void sppmi_count_kernel(const int64_t* __restrict__ indptr, int num_users, int windows, int64_t* __restrict__ pairs) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= num_users) return;
    const int64_t beg = u ? indptr[u - 1] : 0;
    pairs[u] = sppmi_pairs_of(indptr[u] - beg, windows);
}

// one thread per event: its pairs with the `windows` events after it, both orientations
// KeyT: uint32_t when num_items^2 fits 32 bits (half the bytes through the sort), else uint64_t
template <typename KeyT>
__global__ __launch_bounds__(256) void sppmi_lines_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ items, int num_users,
                                                          int64_t num_events, int windows, uint64_t num_items, const int64_t* __restrict__ pair_off,
                                                          KeyT* __restrict__ keys) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= num_events) return;
    const int u = static_cast<int>(lower_bound_dev<int64_t>(indptr, num_users, e + 1));   // first user whose END offset is > e
    const int64_t beg = u ? indptr[u - 1] : 0, L = indptr[u] - beg, i = e - beg;
    int64_t at = 2 * (pair_off[u] + sppmi_pairs_before(i, L, windows));
    const uint64_t w = static_cast<uint32_t>(items[e]);
    for (int64_t j = i + 1; j < i + windows + 1 && j < L; ++j) {
        const uint64_t c = static_cast<uint32_t>(items[beg + j]);
        keys[at++] = static_cast<KeyT>(w * num_items + c);
        keys[at++] = static_cast<KeyT>(c * num_items + w);
    }
}

template <typename KeyT>
__global__ __launch_bounds__(256) void sppmi_appear_kernel(const KeyT* __restrict__ sorted, int64_t n, int num_items, int64_t* __restrict__ app) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= num_items) return;
    const uint64_t ni = static_cast<uint64_t>(num_items);
    // (x + 1) * ni can be one past the key type's range for the last item: its run ends at n
    const int64_t end = x + 1 < num_items ? lower_bound_dev<KeyT>(sorted, n, static_cast<KeyT>((static_cast<uint64_t>(x) + 1) * ni)) : n;
    app[x] = end - lower_bound_dev<KeyT>(sorted, n, static_cast<KeyT>(static_cast<uint64_t>(x) * ni));
}

__host__ __device__ __forceinline__ double sppmi_pow10(int m) {   // exact for 0 <= m <= 22
    double p = 1.0;
    for (int i = 0; i < m; ++i) p *= 10.0;
    return p;
}
// float("%g" % x) for finite x > 0: round to six significant decimal digits, then to the nearest float
__host__ __device__ __forceinline__ float sppmi_text_round_trip(double x) {
    int e = 0;   // 10^e <= x < 10^(e+1)
    if (x >= 1.0) {
        while (e < 22 && x >= sppmi_pow10(e + 1)) ++e;
    } else {
        int m = 1;
        while (m < 22 && x * sppmi_pow10(m) < 1.0) ++m;
        e = -m;
    }
    int m5 = 5 - e;   // x * 10^m5 in [1e5, 1e6)
    double r = rint(m5 >= 0 ? x * sppmi_pow10(m5) : x / sppmi_pow10(-m5));
    if (r >= 1e6) { r = 1e5; m5 -= 1; }
    if (r < 1e5) { r *= 10.0; m5 += 1; }   // x sat a hair below the power of ten the search placed it above
    return static_cast<float>(m5 >= 0 ? r / sppmi_pow10(m5) : r * sppmi_pow10(-m5));
}

// one thread per distinct (a, b): value + number of output entries (0: sppmi <= 0, 2: a == b -- the reference writes that line twice)
template <typename KeyT>
__global__ __launch_bounds__(256) void sppmi_value_kernel(const KeyT* __restrict__ uniq, const unsigned int* __restrict__ cnt, int64_t runs,
                                                          uint64_t num_items, const int64_t* __restrict__ app, double log_d, double log_k,
                                                          float* __restrict__ val, int64_t* __restrict__ emit) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (r >= runs) return;
    const uint64_t key = static_cast<uint64_t>(uniq[r]);
    const uint64_t a = key / num_items, b = key % num_items;
    const uint64_t probe = a > b ? a : b, c = a > b ? b : a;   // fileio.hpp:207-208: the group of the larger id does the pair
    // fileio.hpp:182-250 writes a group when the next id's first line arrives and never flushes at end of file: the group of the
    // largest id that has lines (the first id of the last sorted key) is never a probe, so its pairs are not in the reference's output
    const uint64_t eof_group = static_cast<uint64_t>(uniq[runs - 1]) / num_items;
    const double pmi = log(static_cast<double>(cnt[r])) + log_d - log(static_cast<double>(app[probe])) - log(static_cast<double>(app[c]));
    const double sppmi = pmi - log_k;
    const bool keep = sppmi > 0 && probe != eof_group;
    val[r] = keep ? sppmi_text_round_trip(sppmi) : 0.f;
    emit[r] = keep ? (a == b ? 2 : 1) : 0;
}

template <typename KeyT>
__global__ __launch_bounds__(256) void sppmi_scatter_kernel(const KeyT* __restrict__ uniq, const float* __restrict__ val,
                                                            const int64_t* __restrict__ emit_off, int64_t runs, int64_t nnz, uint64_t num_items,
                                                            int32_t* __restrict__ out_row, int32_t* __restrict__ out_key, float* __restrict__ out_val) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (r >= runs) return;
    const int64_t at = emit_off[r], end = r + 1 < runs ? emit_off[r + 1] : nnz;
    const uint64_t key = static_cast<uint64_t>(uniq[r]);
    for (int64_t p = at; p < end; ++p) {
        out_row[p] = static_cast<int32_t>(key / num_items);
        out_key[p] = static_cast<int32_t>(key % num_items);
        out_val[p] = val[r];
    }
}

// rows ascending: indptr[m] = p + 1 for every row id m in [row(p), row(p+1))
__global__ __launch_bounds__(256) void sppmi_indptr_kernel(const int32_t* __restrict__ rows, int64_t nnz, int num_items, int64_t* __restrict__ indptr) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (p >= nnz) return;
    const int m0 = rows[p], m1 = p + 1 < nnz ? rows[p + 1] : num_items;
    for (int m = m0; m < m1; ++m) indptr[m] = p + 1;
    if (p == 0)
        for (int m = 0; m < m0; ++m) indptr[m] = 0;
}

class SppmiHandle : public HandleBase {
 public:
    ~SppmiHandle() override {
        if (stream) (void)hipStreamDestroy(stream);
    }
    void ensure() {
        BFH_HIP(hipSetDevice(device));
        if (!stream) BFH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }

    void build(const int64_t* indptr, const int32_t* items, int num_users, int num_items, int windows, int k, int64_t* nnz_out, int64_t* lines_out) {
        if (static_cast<uint64_t>(num_items) * static_cast<uint64_t>(num_items) <= (uint64_t(1) << 32))
            build_with<uint32_t>(indptr, items, num_users, num_items, windows, k, nnz_out, lines_out);
        else
            build_with<uint64_t>(indptr, items, num_users, num_items, windows, k, nnz_out, lines_out);
    }
    template <typename KeyT>
    void build_with(const int64_t* indptr, const int32_t* items, int num_users, int num_items, int windows, int k, int64_t* nnz_out, int64_t* lines_out) {
        BFH_REQUIRE(num_users > 0 && num_items > 0 && windows > 0 && k > 0, "sppmi: num_users, num_items, windows and k must be positive");
        ensure();
        const int64_t events = indptr[num_users - 1];
        int64_t prev = 0;
        for (int u = 0; u < num_users; ++u) {
            BFH_REQUIRE(indptr[u] >= prev, "sppmi: indptr must be non-decreasing END offsets");
            prev = indptr[u];
        }
        for (int64_t e = 0; e < events; ++e)
            if (items[e] < 0 || items[e] >= num_items) throw Error(BFH_ERR_INVALID, "sppmi: item id outside [0, num_items) at event " + std::to_string(e));
        num_items_ = num_items;
        nnz_ = 0;
        d_indptr_out_.resize(static_cast<size_t>(num_items));
        BFH_HIP(hipMemsetAsync(d_indptr_out_.get(), 0, d_indptr_out_.bytes(), stream));
        stats = bfh_stats{};
        int64_t total_pairs = 0;
        for (int u = 0; u < num_users; ++u) total_pairs += sppmi_pairs_of(indptr[u] - (u ? indptr[u - 1] : 0), windows);
        const int64_t lines = 2 * total_pairs;
        if (lines_out) *lines_out = lines;
        if (lines == 0) {
            BFH_HIP(hipStreamSynchronize(stream));
            if (nnz_out) *nnz_out = 0;
            return;
        }
        DevBuf<int64_t> d_indptr, d_pairs, d_off, d_app, d_emit, d_emit_off, d_runs;
        DevBuf<int32_t> d_items, d_rows;
        DevBuf<KeyT> d_keys, d_sorted, d_uniq;
        DevBuf<unsigned int> d_cnt;
        DevBuf<float> d_val;
        DevBuf<char> d_tmp;
        d_indptr.resize(num_users); d_pairs.resize(num_users); d_off.resize(num_users); d_items.resize(static_cast<size_t>(events));
        d_keys.resize(static_cast<size_t>(lines)); d_sorted.resize(static_cast<size_t>(lines)); d_app.resize(num_items);
        BFH_HIP(hipMemcpyAsync(d_indptr.get(), indptr, sizeof(int64_t) * num_users, hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(d_items.get(), items, sizeof(int32_t) * events, hipMemcpyHostToDevice, stream));
        const int slot = t_main_.begin(stream);
        hipLaunchKernelGGL(sppmi_count_kernel, dim3((num_users + 255) / 256), dim3(256), 0, stream, d_indptr.get(), num_users, windows, d_pairs.get());
        BFH_HIP(hipGetLastError());
        auto with_tmp = [&](auto&& call) {   // rocPRIM's size query + run
            size_t bytes = 0;
            BFH_HIP(call(nullptr, bytes));
            if (d_tmp.size() < bytes) d_tmp.resize(bytes ? bytes : 1);
            bytes = d_tmp.size();
            BFH_HIP(call(d_tmp.get(), bytes));
        };
        with_tmp([&](void* t, size_t& b) {
            return rocprim::exclusive_scan(t, b, d_pairs.get(), d_off.get(), int64_t(0), static_cast<size_t>(num_users), rocprim::plus<int64_t>(), stream);
        });
        hipLaunchKernelGGL((sppmi_lines_kernel<KeyT>), dim3(static_cast<unsigned>((events + 255) / 256)), dim3(256), 0, stream, d_indptr.get(), d_items.get(),
                           num_users, events, windows, static_cast<uint64_t>(num_items), d_off.get(), d_keys.get());
        BFH_HIP(hipGetLastError());
        int bits = 1;
        while (bits < static_cast<int>(sizeof(KeyT)) * 8 && (uint64_t(1) << bits) < static_cast<uint64_t>(num_items) * static_cast<uint64_t>(num_items)) ++bits;
        with_tmp([&](void* t, size_t& b) {
            return rocprim::radix_sort_keys(t, b, d_keys.get(), d_sorted.get(), static_cast<size_t>(lines), 0u, static_cast<unsigned>(bits), stream);
        });
        hipLaunchKernelGGL((sppmi_appear_kernel<KeyT>), dim3((num_items + 255) / 256), dim3(256), 0, stream, d_sorted.get(), lines, num_items, d_app.get());
        BFH_HIP(hipGetLastError());
        // distinct pairs: at most `lines`; the key buffer is free again and holds them
        d_uniq.resize(static_cast<size_t>(lines)); d_cnt.resize(static_cast<size_t>(lines)); d_runs.resize(1);
        with_tmp([&](void* t, size_t& b) {
            return rocprim::run_length_encode(t, b, d_sorted.get(), static_cast<size_t>(lines), d_uniq.get(), d_cnt.get(), d_runs.get(), stream);
        });
        int64_t runs = 0;
        BFH_HIP(hipMemcpyAsync(&runs, d_runs.get(), sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        d_val.resize(static_cast<size_t>(runs)); d_emit.resize(static_cast<size_t>(runs)); d_emit_off.resize(static_cast<size_t>(runs));
        const unsigned rblocks = static_cast<unsigned>((runs + 255) / 256);
        hipLaunchKernelGGL((sppmi_value_kernel<KeyT>), dim3(rblocks), dim3(256), 0, stream, d_uniq.get(), d_cnt.get(), runs, static_cast<uint64_t>(num_items),
                           d_app.get(), std::log(static_cast<double>(lines)), std::log(static_cast<double>(k)), d_val.get(), d_emit.get());
        BFH_HIP(hipGetLastError());
        with_tmp([&](void* t, size_t& b) {
            return rocprim::exclusive_scan(t, b, d_emit.get(), d_emit_off.get(), int64_t(0), static_cast<size_t>(runs), rocprim::plus<int64_t>(), stream);
        });
        int64_t last_off = 0, last_emit = 0;
        BFH_HIP(hipMemcpyAsync(&last_off, d_emit_off.get() + (runs - 1), sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipMemcpyAsync(&last_emit, d_emit.get() + (runs - 1), sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        nnz_ = last_off + last_emit;
        if (nnz_ > 0) {
            d_rows.resize(static_cast<size_t>(nnz_)); d_key_out_.resize(static_cast<size_t>(nnz_)); d_val_out_.resize(static_cast<size_t>(nnz_));
            hipLaunchKernelGGL((sppmi_scatter_kernel<KeyT>), dim3(rblocks), dim3(256), 0, stream, d_uniq.get(), d_val.get(), d_emit_off.get(), runs, nnz_,
                               static_cast<uint64_t>(num_items), d_rows.get(), d_key_out_.get(), d_val_out_.get());
            hipLaunchKernelGGL(sppmi_indptr_kernel, dim3(static_cast<unsigned>((nnz_ + 255) / 256)), dim3(256), 0, stream, d_rows.get(), nnz_, num_items,
                               d_indptr_out_.get());
            BFH_HIP(hipGetLastError());
        }
        t_main_.end(slot, stream);
        BFH_HIP(hipStreamSynchronize(stream));   // the locals above are freed on return
        stats.samples = lines;
        stats.launches = runs;                   // distinct pairs
        stats.kernel_ms = t_main_.drain();
        stats.h2d_bytes = 8.0 * num_users + 4.0 * events;
        if (nnz_out) *nnz_out = nnz_;
    }

    void fetch(int64_t* indptr_out, int32_t* keys_out, float* vals_out) {
        BFH_REQUIRE(num_items_ > 0, "sppmi: fetch before build");
        ensure();
        // through the library's pinned ring (HostStager: 4 MB chunks, DMA and the copy-out of the previous chunks overlapped) -- the caller's arrays are
        // pageable numpy memory, and a plain hipMemcpy into them was most of round 5's 63.6 ms of wall clock around 7.5 ms of device time
        int dev = 0;
        BFH_HIP(hipGetDevice(&dev));
        BFH_HIP(hipMemcpyAsync(indptr_out, d_indptr_out_.get(), sizeof(int64_t) * num_items_, hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        if (nnz_ > 0) {
            stager_.d2h(keys_out, d_key_out_.get(), sizeof(int32_t) * nnz_, stream, dev);
            stager_.d2h(vals_out, d_val_out_.get(), sizeof(float) * nnz_, stream, dev);
        }
        stats.d2h_bytes += 8.0 * num_items_ + 8.0 * nnz_;
    }

 private:
    int num_items_ = 0;
    int64_t nnz_ = 0;
    DevBuf<int64_t> d_indptr_out_;
    DevBuf<int32_t> d_key_out_;
    DevBuf<float> d_val_out_;
    EventTimer t_main_;
    HostStager stager_;
};

}  // namespace bfh

using bfh::guarded;
using bfh::SppmiHandle;

extern "C" {

void* bfh_sppmi_create(void) {
    try {
        SppmiHandle* h = new SppmiHandle();
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) {
            bfh::g_create_error = "no HIP device available (libbuffalo_hip has no CPU fallback)";
            delete h;
            return nullptr;
        }
        h->device = dev;
        return h;
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return nullptr;
    }
}
void bfh_sppmi_destroy(void* h) { delete static_cast<SppmiHandle*>(h); }
int bfh_sppmi_build(void* h, const int64_t* indptr, const int32_t* items, int num_users, int num_items, int windows, int k, int64_t* nnz,
                    int64_t* total_lines) {
    return guarded(h, [&] { static_cast<SppmiHandle*>(h)->build(indptr, items, num_users, num_items, windows, k, nnz, total_lines); return BFH_OK; });
}
int bfh_sppmi_fetch(void* h, int64_t* indptr_out, int32_t* keys_out, float* vals_out) {
    return guarded(h, [&] { static_cast<SppmiHandle*>(h)->fetch(indptr_out, keys_out, vals_out); return BFH_OK; });
}
int bfh_sppmi_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { *out = static_cast<SppmiHandle*>(h)->stats; return BFH_OK; });
}

}  // extern "C"
