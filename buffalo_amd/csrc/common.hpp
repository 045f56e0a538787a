// Shared host+device plumbing of libbuffalo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/buffalo_hip.h"

namespace bfh {

// ------------------------------------------------------------------------------------------------
// Errors.  The reference throws std::runtime_error from CHECK_CUDA
// (/root/reference/include/buffalo/cuda/utils.cuh:24-31); here exceptions stop at the C ABI.
// ------------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define BFH_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            std::ostringstream os__;                                                               \
            os__ << "HIP error " << hipGetErrorString(e__) << " at " << __FILE__ << ":" << __LINE__ \
                 << " (" #expr ")";                                                                \
            throw ::bfh::Error(BFH_ERR_HIP, os__.str());                                           \
        }                                                                                          \
    } while (0)

#define BFH_REQUIRE(cond, msg)                                                   \
    do {                                                                         \
        if (!(cond)) throw ::bfh::Error(BFH_ERR_INVALID, std::string(msg));      \
    } while (0)

extern thread_local std::string g_create_error;

struct HandleBase {
    std::string last_error;
    int device = 0;
    hipStream_t stream = nullptr;
    bfh_stats stats{};
    bool timing = true;
    virtual ~HandleBase() {}
};

// Runs `fn` and maps exceptions onto the status codes of buffalo_hip.h.
template <typename F>
static inline int guarded(void* h, F&& fn) {
    HandleBase* hb = static_cast<HandleBase*>(h);
    if (!hb) {
        g_create_error = "null handle";
        return BFH_ERR_INVALID;
    }
    try {
        BFH_HIP(hipSetDevice(hb->device));
        return fn();
    } catch (const Error& e) {
        hb->last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        hb->last_error = "out of host memory";
        return BFH_ERR_NOMEM;
    } catch (const std::exception& e) {
        hb->last_error = e.what();
        return BFH_ERR_INVALID;
    }
}

// ------------------------------------------------------------------------------------------------
// Options: the reference hands its backends a JSON file path (buffalo/misc/_aux.py:82-89) that is
// parsed with json11 (lib/algo.cc:19-37, lib/cuda/bpr/bpr.cu:228-243).  Minimal JSON reader; only
// top-level scalars are kept.  Unlike json11, a key the backend needs but the file lacks is an
// error (SURVEY section 5 "Config"), except the documented quirks handled by the callers.
// ------------------------------------------------------------------------------------------------
class Options {
 public:
    bool load(const std::string& path, std::string* err);
    bool has(const std::string& k) const { return num_.count(k) || str_.count(k) || boo_.count(k); }
    double num(const std::string& k) const;
    double num_or(const std::string& k, double dflt) const;
    int integer(const std::string& k) const { return static_cast<int>(num(k)); }
    bool boolean(const std::string& k) const;
    bool boolean_or(const std::string& k, bool dflt) const;
    std::string str(const std::string& k) const;

 private:
    std::map<std::string, double> num_;
    std::map<std::string, std::string> str_;
    std::map<std::string, bool> boo_;
};

// ------------------------------------------------------------------------------------------------
// Device memory
// ------------------------------------------------------------------------------------------------
template <typename T>
class DevBuf {
 public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p_) (void)hipFree(p_);
        p_ = nullptr;
        n_ = 0;
    }
    // grow-only unless exact is requested
    void resize(size_t n, bool zero = false, hipStream_t s = nullptr) {
        if (n != n_) {
            release();
            if (n) {
                hipError_t e = hipMalloc(&p_, n * sizeof(T));
                if (e != hipSuccess) {
                    p_ = nullptr;
                    throw Error(BFH_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
                }
            }
            n_ = n;
        }
        if (zero && n_) BFH_HIP(hipMemsetAsync(p_, 0, n_ * sizeof(T), s));
    }
    T* get() const { return p_; }
    size_t size() const { return n_; }
    size_t bytes() const { return n_ * sizeof(T); }

 private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

// HIP-event stopwatch on the handle's stream; resolved lazily (events are only read after the
// stream was synchronised by the caller).
class EventTimer {
 public:
    ~EventTimer() {
        for (auto& p : pool_) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    }
    // returns slot index
    int begin(hipStream_t s) {
        if (used_ == pool_.size()) {
            hipEvent_t a, b;
            BFH_HIP(hipEventCreate(&a));
            BFH_HIP(hipEventCreate(&b));
            pool_.emplace_back(a, b);
        }
        BFH_HIP(hipEventRecord(pool_[used_].first, s));
        return static_cast<int>(used_++);
    }
    void end(int slot, hipStream_t s) { BFH_HIP(hipEventRecord(pool_[slot].second, s)); }
    // call after the stream is idle; returns summed ms and resets
    double drain() {
        double ms = 0.0;
        for (size_t i = 0; i < used_; ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, pool_[i].first, pool_[i].second) == hipSuccess) ms += t;
        }
        used_ = 0;
        return ms;
    }

 private:
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool_;
    size_t used_ = 0;
};

// Device -> caller memory WITHOUT page-locking the caller's arrays (round 5).  Until round 4 the factor arrays were hipHostRegister'ed for
// the life of a model; a registration of memory the library does not own is only as sound as the allocator behind it -- glibc raises its mmap
// threshold after the first large free, so later 1 .. 32 MB numpy arrays sit in the brk heap beside unrelated objects, a registered range then
// covers pages other allocations come and go in, and the runtime's bookkeeping of such a range (re-validation after the process unmaps, trims or
// migrates part of it) is outside the library's control.  Now the DMA lands in a ring of pinned buffers the LIBRARY owns (hipHostMalloc) and worker
// threads copy every chunk on to its destination while the next chunks travel; "pin_host" = 1 brings the registration back for callers who
// guarantee page-exclusive, long-lived arrays.
class HostStager {
 public:
    HostStager() = default;
    HostStager(const HostStager&) = delete;
    HostStager& operator=(const HostStager&) = delete;
    ~HostStager();
    // returns when dst[0, bytes) holds the device bytes; the stream is idle afterwards
    void d2h(void* dst, const void* src_dev, size_t bytes, hipStream_t s, int device);

 private:
    static constexpr int kSlots = 8;
    static constexpr size_t kChunk = size_t(4) << 20;
    char* ring_ = nullptr;
    hipEvent_t ev_[kSlots] = {};
};

// true when the first and the last page of [p, p + bytes) are mapped in this process (mincore): a caller array that was unmapped behind the
// library's back (freed while the handle still owes it a copy) is reported as an error instead of a fault
bool host_range_mapped(const void* p, size_t bytes);

static inline int vdim_of(int d) { return ((d + 31) / 32) * 32; }  // bpr.cu:266-267, als.cu:251-252

// Auto-residency of chunks the caller hands over on every call (the reference's call pattern, cuda/_bpr.pyx:60-74, _als.pyx:52-67):
// a 64-bit hash over EVERY word of the host buffer decides whether the copy in HBM is still the caller's data -- the
// reference always uses the buffer it is handed, so a change anywhere (re-weighted confidences, another split of the same
// shape, an in-place edit) must be seen.  Four independent multiply-xor lanes per thread, large buffers cut over up to 8
// threads: ~1 ms for ML-20M's 80 MB of keys, against the ~5 ms H2D copy of pageable memory it saves.
static inline uint64_t content_signature_range(const int32_t* keys, int64_t n) {
    const uint64_t k0 = 0x9E3779B97F4A7C15ull, k1 = 0xff51afd7ed558ccdull;
    uint64_t h[4] = {k0, k0 ^ 0x1111, k0 ^ 0x2222, k0 ^ 0x3333};
    const int64_t quads = n / 8;   // 8 int32 = 4 uint64
    const char* p = reinterpret_cast<const char*>(keys);
    for (int64_t i = 0; i < quads; ++i) {
        uint64_t w[4];
        std::memcpy(w, p + i * 32, 32);
        for (int l = 0; l < 4; ++l) {
            h[l] = (h[l] ^ w[l]) * k1;
            h[l] ^= h[l] >> 29;
        }
    }
    uint64_t r = h[0];
    for (int l = 1; l < 4; ++l) r = (r ^ (h[l] + k0 + (r << 6) + (r >> 2))) * k1;
    for (int64_t i = quads * 8; i < n; ++i) r = (r ^ static_cast<uint32_t>(keys[i])) * k1 + 1;
    return r ^ (r >> 33);
}
uint64_t content_signature(const int32_t* keys, int64_t n);   // common.hip (threads)

// Stable LSD radix sort of (uint32 key, int32 value) pairs over the low `bits` key bits, on `s`
// (rocprim::radix_sort_pairs; implemented in ingest.hip so only that file pays for the headers).
void device_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n, int bits,
                           DevBuf<char>& tmp, hipStream_t s);
// (ingest.hip) the same over 64-bit keys with 64-bit payloads (eALS: position of every entry in the other orientation)
void device_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out, int64_t n, int bits,
                           DevBuf<char>& tmp, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Device helpers (wave64)
// ------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)

// Sum over the 64 lanes of a wavefront; every lane receives the total.
// 4 DPP row rotations reduce each 16-lane row, then 3 readlanes combine the 4 rows: no LDS, no
// barrier (the reference needs two __syncthreads per dot: include/buffalo/cuda/utils.cuh:80-114).
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ int lane_id() { return static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); }

// Philox4x32-10 (Salmon et al., SC'11).  Counter layout shared with the oracle's counter sampler
// (oracle/buffalo_oracle.cc counter_draw): ctr = (pos_lo, pos_hi, attempt, epoch<<8 | slot),
// key = (seed, 0x5bf03635 ^ stream).
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1, uint32_t& o0, uint32_t& o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
        const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0;
    o1 = c1;
}

__host__ __device__ __forceinline__ void counter_draw(uint32_t seed, uint32_t stream, uint64_t pos_idx, uint32_t slot,
                                                      uint32_t epoch, uint32_t attempt, uint32_t& o0, uint32_t& o1) {
    philox4x32_10(static_cast<uint32_t>(pos_idx), static_cast<uint32_t>(pos_idx >> 32), attempt,
                  (epoch << 8) | (slot & 0xffu), seed, 0x5bf03635u ^ stream, o0, o1);
}

// first index in [0,n) with a[idx] >= v  (std::lower_bound)
template <typename T>
__device__ __forceinline__ int64_t lower_bound_dev(const T* __restrict__ a, int64_t n, T v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// membership test in a sorted int32 run [beg,end)
__device__ __forceinline__ bool sorted_contains(const int32_t* __restrict__ keys, int64_t beg, int64_t end, int32_t v) {
    int64_t lo = beg, hi = end;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t k = keys[mid];
        if (k < v) lo = mid + 1;
        else hi = mid;
    }
    return lo < end && keys[lo] == v;
}

// hardware fp32 atomic add without return (global_atomic_add_f32)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

#endif  // __HIPCC__

}  // namespace bfh
