// COO -> compressed rows on the device -- kernels + C ABI.
//
// Reference semantics: _sort_and_compressed_binarization (/root/reference/buffalo/data/fileio.hpp:263-420):
// the (row, col, val) records are STABLE-sorted by (major, minor) (`__gnu_parallel::stable_sort`, :328-339;
// duplicates keep their input order and are all kept), `indptr[k]` = number of records whose major id is
// <= k (END offsets, no leading zero, :359-379) and the minor ids / values are written out in sorted order
// (:389-410).  The reference runs it once per orientation (sort_key 1: rowwise, 2: colwise) on 1-based ids
// parsed from text; here the ids are 0-based arrays already in memory.
//
// Device formulation: one 64-bit key (major << 32 | minor) per record, a stable LSD radix sort over only the
// bits the two id ranges need (rocprim::radix_sort_pairs -- the one library primitive on this path, like
// rocBLAS would be for a plain GEMM), then two trivially parallel passes: split the sorted keys
// back into minor ids and fill indptr from the positions where the major id changes.  All HBM-bound integer
// work: 2 x (8 + 4) B per record and radix pass.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

namespace bfh {

__global__ __launch_bounds__(256) void ingest_pack_kernel(const int32_t* __restrict__ major, const int32_t* __restrict__ minor, int64_t n,
                                                          uint64_t* __restrict__ keys) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) keys[i] = (static_cast<uint64_t>(static_cast<uint32_t>(major[i])) << 32) | static_cast<uint32_t>(minor[i]);
}

// keys sorted: out_minor[i] = low word; indptr[m] = i + 1 for every major id m in [major(i), major(i+1))
__global__ __launch_bounds__(256) void ingest_unpack_kernel(const uint64_t* __restrict__ keys, int64_t n, int num_major, int32_t* __restrict__ out_minor,
                                                            int64_t* __restrict__ indptr) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    out_minor[i] = static_cast<int32_t>(k & 0xFFFFFFFFull);
    const int m0 = static_cast<int>(k >> 32);
    const int m1 = i + 1 < n ? static_cast<int>(keys[i + 1] >> 32) : num_major;
    for (int m = m0; m < m1; ++m) indptr[m] = i + 1;
    if (i == 0)
        for (int m = 0; m < m0; ++m) indptr[m] = 0;   // leading empty rows
}

static int bits_for(int64_t range) {   // bits needed for ids in [0, range)
    int b = 1;
    while ((int64_t(1) << b) < range) ++b;
    return b;
}

void device_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n, int bits,
                           DevBuf<char>& tmp, hipStream_t s) {
    size_t bytes = 0;
    const size_t count = static_cast<size_t>(n);
    const unsigned end_bit = static_cast<unsigned>(bits);
    BFH_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, count, 0u, end_bit, s));
    if (tmp.size() < bytes) tmp.resize(bytes ? bytes : 1);
    bytes = tmp.size();
    BFH_HIP(rocprim::radix_sort_pairs(tmp.get(), bytes, keys_in, keys_out, vals_in, vals_out, count, 0u, end_bit, s));
}

void device_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int64_t* vals_in, int64_t* vals_out, int64_t n, int bits,
                           DevBuf<char>& tmp, hipStream_t s) {
    size_t bytes = 0;
    const size_t count = static_cast<size_t>(n);
    const unsigned end_bit = static_cast<unsigned>(bits);
    BFH_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, count, 0u, end_bit, s));
    if (tmp.size() < bytes) tmp.resize(bytes ? bytes : 1);
    bytes = tmp.size();
    BFH_HIP(rocprim::radix_sort_pairs(tmp.get(), bytes, keys_in, keys_out, vals_in, vals_out, count, 0u, end_bit, s));
}

static void coo_to_csr(const int32_t* major, const int32_t* minor, const float* vals, int64_t nnz, int num_major, int num_minor, int64_t* indptr,
                       int32_t* out_minor, float* out_vals, bfh_stats* stats) {
    BFH_REQUIRE(nnz >= 0 && num_major > 0 && num_minor > 0, "coo_to_csr: empty shape");
    for (int64_t i = 0; i < nnz; ++i)
        if (major[i] < 0 || major[i] >= num_major || minor[i] < 0 || minor[i] >= num_minor)
            throw Error(BFH_ERR_INVALID, "coo_to_csr: id outside the matrix at record " + std::to_string(i));
    if (nnz == 0) {
        std::fill(indptr, indptr + num_major, int64_t(0));
        return;
    }
    hipStream_t stream;
    BFH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct Guard {
        hipStream_t s;
        ~Guard() { (void)hipStreamDestroy(s); }
    } guard{stream};
    DevBuf<int32_t> d_major, d_minor;
    DevBuf<float> d_vin, d_vout;
    DevBuf<uint64_t> d_kin, d_kout;
    DevBuf<int64_t> d_indptr;
    DevBuf<char> d_tmp;
    d_major.resize(nnz); d_minor.resize(nnz); d_vin.resize(nnz); d_vout.resize(nnz); d_kin.resize(nnz); d_kout.resize(nnz);
    d_indptr.resize(num_major);
    BFH_HIP(hipMemcpyAsync(d_major.get(), major, nnz * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_minor.get(), minor, nnz * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_vin.get(), vals, nnz * 4, hipMemcpyHostToDevice, stream));
    hipEvent_t e0, e1;
    BFH_HIP(hipEventCreate(&e0));
    BFH_HIP(hipEventCreate(&e1));
    BFH_HIP(hipEventRecord(e0, stream));
    const unsigned blocks = static_cast<unsigned>((nnz + 255) / 256);
    hipLaunchKernelGGL(ingest_pack_kernel, dim3(blocks), dim3(256), 0, stream, d_major.get(), d_minor.get(), nnz, d_kin.get());
    BFH_HIP(hipGetLastError());
    const int end_bit = 32 + bits_for(num_major);   // the minor word is sorted over the bits it uses, the gap above it is all zero
    size_t tmp_bytes = 0;
    const size_t count = static_cast<size_t>(nnz);
    BFH_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_kin.get(), d_kout.get(), d_vin.get(), d_vout.get(), count, 0u, static_cast<unsigned>(end_bit), stream));
    d_tmp.resize(tmp_bytes ? tmp_bytes : 1);
    BFH_HIP(rocprim::radix_sort_pairs(d_tmp.get(), tmp_bytes, d_kin.get(), d_kout.get(), d_vin.get(), d_vout.get(), count, 0u, static_cast<unsigned>(end_bit), stream));
    hipLaunchKernelGGL(ingest_unpack_kernel, dim3(blocks), dim3(256), 0, stream, d_kout.get(), nnz, num_major, d_minor.get(), d_indptr.get());
    BFH_HIP(hipGetLastError());
    BFH_HIP(hipEventRecord(e1, stream));
    BFH_HIP(hipMemcpyAsync(out_minor, d_minor.get(), nnz * 4, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipMemcpyAsync(out_vals, d_vout.get(), nnz * 4, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipMemcpyAsync(indptr, d_indptr.get(), static_cast<size_t>(num_major) * 8, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipStreamSynchronize(stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (stats) {
        *stats = bfh_stats{};
        stats->samples = nnz;
        stats->kernel_ms = ms;
        stats->h2d_bytes = 12.0 * nnz;
        stats->d2h_bytes = 8.0 * nnz + 8.0 * num_major;
    }
}

}  // namespace bfh

extern "C" {

int bfh_coo_to_csr(const int32_t* major, const int32_t* minor, const float* vals, int64_t nnz, int num_major, int num_minor, int64_t* indptr,
                   int32_t* out_minor, float* out_vals, bfh_stats* stats) {
    try {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) throw bfh::Error(BFH_ERR_HIP, "no HIP device available (libbuffalo_hip has no CPU fallback)");
        bfh::coo_to_csr(major, minor, vals, nnz, num_major, num_minor, indptr, out_minor, out_vals, stats);
        return BFH_OK;
    } catch (const bfh::Error& e) {
        bfh::g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return BFH_ERR_HIP;
    }
}

}  // extern "C"
