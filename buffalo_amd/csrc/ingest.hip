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
#include <thread>

#include <cstdio>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

// ------------------------------------------------------------------------------------------------------------------------------
// The working TEXT file of buffalo's data creation parsed on the device (round 5; fileio.hpp:263-330).
//
// Reference semantics: the file holds one "row col val" line per entry (1-based ids; buffalo/data/mm.py:175-234 copies the MatrixMarket
// body lines into it), every line is parsed with sscanf(line, "%d %d %f") (:300-303) -- up to 64 OpenMP threads over 4 MB splits whose
// boundary lines are stitched so that the net effect is "all lines in file order" -- and only the first `total_lines` are kept (:312-320).
//
// Device formulation: (1) newline positions by a count / scan / scatter over 4 KB tiles (one pass of counting, one of writing; the scan of
// the tile counts is the one library primitive); (2) one thread per line parses the two integers and the decimal number.  sscanf's "%f" is
// strtof: the CORRECTLY ROUNDED binary32 of the decimal string.  The kernel reproduces it exactly where one rounding suffices -- Clinger's
// fast paths: <= 2^24 in the digits and 10^|e| <= 10^10 is one exact-operand float operation; <= 2^53 and |e| <= 22 is one exact-operand
// double operation whose result is then rounded to float, which equals strtof unless the double lands within one ulp of a float rounding
// boundary -- and FLAGS every line it cannot guarantee (such a near-tie, > 19 significant digits, exponents outside the fast paths,
// subnormal / overflowing results, "inf" / "nan" / hex floats, integers beyond int, malformed lines).  (3) Flagged lines -- none on
// ordinary rating files -- are re-parsed on the host with the reference's own sscanf call.  Result: bit-identical triples by construction.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kTextTile = 4096;   // bytes per 256-thread block: 16 per thread

__global__ __launch_bounds__(256) void text_count_newlines_kernel(const char* __restrict__ text, int64_t bytes, int64_t* __restrict__ tile_count) {
    __shared__ int s_cnt[4];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kTextTile + threadIdx.x * 16;
    int c = 0;
    if (base + 16 <= bytes) {
        const uint4 w = *reinterpret_cast<const uint4*>(text + base);
        const unsigned v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) c += ((v[k] >> (8 * j)) & 0xffu) == 10u;
    } else {
        for (int64_t i = base; i < bytes && i < base + 16; ++i) c += text[i] == '\n';
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// nl[k] = byte offset of the k-th '\n' (tile_base[] = exclusive scan of the tile counts), written only for k < cap
__global__ __launch_bounds__(256) void text_write_newlines_kernel(const char* __restrict__ text, int64_t bytes, const int64_t* __restrict__ tile_base,
                                                                  int64_t cap, int64_t* __restrict__ nl) {
    __shared__ int s_scan[256];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kTextTile + threadIdx.x * 16;
    unsigned mask = 0;   // bit j: byte base + j is a newline
    for (int j = 0; j < 16; ++j)
        if (base + j < bytes && text[base + j] == '\n') mask |= 1u << j;
    const int mine = __popc(mask);
    s_scan[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // Hillis-Steele inclusive scan over the block's 256 counts
        const int v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t k = tile_base[blockIdx.x] + (s_scan[threadIdx.x] - mine);
    while (mask) {
        const int j = __ffs(mask) - 1;
        mask &= mask - 1;
        if (k < cap) nl[k] = base + j;
        ++k;
    }
}

__device__ __forceinline__ bool txt_space(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; }

// "%d": skip white space, optional sign, digits.  false: not representable here (no digits, beyond int) -> the host re-parses the line
__device__ __forceinline__ bool txt_int(const char* __restrict__ t, int64_t& i, int64_t end, int& out) {
    while (i < end && txt_space(t[i])) ++i;
    bool neg = false;
    if (i < end && (t[i] == '-' || t[i] == '+')) { neg = t[i] == '-'; ++i; }
    int64_t v = 0;
    int nd = 0;
    while (i < end && t[i] >= '0' && t[i] <= '9') {
        v = v * 10 + (t[i] - '0');
        ++nd; ++i;
        if (v > 4294967296ll) return false;
    }
    if (nd == 0) return false;
    v = neg ? -v : v;
    if (v > 2147483647ll || v < -2147483648ll) return false;
    out = static_cast<int>(v);
    return true;
}

// "%f" = strtof on [+-]digits[.digits][(e|E)[+-]digits]; false: not guaranteed bit-exact here (see the header comment)
__device__ __forceinline__ bool txt_float(const char* __restrict__ t, int64_t& i, int64_t end, float& out) {
    while (i < end && txt_space(t[i])) ++i;
    bool neg = false;
    if (i < end && (t[i] == '-' || t[i] == '+')) { neg = t[i] == '-'; ++i; }
    unsigned long long w = 0;
    int sig = 0, nd = 0, e10 = 0;
    bool dot = false;
    for (; i < end; ++i) {
        const char ch = t[i];
        if (ch == '.' && !dot) { dot = true; continue; }
        if (ch < '0' || ch > '9') break;
        ++nd;
        if (sig == 0 && ch == '0') { if (dot) --e10; continue; }   // leading zeros carry no significance
        if (sig >= 19) return false;                                // more digits than one exact integer holds
        w = w * 10 + static_cast<unsigned>(ch - '0');
        ++sig;
        if (dot) --e10;
    }
    if (nd == 0) return false;          // "inf", "nan", ".", garbage
    if (i < end && (t[i] == 'x' || t[i] == 'X')) return false;   // hex float ("0x1p3"): strtof reads on
    if (i < end && (t[i] == 'e' || t[i] == 'E')) {
        int64_t j = i + 1;
        bool eneg = false;
        if (j < end && (t[j] == '-' || t[j] == '+')) { eneg = t[j] == '-'; ++j; }
        int ev = 0, ed = 0;
        while (j < end && t[j] >= '0' && t[j] <= '9') {
            if (ev < 100000) ev = ev * 10 + (t[j] - '0');
            ++ed; ++j;
        }
        if (ed > 0) { e10 += eneg ? -ev : ev; i = j; }   // "1e" / "1e+" : the exponent part is not consumed
    }
    if (w == 0) { out = neg ? -0.0f : 0.0f; return true; }
    const int ae = e10 < 0 ? -e10 : e10;
    if (w <= (1ull << 24) && ae <= 10) {        // both operands exact in binary32: ONE correctly rounded operation
        const float p10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
        const float f = static_cast<float>(static_cast<unsigned>(w));
        const float r = e10 < 0 ? __fdiv_rn(f, p10[ae]) : __fmul_rn(f, p10[ae]);
        out = neg ? -r : r;
        return true;
    }
    if (w <= (1ull << 53) && ae <= 22) {        // both exact in binary64: one correctly rounded double, then the rounding to float
        const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        const double dw = static_cast<double>(w);
        const double d = e10 < 0 ? __ddiv_rn(dw, p10[ae]) : __dmul_rn(dw, p10[ae]);
        if (!(d >= 1.17549435e-38 && d <= 3.4028234e38)) return false;      // subnormal or overflowing binary32: strtof's own business
        const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(d));
        const unsigned low = static_cast<unsigned>(bits & 0x1fffffffull);    // the 29 bits the rounding to float drops
        if (low >= 0x0ffffffeu && low <= 0x10000002u) return false;          // within two double ulps of a float rounding boundary: double rounding could differ
        const float r = static_cast<float>(d);
        out = neg ? -r : r;
        return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void text_parse_kernel(const char* __restrict__ text, int64_t bytes, const int64_t* __restrict__ nl, int64_t n_nl,
                                                         int64_t lines, int32_t* __restrict__ r, int32_t* __restrict__ c, float* __restrict__ v,
                                                         int64_t* __restrict__ redo, int64_t redo_cap, unsigned long long* __restrict__ n_redo) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (k >= lines) return;
    int64_t i = k == 0 ? 0 : nl[k - 1] + 1;
    const int64_t end = k < n_nl ? nl[k] : bytes;
    int rr = 0, cc = 0;
    float vv = 0.f;
    const bool ok = txt_int(text, i, end, rr) && txt_int(text, i, end, cc) && txt_float(text, i, end, vv);
    r[k] = rr; c[k] = cc; v[k] = vv;
    if (!ok) {
        const unsigned long long slot = atomicAdd(n_redo, 1ull);
        if (slot < static_cast<unsigned long long>(redo_cap)) redo[slot] = k;
    }
}

// 1-based (row, col) -> 0-based (major, minor) of the orientation `sort_key` names; out-of-range ids are counted
__global__ __launch_bounds__(256) void text_orient_kernel(const int32_t* __restrict__ r, const int32_t* __restrict__ c, int64_t n, int sort_key, int num_major,
                                                          int num_minor, int32_t* __restrict__ major, int32_t* __restrict__ minor,
                                                          unsigned long long* __restrict__ n_bad) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const int a = (sort_key == 2 ? c[i] : r[i]) - 1, b = (sort_key == 2 ? r[i] : c[i]) - 1;
    major[i] = a;
    minor[i] = b;
    if (a < 0 || a >= num_major || b < 0 || b >= num_minor) atomicAdd(n_bad, 1ull);
}

struct StreamGuard {
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~StreamGuard() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
    }
};

// the sort + compress of device arrays (major / minor / vals 0-based and validated): results into out_* (device), indptr (device)
static void csr_from_device_coo(const int32_t* d_major, int32_t* d_minor /* overwritten with the sorted minors */, const float* d_vin, float* d_vout, int64_t nnz,
                                int num_major, int64_t* d_indptr, DevBuf<uint64_t>& d_kin, DevBuf<uint64_t>& d_kout, DevBuf<char>& d_tmp, hipStream_t stream) {
    const unsigned blocks = static_cast<unsigned>((nnz + 255) / 256);
    d_kin.resize(nnz); d_kout.resize(nnz);
    hipLaunchKernelGGL(ingest_pack_kernel, dim3(blocks), dim3(256), 0, stream, d_major, d_minor, nnz, d_kin.get());
    BFH_HIP(hipGetLastError());
    const int end_bit = 32 + bits_for(num_major);   // the minor word is sorted over the bits it uses, the gap above it is all zero
    size_t tmp_bytes = 0;
    const size_t count = static_cast<size_t>(nnz);
    BFH_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_kin.get(), d_kout.get(), d_vin, d_vout, count, 0u, static_cast<unsigned>(end_bit), stream));
    if (d_tmp.size() < tmp_bytes) d_tmp.resize(tmp_bytes ? tmp_bytes : 1);
    tmp_bytes = d_tmp.size();
    BFH_HIP(rocprim::radix_sort_pairs(d_tmp.get(), tmp_bytes, d_kin.get(), d_kout.get(), d_vin, d_vout, count, 0u, static_cast<unsigned>(end_bit), stream));
    hipLaunchKernelGGL(ingest_unpack_kernel, dim3(blocks), dim3(256), 0, stream, d_kout.get(), nnz, num_major, d_minor, d_indptr);
    BFH_HIP(hipGetLastError());
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
    StreamGuard g;
    BFH_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
    hipStream_t stream = g.s;
    DevBuf<int32_t> d_major, d_minor;
    DevBuf<float> d_vin, d_vout;
    DevBuf<uint64_t> d_kin, d_kout;
    DevBuf<int64_t> d_indptr;
    DevBuf<char> d_tmp;
    d_major.resize(nnz); d_minor.resize(nnz); d_vin.resize(nnz); d_vout.resize(nnz);
    d_indptr.resize(num_major);
    BFH_HIP(hipMemcpyAsync(d_major.get(), major, nnz * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_minor.get(), minor, nnz * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_vin.get(), vals, nnz * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipEventCreate(&g.e0));
    BFH_HIP(hipEventCreate(&g.e1));
    BFH_HIP(hipEventRecord(g.e0, stream));
    csr_from_device_coo(d_major.get(), d_minor.get(), d_vin.get(), d_vout.get(), nnz, num_major, d_indptr.get(), d_kin, d_kout, d_tmp, stream);
    BFH_HIP(hipEventRecord(g.e1, stream));
    BFH_HIP(hipMemcpyAsync(out_minor, d_minor.get(), nnz * 4, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipMemcpyAsync(out_vals, d_vout.get(), nnz * 4, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipMemcpyAsync(indptr, d_indptr.get(), static_cast<size_t>(num_major) * 8, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipStreamSynchronize(stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g.e0, g.e1);
    if (stats) {
        *stats = bfh_stats{};
        stats->samples = nnz;
        stats->kernel_ms = ms;
        stats->h2d_bytes = 12.0 * nnz;
        stats->d2h_bytes = 8.0 * nnz + 8.0 * num_major;
    }
}

// text -> (r, c, v) on the device (1-based ids as sscanf reads them), the first `total_lines` lines.  Returns the number of lines the host re-parsed.
struct TextTriples {
    DevBuf<char> text;
    DevBuf<int64_t> tile_cnt, tile_base, nl, redo;
    DevBuf<int32_t> r, c;
    DevBuf<float> v;
    DevBuf<unsigned long long> counters;   // [0] lines to re-parse, [1] ids outside the matrix
    DevBuf<char> tmp;
    int64_t reparsed = 0;
};

// re-parsed lines back to their places: line idx[j] takes (hr, hc, hv)[j]
__global__ __launch_bounds__(256) void text_scatter_kernel(const int64_t* __restrict__ idx, const int32_t* __restrict__ hr, const int32_t* __restrict__ hc,
                                                           const float* __restrict__ hv, int64_t m, int64_t total_lines, int32_t* __restrict__ r,
                                                           int32_t* __restrict__ c, float* __restrict__ v) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t k = idx[j];
    if (k < 0 || k >= total_lines) return;
    r[k] = hr[j]; c[k] = hc[j]; v[k] = hv[j];
}

static void parse_text_on_device(const char* text, int64_t bytes, int64_t total_lines, TextTriples& T, hipStream_t stream) {
    BFH_REQUIRE(text && bytes > 0 && total_lines > 0, "text parse: empty input");
    T.text.resize(static_cast<size_t>(bytes) + 16);
    BFH_HIP(hipMemcpyAsync(T.text.get(), text, static_cast<size_t>(bytes), hipMemcpyHostToDevice, stream));
    const int64_t tiles = (bytes + kTextTile - 1) / kTextTile;
    T.tile_cnt.resize(tiles + 1); T.tile_base.resize(tiles + 1);
    T.counters.resize(2, true, stream);
    BFH_HIP(hipMemsetAsync(T.tile_cnt.get() + tiles, 0, sizeof(int64_t), stream));
    hipLaunchKernelGGL(text_count_newlines_kernel, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, stream, T.text.get(), bytes, T.tile_cnt.get());
    BFH_HIP(hipGetLastError());
    size_t tb = 0;
    BFH_HIP(rocprim::exclusive_scan(nullptr, tb, T.tile_cnt.get(), T.tile_base.get(), int64_t(0), static_cast<size_t>(tiles + 1), rocprim::plus<int64_t>(), stream));
    if (T.tmp.size() < tb) T.tmp.resize(tb ? tb : 1);
    tb = T.tmp.size();
    BFH_HIP(rocprim::exclusive_scan(T.tmp.get(), tb, T.tile_cnt.get(), T.tile_base.get(), int64_t(0), static_cast<size_t>(tiles + 1), rocprim::plus<int64_t>(), stream));
    int64_t n_nl = 0;
    BFH_HIP(hipMemcpyAsync(&n_nl, T.tile_base.get() + tiles, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipStreamSynchronize(stream));
    const int64_t lines_in_file = n_nl + (text[bytes - 1] != '\n' ? 1 : 0);   // getline also returns an unterminated last line
    BFH_REQUIRE(lines_in_file >= total_lines, "text parse: the file holds " + std::to_string(lines_in_file) + " lines, total_lines says " +
                                                  std::to_string(total_lines) + " (fileio.hpp:322 asserts the same)");
    const int64_t cap = std::min(n_nl, total_lines);
    T.nl.resize(std::max<int64_t>(1, cap));
    hipLaunchKernelGGL(text_write_newlines_kernel, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, stream, T.text.get(), bytes, T.tile_base.get(), cap, T.nl.get());
    BFH_HIP(hipGetLastError());
    T.r.resize(total_lines); T.c.resize(total_lines); T.v.resize(total_lines);
    const int64_t redo_cap = std::min<int64_t>(total_lines, int64_t(1) << 22);
    T.redo.resize(redo_cap);
    hipLaunchKernelGGL(text_parse_kernel, dim3(static_cast<unsigned>((total_lines + 255) / 256)), dim3(256), 0, stream, T.text.get(), bytes, T.nl.get(), cap,
                       total_lines, T.r.get(), T.c.get(), T.v.get(), T.redo.get(), redo_cap, T.counters.get());
    BFH_HIP(hipGetLastError());
    unsigned long long n_redo = 0;
    BFH_HIP(hipMemcpyAsync(&n_redo, T.counters.get(), sizeof(n_redo), hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipStreamSynchronize(stream));
    T.reparsed = static_cast<int64_t>(n_redo);
    if (n_redo == 0) return;
    // the lines the kernel would not vouch for: the reference's own call, line by line (fileio.hpp:300-303)
    std::vector<int64_t> idx;
    std::vector<int64_t> nlh(static_cast<size_t>(cap));
    if (cap) BFH_HIP(hipMemcpy(nlh.data(), T.nl.get(), static_cast<size_t>(cap) * 8, hipMemcpyDeviceToHost));
    if (n_redo <= static_cast<unsigned long long>(redo_cap)) {
        idx.resize(n_redo);
        BFH_HIP(hipMemcpy(idx.data(), T.redo.get(), n_redo * 8, hipMemcpyDeviceToHost));
    } else {   // more flagged lines than the list holds (a file in a format the kernel does not speak): every line goes the host way
        idx.resize(static_cast<size_t>(total_lines));
        for (int64_t k = 0; k < total_lines; ++k) idx[k] = k;
        T.reparsed = total_lines;
    }
    std::vector<int32_t> hr(idx.size()), hc(idx.size());
    std::vector<float> hv(idx.size());
    // every line is independent: above a few thousand of them the host's share is cut over up to 8 threads (a file of 17-digit reprs hands MOST of its
    // lines back; one thread's sscanf is ~5 M lines per second)
    auto reparse = [&](size_t j0, size_t j1) {
        std::string line;
        for (size_t j = j0; j < j1; ++j) {
            const int64_t k = idx[j];
            const int64_t beg = k == 0 ? 0 : nlh[k - 1] + 1, end = k < cap ? nlh[k] : bytes;
            line.assign(text + beg, text + end);
            int rr = 0, cc = 0;
            float vv = 0.f;
            sscanf(line.c_str(), "%d %d %f", &rr, &cc, &vv);
            hr[j] = rr; hc[j] = cc; hv[j] = vv;
        }
    };
    {
        const size_t m0 = idx.size();
        unsigned hw = std::thread::hardware_concurrency();
        const size_t parts = std::max<size_t>(1, std::min<size_t>({8, hw ? hw : 1, m0 / 4096}));
        if (parts == 1) {
            reparse(0, m0);
        } else {
            std::vector<std::thread> th;
            th.reserve(parts);
            size_t started_to = 0;
            try {
                for (size_t t = 0; t < parts; ++t) {
                    const size_t a = m0 * t / parts, b = m0 * (t + 1) / parts;
                    th.emplace_back(reparse, a, b);
                    started_to = b;
                }
            } catch (...) {   // a thread could not be started: the calling thread takes what is left
            }
            if (started_to < m0) reparse(started_to, m0);
            for (auto& x : th) x.join();
        }
    }
    // scatter back: the list and the re-parsed values go up ONCE and a small kernel puts every line where it belongs.  (Round 5 copied 3 x 4
    // bytes per line synchronously -- millions of blocking copies for a file of 17-digit values -- and, when EVERY line was flagged but the list
    // still held them, wrote the values in the list's order, which is the order of the flagging atomics, not of the file.)
    const size_t m = idx.size();
    DevBuf<int64_t> d_idx;
    DevBuf<int32_t> d_r, d_c;
    DevBuf<float> d_v;
    d_idx.resize(m); d_r.resize(m); d_c.resize(m); d_v.resize(m);
    BFH_HIP(hipMemcpyAsync(d_idx.get(), idx.data(), m * 8, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_r.get(), hr.data(), m * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_c.get(), hc.data(), m * 4, hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(d_v.get(), hv.data(), m * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(text_scatter_kernel, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, stream, d_idx.get(), d_r.get(), d_c.get(), d_v.get(),
                       static_cast<int64_t>(m), total_lines, T.r.get(), T.c.get(), T.v.get());
    BFH_HIP(hipGetLastError());
    BFH_HIP(hipStreamSynchronize(stream));   // the staging vectors and buffers die with this frame
}

static void parse_triples(const char* text, int64_t bytes, int64_t total_lines, int32_t* rows, int32_t* cols, float* vals, bfh_stats* stats) {
    StreamGuard g;
    BFH_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
    BFH_HIP(hipEventCreate(&g.e0));
    BFH_HIP(hipEventCreate(&g.e1));
    TextTriples T;
    BFH_HIP(hipEventRecord(g.e0, g.s));
    parse_text_on_device(text, bytes, total_lines, T, g.s);
    BFH_HIP(hipEventRecord(g.e1, g.s));
    BFH_HIP(hipMemcpyAsync(rows, T.r.get(), total_lines * 4, hipMemcpyDeviceToHost, g.s));
    BFH_HIP(hipMemcpyAsync(cols, T.c.get(), total_lines * 4, hipMemcpyDeviceToHost, g.s));
    BFH_HIP(hipMemcpyAsync(vals, T.v.get(), total_lines * 4, hipMemcpyDeviceToHost, g.s));
    BFH_HIP(hipStreamSynchronize(g.s));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g.e0, g.e1);
    if (stats) {
        *stats = bfh_stats{};
        stats->samples = total_lines;
        stats->kernel_ms = ms;                 // upload of the text + newline index + parse (+ the host's share, if any line was flagged)
        stats->merges = T.reparsed;            // lines re-parsed on the host with sscanf
        stats->h2d_bytes = static_cast<double>(bytes);
        stats->d2h_bytes = 12.0 * total_lines;
    }
}

static void text_to_csr(const char* text, int64_t bytes, int64_t total_lines, int num_major, int num_minor, int sort_key, int64_t* indptr,
                        int32_t* out_minor, float* out_vals, bfh_stats* stats) {
    BFH_REQUIRE(sort_key == 1 || sort_key == 2, "text_to_csr: sort_key is 1 (rowwise) or 2 (colwise)");
    BFH_REQUIRE(num_major > 0 && num_minor > 0, "text_to_csr: empty shape");
    StreamGuard g;
    BFH_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
    BFH_HIP(hipEventCreate(&g.e0));
    BFH_HIP(hipEventCreate(&g.e1));
    hipStream_t stream = g.s;
    TextTriples T;
    BFH_HIP(hipEventRecord(g.e0, stream));
    parse_text_on_device(text, bytes, total_lines, T, stream);
    const int64_t nnz = total_lines;
    DevBuf<int32_t> d_major, d_minor;
    DevBuf<float> d_vout;
    DevBuf<uint64_t> d_kin, d_kout;
    DevBuf<int64_t> d_indptr;
    d_major.resize(nnz); d_minor.resize(nnz); d_vout.resize(nnz); d_indptr.resize(num_major);
    hipLaunchKernelGGL(text_orient_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0, stream, T.r.get(), T.c.get(), nnz, sort_key, num_major,
                       num_minor, d_major.get(), d_minor.get(), T.counters.get() + 1);
    BFH_HIP(hipGetLastError());
    unsigned long long bad = 0;
    BFH_HIP(hipMemcpyAsync(&bad, T.counters.get() + 1, sizeof(bad), hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipStreamSynchronize(stream));
    BFH_REQUIRE(bad == 0, "text_to_csr: " + std::to_string(bad) + " records with an id outside the matrix");
    T.text.release();   // the bytes are not needed any more: room for the sort
    csr_from_device_coo(d_major.get(), d_minor.get(), T.v.get(), d_vout.get(), nnz, num_major, d_indptr.get(), d_kin, d_kout, T.tmp, stream);
    BFH_HIP(hipEventRecord(g.e1, stream));
    BFH_HIP(hipMemcpyAsync(out_minor, d_minor.get(), nnz * 4, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipMemcpyAsync(out_vals, d_vout.get(), nnz * 4, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipMemcpyAsync(indptr, d_indptr.get(), static_cast<size_t>(num_major) * 8, hipMemcpyDeviceToHost, stream));
    BFH_HIP(hipStreamSynchronize(stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g.e0, g.e1);
    if (stats) {
        *stats = bfh_stats{};
        stats->samples = nnz;
        stats->kernel_ms = ms;
        stats->merges = T.reparsed;
        stats->h2d_bytes = static_cast<double>(bytes);
        stats->d2h_bytes = 8.0 * nnz + 8.0 * num_major;
    }
}

}  // namespace bfh

template <typename F>
static int stateless_call(F&& fn) {
    try {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) throw bfh::Error(BFH_ERR_HIP, "no HIP device available (libbuffalo_hip has no CPU fallback)");
        fn();
        return BFH_OK;
    } catch (const bfh::Error& e) {
        bfh::g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        bfh::g_create_error = e.what();
        return BFH_ERR_HIP;
    }
}


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

int bfh_parse_triples(const char* text, int64_t bytes, int64_t total_lines, int32_t* rows, int32_t* cols, float* vals, bfh_stats* stats) {
    return stateless_call([&] { bfh::parse_triples(text, bytes, total_lines, rows, cols, vals, stats); });
}

int bfh_text_to_csr(const char* text, int64_t bytes, int64_t total_lines, int num_major, int num_minor, int sort_key, int64_t* indptr,
                    int32_t* out_minor, float* out_vals, bfh_stats* stats) {
    return stateless_call([&] { bfh::text_to_csr(text, bytes, total_lines, num_major, num_minor, sort_key, indptr, out_minor, out_vals, stats); });
}

}  // extern "C"
