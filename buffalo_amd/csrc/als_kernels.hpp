// ALS kernels + handle (gfx950).
//
// Reference numerics (all paths relative to /root/reference/):
//   precompute            CALS::precompute                lib/algo_impl/als/als.cc:86-93
//   dense row update      CALS::_partial_update           lib/algo_impl/als/als.cc:107-209
//   row solvers           Algorithm::_leastsquare 0,1,2   lib/algo.cc:52-82
//   iALS++ block update   CALS::_partial_update_ialspp    lib/algo_impl/als/als.cc:211-358
// Object surface: CuALS   include/buffalo/cuda/als/als.hpp:20-35.
//
// Data layout: factor rows are vdim floats; a wave holds a row as K = vdim/64 dwords per lane
// (element k*64+lane) so every row access is K fully coalesced 256-B transactions and a dot
// product is K FMAs + a DPP row reduction.  One wave owns one row of the side being solved; rows
// are handed out through an atomic ticket (the reference uses omp schedule(dynamic,4)).
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <tuple>

#include <type_traits>

#include "common.hpp"
#include "comm.hpp"

namespace bfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AlsParams {
    float* P;              // side being solved   [rows, vdim]
    const float* Q;        // other side          [op_rows, vdim]
    const float* FF;       // [vdim, vdim] Gramian of the other side (symmetric)
    const int64_t* indptr; // full-matrix end offsets of the side being solved
    const int32_t* keys;   // chunk-local
    const float* vals;     // chunk-local
    float* yui;            // chunk-local scratch (iALS++)
    int64_t shift;
    int start_x, next_x;
    int d, vdim, op_rows, block_size;
    float alpha, reg, eps, cg_tol;
    int adaptive_reg, compute_loss, axis, num_cg_max_iters;
    double* loss;          // [0] nume, [1] deno
    int* ticket;
    int debug;             // profiling ablations: 1 skip dense solve, 4 skip M/FF staging
    int solver;            // 0 llt, 1 ldlt, 2 manual_cg, 8 ialspp
    // generalisations used by CFR (cfr_impl.hpp); ALS sets ctx = accumulate = 0, out_scale = ff_scale = 1
    int ctx;               // Gramian pass with weight 1 and coefficient (v - bias_self[row] - bias_other[key]) (cfr.cc:209-225, 283-289)
    const float* bias_self;
    const float* bias_other;
    float out_scale;       // the pass's tiles / vector are multiplied by this before they reach the scratch slot (cfr.cc:130-131)
    const float* F0;       // als_gram_kernel<SPLIT>: row (x - start_x) = FF p0 of row x (als_rowff_kernel), so the one-wave-per-SIMD pass does not form it
    int batch;             // als_gram_kernel<SPLIT>: work items drawn per ticket, at most
    const float* split;    // als_gram_kernel<SPLIT>: {S, S^2, 1/S^2, weight cut} written by als_split_scale_kernel (device-side, no host round trip)
    int accumulate;        // add into the row's (zeroed) slot instead of overwriting it: two passes build one system
    float ff_scale;        // the solve kernel's M = ff_scale * FF + slot
    const float* Qi;       // als_wide_kernel<SPLIT>: the block-interleaved copy of Q (als_interleave_stats_kernel: Qi[row][T col + b] = Q[row][32 b + col])
};

template <int K>
struct ARow {
    float v[K];
};
template <int K>
__device__ __forceinline__ void aload(ARow<K>& r, const float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        r.v[k] = (e < vdim) ? base[e] : 0.0f;
    }
}
template <int K>
__device__ __forceinline__ void astore(const ARow<K>& r, float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        if (e < vdim) base[e] = r.v[k];
    }
}
template <int K>
__device__ __forceinline__ float adot(const ARow<K>& a, const ARow<K>& b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += a.v[k] * b.v[k];
    return wave_sum(s);
}
// element e (uniform) of a row held in K dwords per lane
template <int K>
__device__ __forceinline__ float aget(const ARow<K>& r, int e) {
    float out = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((e >> 6) == k) out = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.v[k]), e & 63));
    return out;
}
// out = x * FF  (row vector times symmetric matrix), d rows of FF streamed (L1/L2 resident)
template <int K>
__device__ __forceinline__ void avecmat(ARow<K>& out, const ARow<K>& x, const float* __restrict__ FF, int d, int vdim, int lane) {
#pragma unroll
    for (int k = 0; k < K; ++k) out.v[k] = 0.f;
    for (int i = 0; i < d; ++i) {
        const float xi = aget<K>(x, i);
        ARow<K> f;
        aload<K>(f, FF + static_cast<size_t>(i) * vdim, lane, vdim);
#pragma unroll
        for (int k = 0; k < K; ++k) out.v[k] += xi * f.v[k];
    }
}

__device__ __forceinline__ int next_row(int* ticket, int lane) {
    int r = 0;
    if (lane == 0) r = atomicAdd(ticket, 1);
    return __builtin_amdgcn_readfirstlane(r);
}

// ------------------------------------------------------------------------------------------------
// FF = F^T F on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32).  One wave produces a
// 32 x (32*NT) strip for a slice of rows; A operand = F[row][bi*32 + (lane&31)], row = r + (lane>>5),
// B operands = the same two rows at column tiles bj..bj+NT-1.  The slices' partials are combined with
// fp64 atomics into a zeroed accumulator and rounded to fp32 once (als_gramian_round_kernel): the order
// the slices arrive in then perturbs the sum at the 1e-16 level, i.e. FF is reproducible run to run
// (cublasSgemm in the reference, lib/cuda/als/als.cu:315-317, is deterministic too; fp32 atomics were
// not, and iALS++ amplifies a 1e-7 wobble of FF through the cancellation in its gradient).
// ------------------------------------------------------------------------------------------------
template <int NT, int UPG = 8>
__global__ __launch_bounds__(64) void als_gramian_kernel(const float* __restrict__ F, int rows, int vdim, int rows_per_slice,
                                                          double* __restrict__ FF) {
    const int lane = threadIdx.x;
    const int T = vdim / 32;
    const int bi = blockIdx.y;
    const int bj0 = blockIdx.z * NT;
    const int r0 = blockIdx.x * rows_per_slice;
    const int r1 = (r0 + rows_per_slice < rows) ? r0 + rows_per_slice : rows;
    f32x16 acc[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    const int half = lane >> 5, col = lane & 31;
    // UPG row pairs per trip, all their loads issued before the first matrix instruction (round 5: with one pair per trip every trip paid a full
    // memory round trip), and -- round 6 -- the loads of the NEXT trip issued before this trip's matrix instructions (two register sets): with two
    // waves per SIMD a trip's 2.5 us round trip was still most of its time (0.09 / 0.13 ms for ML-20M's items / users where the matrix
    // instructions need 6 / 30 us; more slices do not help -- their fp64 atomics cost more than they hide: profiles/r06_als_gramian.txt).
    // The accumulation order per accumulator -- pair by pair -- is unchanged, so FF keeps its bits.
    auto load_trip = [&](int r, float (&a)[UPG], float (&b)[UPG][NT]) {
#pragma unroll
        for (int u = 0; u < UPG; ++u) {
            const int row = r + 2 * u + half;
            const bool ok = row < r1;
            const float* fr = F + static_cast<size_t>(ok ? row : r0) * vdim;
            a[u] = ok ? fr[bi * 32 + col] : 0.f;
#pragma unroll
            for (int g = 0; g < NT; ++g) b[u][g] = (ok && bj0 + g < T) ? fr[(bj0 + g) * 32 + col] : 0.f;
        }
    };
    auto mfma_trip = [&](const float (&a)[UPG], const float (&b)[UPG][NT]) {
#pragma unroll
        for (int u = 0; u < UPG; ++u)
#pragma unroll
            for (int g = 0; g < NT; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][g], acc[g], 0, 0, 0);
    };
    float a0[UPG], b0[UPG][NT], a1[UPG], b1[UPG][NT];
    load_trip(r0, a0, b0);
    for (int r = r0; r < r1; r += 4 * UPG) {
        load_trip(r + 2 * UPG, a1, b1);     // (past the slice's end: zeros, no loads)
        mfma_trip(a0, b0);
        if (r + 2 * UPG >= r1) break;
        load_trip(r + 4 * UPG, a0, b0);
        mfma_trip(a1, b1);
    }
    // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        if (bj0 + g >= T) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
            atomicAdd(FF + static_cast<size_t>(bi * 32 + i) * vdim + (bj0 + g) * 32 + col, static_cast<double>(acc[g][e]));
        }
    }
}

__global__ __launch_bounds__(256) void als_gramian_round_kernel(const double* __restrict__ acc, float* __restrict__ FF, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) FF[e] = static_cast<float>(acc[e]);
}

// ------------------------------------------------------------------------------------------------
// manual_cg row update (optimizer "manual_cg", the default for d < 128): als.cc:107-209 with
// _leastsquare case 2 (algo.cc:58-82, Q-17), matrix-free: the d x d system matrix
// A = FF + alpha*sum v q q^T + reg*ada*I is applied as x*FF + alpha*sum v (x.q) q + reg*ada*x.
// ------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void als_apply(ARow<K>& out, const ARow<K>& x, const AlsParams& p, int64_t beg, int64_t n,
                                          float regada, int lane, float* dots_first /* optional: x.q_k of first pass */,
                                          double* nume, double* deno, bool loss_terms) {
    avecmat<K>(out, x, p.FF, p.d, p.vdim, lane);
    if (loss_terms) {  // als.cc:175-178
        *nume += static_cast<double>(adot<K>(x, out));
        *deno += static_cast<double>(p.op_rows);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) out.v[k] += regada * x.v[k];
    for (int64_t k0 = 0; k0 < n; k0 += 64) {
        const int64_t kk = k0 + lane;
        int myc = 0;
        float myv = 0.f;
        if (kk < n) {
            myc = p.keys[beg + kk];
            myv = p.vals[beg + kk];
        }
        const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
        for (int j = 0; j < nh; ++j) {
            const int c = __builtin_amdgcn_readlane(myc, j);
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), j));
            ARow<K> q;
            aload<K>(q, p.Q + static_cast<size_t>(c) * p.vdim, lane, p.vdim);
            const float dot = adot<K>(x, q);
            const float coef = p.alpha * v * dot;
#pragma unroll
            for (int k = 0; k < K; ++k) out.v[k] += coef * q.v[k];
            if (loss_terms) {  // als.cc:187-192
                *nume -= static_cast<double>(dot * dot);
                *nume += static_cast<double>((dot - 1) * (dot - 1)) * (1.0 + static_cast<double>(v * p.alpha));
                *deno += static_cast<double>(v * p.alpha);
            }
        }
    }
    (void)dots_first;
}

template <int K>
__global__ __launch_bounds__(256) void als_cg_kernel(AlsParams p) {
    const int lane = threadIdx.x & 63;
    const int vdim = p.vdim;
    double nume = 0.0, deno = 0.0;
    const int nrows = p.next_x - p.start_x;
    for (int i = next_row(p.ticket, lane); i < nrows; i = next_row(p.ticket, lane)) {
        const int u = p.start_x + i;
        const int64_t beg = (u == 0 ? 0 : p.indptr[u - 1]) - p.shift;
        const int64_t n = p.indptr[u] - p.shift - beg;
        if (n == 0) continue;  // Q-16: empty rows stay unchanged
        float* Pu = p.P + static_cast<size_t>(u) * vdim;
        ARow<K> x, y, r, pv, Ap;
        aload<K>(x, Pu, lane, vdim);
        const float ada = p.adaptive_reg ? static_cast<float>(n) : 1.0f;
        const float regada = p.reg * ada;
        // y = sum (1 + v*alpha) q   (als.cc:183-185)
#pragma unroll
        for (int k = 0; k < K; ++k) y.v[k] = 0.f;
        for (int64_t k0 = 0; k0 < n; k0 += 64) {
            const int64_t kk = k0 + lane;
            int myc = 0;
            float myv = 0.f;
            if (kk < n) {
                myc = p.keys[beg + kk];
                myv = p.vals[beg + kk];
            }
            const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
            for (int j = 0; j < nh; ++j) {
                const int c = __builtin_amdgcn_readlane(myc, j);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), j));
                ARow<K> q;
                aload<K>(q, p.Q + static_cast<size_t>(c) * vdim, lane, vdim);
                const float coef = static_cast<float>(1.0 + static_cast<double>(v * p.alpha));
#pragma unroll
                for (int k = 0; k < K; ++k) y.v[k] += q.v[k] * coef;
            }
        }
        if (p.compute_loss) nume += static_cast<double>(ada * p.reg * adot<K>(x, x));  // als.cc:198-200
        // r = y - x*A   (algo.cc:61)
        als_apply<K>(Ap, x, p, beg, n, regada, lane, nullptr, &nume, &deno, p.compute_loss && p.axis == 1);
#pragma unroll
        for (int k = 0; k < K; ++k) r.v[k] = y.v[k] - Ap.v[k];
        if (adot<K>(y, y) < adot<K>(r, r)) {  // algo.cc:63-66
#pragma unroll
            for (int k = 0; k < K; ++k) {
                x.v[k] = 0.f;
                r.v[k] = y.v[k];
            }
        }
        pv = r;
        float rs_old = adot<K>(r, r);
        for (int it = 0; it < p.num_cg_max_iters; ++it) {
            double dn = 0, dd = 0;
            als_apply<K>(Ap, pv, p, beg, n, regada, lane, nullptr, &dn, &dd, false);
            const float a = rs_old / (adot<K>(Ap, pv) + p.eps);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                x.v[k] += a * pv.v[k];
                r.v[k] -= a * Ap.v[k];
            }
            const float rs_new = adot<K>(r, r);
            if (rs_new < p.cg_tol) break;
            const float beta = rs_new / (rs_old + p.eps);
#pragma unroll
            for (int k = 0; k < K; ++k) pv.v[k] = r.v[k] + beta * pv.v[k];
            rs_old = rs_new;
        }
        astore<K>(x, Pu, lane, vdim);
    }
    if (p.compute_loss && lane == 0) {
        if (nume != 0.0) atomicAdd(p.loss, nume);
        if (deno != 0.0) atomicAdd(p.loss + 1, deno);
    }
}

// ------------------------------------------------------------------------------------------------
// llt / ldlt row update: explicit normal equations in LDS + Cholesky by one wave
// (als.cc:180-204 + algo.cc:52-57).  Used for d < 128 only (d >= 128 is forced to iALS++, Q-13),
// so A (vdim x vdim, padded stride) fits LDS: 96*97*4 = 37 KB.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(64) void als_chol_kernel(AlsParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int vdim = p.vdim, D = p.d;
    const int ld = vdim + 1;
    float* A = lds;            // [vdim][ld]
    float* z = lds + vdim * ld;  // [vdim]
    double nume = 0.0, deno = 0.0;
    const int nrows = p.next_x - p.start_x;
    for (int i = next_row(p.ticket, lane); i < nrows; i = next_row(p.ticket, lane)) {
        const int u = p.start_x + i;
        const int64_t beg = (u == 0 ? 0 : p.indptr[u - 1]) - p.shift;
        const int64_t n = p.indptr[u] - p.shift - beg;
        if (n == 0) continue;
        float* Pu = p.P + static_cast<size_t>(u) * vdim;
        ARow<K> x, y;
        aload<K>(x, Pu, lane, vdim);
        const float ada = p.adaptive_reg ? static_cast<float>(n) : 1.0f;
        __syncthreads();
        // A = 0 (the alpha-scaled sum is built first, FF and the ridge are added afterwards: als.cc:194-202)
        for (int e = lane; e < vdim * ld; e += 64) A[e] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) y.v[k] = 0.f;
        if (p.compute_loss && p.axis == 1) {
            ARow<K> t;
            avecmat<K>(t, x, p.FF, D, vdim, lane);
            nume += static_cast<double>(adot<K>(x, t));
            deno += static_cast<double>(p.op_rows);
        }
        __syncthreads();
        for (int64_t k0 = 0; k0 < n; k0 += 64) {
            const int64_t kk = k0 + lane;
            int myc = 0;
            float myv = 0.f;
            if (kk < n) {
                myc = p.keys[beg + kk];
                myv = p.vals[beg + kk];
            }
            const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
            for (int j = 0; j < nh; ++j) {
                const int c = __builtin_amdgcn_readlane(myc, j);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), j));
                ARow<K> q;
                aload<K>(q, p.Q + static_cast<size_t>(c) * vdim, lane, vdim);
                const float coef = static_cast<float>(1.0 + static_cast<double>(v * p.alpha));
#pragma unroll
                for (int k = 0; k < K; ++k) y.v[k] += q.v[k] * coef;
                // rank-1 update: A[a][e] += (v*q_a) * q_e ; each lane owns columns e = k*64+lane
                for (int a = 0; a < D; ++a) {
                    const float va = v * aget<K>(q, a);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int e = k * 64 + lane;
                        if (e < vdim) A[a * ld + e] += va * q.v[k];
                    }
                }
                if (p.compute_loss && p.axis == 1) {
                    const float dot = adot<K>(x, q);
                    nume -= static_cast<double>(dot * dot);
                    nume += static_cast<double>((dot - 1) * (dot - 1)) * (1.0 + static_cast<double>(v * p.alpha));
                    deno += static_cast<double>(v * p.alpha);
                }
            }
        }
        if (p.compute_loss) nume += static_cast<double>(ada * p.reg * adot<K>(x, x));
        // m = FF + FiF*alpha ; m(d,d) += reg*ada
        for (int a = 0; a < D; ++a) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int e = k * 64 + lane;
                if (e < D) {
                    float m = p.FF[static_cast<size_t>(a) * vdim + e] + A[a * ld + e] * p.alpha;
                    if (a == e) m += p.reg * ada;
                    A[a * ld + e] = m;
                }
            }
        }
        // z = y
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = k * 64 + lane;
            if (e < vdim) z[e] = y.v[k];
        }
        __syncthreads();
        // in-place Cholesky (lower), one wave, lanes over rows
        for (int j = 0; j < D; ++j) {
            const float ljj = sqrtf(A[j * ld + j]);
            __syncthreads();
            for (int r = j + lane; r < D; r += 64) A[r * ld + j] = (r == j) ? ljj : A[r * ld + j] / ljj;
            __syncthreads();
            // trailing update of the lower triangle: A[r][c] -= L[r][j]*L[c][j],  j < c <= r
            for (int r = j + 1 + lane; r < D; r += 64) {
                const float lrj = A[r * ld + j];
                for (int c2 = j + 1; c2 <= r; ++c2) A[r * ld + c2] -= lrj * A[c2 * ld + j];
            }
            __syncthreads();
        }
        // forward substitution L w = y
        for (int r = 0; r < D; ++r) {
            float s = 0.f;
            for (int c2 = lane; c2 < r; c2 += 64) s += A[r * ld + c2] * z[c2];
            s = wave_sum(s);
            __syncthreads();
            if (lane == 0) z[r] = (z[r] - s) / A[r * ld + r];
            __syncthreads();
        }
        // back substitution L^T x = w
        for (int r = D - 1; r >= 0; --r) {
            float s = 0.f;
            for (int c2 = r + 1 + lane; c2 < D; c2 += 64) s += A[c2 * ld + r] * z[c2];
            s = wave_sum(s);
            __syncthreads();
            if (lane == 0) z[r] = (z[r] - s) / A[r * ld + r];
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = k * 64 + lane;
            if (e < D) Pu[e] = z[e];
        }
    }
    if (p.compute_loss && lane == 0) {
        if (nume != 0.0) atomicAdd(p.loss, nume);
        if (deno != 0.0) atomicAdd(p.loss + 1, deno);
    }
}

// ------------------------------------------------------------------------------------------------
// iALS++ (als.cc:211-358, Q-14): per row, Yui = P_u . Q_c for every nnz, then for each block of
// `block_size` latent dims: gradient b, 3 CG steps on A_blk + sum alpha*v q_blk q_blk^T (matrix
// free), p_blk -= x, Yui -= q_blk . x.  Block vectors: element j lives in lane j%64, dword j/64.
// ------------------------------------------------------------------------------------------------
template <int K, int KB>
__global__ __launch_bounds__(256) void als_ialspp_kernel(AlsParams p) {
    const int lane = threadIdx.x & 63;
    const int vdim = p.vdim, D = p.d;
    double nume = 0.0, deno = 0.0;
    const int nrows = p.next_x - p.start_x;
    const int bs0 = p.block_size < D ? p.block_size : D;
    for (int i = next_row(p.ticket, lane); i < nrows; i = next_row(p.ticket, lane)) {
        const int u = p.start_x + i;
        const int64_t beg = (u == 0 ? 0 : p.indptr[u - 1]) - p.shift;
        const int64_t n = p.indptr[u] - p.shift - beg;
        if (n == 0) continue;
        float* Pu = p.P + static_cast<size_t>(u) * vdim;
        float* Y = p.yui + beg;
        ARow<K> x;
        aload<K>(x, Pu, lane, vdim);
        const float ada = p.adaptive_reg ? static_cast<float>(n) : 1.0f;
        if (p.compute_loss && p.axis == 1) {  // als.cc:288-291
            ARow<K> t;
            avecmat<K>(t, x, p.FF, D, vdim, lane);
            nume += static_cast<double>(adot<K>(x, t));
            deno += static_cast<double>(p.op_rows);
        }
        // ---- Yui (als.cc:256-266) + positive-sample loss terms (als.cc:298-303) ----
        for (int64_t k0 = 0; k0 < n; k0 += 64) {
            const int64_t kk = k0 + lane;
            int myc = 0;
            float myv = 0.f, myy = 0.f;
            if (kk < n) {
                myc = p.keys[beg + kk];
                myv = p.vals[beg + kk];
            }
            const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
            for (int j = 0; j < nh; ++j) {
                const int c = __builtin_amdgcn_readlane(myc, j);
                ARow<K> q;
                aload<K>(q, p.Q + static_cast<size_t>(c) * vdim, lane, vdim);
                const float dot = adot<K>(x, q);
                if (lane == j) myy = dot;
                if (p.compute_loss && p.axis == 1) {
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), j));
                    nume -= static_cast<double>(dot * dot);
                    nume += static_cast<double>((dot - 1) * (dot - 1)) * (1.0 + static_cast<double>(v * p.alpha));
                    deno += static_cast<double>(v * p.alpha);
                }
            }
            if (kk < n) Y[kk] = myy;
        }
        if (p.compute_loss) nume += static_cast<double>(ada * p.reg * adot<K>(x, x));  // als.cc:306-309

        for (int bb = 0; bb < D; bb += bs0) {
            int bs = bs0;
            if (bb + bs >= D) bs = D - bb;
            // block-layout vectors
            ARow<KB> b, xs, r, pv, Ap, pblk;
            bool act[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                act[k] = (k * 64 + lane) < bs;
                pblk.v[k] = act[k] ? Pu[bb + k * 64 + lane] : 0.f;
            }
            aload<K>(x, Pu, lane, vdim);  // p: the whole row at block start (als.cc:285)
            // b = p * FF[:, blk] + reg * p_blk   (als.cc:286)
#pragma unroll
            for (int k = 0; k < KB; ++k) b.v[k] = 0.f;
            for (int dd = 0; dd < D; ++dd) {
                const float pd = aget<K>(x, dd);
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    if (act[k]) b.v[k] += pd * p.FF[static_cast<size_t>(dd) * vdim + bb + k * 64 + lane];
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) b.v[k] += p.reg * pblk.v[k];
            // b += (Yui - 1) * v * alpha * q_blk   (als.cc:293-297)
            for (int64_t k0 = 0; k0 < n; k0 += 64) {
                const int64_t kk = k0 + lane;
                int myc = 0;
                float myv = 0.f, myy = 0.f;
                if (kk < n) {
                    myc = p.keys[beg + kk];
                    myv = p.vals[beg + kk];
                    myy = Y[kk];
                }
                const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
                for (int j = 0; j < nh; ++j) {
                    const int c = __builtin_amdgcn_readlane(myc, j);
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), j));
                    const float yv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myy), j));
                    const float coef = (yv - 1.0f) * v * p.alpha;
                    const float* qb = p.Q + static_cast<size_t>(c) * vdim + bb;
#pragma unroll
                    for (int k = 0; k < KB; ++k)
                        if (act[k]) b.v[k] += coef * qb[k * 64 + lane];
                }
            }
            // ---- CG: 3 hard-coded steps, plain reg, no eps, rs in double (als.cc:313-345) ----
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                xs.v[k] = 0.f;
                r.v[k] = b.v[k];
                pv.v[k] = b.v[k];
            }
            double rsold = static_cast<double>(adot<KB>(r, r));
            if (rsold > static_cast<double>(p.cg_tol)) {
                for (int step = 0; step < 3; ++step) {
                    // Ap = (FF[blk,blk] + reg I) * pv
#pragma unroll
                    for (int k = 0; k < KB; ++k) Ap.v[k] = p.reg * pv.v[k];
                    for (int c2 = 0; c2 < bs; ++c2) {
                        const float pc = aget<KB>(pv, c2);
#pragma unroll
                        for (int k = 0; k < KB; ++k)
                            if (act[k]) Ap.v[k] += pc * p.FF[static_cast<size_t>(bb + c2) * vdim + bb + k * 64 + lane];
                    }
                    for (int64_t k0 = 0; k0 < n; k0 += 64) {
                        const int64_t kk = k0 + lane;
                        int myc = 0;
                        float myv = 0.f;
                        if (kk < n) {
                            myc = p.keys[beg + kk];
                            myv = p.vals[beg + kk];
                        }
                        const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
                        for (int j = 0; j < nh; ++j) {
                            const int c = __builtin_amdgcn_readlane(myc, j);
                            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), j));
                            const float* qb = p.Q + static_cast<size_t>(c) * vdim + bb;
                            ARow<KB> q;
#pragma unroll
                            for (int k = 0; k < KB; ++k) q.v[k] = act[k] ? qb[k * 64 + lane] : 0.f;
                            const float coef = v * p.alpha * adot<KB>(q, pv);
#pragma unroll
                            for (int k = 0; k < KB; ++k) Ap.v[k] += coef * q.v[k];
                        }
                    }
                    const float step_size = static_cast<float>(rsold / static_cast<double>(adot<KB>(pv, Ap)));
#pragma unroll
                    for (int k = 0; k < KB; ++k) {
                        xs.v[k] += step_size * pv.v[k];
                        r.v[k] -= step_size * Ap.v[k];
                    }
                    const double rsnew = static_cast<double>(adot<KB>(r, r));
                    if (rsnew < static_cast<double>(p.cg_tol)) break;
                    const float ratio = static_cast<float>(rsnew / rsold);
#pragma unroll
                    for (int k = 0; k < KB; ++k) pv.v[k] = r.v[k] + ratio * pv.v[k];
                    rsold = rsnew;
                }
            }
            // p_blk -= x   (als.cc:346)
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (act[k]) Pu[bb + k * 64 + lane] = pblk.v[k] - xs.v[k];
            // Yui -= q_blk . x   (als.cc:347-350)
            for (int64_t k0 = 0; k0 < n; k0 += 64) {
                const int64_t kk = k0 + lane;
                int myc = 0;
                float myy = 0.f;
                if (kk < n) {
                    myc = p.keys[beg + kk];
                    myy = Y[kk];
                }
                const int nh = static_cast<int>((n - k0) < 64 ? (n - k0) : 64);
                for (int j = 0; j < nh; ++j) {
                    const int c = __builtin_amdgcn_readlane(myc, j);
                    const float* qb = p.Q + static_cast<size_t>(c) * vdim + bb;
                    ARow<KB> q;
#pragma unroll
                    for (int k = 0; k < KB; ++k) q.v[k] = act[k] ? qb[k * 64 + lane] : 0.f;
                    const float dx = adot<KB>(q, xs);
                    if (lane == j) myy -= dx;
                }
                if (kk < n) Y[kk] = myy;
            }
        }
    }
    if (p.compute_loss && lane == 0) {
        if (nume != 0.0) atomicAdd(p.loss, nume);
        if (deno != 0.0) atomicAdd(p.loss + 1, deno);
    }
}

// ================================================================================================
// Gramian-on-MFMA path (vdim <= 128: every d <= 128, i.e. BASELINE config #3 and all d < 128 solvers)
//
// One WAVE owns a row (or a 4096-nnz chunk of a heavy row).  ONE pass over the row's nnz builds, on the
// matrix cores (v_mfma_f32_32x32x2_f32, exact fp32), the upper triangle of
//     G = alpha * sum_k v_k q_k q_k^T   (vdim x vdim: T(T+1)/2 tiles of 32x32, 16 accumulator registers each)
//     g = sum_k coef_k q_k              (coef = alpha*v for iALS++, 1 + alpha*v for the dense solvers)
// with q rows streamed straight from HBM/L2 into the MFMA operand layout (lane = (k&1)*32 + i reads
// q_k[32*b + i]: a 128-byte segment per half-wave, no LDS staging, no transposition) -- als_gram_kernel.
//   * iALS++ (als.cc:269-352), block_size 32, d % 32 == 0 (the d >= 128 default, BASELINE config #3):
//     the reference tracks Yui_k = p.q_k incrementally over 5 passes of the nnz per block; with
//     M = FF + G in hand, sum_k alpha v_k (Yui_k - 1) q_k == G p - g, so the block gradient is
//     (M p)_blk - g_blk + reg p_blk and the CG matrix M[blk,blk] + reg I -- the same recurrence, evaluated
//     from the accumulator registers without M ever leaving the wave (als_ialspp_inreg);
//   * everything else (manual_cg / llt / ldlt, als.cc:180-204 + algo.cc:52-82: A = M + reg*ada*I; iALS++
//     with other block sizes): the tiles go to an HBM scratch slot and als_solve_kernel runs the dense
//     algebra from LDS (als_dense_solve).
// Heavy rows (item "Star Wars" has 10^5 users) are cut into chunks whose partial G/g are combined
// with fp32 atomics in a scratch slot and solved by als_solve_kernel.
// ================================================================================================
struct AlsWork {
    int row;
    int kbeg, kend;  // chunk-local nnz range
    int slot;        // -1: whole row in this item (solve in place); >= 0: partial of heavy row `slot`
};

constexpr int ALS_LD_PAD = 1;

// the dense phase runs on ONE wave whose lanes exchange data through LDS: order the DS traffic
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dense phase, executed by ONE wave on LDS-resident M (ld = vdim+1), g, p.  `mode`: 8 ialspp, 2 manual_cg, 0/1 cholesky
__device__ __forceinline__ void als_dense_solve(float* M, float* gv, float* pv_lds, const float* p0, const float* f0, float* w0, float* w1,
                                                float* w2, float* w3, float* w4, const AlsParams& p, int lane, float regada, int mode) {
    const int D = p.d, ld = p.vdim + ALS_LD_PAD;
    if (mode == 8) {
        const int bs0 = p.block_size < D ? p.block_size : D;
        if (bs0 <= 32) {
            // fast path (default block_size 32): lane = (i, h) -- row i of the block, h = column parity.
            // x, r, pv live in registers; only pv is exchanged through LDS.  delta = p - p0 is non-zero
            // only in the blocks already solved, so the gradient touches bb columns, not D.
            const int i = lane & 31, h = lane >> 5;
            float* pvs = w3;
            for (int bb = 0; bb < D; bb += bs0) {
                int bs = bs0;
                if (bb + bs >= D) bs = D - bb;
                const bool act = i < bs;
                const float* Mi = M + (bb + (act ? i : 0)) * ld;
                float sum = 0.f;
#pragma unroll 8
                for (int j = h; j < bb; j += 2) sum += Mi[j] * (pv_lds[j] - p0[j]);
                sum += __shfl_xor(sum, 32, 64);
                float bi = act ? sum + f0[bb + i] + p.reg * pv_lds[bb + i] + gv[bb + i] : 0.f;
                float xr = 0.f, rr = bi, pvr = bi;
                double rsold = static_cast<double>(wave_sum(h == 0 ? rr * rr : 0.f));
                if (rsold > static_cast<double>(p.cg_tol)) {
                    for (int step = 0; step < 3; ++step) {
                        wave_lds_sync();
                        if (h == 0 && act) pvs[i] = pvr;
                        wave_lds_sync();
                        float ap = 0.f;
#pragma unroll 8
                        for (int j = h; j < bs; j += 2) ap += Mi[bb + j] * pvs[j];
                        ap += __shfl_xor(ap, 32, 64);
                        ap = act ? ap + p.reg * pvr : 0.f;
                        const float pap = wave_sum(h == 0 ? pvr * ap : 0.f);
                        const float step_size = static_cast<float>(rsold / static_cast<double>(pap));
                        xr += step_size * pvr;
                        rr -= step_size * ap;
                        const double rsnew = static_cast<double>(wave_sum(h == 0 ? rr * rr : 0.f));
                        if (rsnew < static_cast<double>(p.cg_tol)) break;
                        pvr = rr + static_cast<float>(rsnew / rsold) * pvr;
                        rsold = rsnew;
                    }
                }
                wave_lds_sync();
                if (h == 0 && act) pv_lds[bb + i] -= xr;
                wave_lds_sync();
            }
            return;
        }
        for (int bb = 0; bb < D; bb += bs0) {
            int bs = bs0;
            if (bb + bs >= D) bs = D - bb;
            float* b = w0; float* x = w1; float* r = w2; float* pv = w3; float* ap = w4;
            // b = M[blk,:] (p - p0) + (FF p0)_blk + reg p_blk + r0_blk   -- delta form: no G p - g cancellation
            float rs = 0.f;
            for (int i = lane; i < bs; i += 64) {
                const float* Mi = M + (bb + i) * ld;
                float sum = 0.f;
#pragma unroll 8
                for (int j = 0; j < bb; ++j) sum += Mi[j] * (pv_lds[j] - p0[j]);
                sum += f0[bb + i] + p.reg * pv_lds[bb + i] + gv[bb + i];
                b[i] = sum; r[i] = sum; pv[i] = sum; x[i] = 0.f;
                rs += sum * sum;
            }
            (void)b;
            double rsold = static_cast<double>(wave_sum(rs));
            wave_lds_sync();
            if (rsold > static_cast<double>(p.cg_tol)) {
                for (int step = 0; step < 3; ++step) {
                    float pap = 0.f;
                    for (int i = lane; i < bs; i += 64) {
                        const float* Mi = M + (bb + i) * ld + bb;
                        float sum = p.reg * pv[i];
#pragma unroll 8
                        for (int j = 0; j < bs; ++j) sum += Mi[j] * pv[j];
                        ap[i] = sum;
                        pap += pv[i] * sum;
                    }
                    pap = wave_sum(pap);
                    const float step_size = static_cast<float>(rsold / static_cast<double>(pap));
                    float rn = 0.f;
                    for (int i = lane; i < bs; i += 64) {
                        x[i] += step_size * pv[i];
                        const float ri = r[i] - step_size * ap[i];
                        r[i] = ri;
                        rn += ri * ri;
                    }
                    const double rsnew = static_cast<double>(wave_sum(rn));
                    if (rsnew < static_cast<double>(p.cg_tol)) break;
                    const float ratio = static_cast<float>(rsnew / rsold);
                    wave_lds_sync();
                    for (int i = lane; i < bs; i += 64) pv[i] = r[i] + ratio * pv[i];
                    wave_lds_sync();
                    rsold = rsnew;
                }
            }
            for (int i = lane; i < bs; i += 64) pv_lds[bb + i] -= x[i];
            wave_lds_sync();
        }
        return;
    }
    // explicit system A = M + regada I (in place)
    for (int i = lane; i < D; i += 64) M[i * ld + i] += regada;
    wave_lds_sync();
    if (mode == 2) {  // manual_cg, algo.cc:58-82 (Q-17); x = pv_lds, y = gv
        float* r = w0; float* q = w1; float* Ap = w2;
        auto matvec = [&](const float* v, float* out) {  // out = v * A (A symmetric)
            for (int i = lane; i < D; i += 64) {
                const float* Ai = M + i * ld;
                float sum = 0.f;
#pragma unroll 8
                for (int j = 0; j < D; ++j) sum += Ai[j] * v[j];
                out[i] = sum;
            }
            wave_lds_sync();
        };
        matvec(pv_lds, Ap);
        float yy = 0.f, rr = 0.f;
        for (int i = lane; i < D; i += 64) {
            const float ri = gv[i] - Ap[i];
            r[i] = ri;
            yy += gv[i] * gv[i];
            rr += ri * ri;
        }
        yy = wave_sum(yy);
        rr = wave_sum(rr);
        if (yy < rr) {
            for (int i = lane; i < D; i += 64) { pv_lds[i] = 0.f; r[i] = gv[i]; }
            rr = yy;
        }
        for (int i = lane; i < D; i += 64) q[i] = r[i];
        wave_lds_sync();
        float rs_old = rr;
        for (int it = 0; it < p.num_cg_max_iters; ++it) {
            matvec(q, Ap);
            float pap = 0.f;
            for (int i = lane; i < D; i += 64) pap += Ap[i] * q[i];
            pap = wave_sum(pap);
            const float a = rs_old / (pap + p.eps);
            float rn = 0.f;
            for (int i = lane; i < D; i += 64) {
                pv_lds[i] += a * q[i];
                const float ri = r[i] - a * Ap[i];
                r[i] = ri;
                rn += ri * ri;
            }
            const float rs_new = wave_sum(rn);
            if (rs_new < p.cg_tol) break;
            const float beta = rs_new / (rs_old + p.eps);
            for (int i = lane; i < D; i += 64) q[i] = r[i] + beta * q[i];
            wave_lds_sync();
            rs_old = rs_new;
        }
        return;
    }
    // llt / ldlt: in-place Cholesky of A, then two triangular solves on z = y
    float* z = w0;
    for (int i = lane; i < D; i += 64) z[i] = gv[i];
    wave_lds_sync();
    for (int j = 0; j < D; ++j) {
        const float ljj = sqrtf(M[j * ld + j]);
        wave_lds_sync();
        for (int r = j + lane; r < D; r += 64) M[r * ld + j] = (r == j) ? ljj : M[r * ld + j] / ljj;
        wave_lds_sync();
        for (int r = j + 1 + lane; r < D; r += 64) {
            const float lrj = M[r * ld + j];
            for (int c2 = j + 1; c2 <= r; ++c2) M[r * ld + c2] -= lrj * M[c2 * ld + j];
        }
        wave_lds_sync();
    }
    for (int r = 0; r < D; ++r) {
        float sum = 0.f;
        for (int c2 = lane; c2 < r; c2 += 64) sum += M[r * ld + c2] * z[c2];
        sum = wave_sum(sum);
        wave_lds_sync();
        if (lane == 0) z[r] = (z[r] - sum) / M[r * ld + r];
        wave_lds_sync();
    }
    for (int r = D - 1; r >= 0; --r) {
        float sum = 0.f;
        for (int c2 = r + 1 + lane; c2 < D; c2 += 64) sum += M[c2 * ld + r] * z[c2];
        sum = wave_sum(sum);
        wave_lds_sync();
        if (lane == 0) z[r] = (z[r] - sum) / M[r * ld + r];
        wave_lds_sync();
    }
    for (int i = lane; i < D; i += 64) pv_lds[i] = z[i];
    wave_lds_sync();
}

// LDS carve: M[vdim][vdim+1] | g[vdim] | p[vdim] | 4 work vectors[vdim]
__host__ __device__ inline size_t als_gs_lds_bytes(int vdim) { return (static_cast<size_t>(vdim) * (vdim + ALS_LD_PAD) + 9 * vdim + 4) * sizeof(float); }

// floats per scratch slot: G [vdim][vdim] | g [vdim] | g1 = sum q (loss only) [vdim]
__host__ __device__ inline size_t als_slot_floats(int vdim) { return static_cast<size_t>(vdim) * vdim + 2 * static_cast<size_t>(vdim); }

__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// row-level loss terms by one wave: reg*ada*|p|^2 (both half-epochs, als.cc:306-309) and, on the item
// half-epoch, p^T M p - 2 p.(g_w + g_1) (see als_gram_kernel) + one count of the other side's rows.
// Mp = M p; gsum_a - gsum_b = g_w + g_1 (gsum_b == nullptr: gsum_a alone).
__device__ __forceinline__ void als_row_loss(const AlsParams& p, const float* pl, const float* Mp, const float* gsum_a, const float* gsum_b, float ada,
                                             int lane, double& nume, double& deno) {
    float pp = 0.f, pmp = 0.f, pg = 0.f;
    for (int i = lane; i < p.vdim; i += 64) {
        pp += pl[i] * pl[i];
        if (p.axis == 1) {
            pmp += pl[i] * Mp[i];
            pg += pl[i] * (gsum_b ? gsum_a[i] - gsum_b[i] : gsum_a[i]);
        }
    }
    pp = wave_sum(pp);
    nume += static_cast<double>(ada * p.reg * pp);
    if (p.axis == 1) {
        pmp = wave_sum(pmp);
        pg = wave_sum(pg);
        nume += static_cast<double>(pmp) - 2.0 * static_cast<double>(pg);
        deno += static_cast<double>(p.op_rows);
    }
}

// ------------------------------------------------------------------------------------------------
// In-register iALS++ (als.cc:269-352) for the wave-per-row Gramian kernel: when the pass over the nnz
// ends, the wave holds the upper triangle of M = FF + G in its accumulators (they were initialised with
// the FF tiles) in the MFMA C layout  tile[e] @ lane (half, col) = M[32a + r(e) + 4 half][32b + col],
// r(e) = (e&3) + 8(e>>2).  Both products the block recurrence needs come straight out of that layout:
//   * "column" product  y[col] = sum_r tile[r][col] x[r]   -- 16 FMAs per lane against an LDS-broadcast
//     x, then one cross-half add: used for tiles above the diagonal block (M[blk][j<blk] = M[j][blk]^T)
//     and for the symmetric diagonal tile (the CG matrix);
//   * "row" product     y[r]   = sum_col tile[r][col] x[col] -- 16 per-lane products, then a transpose-
//     reduce over the 32 lanes of each half (16+8+4+2+1 shuffles) that leaves row (lane>>1)&15 in each
//     lane pair: used for the tiles right of the diagonal block.
// So M never goes to LDS or HBM: no scratch round trip, no second kernel, and the solve of one wave
// overlaps the Gramian passes of the other waves on the CU.  Needs block_size == 32 and d % 32 == 0.
// ------------------------------------------------------------------------------------------------
template <int T>
__host__ __device__ constexpr int als_tri(int a, int b) { return a * T - a * (a - 1) / 2 + (b - a); }   // index of tile (a, b), a <= b

__device__ __forceinline__ float als_tile_colpart(const f32x16& t, const float* x /* LDS, 32 floats, 16-B aligned */, int half) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(x + 8 * k + 4 * half);
        s += t[4 * k + 0] * v.x + t[4 * k + 1] * v.y + t[4 * k + 2] * v.z + t[4 * k + 3] * v.w;
    }
    return s;   // caller adds the other half's share (rows 4..7 mod 8)
}

// z[e] = per-lane products; leaves sum over the 32 lanes of this half of row (lane>>1)&15 (in e-numbering) in every lane
__device__ __forceinline__ float als_rows_reduce(float (&z)[16], int lane) {
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float keep = b4 ? z[e + 8] : z[e], send = b4 ? z[e] : z[e + 8];
        z[e] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float keep = b3 ? z[e + 4] : z[e], send = b3 ? z[e] : z[e + 4];
        z[e] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float keep = b2 ? z[e + 2] : z[e], send = b2 ? z[e] : z[e + 2];
        z[e] = keep + __shfl_xor(send, 4, 64);
    }
    {
        const float keep = b1 ? z[1] : z[0], send = b1 ? z[0] : z[1];
        z[0] = keep + __shfl_xor(send, 2, 64);
    }
    return z[0] + __shfl_xor(z[0], 1, 64);
}

// two floats -> two packed f16 pairs h, l: h = x rounded to nearest f16 (11 bits, error <= 2^-12 |x|), l = the (exact fp32) remainder
// rounded the same way (error <= 2^-12 |remainder| <= 2^-24 |x|): h + l carries x to fp32's own precision.  gfx950's
// v_cvt_pk_f16_f32 rounds a pair per instruction.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void als_split_f16(float x0, float x1, unsigned& h, unsigned& l) {
    const f32x2_t x = {x0, x1};
    const f16x2_t hh = __builtin_convertvector(x, f16x2_t);
    const f32x2_t r = {x0 - static_cast<float>(hh[0]), x1 - static_cast<float>(hh[1])};
    const f16x2_t ll = __builtin_convertvector(r, f16x2_t);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// The same cut with the scaling folded in (als_pc.hpp's pc_split_pair, als_wide_kernel<SPLIT>): (q0 s0, q1 s1) -> packed f16 pairs h (the
// product rounded to nearest ONCE, inside the fused operation) and l (the remainder q s - h, exact in the fused multiply-add, rounded the same
// way) -- four v_fma_mix{lo,hi}_f16 per pair where the form above takes seven instructions with its multiplies.
__device__ __forceinline__ void als_split_pair_mix(float q0, float s0, float q1, float s1, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(q0), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(q1), "v"(s1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(q0), "v"(s0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(q1), "v"(s1), "v"(h));
}

// Scale of the split pass (als_gram_kernel<SPLIT>), decided on the device from ONE fixed-order reduction, qmax = max |Q| over the
// other factor matrix -- nothing that depends on how the rows are chunked or sharded, so every chunking and every rank of a
// sharded run works with the same numbers:
//   S    = the power of two that puts S qmax in [2^6, 2^7].  An entry's x = S sqrt(alpha v) q then stays below f16's 65504 for
//          weights alpha v up to 2^16 / 2, and keeps a full 11-bit remainder (|x| >= 2^-3) down to |q| ~ 1e-3 qmax at weight 1;
//   wcut = 2^15: heavier entries (and negative ones) go through the fp32 instruction instead.
// `part`: [ALS_STAT_BLOCKS] floats written by the first kernel.
constexpr int ALS_STAT_BLOCKS = 1024;
__global__ __launch_bounds__(256) void als_split_stats_kernel(const float* __restrict__ Q, size_t nq, float* __restrict__ part) {
    __shared__ float smq[256];
    float qm = 0.f;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const float4* Q4 = reinterpret_cast<const float4*>(Q);   // hipMalloc'd, vdim % 32 == 0
    for (size_t e = tid; e < nq / 4; e += stride) {
        const float4 v = Q4[e];
        qm = fmaxf(qm, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    smq[threadIdx.x] = qm;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) smq[threadIdx.x] = fmaxf(smq[threadIdx.x], smq[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = smq[0];
}
__global__ __launch_bounds__(64) void als_split_scale_kernel(const float* __restrict__ part, int nblocks, float wcut, float* __restrict__ out) {
    float qm = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 64) qm = fmaxf(qm, part[b]);
    for (int st = 32; st > 0; st >>= 1) qm = fmaxf(qm, __shfl_xor(qm, st, 64));
    if (threadIdx.x != 0) return;
    int e = 0;
    if (qm > 0.f && qm < 3e38f) {   // a max is the same number in any order: the same S on every run and every rank
        e = static_cast<int>(floor(log2(128.0 / static_cast<double>(qm))));
        if (e > 40) e = 40;
        if (e < -40) e = -40;
    }
    out[0] = static_cast<float>(ldexp(1.0, e));
    out[1] = static_cast<float>(ldexp(1.0, 2 * e));
    out[2] = static_cast<float>(ldexp(1.0, -2 * e));
    out[3] = wcut;
}

// sum over the 32 lanes of each half-wave (every lane receives its half's total): 4 DPP row rotations leave each 16-lane row's
// sum in all of its lanes, one ds_swizzle (bit mode, xor 16: no LDS memory, no SGPR round trip) fetches the half's other row
__device__ __forceinline__ float half_sum(float v, int /*half*/) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));               // and 0x1f, or 0, xor 0x10
}

// sum_col a[col] b[col] over the 32 lanes of half 0 for vectors BOTH halves hold alike (the block solve's: lane (half, col) carries element col), handed
// to every lane: the bits of wave_sum(half == 0 ? a * b : 0) -- (r0 + r1) + (0 + 0) -- without the select, two of the four readlanes and two adds
// (28 such sums per row: the block solve is issue-bound).  The product is rounded BEFORE the first addition, as the select made it:
// left to -ffp-contract=fast it fuses into the first butterfly step, and that one rounding moved two ill-conditioned parity cases (d = 96 tiny,
// outliers) from 1-2x to 4-10x of the oracle's distance from float64 (GPU call 11) -- three CG steps on blocks conditioned ~1e4 amplify it.
__device__ __forceinline__ float als_dot32_uniform(float a, float b) {
    float v;
    {
#pragma clang fp contract(off)
        v = a * b;
    }
    asm volatile("" : "+v"(v));   // (and the rounded product stays a value of its own)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    return r0 + r1;
}

// (A x)[32 blk + col] for the symmetric matrix whose upper-triangle tiles are `acc` and an LDS vector x (32 T floats);
// `out`: 32 LDS floats of exchange space
template <int T>
__device__ __forceinline__ float als_block_matvec(const f32x16 (&acc)[T * (T + 1) / 2], const float* x, float* out, int blk, int lane, int half, int col) {
    float part = 0.f;
#pragma unroll
    for (int ja = 0; ja < T; ++ja)
        if (ja <= blk) part += als_tile_colpart(acc[als_tri<T>(ja < blk ? ja : blk, blk)], x + ja * 32, half);
    float s = part + __shfl_xor(part, 32, 64);
    if (blk < T - 1) {
        float z[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
        for (int jb = 1; jb < T; ++jb)
            if (jb > blk) {
                const float xv = x[jb * 32 + col];
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] += acc[als_tri<T>(blk < jb ? blk : jb, jb)][e] * xv;
            }
        const float y = als_rows_reduce(z, lane);
        const int es = (lane >> 1) & 15;
        wave_lds_sync();
        if (!(lane & 1)) out[(es & 3) + 8 * (es >> 2) + 4 * half] = y;
        wave_lds_sync();
        s += out[col];
    }
    return s;
}

// `pc`: LDS copy of the row (vdim floats; p0 at entry), `dl`: LDS delta = p - p0 (vdim floats, zero at entry), `tmp`: 64 LDS floats.
// `h` = sum_k alpha v_k (q_k.p0 - 1) q_k (the residual-weighted sum of the reference, als.cc:296, formed per entry from the row AT
// ENTRY), `f0` = FF p0 (taken from the FF tiles before the pass), both as this lane's element [32 blk + col].  The gradient of block
// blk at the current row is then  f0 + h + (M delta) + reg p  -- the reference's recurrence with its tracked Yui written out:
// Yui_k - 1 = (q_k.p0 - 1) + q_k.delta.  (Round 2 evaluated (M p) - g_w instead: the same number, but G p and g_w nearly cancel
// once the model fits -- Yui ~ 1 -- and the rows lost 1-2 digits against the reference path: profiles/r03_als_config3_warm_epoch.txt.)
// On return pc holds the updated row.
template <int T>
__device__ __forceinline__ void als_ialspp_inreg(const f32x16 (&acc)[T * (T + 1) / 2], const float (&h)[T], const float (&g1)[T], const float (&f0)[T],
                                                 const AlsParams& p, float* pc, float* dl, float* tmp, int lane, int half, int col, float ada,
                                                 double& nume, double& deno, float ms = 1.0f /* acc holds M / ms (split pass: ms = 1/S^2) */) {
    float* out = tmp;        // 32: row-product results
    float* pvs = tmp + 32;   // 32: CG direction
    if (p.compute_loss) {   // als.cc:288-309 on the row at entry: with g_w = G p0 - h,
        // p FF p + [p G p - 2 p.(g_w + g_1)] = 2 p.f0 - p M p + 2 p.(h - g_1)
        float pp = 0.f, pmp = 0.f, pg = 0.f;
#pragma unroll
        for (int blk = 0; blk < T; ++blk) {
            const float pv = pc[blk * 32 + col];
            pp += pv * pv;
            if (p.axis == 1) {
                pmp += pv * (2.0f * f0[blk] - ms * als_block_matvec<T>(acc, pc, out, blk, lane, half, col));
                pg += pv * (g1[blk] - h[blk]);
            }
        }
        pp = wave_sum(half == 0 ? pp : 0.f);
        nume += static_cast<double>(ada * p.reg * pp);
        if (p.axis == 1) {
            pmp = wave_sum(half == 0 ? pmp : 0.f);
            pg = wave_sum(half == 0 ? pg : 0.f);
            nume += static_cast<double>(pmp) - 2.0 * static_cast<double>(pg);
            deno += static_cast<double>(p.op_rows);
        }
    }
#pragma unroll
    for (int blk = 0; blk < T; ++blk) {
        const float pblk = pc[blk * 32 + col];
        float md = 0.f;   // (M delta)[32 blk + col]: delta is non-zero only in the blocks already solved (tiles above the diagonal block)
#pragma unroll
        for (int ja = 0; ja < T; ++ja)
            if (ja < blk) md += als_tile_colpart(acc[als_tri<T>(ja, blk)], dl + ja * 32, half);
        md = ms * (md + __shfl_xor(md, 32, 64));
        const float bi = f0[blk] + h[blk] + md + p.reg * pblk;   // als.cc:286-297: gradient of the block at the current row
        float xr = 0.f, rr = bi, pvr = bi;
        // (the reference keeps rsold / rsnew in double; they only ever hold float values, and every use -- the comparisons with the float cg_tol, the
        //  quotients -- gives the same result on the floats themselves: round 6 dropped the conversions and the double compares from the chain)
        float rsold = als_dot32_uniform(rr, rr);
        if (rsold > p.cg_tol) {   // als.cc:313-345: 3 CG steps on M[blk,blk] + reg I
            for (int step = 0; step < 3; ++step) {
                wave_lds_sync();
                if (half == 0) pvs[col] = pvr;
                wave_lds_sync();
                float ap = als_tile_colpart(acc[als_tri<T>(blk, blk)], pvs, half);
                ap = ms * (ap + __shfl_xor(ap, 32, 64));
                ap += p.reg * pvr;
                const float pap = als_dot32_uniform(pvr, ap);
                // als.cc:328: double / float rounded to float.  Both operands hold float values, and a quotient of two floats taken in
                // double and rounded to float IS the correctly rounded float quotient (53 >= 2*24 + 2): one fp32 division, same bits
                const float step_size = rsold / pap;
                xr += step_size * pvr;
                rr -= step_size * ap;
                const float rsnew = als_dot32_uniform(rr, rr);
                if (rsnew < p.cg_tol) break;
                pvr = rr + (rsnew / rsold) * pvr;
                rsold = rsnew;
            }
        }
        wave_lds_sync();
        if (half == 0) {
            pc[blk * 32 + col] = pblk - xr;   // als.cc:346
            dl[blk * 32 + col] = -xr;
        }
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// als_gram_kernel -- ONE WAVE PER ROW, no big LDS.  The wave holds the whole upper triangle of G
// (T(T+1)/2 tiles x 16 accumulator registers: 160 at vdim 128) and streams each q row through the matrix
// cores exactly once:
//     acc[(a,b)] += (alpha v q[block a])^T q[block b]     v_mfma_f32_32x32x2_f32: one instruction eats TWO nnz
//     gpart[a]   += c q[32a + col],  c = alpha v (iALS++: g_w = sum alpha v q) or 1 + alpha v (dense solvers: y)
// The 64 keys/vals of a chunk sit one-per-lane; a pair's two row ids are wave-uniform so they come out
// with v_readlane (SALU), the operand rows are plain dword loads (half-wave = one 128-B line per tile, a
// 32-bit per-lane row offset off one SGPR base), and the loop is register double-buffered: the T loads
// of each of the UP pairs of group j+1 are in flight while group j feeds UP * T(T+1)/2 MFMAs (~2.5k
// cycles at vdim 128) -- enough to hide an Infinity-Cache round trip with 2 waves per SIMD -- and no q
// row is fetched twice.
//
// Loss (als.cc:187-192 / 298-303, item half-epoch only): sum_k [-y_k^2 + (y_k-1)^2 (1+w_k)] with
// y_k = p.q_k, w_k = alpha v_k equals  p^T G p - 2 p.(g_w + g_1) + sum_k (1 + w_k)  (g_1 = sum q), so
// nothing per-nnz is needed beyond one more FMA (g1part, iALS++ only: the dense solvers' y is
// g_w + g_1 already); the constant and the denominator sum_k w_k are added where the vals are read, the
// quadratic form (with p^T FF p, as p^T M p) by whoever holds M.
//
// INREG (iALS++ with block_size 32, d == vdim): the accumulators start from the FF tiles, rows that fit
// one work item are SOLVED IN PLACE (als_ialspp_inreg) and only the chunks of heavy rows go to scratch
// (slot_base = 0).  Otherwise the computed tiles of G, g (and g1) go to the row's HBM scratch slot
// (row - start_x; heavy rows: slot_base + wk.slot, zeroed by the host, fp32 atomics) and als_solve_kernel
// -- a 256-thread block per row -- rebuilds M = FF + G in LDS and runs als_dense_solve.
// ------------------------------------------------------------------------------------------------
// BIG: the other factor matrix is 4 GiB or larger -- 64-bit gather offsets (a few % slower, 19 % in the wide kernel)
// LOSS = false: compile-time promise that no loss terms are wanted (compute_loss_on_training off, or the user half-epoch whose
// per-entry terms are zero): drops the g_1 accumulation -- T FMAs per entry pair and T registers -- from the hot loop
template <int N, int I = 0, class F>
__device__ __forceinline__ void als_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        als_static_for<N, I + 1>(f);
    }
}

// row / column block of upper-triangle tile t (tiles numbered row by row: (0,0) (0,1) .. (0,T-1) (1,1) ..)
template <int T>
__host__ __device__ constexpr int als_tile_row(int t) {
    int a = 0;
    while (t >= T - a) { t -= T - a; ++a; }
    return a;
}
template <int T>
__host__ __device__ constexpr int als_tile_col(int t) {
    int a = 0;
    while (t >= T - a) { t -= T - a; ++a; }
    return a + t;
}

// F0[x - start_x] = P[x] FF for the rows of one call (iALS++'s "FF p0", als.cc:286): formed here at full occupancy -- four rows per
// wave against one pass over FF (L2-resident, a half-wave reads 128 contiguous bytes) -- instead of inside the split pass, whose one
// wave per SIMD pays every dependent instruction in full.  ~2.3 G FMAs for 138,493 rows at vdim 128: tens of microseconds.
template <int T>
__global__ __launch_bounds__(256) void als_rowff_kernel(const float* __restrict__ P, int start_x, int nrows, const float* __restrict__ FF, float* __restrict__ F0) {
    constexpr int VD = 32 * T;
    __shared__ __attribute__((aligned(16))) float s_rows[4][VD * 4];   // per wave: four rows, transposed [j][row]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    float* sp = s_rows[wave];
    const int nquads = (nrows + 3) / 4;
    for (int qd = blockIdx.x * 4 + wave; qd < nquads; qd += gridDim.x * 4) {
        const int r0 = qd * 4;
        wave_lds_sync();
#pragma unroll
        for (int rw = 0; rw < 4; ++rw) {
            const int row = r0 + rw;
            for (int j = lane; j < VD; j += 64) sp[j * 4 + rw] = row < nrows ? P[static_cast<size_t>(start_x + row) * VD + j] : 0.f;
        }
        wave_lds_sync();
        float acc[4][T];
#pragma unroll
        for (int rw = 0; rw < 4; ++rw)
#pragma unroll
            for (int b = 0; b < T; ++b) acc[rw][b] = 0.f;
        const float* Fl = FF + col;
#pragma unroll 4
        for (int jj = 0; jj < VD / 2; ++jj) {
            const int j = half * (VD / 2) + jj;
            const float4 pv = *reinterpret_cast<const float4*>(sp + j * 4);   // the same address in every lane of the half: a broadcast
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const float f = Fl[static_cast<size_t>(j) * VD + b * 32];
                acc[0][b] = __builtin_fmaf(pv.x, f, acc[0][b]);
                acc[1][b] = __builtin_fmaf(pv.y, f, acc[1][b]);
                acc[2][b] = __builtin_fmaf(pv.z, f, acc[2][b]);
                acc[3][b] = __builtin_fmaf(pv.w, f, acc[3][b]);
            }
        }
#pragma unroll
        for (int rw = 0; rw < 4; ++rw)
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const float v = acc[rw][b] + __shfl_xor(acc[rw][b], 32, 64);
                if (half == 0 && r0 + rw < nrows) F0[static_cast<size_t>(r0 + rw) * VD + b * 32 + col] = v;
            }
    }
}

// SPLIT (iALS++ in place only): the Gramian through the f16 matrix cores at fp32 accuracy.  x = S sqrt(alpha v) q is cut into two
// f16 pieces h + l (each rounded to nearest, the remainder exact in fp32: h + l = x to 2^-24) and x x^T is taken as hh + hl + lh
// (what is dropped is <= 2^-24 of a term; measured against f64 the rows are as close as the fp32 instruction's,
// profiles/r03_als_split_f16.txt):
// products of f16 pairs are exact in fp32 and v_mfma_f32_32x32x16_f16 accumulates in fp32.  One instruction eats SIXTEEN entries
// in 32 cycles where v_mfma_f32_32x32x2_f32 eats two in 64: 3 x 32 / 16 = 6 matrix-core cycles per entry and tile instead of 32.
// A and B of a 32x32x16 step hold, per lane, element [32 a + (lane & 31)] of eight entries (k = 8 (lane >> 5) + r) -- the same
// registers serve as A and B, and since a Gramian sums over k ANY k order is right as long as both operands use the same one.
// S (a power of two, als_split_scale_kernel) keeps S x inside f16's range; the accumulators hold S^2 M (the FF tiles are scaled
// when they are copied to LDS) and every product read back out of them is multiplied by 1/S^2 -- exact, S being a power of two.
// Entries with a negative weight (no square root) or heavier than 2^15 go through the fp32 instruction in a side pass
// per 64-entry chunk.  The pass is VALU-bound now (~230 VALU instructions per 16 entries against 30 matrix instructions), so the loop
// is software-pipelined inside the wave: while the pieces of group j feed the matrix cores, the rows of group j+1 are weighted and
// cut and the rows of group j+2 are on their way (`fused`); at T >= 3 that takes the 512-register file: ONE wave per SIMD, which
// therefore pays every per-row step in full -- hence als_rowff_kernel, the drawn-ahead tickets and the prefetched keys below.
// Counters (profiles/r03_als_split_counters.txt): per wave 47 % of the cycles issue VALU, 32 % wait on memory, the matrix pipe is
// busy 22-32 %.
template <int T, bool IALS, bool INREG, bool BIG, bool LOSS = true, bool SPLIT = false>
__global__ __launch_bounds__(256, (SPLIT && T >= 3) ? 1 : 2) void als_gram_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_items, float* __restrict__ scratch,
                                                          int slot_base) {
    static_assert(!INREG || IALS, "the in-register solve is the iALS++ recurrence");
    static_assert(!SPLIT || INREG, "the split-f16 pass is written for the in-place iALS++ rows");
    __shared__ __attribute__((aligned(16))) float s_vec[INREG ? 4 * (2 * 32 * T + 64) : 4];   // per wave: p | delta | 64 exchange floats
    // the FF tiles every in-place row starts from: one copy per block in LDS (64 KB at vdim 128; two blocks a CU fit the 160 KB),
    // read back conflict-free (a half-wave reads 32 consecutive floats) instead of 40 KB of L2 round trips per row
    __shared__ __attribute__((aligned(16))) float s_ff[INREG ? (32 * T) * (32 * T) : 4];
    constexpr int NT = T * (T + 1) / 2;
    constexpr int UP = 4;
    constexpr unsigned row_bytes = 32u * T * 4u;   // vdim == 32*T
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31;
    // split pass: S, S^2, 1/S^2 and the weight cut (als_split_scale_kernel); 1, 1, 1, +inf otherwise
    const float sS = SPLIT ? p.split[0] : 1.0f, sS2 = SPLIT ? p.split[1] : 1.0f, sI2 = SPLIT ? p.split[2] : 1.0f;
    const float wcut = SPLIT ? p.split[3] : 0.f;
    if (INREG && (SPLIT || !(p.debug & 8))) {
        const float4* src = reinterpret_cast<const float4*>(p.FF);
        float4* dst = reinterpret_cast<float4*>(s_ff);
        for (int e = threadIdx.x; e < (32 * T) * (32 * T) / 4; e += blockDim.x) {
            float4 v = src[e];
            if (SPLIT) { v.x *= sS2; v.y *= sS2; v.z *= sS2; v.w *= sS2; }
            dst[e] = v;
        }
        __syncthreads();
    }
    const bool lossk = LOSS && p.compute_loss && p.axis == 1;
    double nume_k = 0.0, deno_k = 0.0;
    const char* qbase = reinterpret_cast<const char*>(p.Q);
    // SPLIT draws several work items per ticket (same-address atomics serialise at ~12 ns each: 138,493 one-row draws alone are
    // 1.7 ms).  The list is sorted by length, longest first, so a draw made while working on a row of n entries takes
    // min(p.batch, 1024 / n) rows -- none longer than n, about 1024 entries of work at most: single rows at the head of the list
    // (the longest-first balance stays), up to p.batch of the short ones.
    auto ticket = [&](int rows) {
        int it = 0;
        if (lane == 0) it = atomicAdd(p.ticket, rows);
        return it;   // lane 0's value; readfirstlane where it is needed
    };
    // SPLIT runs one wave per SIMD, so nothing else hides a row's start-up chain (ticket -> work item -> keys -> rows): the next row's
    // ticket is drawn when this row starts, its work item is read before the entry loop, its first 128 keys / values and its
    // factors at entry are fetched before the solve -- each has landed by the time the next step needs it.
    int item = __builtin_amdgcn_readfirstlane(ticket(1));
    int batch_end = item + 1;
    AlsWork wk_cur = work[item < n_items ? item : 0];
    bool have_pf = false;
    int pf_c0 = 0, pf_c1 = 0;
    float pf_v0 = 0.f, pf_v1 = 0.f, pf_p0[T], pf_f0[T];
#pragma unroll
    for (int b = 0; b < T; ++b) { pf_p0[b] = 0.f; pf_f0[b] = 0.f; }
    while (item < n_items) {
        const bool draw = item + 1 >= batch_end;
        int rows_nx = 1;
        if (SPLIT) {
            const int64_t len = wk_cur.kend - wk_cur.kbeg;
            const int64_t fit = 1024 / (len > 0 ? len : 1);
            rows_nx = static_cast<int>(fit < 1 ? 1 : (fit > p.batch ? p.batch : fit));
            if (rows_nx < 1) rows_nx = 1;
        }
        const int item_nx_raw = SPLIT ? (draw ? ticket(rows_nx) : item + 1) : 0;
        int item_nx = 0;
        AlsWork wk_nx = wk_cur;
        const AlsWork wk = wk_cur;
        const bool solve_here = INREG && wk.slot < 0;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // "als_debug" ablation bits (timing studies only, results are wrong): 1 no block solve, 2 no FF tiles / FF p0 before the pass,
        // 4 no per-entry residual dot
        if (solve_here && !(p.debug & 2)) {   // M = FF + G: start the accumulators from the FF tiles (64 KB, L2-resident)
            if (!SPLIT && (p.debug & 8)) {   // bit 8: the tiles from L2 as before this round (A/B only, same values)
                const float* Fl = p.FF + half * 4 * (32 * T) + col;
                asm volatile("" : "+v"(Fl));   // keep the 10 tiles' address arithmetic inside the item loop (hoisted, it spills)
                int t = 0;
#pragma unroll
                for (int a = 0; a < T; ++a)
#pragma unroll
                    for (int b = a; b < T; ++b, ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][e] = Fl[(a * 32 + (e & 3) + 8 * (e >> 2)) * (32 * T) + b * 32];
            } else {
                const float* Fl = s_ff + half * 4 * (32 * T) + col;
                int t = 0;
#pragma unroll
                for (int a = 0; a < T; ++a)
#pragma unroll
                    for (int b = a; b < T; ++b, ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][e] = Fl[(a * 32 + (e & 3) + 8 * (e >> 2)) * (32 * T) + b * 32];
            }
        }
        float gpart[T], g1part[T];
#pragma unroll
        for (int a = 0; a < T; ++a) { gpart[a] = 0.f; g1part[a] = 0.f; }
        // iALS++: the row at entry, this lane's elements [32 b + col] -- every entry's residual q_k.p0 - 1 is formed against it
        float p0r[T], f0r[T];
#pragma unroll
        for (int b = 0; b < T; ++b) { p0r[b] = 0.f; f0r[b] = 0.f; }
        constexpr int VD0 = 32 * T;
        float* pc = s_vec + (INREG ? (threadIdx.x >> 6) * (2 * VD0 + 64) : 0);
        if (IALS) {
            if (SPLIT && have_pf) {
#pragma unroll
                for (int b = 0; b < T; ++b) { p0r[b] = pf_p0[b]; f0r[b] = pf_f0[b]; }
            } else {
                const float* Pu0 = p.P + static_cast<size_t>(wk.row) * VD0;
#pragma unroll
                for (int b = 0; b < T; ++b) p0r[b] = Pu0[b * 32 + col];
                if (SPLIT) {   // FF p0 of this row, formed by als_rowff_kernel
                    const float* Fu0 = p.F0 + static_cast<size_t>(wk.row - p.start_x) * VD0;
#pragma unroll
                    for (int b = 0; b < T; ++b) f0r[b] = Fu0[b * 32 + col];
                }
            }
        }
        if (solve_here && !(p.debug & 2)) {   // f0 = FF p0 while the accumulators still hold FF alone
            wave_lds_sync();
            if (half == 0) {
#pragma unroll
                for (int b = 0; b < T; ++b) { pc[b * 32 + col] = p0r[b]; pc[VD0 + b * 32 + col] = 0.f; }
            }
            wave_lds_sync();
            if (!SPLIT) {
#pragma unroll
                for (int b = 0; b < T; ++b) f0r[b] = als_block_matvec<T>(acc, pc, pc + 2 * VD0, b, lane, half, col);
            }
        }
        const int64_t n = wk.kend - wk.kbeg;
        const int64_t nchunks = (n + 63) / 64;
        auto fetch_keys_of = [&](int64_t kbeg, int64_t n, int64_t chunk, int& cc, float& vvv) {
            const int64_t kk = chunk * 64 + lane;
            cc = 0;        // padding lanes: row 0 of the other factor with weight 0
            vvv = 0.f;
            if (kk < n) {
                cc = p.keys[kbeg + kk];
                vvv = p.vals[kbeg + kk];
                if (!IALS && p.ctx) vvv -= p.bias_other[cc];
                if (lossk) {   // constant and denominator of the loss, see als_gram_kernel
                    const double w = static_cast<double>(vvv * p.alpha);
                    deno_k += w;
                    nume_k += 1.0 + w;
                }
            }
        };
        auto fetch_keys = [&](int64_t chunk, int& cc, float& vvv) { fetch_keys_of(wk.kbeg, n, chunk, cc, vvv); };
        auto load_pair = [&](int myc, float myv, int pr, float (&q)[T], float& v) {
            const int c0 = __builtin_amdgcn_readlane(myc, 2 * pr), c1 = __builtin_amdgcn_readlane(myc, 2 * pr + 1);
            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), 2 * pr));
            const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), 2 * pr + 1));
            v = half ? v1 : v0;
            using off_t = typename std::conditional<BIG, size_t, unsigned>::type;   // 32-bit: one SGPR base + a VGPR offset
            const off_t voff = static_cast<off_t>(static_cast<unsigned>(half ? c1 : c0)) * row_bytes + static_cast<unsigned>(col) * 4u;
            const float* q_ = reinterpret_cast<const float*>(qbase + voff);
#pragma unroll
            for (int b = 0; b < T; ++b) q[b] = q_[b * 32];
        };
        const bool ctx = !IALS && p.ctx != 0;
        const float bself = ctx ? p.bias_self[wk.row] : 0.f;
        auto consume = [&](const float (&q)[T], float v, float one) {
            const float wgt = ctx ? one : p.alpha * v;
            const float cdense = ctx ? (v - bself) * one : one + wgt;
            float cial = wgt;
            if (IALS && !(p.debug & 4)) {   // als.cc:292-296: residual = Yui - 1 against the row at entry, coefficient residual * val * alpha
                float part = 0.f;
#pragma unroll
                for (int b = 0; b < T; ++b) part += q[b] * p0r[b];
                cial = wgt * (half_sum(part, half) - 1.0f);
            }
            int t = 0;
#pragma unroll
            for (int a = 0; a < T; ++a) {
                const float av = wgt * q[a];
#pragma unroll
                for (int b = a; b < T; ++b, ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, q[b], acc[t], 0, 0, 0);
                // als.cc:184: float(1.0 + double(v*alpha)) == 1.0f + v*alpha (the exact sum rounded once either way)
                gpart[a] += (IALS ? cial : cdense) * q[a];   // iALS++: h = sum alpha v (q.p0 - 1) q;  dense solvers: y
                if (IALS && LOSS) g1part[a] += (lossk ? one : 0.f) * q[a];   // unconditional at run time: keeps the loop body one basic block
            }
        };
        auto nnz_of = [&](int64_t ch) { return ch < nchunks ? static_cast<int>((n - ch * 64) < 64 ? (n - ch * 64) : 64) : 0; };
        auto groups_of = [&](int64_t ch) { return (nnz_of(ch) >> 1) / UP; };   // groups hold COMPLETE pairs only (no padding lane)
        if constexpr (!SPLIT) {
            int myc, myc_n;
            float myv, myv_n;
            fetch_keys(0, myc, myv);
            fetch_keys(1, myc_n, myv_n);
            float qa[UP][T], va[UP];
            if (groups_of(0) > 0) {
    #pragma unroll
                for (int uu = 0; uu < UP; ++uu) load_pair(myc, myv, uu, qa[uu], va[uu]);
            }
            for (int64_t ch = 0; ch < nchunks; ++ch) {
                const int npairs = (nnz_of(ch) + 1) >> 1;
                const int ngroups = groups_of(ch);
                for (int gidx = 0; gidx < ngroups; ++gidx) {
                    // Branch-free body: the loads of the NEXT group (the following group of this chunk, else the first
                    // group of the next chunk; harmless rows when the item ends here) and the MFMAs of the current one
                    // sit in one basic block, and the scheduler is told to interleave them -- left alone it emits the
                    // VALU/VMEM work as one clump and then UP*NT MFMAs back to back, and since both waves of a SIMD
                    // run the same loop they fall into step: while one clump issues the matrix core idles
                    // (measured: 66 % MFMA-busy with every busy cycle stalling BOTH waves; 8.2 -> 7.7 ms per epoch).
                    float qb[UP][T], vb[UP];
                    const bool here = gidx + 1 < ngroups;
                    const int src_c = here ? myc : myc_n;
                    const float src_v = here ? myv : myv_n;
                    const int pr0 = here ? (gidx + 1) * UP : 0;
    #pragma unroll
                    for (int uu = 0; uu < UP; ++uu) load_pair(src_c, src_v, pr0 + uu, qb[uu], vb[uu]);
    #pragma unroll
                    for (int uu = 0; uu < UP; ++uu) consume(qa[uu], va[uu], 1.0f);
    #pragma unroll
                    for (int uu = 0; uu < UP; ++uu) {
                        va[uu] = vb[uu];
    #pragma unroll
                        for (int b = 0; b < T; ++b) qa[uu][b] = qb[uu][b];
                    }
    #pragma unroll
                    for (int i = 0; i < UP * NT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x006, IALS ? 3 : 2, 0);                          // two VALU / SALU (three with the per-entry residual dot)
                        if (i % 2 == 0 && i / 2 < UP * T) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one of the UP*T row loads
                    }
                }
                // <= UP leftover pairs: only the last chunk of the item has them; its last pair may be half padding
                for (int pr = ngroups * UP; pr < npairs; ++pr) {
                    float q1[T], v1;
                    load_pair(myc, myv, pr, q1, v1);
                    consume(q1, v1, (ch * 64 + 2 * pr + half < n) ? 1.0f : 0.f);
                }
                myc = myc_n;
                myv = myv_n;
                fetch_keys(ch + 2, myc_n, myv_n);
            }
        } else {
            // ---- split-f16 pass (see the kernel's header): groups of 16 entries, two register sets, no copies ----
            const int64_t ngroups = (n + 15) >> 4;   // the last group is padded with weight-0 entries of row 0
            // per 64-entry chunk, one entry per lane: row id, weight alpha v, S sqrt(weight) (0: not on the f16 path)
            int myc, myc_n;
            float myw, myw_n, mys, mys_n;
            auto weigh = [&](float vvv, float& ww, float& ss) {
                ww = p.alpha * vvv;
                ss = (ww > 0.f && ww <= wcut) ? sS * __builtin_amdgcn_sqrtf(ww) : 0.f;
            };
            auto fetch = [&](int64_t chunk, int& cc, float& ww, float& ss) {
                float vvv;
                fetch_keys(chunk, cc, vvv);
                weigh(vvv, ww, ss);
            };
            auto fix_outliers = [&](int cc, float ww, float ss) {   // negative / very heavy entries of a chunk: fp32 instruction, pairwise
                const bool out = ss == 0.f && ww != 0.f;
                if (__builtin_amdgcn_ballot_w64(out) == 0) return;
                const float wfix = out ? ww * sS2 : 0.f;
                for (int pr = 0; pr < 32; ++pr) {
                    float q1[T], w1;
                    load_pair(cc, wfix, pr, q1, w1);
                    int t = 0;
#pragma unroll
                    for (int a = 0; a < T; ++a) {
                        const float av = w1 * q1[a];
#pragma unroll
                        for (int b = a; b < T; ++b, ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, q1[b], acc[t], 0, 0, 0);
                    }
                }
            };
            // lane (col, half) works on the entries 16 g + 8 half + r, r = 0 .. 7, of a chunk: ds_bpermute fetches their per-lane values
            const int bsel = 32 * half;   // byte index of lane 8 half
            auto load_group = [&](int src_c, int g, float (&q)[8][T]) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int c = __builtin_amdgcn_ds_bpermute(bsel + 64 * g + 4 * r, src_c);
                    using off_t = typename std::conditional<BIG, size_t, unsigned>::type;
                    const off_t voff = static_cast<off_t>(static_cast<unsigned>(c)) * row_bytes + static_cast<unsigned>(col) * 4u;
                    const float* q_ = reinterpret_cast<const float*>(qbase + voff);
#pragma unroll
                    for (int b = 0; b < T; ++b) q[r][b] = q_[b * 32];
                }
            };
            // the per-entry residual and h (als.cc:292-296), then x = S sqrt(alpha v) q cut into pieces; k0 = the group's first entry
            // Written in stages over the eight entries (all fetches, all dots, the eight lane reductions step by step, ...): a wave
            // alone on its SIMD has nobody to hide a dependent chain behind, so the chains are laid side by side.
            auto prep = [&](float src_w, float src_s, int64_t k0, int g, float (&q)[8][T], u32x4 (&H)[T], u32x4 (&L)[T]) {
                float wgt[8], sw[8], y[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    wgt[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 64 * g + 4 * r, __builtin_bit_cast(int, src_w)));
                    sw[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 64 * g + 4 * r, __builtin_bit_cast(int, src_s)));
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    y[r] = 0.f;
#pragma unroll
                    for (int b = 0; b < T; ++b) y[r] = __builtin_fmaf(q[r][b], p0r[b], y[r]);
                }
                // sum over the 32 lanes of each half (half_sum), the eight chains interleaved
#pragma unroll
                for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x128, 0xf, 0xf, false));
#pragma unroll
                for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x124, 0xf, 0xf, false));
#pragma unroll
                for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x122, 0xf, 0xf, false));
#pragma unroll
                for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x121, 0xf, 0xf, false));
                float yo[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) yo[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, y[r]), 0x401F));
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float cial = __builtin_fmaf(wgt[r], y[r] + yo[r], -wgt[r]);   // alpha v (q.p0 - 1)
                    const float one = (LOSS && lossk && k0 + 8 * half + r < n) ? 1.0f : 0.f;
#pragma unroll
                    for (int b = 0; b < T; ++b) {
                        gpart[b] = __builtin_fmaf(cial, q[r][b], gpart[b]);
                        if (LOSS) g1part[b] = __builtin_fmaf(one, q[r][b], g1part[b]);
                        q[r][b] *= sw[r];
                    }
                }
#pragma unroll
                for (int b = 0; b < T; ++b)
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        unsigned h_, l_;
                        als_split_f16(q[2 * j2][b], q[2 * j2 + 1][b], h_, l_);
                        H[b][j2] = h_;
                        L[b][j2] = l_;
                    }
            };
            // One group's matrix instructions with the NEXT group's preparation and the rows of the one after laid between them BY HAND:
            // a wave alone on its SIMD overlaps the two pipes only if the instruction stream alternates -- one matrix instruction
            // (32 cycles in its pipe), then ~32 cycles of other work -- and the scheduler, asked with sched_group_barrier, clumps
            // (measured in the assembly: runs of 3-10 matrix instructions, then 80 VALU).  So the preparation is cut into NS steps,
            // spread evenly over the NM matrix instructions, with a full scheduling fence after every slot.
            // Steps: 8 x (row id, weight, S sqrt(weight) of entry r)  |  8 x (the T loads of row r, group after next)  |  8 x (q_r . p0)
            //        | 4 levels of the eight lane reductions  |  the eight swizzles  |  8 x (residual, h, scaling of entry r)  |  4T cuts.
            constexpr int NM = 3 * NT, NS = 37 + 4 * T;
            auto fused = [&](int src_c, float src_w, float src_s, int64_t k0, int gl, int gp, float (&ql)[8][T], float (&qp)[8][T],
                             u32x4 (&Ho)[T], u32x4 (&Lo)[T], const u32x4 (&Hi)[T], const u32x4 (&Li)[T]) {
                int cid[8];
                float wgt[8], sw[8], y[8], yo[8];
                als_static_for<NM>([&](auto Ic) {
                    constexpr int i = decltype(Ic)::value;
                    {
                        constexpr int pr = i / NT, t = i % NT;   // small terms first: l h, h l, h h
                        constexpr int a = als_tile_row<T>(t), b = als_tile_col<T>(t);
                        // (pure instructions carry no ordering of their own: instruction selection is free to lay them on either side of
                        //  a fence.  Empty asm statements -- ordered among themselves and with the fences -- tie every step's inputs
                        //  and results to its slot.)
                        const u32x4 X = pr == 0 ? Li[a] : Hi[a];
                        const u32x4 Y = pr == 1 ? Li[b] : Hi[b];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, X), __builtin_bit_cast(f16x8_t, Y), acc[t], 0, 0, 0);
                    }
                    constexpr int k_lo = i * NS / NM, k_hi = (i + 1) * NS / NM;
                    als_static_for<k_hi - k_lo>([&](auto Jc) {
                        constexpr int k = k_lo + decltype(Jc)::value;
                        if constexpr (k < 8) {
                            constexpr int r = k;
                            cid[r] = __builtin_amdgcn_ds_bpermute(bsel + 64 * gl + 4 * r, src_c);
                            wgt[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 64 * gp + 4 * r, __builtin_bit_cast(int, src_w)));
                            sw[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 64 * gp + 4 * r, __builtin_bit_cast(int, src_s)));
                        } else if constexpr (k < 16) {
                            constexpr int r = k - 8;
                            using off_t = typename std::conditional<BIG, size_t, unsigned>::type;
                            const off_t voff = static_cast<off_t>(static_cast<unsigned>(cid[r])) * row_bytes + static_cast<unsigned>(col) * 4u;
                            const float* q_ = reinterpret_cast<const float*>(qbase + voff);
#pragma unroll
                            for (int b = 0; b < T; ++b) ql[r][b] = q_[b * 32];
                        } else if constexpr (k < 24) {
                            constexpr int r = k - 16;
                            y[r] = qp[r][0] * p0r[0];
#pragma unroll
                            for (int b = 1; b < T; ++b) y[r] = __builtin_fmaf(qp[r][b], p0r[b], y[r]);
                        } else if constexpr (k < 28) {
                            constexpr int ctrl = k == 24 ? 0x128 : k == 25 ? 0x124 : k == 26 ? 0x122 : 0x121;   // row_ror 8, 4, 2, 1
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), ctrl, 0xf, 0xf, false));
                            }
                        } else if constexpr (k < 29) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) yo[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, y[r]), 0x401F));
                        } else if constexpr (k < 37) {
                            constexpr int r = k - 29;
                            const float cial = __builtin_fmaf(wgt[r], y[r] + yo[r], -wgt[r]);   // alpha v (q.p0 - 1)
                            const float one = (LOSS && lossk && k0 + 8 * half + r < n) ? 1.0f : 0.f;
#pragma unroll
                            for (int b = 0; b < T; ++b) {
                                gpart[b] = __builtin_fmaf(cial, qp[r][b], gpart[b]);
                                if (LOSS) g1part[b] = __builtin_fmaf(one, qp[r][b], g1part[b]);
                                qp[r][b] *= sw[r];
                            }
                        } else {
                            constexpr int b = (k - 37) / 4, j2 = (k - 37) % 4;
                            unsigned h_, l_;
                            asm volatile("" : "+v"(qp[2 * j2][b]));
                            als_split_f16(qp[2 * j2][b], qp[2 * j2 + 1][b], h_, l_);
                            asm volatile("" : "+v"(h_), "+v"(l_));
                            Ho[b][j2] = h_;
                            Lo[b][j2] = l_;
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (have_pf) {   // fetched while the previous row was being solved
                myc = pf_c0;
                myc_n = pf_c1;
                weigh(pf_v0, myw, mys);
                weigh(pf_v1, myw_n, mys_n);
            } else {
                fetch(0, myc, myw, mys);
                fetch(1, myc_n, myw_n, mys_n);
            }
            float qA[8][T], qB[8][T];
            u32x4 HA[T], LA[T], HB[T], LB[T];
            if (ngroups > 0) {
                fix_outliers(myc, myw, mys);
                load_group(myc, 0, qA);
                load_group(myc, 1, qB);
                prep(myw, mys, 0, 0, qA, HA, LA);
            }
            item_nx = __builtin_amdgcn_readfirstlane(item_nx_raw);
            if (draw) batch_end = item_nx + rows_nx;
            wk_nx = work[item_nx < n_items ? item_nx : n_items - 1];
            int64_t jg = 0;
            auto advance = [&]() {   // jg entered a new 64-entry chunk
                myc = myc_n;
                myw = myw_n;
                mys = mys_n;
                fetch((jg >> 2) + 1, myc_n, myw_n, mys_n);
                fix_outliers(myc, myw, mys);
            };
            while (jg < ngroups) {
                {   // pieces A = group jg; rows B = group jg + 1; rows A <- group jg + 2.  (Past the row's end the keys are padding:
                    // the last pass prepares a group of zeros.  Branching around it makes a second copy of the matrix instructions,
                    // and the register allocator then shuttles the accumulators between the two: measured in the assembly, not worth it.)
                    const int s = static_cast<int>(jg & 3);
                    fused(s < 2 ? myc : myc_n, s < 3 ? myw : myw_n, s < 3 ? mys : mys_n, (jg + 1) * 16, (s + 2) & 3, (s + 1) & 3, qA, qB, HB, LB, HA, LA);
                }
                ++jg;
                if ((jg & 3) == 0) advance();
                if (jg >= ngroups) break;
                {   // the same with the two register sets swapped
                    const int s = static_cast<int>(jg & 3);
                    fused(s < 2 ? myc : myc_n, s < 3 ? myw : myw_n, s < 3 ? mys : mys_n, (jg + 1) * 16, (s + 2) & 3, (s + 1) & 3, qB, qA, HA, LA, HB, LB);
                }
                ++jg;
                if ((jg & 3) == 0) advance();
            }
            have_pf = item_nx < n_items;
            if (have_pf) {
                const int64_t nn = wk_nx.kend - wk_nx.kbeg;
                fetch_keys_of(wk_nx.kbeg, nn, 0, pf_c0, pf_v0);
                fetch_keys_of(wk_nx.kbeg, nn, 1, pf_c1, pf_v1);
                const float* Pn = p.P + static_cast<size_t>(wk_nx.row) * VD0;
                const float* Fn = p.F0 + static_cast<size_t>(wk_nx.row - p.start_x) * VD0;
#pragma unroll
                for (int b = 0; b < T; ++b) { pf_p0[b] = Pn[b * 32 + col]; pf_f0[b] = Fn[b * 32 + col]; }
            }
        }
        constexpr int VD = 32 * T;   // vdim == 32*T on this path
        if (solve_here) {
            float* Pu = p.P + static_cast<size_t>(wk.row) * VD;
            float gs[T], g1s[T];
#pragma unroll
            for (int a = 0; a < T; ++a) {   // the two halves hold the k-parities of the same element
                gs[a] = gpart[a] + __shfl_xor(gpart[a], 32, 64);
                g1s[a] = g1part[a] + __shfl_xor(g1part[a], 32, 64);
            }
            double nume = 0.0, deno = 0.0;
            if (!(p.debug & 1)) {   // pc = p0 and delta = 0 were put in place before the pass
                als_ialspp_inreg<T>(acc, gs, g1s, f0r, p, pc, pc + VD, pc + 2 * VD, lane, half, col, p.adaptive_reg ? static_cast<float>(n) : 1.0f, nume, deno, sI2);
                wave_lds_sync();
                for (int e = lane; e < VD; e += 64) Pu[e] = pc[e];
            }
            if (p.compute_loss && lane == 0) {   // row-level terms ride on lane 0's share of the per-nnz sums
                nume_k += nume;
                deno_k += deno;
            }
            if (SPLIT) { item = item_nx; wk_cur = wk_nx; }
            else { item = __builtin_amdgcn_readfirstlane(ticket(1)); wk_cur = work[item < n_items ? item : 0]; }
            continue;
        }
        // upper-triangle tiles, g, g1 -> the row's scratch slot
        // (every element offset below is a compile-time constant off ONE lane base)
        float* S = scratch + static_cast<size_t>(wk.slot >= 0 && !p.accumulate ? slot_base + wk.slot : wk.row - p.start_x) * als_slot_floats(VD);
        float* Sl = S + half * 4 * VD + col;
        const bool atomic = wk.slot >= 0 || p.accumulate;   // chunk of a heavy row / second pass: summed into the slot (zeroed by the host)
        const float osc = p.out_scale * sI2;   // the tiles leave in M's units (gpart is not scaled: osg)
        const float osg = p.out_scale;
        int t = 0;
#pragma unroll
        for (int a = 0; a < T; ++a) {
#pragma unroll
            for (int b = a; b < T; ++b, ++t) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* dst = Sl + (a * 32 + (e & 3) + 8 * (e >> 2)) * VD + b * 32;
                    if (atomic) atomic_add_f32(dst, acc[t][e] * osc);
                    else *dst = acc[t][e] * osc;
                }
            }
            const float gs = (gpart[a] + __shfl_xor(gpart[a], 32, 64)) * osg;   // the two halves hold the k-parities of the same element
            const float g1s = g1part[a] + __shfl_xor(g1part[a], 32, 64);
            if (half == 0) {
                float* gdst = S + VD * VD + a * 32 + col;
                if (atomic) {
                    atomic_add_f32(gdst, gs);
                    if (IALS && lossk) atomic_add_f32(gdst + VD, g1s);
                } else {
                    *gdst = gs;
                    if (IALS && lossk) gdst[VD] = g1s;
                }
            }
        }
        if (SPLIT) { item = item_nx; wk_cur = wk_nx; }
        else { item = __builtin_amdgcn_readfirstlane(ticket(1)); wk_cur = work[item < n_items ? item : 0]; }
    }
    if (lossk || (INREG && p.compute_loss)) {
        nume_k = wave_sum_f64(nume_k);
        deno_k = wave_sum_f64(deno_k);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 128 < vdim <= 256 (iALS++ at d = 160 ... 256, block_size 32, d % 32 == 0): the upper triangle of M has
// T(T+1)/2 = 15 ... 36 tiles -- too many accumulators for one wave, and M (up to 263 KB) does not fit LDS.
// A block of W = ceil(T/2) waves owns the row: wave w keeps tile-rows w and T-1-w (T+1 tiles, <= 144
// accumulator registers; the middle row alone when T is odd), loads only the column blocks w..T-1 of each
// q row, and the block recurrence of als_ialspp_inreg runs across the waves: for block blk every wave adds
// the products of ITS tiles in column blk / row blk into an LDS vector, the wave that owns the diagonal tile
// runs the three CG steps, the new p_blk goes back through LDS.  Two barriers per block, M never leaves the
// registers.  Heavy rows: their chunks add into a scratch slot and a second launch (`finalize`) starts from
// FF + slot instead of the nnz pass.  The wave index picks one of W instantiations so that every tile,
// operand and accumulator index is a compile-time constant.
// ------------------------------------------------------------------------------------------------
template <int T, int WV>
struct AlsWide {
    static constexpr int W = (T + 1) / 2;
    static constexpr int R0 = WV, R1 = T - 1 - WV;
    static constexpr bool TWO = R1 != R0;
    static constexpr int N0 = T - R0;                 // tiles of row R0: columns R0 .. T-1
    static constexpr int N1 = TWO ? T - R1 : 0;       // tiles of row R1: columns R1 .. T-1
    static constexpr int NTW = N0 + N1;
    static constexpr int NB = T - R0;                 // column blocks R0 .. T-1 are loaded (R1 >= R0)
    // slot of tile (a, b), a in {R0, R1}, b >= a
    static constexpr int slot(int a, int b) { return a == R0 ? b - R0 : N0 + (b - R1); }
    static constexpr bool owns_row(int a) { return a == R0 || (TWO && a == R1); }
};

__host__ __device__ inline size_t als_wide_ring_offset_floats(int vdim) {
    const size_t upto = 3 * static_cast<size_t>(vdim) + 4 * 32 + 32 + 32 + 8 + 4 + static_cast<size_t>(vdim);   // ... | g_1
    return (upto + 3) & ~size_t(3);
}
struct AlsWideLds {          // carved from dynamic LDS: 3 vdim | W*32 | 32 | 32 | 8 floats
    float* pc;               // the row (current iterate)
    float* dl;               // delta = p - p0 (zero in the blocks not solved yet)
    float* hv;               // h = sum_k alpha v_k (q_k.p0 - 1) q_k, all blocks (formed by wave 0, which loads every block of the q rows)
    float* contrib;          // [W][32] column-product partials of the waves
    float* rowres;           // [32] row-product result of the diagonal tile's owner
    float* pvs;              // [32] CG direction
    float* red;              // [8] loss partials
    float* g1v;              // SPLIT: [vdim] g_1 = sum q of the row (loss only), formed by the producer
    u32x4* ring;             // SPLIT: two slots of one group of 16 entries: [16][vdim] fp32 rows + 16 scales S sqrt(alpha v)
};

// SPLIT (round 5, "als_wide_split"): the Gramian through the f16 matrix cores at fp32 accuracy, with the rows gathered ONCE per block.  The fp32
// pass has every wave load the blocks of the q rows it needs itself -- 14 block loads per row at T = 5 where the row has 5: the first split-f16
// version of this kernel kept that and was bound by exactly this traffic (10.0 ms per ML-20M epoch at d = 160, 72 GB out of L2 / Infinity
// Cache; four waves with two row sets each: 11.1).  Now a block carries one more wave, the PRODUCER (wave W): it gathers the 16 rows of a
// group, forms every entry's residual and h (it holds p0), cuts x = S sqrt(alpha v) q into the f16 pieces h + l for all T blocks and parks
// them in one of two LDS slots (T x 2 KB each); the W consumer waves read the pieces of the blocks their tiles need (conflict-free 16-byte
// reads) and issue l h + h l + h h per tile.  One block barrier per group: the producer fills slot (g + 1) & 1 while the consumers drain
// slot g & 1.  The accumulators hold S^2 G during the pass and come back (x 1 / S^2, + the FF tile) behind it.  Calls with weights outside
// the f16 path (als_defer_scan_kernel) keep the fp32 instantiation.
typedef __attribute__((address_space(1))) float AlsGlobalF;   // a pointer the optimiser lost track of (asm launder) is a FLAT pointer until told otherwise

// The lane id from the hardware, opaque to the optimiser: als_wide_item<SPLIT> runs at three blocks per CU (168 registers), where every
// lane constant the compiler hoists out of the row loop (col * 4, thread * 4, ...) is spilled across the pass and reloaded from scratch before
// each of its ~100 uses in the row end -- +12 us per row, the whole gain of the third block.  Re-deriving them behind the pass costs 2 instructions.
__device__ __forceinline__ int als_fresh_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

template <int T, int WV, bool BIG, bool SPLIT = false>
__device__ __forceinline__ void als_wide_item(const AlsParams& p, const AlsWork& wk, bool finalize, float* __restrict__ scratch, const AlsWideLds& L,
                                              int lane, int half, int col, double& nume_k, double& deno_k) {
    constexpr bool PROD = SPLIT && WV == (T + 1) / 2;     // the producer wave: no tiles
    using C = AlsWide<T, PROD ? 0 : WV>;
    constexpr int NTW = PROD ? 1 : C::NTW, NB = C::NB, R0 = C::R0, R1 = C::R1, VD = 32 * T, W = C::W;
    constexpr bool TWO = C::TWO && !PROD;
    // the wave that forms the per-entry residual and h (it needs EVERY block of the q rows): wave 0, which loads them all anyway; SPLIT: the LAST
    // consumer (the middle tile row(s): fewest tiles), which reads the blocks below its own from the slot as well -- on the producer that chain was a
    // quarter of an iteration nothing else in the block could overlap
    constexpr bool HWAVE = SPLIT ? (!PROD && WV == (T + 1) / 2 - 1) : (WV == 0);
    constexpr int UP = 4;
    constexpr unsigned row_bytes = VD * 4u;
    const bool lossk = p.compute_loss && p.axis == 1;
    const bool partial = !finalize && wk.slot >= 0;   // chunk of a heavy row: tiles go to the scratch slot, no solve
    float* Pu = p.P + static_cast<size_t>(wk.row) * VD;

    f32x16 acc[NTW];
    float g10 = 0.f, g11 = 0.f;   // g1 = sum q shares of rows R0, R1 (loss only)
    // Residual-first gradient (als.cc:292-296, like als_ialspp_inreg): every entry's q_k.p0 - 1 against the row AT ENTRY, summed as
    // h = sum alpha v (q.p0 - 1) q.  The dot needs every block of the q row: wave 0 loads them all (R0 = 0), so it forms h for all T
    // blocks and hands it over through LDS (the scratch slot for the chunks of a heavy row).  [Rounds 1-3 evaluated (M p) - g_w here:
    // the same number, but once the model fits G p and g_w nearly cancel -- 10x envelope, profiles/r03_als_config3_warm_epoch.txt.]
    float p0r[HWAVE ? T : 1], hb[HWAVE ? T : 1];
#pragma unroll
    for (int b = 0; b < (HWAVE ? T : 1); ++b) { p0r[b] = HWAVE ? Pu[b * 32 + col] : 0.f; hb[b] = 0.f; }
    float g1all[(SPLIT && HWAVE) ? T : 1];   // SPLIT: g_1 = sum q (loss only) is formed for every block by the wave that reads every block
#pragma unroll
    for (int b = 0; b < ((SPLIT && HWAVE) ? T : 1); ++b) g1all[b] = 0.f;
    if constexpr (!PROD) {   // accumulators start from FF (+ the heavy row's summed chunk tiles when finalizing); zero for a chunk (SPLIT: FF joins behind the pass)
        const float* Fl_ = p.FF + half * 4 * VD + col;
        const float* Sl = scratch + static_cast<size_t>(wk.slot >= 0 ? wk.slot : 0) * als_slot_floats(VD) + half * 4 * VD + col;
        asm volatile("" : "+v"(Fl_));   // (the address arithmetic stays inside the row loop; the cast says "global" again: flat loads otherwise)
        const AlsGlobalF* Fl = (const AlsGlobalF*)Fl_;
        // three straight loops behind wave-uniform branches (a per-element select around a load compiles into one two-instruction block per element
        // with its own wait: 96 taken branches per row and wave on the split path, where the tiles start from zero)
        auto tile_off = [&](int s, int e) {
            const int a = s < C::N0 ? R0 : R1, b = s < C::N0 ? R0 + s : R1 + (s - C::N0);
            return (a * 32 + (e & 3) + 8 * (e >> 2)) * VD + b * 32;
        };
        if (partial || (SPLIT && !finalize)) {
#pragma unroll
            for (int s = 0; s < NTW; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
        } else if (finalize) {   // (heavy rows only: a tile at a time, the loads of all NTW at once would not fit the registers)
#pragma unroll
            for (int s = 0; s < NTW; ++s) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] = Fl[tile_off(s, e)] + Sl[tile_off(s, e)];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NTW; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] = Fl[tile_off(s, e)];
        }
    }
    if (!finalize) {
        const int64_t n = wk.kend - wk.kbeg;
        const int64_t nchunks = (n + 63) / 64;
        const char* qbase = reinterpret_cast<const char*>(p.Q + R0 * 32);   // blocks R0 .. T-1
        auto fetch_keys = [&](int64_t chunk, int& cc, float& vvv) {
            const int64_t kk = chunk * 64 + lane;
            cc = 0;
            vvv = 0.f;
            if (kk < n) {
                cc = p.keys[wk.kbeg + kk];
                vvv = p.vals[wk.kbeg + kk];
                if (lossk && (SPLIT ? PROD : HWAVE)) {   // once per entry: by the wave that walks the keys
                    const double w = static_cast<double>(vvv * p.alpha);
                    deno_k += w;
                    nume_k += 1.0 + w;
                }
            }
        };
        auto load_pair = [&](int myc, float myv, int pr, float (&q)[NB], float& v) {
            const int c0 = __builtin_amdgcn_readlane(myc, 2 * pr), c1 = __builtin_amdgcn_readlane(myc, 2 * pr + 1);
            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), 2 * pr));
            const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), 2 * pr + 1));
            v = half ? v1 : v0;
            using off_t = typename std::conditional<BIG, size_t, unsigned>::type;   // 32-bit: one SGPR base + a VGPR offset
            const off_t voff = static_cast<off_t>(static_cast<unsigned>(half ? c1 : c0)) * row_bytes + static_cast<unsigned>(col) * 4u;
            const float* q_ = reinterpret_cast<const float*>(qbase + voff);
#pragma unroll
            for (int b = 0; b < NB; ++b) q[b] = q_[b * 32];
        };
        auto consume = [&](const float (&q)[NB], float v, float one) {
            const float wgt = p.alpha * v;
            const float lo = lossk ? one : 0.f;
            const float a0 = wgt * q[0];                       // block R0
#pragma unroll
            for (int s = 0; s < C::N0; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q[s], acc[s], 0, 0, 0);
            g10 += lo * q[0];
            if (TWO) {
                const float a1 = wgt * q[R1 - R0];             // block R1
#pragma unroll
                for (int s = 0; s < C::N1; ++s)
                    acc[C::N0 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q[R1 - R0 + s], acc[C::N0 + s], 0, 0, 0);
                g11 += lo * q[R1 - R0];
            }
            if constexpr (HWAVE) {   // NB == T here (fp32 pass: wave 0)
                float part = q[0] * p0r[0];
#pragma unroll
                for (int b = 1; b < T; ++b) part = __builtin_fmaf(q[b], p0r[b], part);
                const float cial = __builtin_fmaf(wgt, half_sum(part, half), -wgt);   // alpha v (q.p0 - 1)
#pragma unroll
                for (int b = 0; b < T; ++b) hb[b] = __builtin_fmaf(cial, q[b], hb[b]);
            }
        };
        if constexpr (SPLIT) {
            const float sS = p.split[0], sI2 = p.split[2], wcut = p.split[3];
            // wave-uniform loop state in SGPRs (the work item came through a vector load; a 64-bit counter in VGPRs is what gets spilled)
            const int n32 = __builtin_amdgcn_readfirstlane(static_cast<int>(n));
            const int ngroups = (p.debug & 16) ? 0 : (n32 + 15) >> 4;     // the last group is padded with weight-0 entries of row 0 ("als_debug" bit 16: timing study, no pass)
            float* const ringf = reinterpret_cast<float*>(L.ring);   // two slots of [16 entries][vdim] fp32 rows + 16 scales S sqrt(alpha v) + 16 weights alpha v
            constexpr int SLOTF = 16 * VD + 32;
            if constexpr (PROD) {
                const int bsel = 32 * half;          // lane (col, half) works on the entries 16 g + 8 half + r, r = 0 .. 7, of a chunk
                // rows come from the block-interleaved copy (Qi[row][T col + b]): a lane's T elements of an entry are contiguous -- T / 4 + 1 load
                // instructions per entry instead of T, which keeps three sets of rows under the 63 loads the hardware counter can track (with one
                // dword per block and entry, two sets were already more than it counts: the waits then drained the prefetch)
                const char* qb0 = reinterpret_cast<const char*>(p.Qi);
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                int myc, myc_n;
                float myw, myw_n, mys, mys_n;
                auto fetch = [&](int64_t chunk, int& cc, float& ww, float& ss) {
                    float vvv;
                    fetch_keys(chunk, cc, vvv);
                    ww = p.alpha * vvv;
                    ss = (ww > 0.f && ww <= wcut) ? sS * __builtin_amdgcn_sqrtf(ww) : 0.f;   // (the host sends calls with other weights to the fp32 kernel)
                };
                fetch(0, myc, myw, mys);
                fetch(1, myc_n, myw_n, mys_n);
                // three sets of rows: while group pg is prepared from one, the rows of pg + 1 and pg + 2 are landing in the others and those of
                // pg + 3 leave into the one just consumed
                // (two sets at T >= 7 were tried in round 6: the spill counts of T = 7 / 8 did not move -- the producer's rows are not what spills)
                constexpr int NSET = 3;
                float qA[8][T], qB[8][T], qC[NSET == 3 ? 8 : 1][NSET == 3 ? T : 1];
                // the row ids of a group's entries, lane (col, half) <- entries 16 g + 8 half + r: all eight exchanges issued back to back (left to the
                // compiler each exchange was followed by its wait and its load -- 24 serialised LDS round trips per group, 1.7 us of the 2.2 a group took)
                auto group_ids = [&](int (&cid)[8], int src_c, int g) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) cid[r] = __builtin_amdgcn_ds_bpermute(bsel + 64 * g + 4 * r, src_c);
                };
                auto load_group = [&](float (&q)[8][T], const int (&cid)[8]) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        using off_t = typename std::conditional<BIG, size_t, unsigned>::type;
                        const off_t voff = static_cast<off_t>(static_cast<unsigned>(cid[r])) * row_bytes + static_cast<unsigned>(col) * (4u * T);
                        const char* q_ = qb0 + voff;
#pragma unroll
                        for (int b4 = 0; b4 + 4 <= T; b4 += 4) {
                            const f4u v4 = *reinterpret_cast<const f4u*>(q_ + 4 * b4);
                            q[r][b4] = v4[0]; q[r][b4 + 1] = v4[1]; q[r][b4 + 2] = v4[2]; q[r][b4 + 3] = v4[3];
                        }
#pragma unroll
                        for (int b = T & ~3; b < T; ++b) q[r][b] = *reinterpret_cast<const float*>(q_ + 4 * b);
                    }
                };
                {
                    int c0[8], c1[8], c2[8];
                    group_ids(c0, myc, 0);
                    group_ids(c1, myc, 1);   // (past the row's end: padding keys, row 0 with weight 0)
                    group_ids(c2, myc, 2);
                    load_group(qA, c0);
                    load_group(qB, c1);
                    if constexpr (NSET == 3) load_group(qC, c2);
                }
                // group pg: residual + h + g_1 from the rows, the pieces of all T blocks into `slot`, then the rows of group pg + 2 into the set
                // keys of chunk 2, in flight since before the first preparation (see the commit in `prepare`)
                int pend_c = 0;
                float pend_v = 0.f;
                bool pend_in = false;
                auto prepare = [&](float (&q)[8][T], int pg, int slot) {
#ifdef BFH_ALS_WIDE_STUDY   // (BFH_EXTRA_FLAGS=-DBFH_ALS_WIDE_STUDY: the two branches cost the default build 150 spilled registers and 1 ms per epoch)
                    if (p.debug & 64) return;   // timing study: the producer only keeps the barriers company (results are wrong)
#endif
                    const int g = pg & 3;
                    // the chunk after next: its keys are fetched on EVERY preparation (one control-flow path: the compiler counts the loads in flight
                    // exactly; a conditional fetch turns the waits into vmcnt(0)) and USED one preparation later -- the commit below reads what the
                    // previous preparation asked for (the same chunk: a commit happens at a chunk's fourth group), never its own request, whose
                    // memory round trip would otherwise be paid in full by every group (measured: 1.7 us per group with everything else switched off)
                    const int nc = pend_c;
                    const float nv = pend_v;
                    const bool kin = pend_in;
                    {
                        const int64_t kk = static_cast<int64_t>(((pg + 2) >> 2) + 1) * 64 + lane;   // (asked for one preparation ahead of its use)
                        pend_in = kk < n;
                        pend_c = pend_in ? p.keys[wk.kbeg + kk] : 0;
                        pend_v = pend_in ? p.vals[wk.kbeg + kk] : 0.f;
                    }
                    float wgt[8], sw[8];
                    int cid[8];   // the rows of group pg + NSET (the keys of a chunk's last NSET groups + NSET sit in the next chunk)
                    group_ids(cid, g < 4 - NSET ? myc : myc_n, (g + NSET) & 3);
#pragma unroll
                    for (int r = 0; r < 8; ++r) wgt[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 64 * g + 4 * r, __builtin_bit_cast(int, myw)));
#pragma unroll
                    for (int r = 0; r < 8; ++r) sw[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 64 * g + 4 * r, __builtin_bit_cast(int, mys)));
                    __builtin_amdgcn_sched_barrier(0);   // (the 24 exchanges stay together, ahead of everything that waits on one of them)
                    // the group's rows (fp32) and scales into the slot: [entry 8 half + r][block][col], then the 16 scales -- the consumers cut the
                    // blocks they need themselves (in parallel on their own SIMDs; cutting all T blocks here made this wave the bottleneck: 11.8 ms)
                    float* const dst = ringf + slot * SLOTF + (8 * half) * VD + col;
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int b = 0; b < T; ++b) dst[r * VD + b * 32] = q[r][b];
                    if (col == 0) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            ringf[slot * SLOTF + 16 * VD + 8 * half + r] = sw[r];
                            ringf[slot * SLOTF + 16 * VD + 16 + 8 * half + r] = wgt[r];   // alpha v: the residual's weight (last consumer)
                        }
                    }
                    // the rows of group pg + NSET leave into the set just consumed
                    load_group(q, cid);
                    {   // commit the keys fetched at the top of this preparation when it ends a chunk
                        const bool adv = g == 3;   // the next group opens a new 64-entry chunk
                        if (lossk && adv && kin) {
                            const double w = static_cast<double>(nv * p.alpha);
                            deno_k += w;
                            nume_k += 1.0 + w;
                        }
                        const float nw = p.alpha * nv;
                        const float ns = (nw > 0.f && nw <= wcut) ? sS * __builtin_amdgcn_sqrtf(nw) : 0.f;
                        myc = adv ? myc_n : myc; myw = adv ? myw_n : myw; mys = adv ? mys_n : mys;
                        myc_n = adv ? nc : myc_n; myw_n = adv ? nw : myw_n; mys_n = adv ? ns : mys_n;
                    }
                };
                prepare(qA, 0, 0);
                __syncthreads();                      // slot 0 is ready
                for (int jg = 0; jg < ngroups;) {     // while the consumers drain slot jg & 1: group jg + 1 (past the end: a padding group nobody reads)
                    prepare(qB, jg + 1, (jg + 1) & 1);
                    __syncthreads();
                    if (++jg >= ngroups) break;
                    if constexpr (NSET == 3) {
                        prepare(qC, jg + 1, (jg + 1) & 1);
                        __syncthreads();
                        if (++jg >= ngroups) break;
                    }
                    prepare(qA, jg + 1, (jg + 1) & 1);
                    __syncthreads();
                    ++jg;
                }
            } else {
                __syncthreads();                      // slot 0 is ready
                for (int jg = 0; jg < ngroups; ++jg) {
#ifdef BFH_ALS_WIDE_STUDY
                    if (p.debug & 32) { __syncthreads(); continue; }   // timing study: the consumers only keep the barriers company (results are wrong)
#endif
                    constexpr int QB0 = HWAVE ? 0 : R0, NQ = T - QB0, PO = R0 - QB0;   // blocks read from the slot, index of block R0 among them
                    const float* const src = ringf + (jg & 1) * SLOTF + (8 * half) * VD + QB0 * 32 + col;
                    const float* const ssw = ringf + (jg & 1) * SLOTF + 16 * VD + 8 * half;
                    u32x4 H[T <= 6 ? NB : 2], Lo[T <= 6 ? NB : 2];   // T = 7, 8: the pieces of the tile rows' own blocks R0, R1 only
                    if constexpr (T <= 6) {
                        float qv[8][NQ], sw[8];
    #pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            sw[r] = ssw[r];
    #pragma unroll
                            for (int b = 0; b < NQ; ++b) qv[r][b] = src[r * VD + b * 32];
                        }
                        if constexpr (HWAVE) {   // the residual q.p0 - 1 of every entry, h and (loss) g_1 -- NQ == T here; the eight chains side by side
                            float wgt[8], y[8];
    #pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                wgt[r] = ssw[16 + r];
                                y[r] = qv[r][0] * p0r[0];
    #pragma unroll
                                for (int b = 1; b < T; ++b) y[r] = __builtin_fmaf(qv[r][b], p0r[b], y[r]);
                            }
    #pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x128, 0xf, 0xf, false));
    #pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x124, 0xf, 0xf, false));
    #pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x122, 0xf, 0xf, false));
    #pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x121, 0xf, 0xf, false));
    #pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const float yo = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, y[r]), 0x401F));
                                const float cial = __builtin_fmaf(wgt[r], y[r] + yo, -wgt[r]);   // alpha v (q.p0 - 1)
    #pragma unroll
                                for (int b = 0; b < T; ++b) hb[b] = __builtin_fmaf(cial, qv[r][b], hb[b]);
                            }
                            if (lossk) {   // g_1 = sum q over the real entries (wave-uniform branch)
    #pragma unroll
                                for (int r = 0; r < 8; ++r) {
                                    const float one = (jg * 16 + 8 * half + r < n32) ? 1.0f : 0.f;
    #pragma unroll
                                    for (int b = 0; b < T; ++b) g1all[b] = __builtin_fmaf(one, qv[r][b], g1all[b]);
                                }
                            }
                        }
    #pragma unroll
                        for (int b = 0; b < NB; ++b)
    #pragma unroll
                            for (int j2 = 0; j2 < 4; ++j2) {
                                unsigned h_, l_;
                                als_split_pair_mix(qv[2 * j2][PO + b], sw[2 * j2], qv[2 * j2 + 1][PO + b], sw[2 * j2 + 1], h_, l_);
                                H[b][j2] = h_;
                                Lo[b][j2] = l_;
                            }

                    } else {
                        // T = 7, 8 (128 / 144 accumulators): the blocks of the slot are taken ONE AT A TIME -- eight registers of rows in flight instead of
                        // 8 x NQ (64 at T = 8), without which the allocator spills 600 registers.  The wave that forms the residuals walks the slot twice:
                        // first the dots q.p0 block by block, then -- with the eight coefficients alpha v (q.p0 - 1) in hand -- h, g_1 and the cut.
                        float sw[8], cial[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) { sw[r] = ssw[r]; cial[r] = 0.f; }
                        if constexpr (HWAVE) {   // (QB0 == 0: `src` starts at block 0)
                            float wgt[8], y[8];
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                wgt[r] = ssw[16 + r];
                                y[r] = src[r * VD] * p0r[0];
                            }
#pragma unroll
                            for (int b = 1; b < T; ++b)
#pragma unroll
                                for (int r = 0; r < 8; ++r) y[r] = __builtin_fmaf(src[r * VD + b * 32], p0r[b], y[r]);
#pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x128, 0xf, 0xf, false));
#pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x124, 0xf, 0xf, false));
#pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x122, 0xf, 0xf, false));
#pragma unroll
                            for (int r = 0; r < 8; ++r) y[r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[r]), 0x121, 0xf, 0xf, false));
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const float yo = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, y[r]), 0x401F));
                                cial[r] = __builtin_fmaf(wgt[r], y[r] + yo, -wgt[r]);   // alpha v (q.p0 - 1)
                            }
                        }
#pragma unroll
                        for (int b = 0; b < NQ; ++b) {
                            float qb[8];
#pragma unroll
                            for (int r = 0; r < 8; ++r) qb[r] = src[r * VD + b * 32];
                            if constexpr (HWAVE) {   // entry by entry, like the all-at-once form above: the same sums in the same order
#pragma unroll
                                for (int r = 0; r < 8; ++r) hb[b] = __builtin_fmaf(cial[r], qb[r], hb[b]);
                                if (lossk) {
#pragma unroll
                                    for (int r = 0; r < 8; ++r) {
                                        const float one = (jg * 16 + 8 * half + r < n32) ? 1.0f : 0.f;
                                        g1all[b] = __builtin_fmaf(one, qb[r], g1all[b]);
                                    }
                                }
                            }
                            if (b >= PO) {
                                // ... and its products are issued at once: only the pieces of the two tile rows' OWN blocks (the X operands) stay, a
                                // block's pieces (the Y operand) live for its eight matrix instructions -- 24 registers of pieces instead of 16 per block
                                // (128 at T = 8, which with 144 accumulators spilled 600 registers into the group loop).  Per accumulator the order is
                                // l l, l h, h l, h h as in the all-at-once form; the two tile rows' instructions alternate (no back-to-back dependence).
                                const int bb = b - PO;   // block R0 + bb
                                u32x4 Yh, Yl;
#pragma unroll
                                for (int j2 = 0; j2 < 4; ++j2) {
                                    unsigned h_, l_;
                                    als_split_pair_mix(qb[2 * j2], sw[2 * j2], qb[2 * j2 + 1], sw[2 * j2 + 1], h_, l_);
                                    Yh[j2] = h_;
                                    Yl[j2] = l_;
                                }
                                if (bb == 0) { H[0] = Yh; Lo[0] = Yl; }
                                if (TWO && bb == R1 - R0) { H[1] = Yh; Lo[1] = Yl; }
                                const bool second = TWO && bb >= R1 - R0;
#pragma unroll
                                for (int pr = -1; pr < 3; ++pr) {
                                    const u32x4 Y = (pr == 1 || pr == -1) ? Yl : Yh;
                                    acc[bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, pr <= 0 ? Lo[0] : H[0]), __builtin_bit_cast(f16x8_t, Y), acc[bb], 0, 0, 0);
                                    if (second)
                                        acc[C::N0 + bb - (R1 - R0)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, pr <= 0 ? Lo[1] : H[1]), __builtin_bit_cast(f16x8_t, Y),
                                                                                                             acc[C::N0 + bb - (R1 - R0)], 0, 0, 0);
                                }
                            }
                        }
                    }
                    // T >= 6: the fourth product l l as well (2^-22 of a term: what the three-product form drops).  With it the products are exact to
                    // 2^-33 like the fp32 instruction's, and the one ill-conditioned tiny case that kept d = 192 on the fp32 form in round 5 (5.9x the
                    // oracle's distance from float64 against 3.4x for the fp32 instruction, bound 4x) lands where the fp32 instruction does; a third more
                    // matrix instructions, still a fraction of the fp32 form's (1/16 of the f16 rate)
                    constexpr int NPROD = T >= 6 ? 4 : 3;
                    if constexpr (T <= 6)   // (T = 7, 8: issued block by block above)
#pragma unroll
                    for (int pr0 = 0; pr0 < NPROD; ++pr0) {   // small terms first: (l l,) l h, h l, h h
                        const int pr = NPROD == 4 ? pr0 - 1 : pr0;   // -1: l l
#pragma unroll
                        for (int sidx = 0; sidx < C::N0; ++sidx) {
                            const u32x4 X = pr <= 0 ? Lo[0] : H[0];
                            const u32x4 Y = (pr == 1 || pr == -1) ? Lo[sidx] : H[sidx];
                            acc[sidx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, X), __builtin_bit_cast(f16x8_t, Y), acc[sidx], 0, 0, 0);
                        }
                        if constexpr (TWO) {
#pragma unroll
                            for (int sidx = 0; sidx < C::N1; ++sidx) {
                                const u32x4 X = pr <= 0 ? Lo[R1 - R0] : H[R1 - R0];
                                const u32x4 Y = (pr == 1 || pr == -1) ? Lo[R1 - R0 + sidx] : H[R1 - R0 + sidx];
                                acc[C::N0 + sidx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, X), __builtin_bit_cast(f16x8_t, Y), acc[C::N0 + sidx], 0, 0, 0);
                            }
                        }
                    }
                    __syncthreads();                  // the slot may be refilled; the next one is ready
                }
                // back to M's units, and (whole rows) the FF tiles join -- one wave-uniform branch around two straight loops (a per-element
                // select around the load compiles into hundreds of two-instruction blocks with a spill reload each)
                if (partial) {
#pragma unroll
                    for (int sidx = 0; sidx < NTW; ++sidx)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[sidx][e] *= sI2;
                } else {
                    const float* Fl2_ = p.FF + half * 4 * VD + col;
                    asm volatile("" : "+v"(Fl2_));
                    const AlsGlobalF* Fl2 = (const AlsGlobalF*)Fl2_;
#pragma unroll
                    for (int sidx = 0; sidx < NTW; ++sidx) {
                        const int a = sidx < C::N0 ? R0 : R1, b = sidx < C::N0 ? R0 + sidx : R1 + (sidx - C::N0);
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[sidx][e] = __builtin_fmaf(acc[sidx][e], sI2, Fl2[(a * 32 + (e & 3) + 8 * (e >> 2)) * VD + b * 32]);
                    }
                }
            }
        } else {
            auto nnz_of = [&](int64_t ch) { return ch < nchunks ? static_cast<int>((n - ch * 64) < 64 ? (n - ch * 64) : 64) : 0; };
            auto groups_of = [&](int64_t ch) { return (nnz_of(ch) >> 1) / UP; };
            int myc, myc_n;
            float myv, myv_n;
            fetch_keys(0, myc, myv);
            fetch_keys(1, myc_n, myv_n);
            float qa[UP][NB], va[UP];
            if (groups_of(0) > 0) {
    #pragma unroll
                for (int uu = 0; uu < UP; ++uu) load_pair(myc, myv, uu, qa[uu], va[uu]);
            }
            for (int64_t ch = 0; ch < nchunks; ++ch) {
                const int npairs = (nnz_of(ch) + 1) >> 1;
                const int ngroups = groups_of(ch);
                for (int gidx = 0; gidx < ngroups; ++gidx) {   // same pipelining as als_gram_kernel
                    float qb[UP][NB], vb[UP];
                    const bool here = gidx + 1 < ngroups;
                    const int src_c = here ? myc : myc_n;
                    const float src_v = here ? myv : myv_n;
                    const int pr0 = here ? (gidx + 1) * UP : 0;
    #pragma unroll
                    for (int uu = 0; uu < UP; ++uu) load_pair(src_c, src_v, pr0 + uu, qb[uu], vb[uu]);
    #pragma unroll
                    for (int uu = 0; uu < UP; ++uu) consume(qa[uu], va[uu], 1.0f);
    #pragma unroll
                    for (int uu = 0; uu < UP; ++uu) {
                        va[uu] = vb[uu];
    #pragma unroll
                        for (int b = 0; b < NB; ++b) qa[uu][b] = qb[uu][b];
                    }
    #pragma unroll
                    for (int i = 0; i < UP * NTW; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
                        if (i % 2 == 0 && i / 2 < UP * NB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                for (int pr = ngroups * UP; pr < npairs; ++pr) {
                    float q1[NB], v1;
                    load_pair(myc, myv, pr, q1, v1);
                    consume(q1, v1, (ch * 64 + 2 * pr + half < n) ? 1.0f : 0.f);
                }
                myc = myc_n;
                myv = myv_n;
                fetch_keys(ch + 2, myc_n, myv_n);
            }
        }
    }
    int tid = threadIdx.x;
    if constexpr (SPLIT) {   // the row end works on lane constants made HERE (als_fresh_lane)
        lane = als_fresh_lane();
        half = lane >> 5;
        col = lane & 31;
        tid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 64 + lane;
    }
    // the two halves hold the k-parities of the same element
    g10 += __shfl_xor(g10, 32, 64); g11 += __shfl_xor(g11, 32, 64);
    if constexpr (HWAVE) {
#pragma unroll
        for (int b = 0; b < T; ++b) hb[b] += __shfl_xor(hb[b], 32, 64);
    }
    if constexpr (SPLIT && HWAVE) {
#pragma unroll
        for (int b = 0; b < T; ++b) g1all[b] += __shfl_xor(g1all[b], 32, 64);
    }

    float* S = scratch + static_cast<size_t>(wk.slot >= 0 ? wk.slot : 0) * als_slot_floats(VD);
    if (partial) {   // add this chunk's tiles / vector shares into the (zeroed) slot
        if constexpr (!PROD) {
            float* Sl = S + half * 4 * VD + col;
#pragma unroll
            for (int s = 0; s < NTW; ++s) {
                const int a = s < C::N0 ? R0 : R1, b = s < C::N0 ? R0 + s : R1 + (s - C::N0);
#pragma unroll
                for (int e = 0; e < 16; ++e) atomic_add_f32(Sl + (a * 32 + (e & 3) + 8 * (e >> 2)) * VD + b * 32, acc[s][e]);
            }
        }
        if (half == 0) {
            if constexpr (HWAVE) {
#pragma unroll
                for (int b = 0; b < T; ++b) atomic_add_f32(S + VD * VD + b * 32 + col, hb[b]);
            }
            if constexpr (SPLIT) {
                if constexpr (HWAVE) {
                    if (lossk) {
#pragma unroll
                        for (int b = 0; b < T; ++b) atomic_add_f32(S + VD * VD + VD + b * 32 + col, g1all[b]);
                    }
                }
            } else {
                if (lossk) atomic_add_f32(S + VD * VD + VD + R0 * 32 + col, g10);
                if (TWO && lossk) atomic_add_f32(S + VD * VD + VD + R1 * 32 + col, g11);
            }
        }
        return;
    }
    if (finalize) {   // g1 was summed in the slot (h: below)
        g10 = lossk ? S[VD * VD + VD + R0 * 32 + col] : 0.f;
        if (TWO) g11 = lossk ? S[VD * VD + VD + R1 * 32 + col] : 0.f;
    }

    // ---------------- iALS++ across the W waves (als.cc:269-352, see als_ialspp_inreg) ----------------
    for (int e = tid; e < VD; e += 64 * (W + (SPLIT ? 1 : 0))) {
        L.pc[e] = Pu[e];
        L.dl[e] = 0.f;
        if (finalize) L.hv[e] = S[VD * VD + e];
    }
    if constexpr (HWAVE) {
        if (!finalize && half == 0) {
#pragma unroll
            for (int b = 0; b < T; ++b) L.hv[b * 32 + col] = hb[b];
        }
    }
    if constexpr (SPLIT && HWAVE) {   // g_1 of every block for the waves that own the rows (loss only)
        if (!finalize && half == 0 && lossk) {
#pragma unroll
            for (int b = 0; b < T; ++b) L.g1v[b * 32 + col] = g1all[b];
        }
    }
    // f0 = FF p0 of this wave's rows (als_rowff_kernel)
    const float* Fu0 = p.F0 + static_cast<size_t>(wk.row - p.start_x) * VD;
    const float f00 = PROD ? 0.f : Fu0[R0 * 32 + col], f01 = TWO ? Fu0[R1 * 32 + col] : 0.f;
    __syncthreads();
    if constexpr (SPLIT && !PROD) {
        if (!finalize && lossk) { g10 = L.g1v[R0 * 32 + col]; if (TWO) g11 = L.g1v[R1 * 32 + col]; }
    }
    // this wave's share of (M x)[32 blk + col] for x = pc: column products of its tiles in column blk go to
    // contrib[WV], the row products of row blk (if it owns it) to rowres; the caller sums after a barrier
    // DELTA: x is delta = p - p0, which is non-zero only in the blocks ALREADY solved (< blk): the tiles whose operand block is >= blk multiply
    // exact zeros and are skipped -- all of the row products, and the column products of the wave's own rows >= blk (round 5: a third of the row end)
    auto block_partials = [&](auto blk_c, const float* x, auto delta_c) {
        constexpr int blk = decltype(blk_c)::value;
        constexpr bool DELTA = decltype(delta_c)::value;
        if constexpr (PROD) return;   // the producer holds no tiles: it only keeps the block's barriers company
        float part = 0.f;
        if constexpr (DELTA ? (R0 < blk) : (R0 <= blk)) part += als_tile_colpart(acc[C::slot(R0, blk)], x + R0 * 32, half);
        if constexpr (TWO && (DELTA ? (R1 < blk) : (R1 <= blk))) part += als_tile_colpart(acc[C::slot(R1, blk)], x + R1 * 32, half);
        part += __shfl_xor(part, 32, 64);
        if (half == 0) L.contrib[WV * 32 + col] = part;
        if constexpr (!PROD && C::owns_row(blk)) {
            float y = 0.f;
            if constexpr (blk < T - 1 && !DELTA) {
                float z[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
                for (int jb = blk + 1; jb < T; ++jb) {
                    const float xv = x[jb * 32 + col];
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] += acc[C::slot(blk, jb)][e] * xv;
                }
                y = als_rows_reduce(z, lane);
            }
            const int es = (lane >> 1) & 15;
            if (!(lane & 1)) L.rowres[(es & 3) + 8 * (es >> 2) + 4 * half] = y;
        }
    };
    auto block_sum = [&]() {
        float s = L.rowres[col];
#pragma unroll
        for (int w = 0; w < W; ++w) s += L.contrib[w * 32 + col];
        return s;
    };
    const float ada = p.adaptive_reg ? static_cast<float>(wk.kend - wk.kbeg) : 1.0f;
    (void)ada;
    if (p.compute_loss) {   // als.cc:288-309 on the row at entry: reg*ada*|p|^2 and, on the item side, with g_w = G p0 - h:
        // p FF p + [p G p - 2 p.(g_w + g_1)] = 2 p.f0 - p M p + 2 p.(h - g_1)   (als_ialspp_inreg)
        float pp = 0.f, pmp = 0.f, pg = 0.f;
        als_static_for<T>([&](auto blk_c) {
            constexpr int blk = decltype(blk_c)::value;
            float mp = 0.f;
            if (p.axis == 1) {
                block_partials(blk_c, L.pc, std::false_type{});
                __syncthreads();
                mp = block_sum();
                __syncthreads();
            }
            if constexpr (!PROD && C::owns_row(blk)) {
                const float pv = L.pc[blk * 32 + col];
                pp += pv * pv;
                pmp += pv * (2.0f * (blk == R0 ? f00 : f01) - mp);
                pg += pv * ((blk == R0 ? g10 : g11) - L.hv[blk * 32 + col]);
            }
        });
        pp = wave_sum(half == 0 ? pp : 0.f);
        pg = wave_sum(half == 0 ? pg : 0.f);
        pmp = wave_sum(half == 0 ? pmp : 0.f);
        if (lane == 0) {
            // heavy rows: ada uses the full row length, which the finalize item carries in kend - kbeg
            nume_k += static_cast<double>(ada * p.reg * pp);
            if (p.axis == 1) {
                nume_k += static_cast<double>(pmp) - 2.0 * static_cast<double>(pg);
                if (HWAVE) deno_k += static_cast<double>(p.op_rows);
            }
        }
    }
    als_static_for<T>([&](auto blk_c) {
        constexpr int blk = decltype(blk_c)::value;
        if constexpr (blk > 0) {   // (M delta)[blk]: delta is non-zero only in the blocks already solved -- nothing at all for the first block
            block_partials(blk_c, L.dl, std::true_type{});
            __syncthreads();
        }
        if constexpr (!PROD && C::owns_row(blk)) {   // this wave holds the diagonal tile: gradient of the block at the current row + 3 CG steps (als.cc:286-346)
            const float pblk = L.pc[blk * 32 + col];
            const float bi = (blk == R0 ? f00 : f01) + L.hv[blk * 32 + col] + (blk > 0 ? block_sum() : 0.f) + p.reg * pblk;
            float xr = 0.f, rr = bi, pvr = bi;
            double rsold = static_cast<double>(wave_sum(half == 0 ? rr * rr : 0.f));
            if (rsold > static_cast<double>(p.cg_tol)) {
                for (int step = 0; step < 3; ++step) {
                    wave_lds_sync();
                    if (half == 0) L.pvs[col] = pvr;
                    wave_lds_sync();
                    float ap = als_tile_colpart(acc[C::slot(blk, blk)], L.pvs, half);
                    ap += __shfl_xor(ap, 32, 64);
                    ap += p.reg * pvr;
                    const float pap = wave_sum(half == 0 ? pvr * ap : 0.f);
                    // als.cc:328: double / float rounded to float.  Both operands hold float values, and a quotient of two floats taken in double
                    // and rounded to float IS the correctly rounded float quotient (53 >= 2 * 24 + 2): one fp32 division, same bits (als_ialspp_inreg)
                    const float step_size = static_cast<float>(rsold) / pap;
                    xr += step_size * pvr;
                    rr -= step_size * ap;
                    const double rsnew = static_cast<double>(wave_sum(half == 0 ? rr * rr : 0.f));
                    if (rsnew < static_cast<double>(p.cg_tol)) break;
                    pvr = rr + (static_cast<float>(rsnew) / static_cast<float>(rsold)) * pvr;
                    rsold = rsnew;
                }
            }
            if (half == 0 && !(p.debug & 1)) {
                L.pc[blk * 32 + col] = pblk - xr;
                L.dl[blk * 32 + col] = -xr;
            }
        }
        __syncthreads();
    });
    for (int e = tid; e < VD; e += 64 * (W + (SPLIT ? 1 : 0))) Pu[e] = L.pc[e];
    __syncthreads();
}

// workgroups of als_wide_kernel a CU holds: the fp32 form two (W waves of <= 144 accumulators); the split form three at T = 5 (96 accumulators: 168
// registers), two at T = 6 (112) and one above (128 / 144 accumulators + the pieces of up to eight blocks want all 256 registers)
__host__ __device__ constexpr int als_wide_blocks_per_cu(int T, bool split) { return split ? (T <= 5 ? 3 : (T == 6 ? 2 : 1)) : 2; }

template <int T, bool BIG, bool SPLIT = false>
__global__ __launch_bounds__((64 * ((T + 1) / 2 + (SPLIT ? 1 : 0))), als_wide_blocks_per_cu(T, SPLIT)) void als_wide_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_items,
                                                                                            float* __restrict__ scratch, int finalize) {
    constexpr int W = (T + 1) / 2, VD = 32 * T;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    AlsWideLds L;
    L.pc = lds;
    L.dl = lds + VD;
    L.hv = lds + 2 * VD;
    L.contrib = lds + 3 * VD;
    L.rowres = L.contrib + W * 32;
    L.pvs = L.rowres + 32;
    L.red = L.pvs + 32;
    int* s_item = reinterpret_cast<int*>(L.red + 8);
    L.g1v = L.red + 12;
    L.ring = reinterpret_cast<u32x4*>(lds + als_wide_ring_offset_floats(VD));
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    double nume_k = 0.0, deno_k = 0.0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) *s_item = atomicAdd(p.ticket, 1);
        __syncthreads();
        const int item = *s_item;
        if (item >= n_items) break;
        const AlsWork wk = work[item];
        // SPLIT: the roles (W consumers with different tile counts + the producer) ROTATE over a block's waves from row to row -- the tiles live in
        // registers only inside a row -- so that every SIMD sees the same mix of work whatever the placement of the waves (with fixed roles both
        // blocks of a CU had their producer on the same SIMD, whose VALU then bounded the pass: 6.0 ms of 9.6 at d = 160)
        const int role = SPLIT ? (wv + item) % (W + 1) : wv;
        if (SPLIT && role == W) als_wide_item<T, W, BIG, SPLIT>(p, wk, finalize != 0, scratch, L, lane, half, col, nume_k, deno_k);   // the producer
        else if (role == 0) als_wide_item<T, 0, BIG, SPLIT>(p, wk, finalize != 0, scratch, L, lane, half, col, nume_k, deno_k);
        else if (role == 1) als_wide_item<T, 1, BIG, SPLIT>(p, wk, finalize != 0, scratch, L, lane, half, col, nume_k, deno_k);
        else if (role == 2) als_wide_item<T, 2, BIG, SPLIT>(p, wk, finalize != 0, scratch, L, lane, half, col, nume_k, deno_k);
        else als_wide_item<T, (W > 3 ? 3 : 0), BIG, SPLIT>(p, wk, finalize != 0, scratch, L, lane, half, col, nume_k, deno_k);
    }
    if (p.compute_loss) {
        nume_k = wave_sum_f64(nume_k);
        deno_k = wave_sum_f64(deno_k);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}
// dynamic LDS: 3 vdim | 4 x 32 contrib | 32 | 32 | 8 loss | 4 (ticket) | [SPLIT: vdim g_1, then -- 16-byte aligned -- the two slots of pieces]
__host__ __device__ inline size_t als_wide_lds_bytes(int vdim, bool split = false) {
    const size_t base = 3 * static_cast<size_t>(vdim) + 4 * 32 + 32 + 32 + 8 + 4;
    return (split ? als_wide_ring_offset_floats(vdim) + 2 * (16 * static_cast<size_t>(vdim) + 32) : base) * sizeof(float);
}

struct AlsHeavy {
    int row, slot;
    int64_t n;
};

// Dense phase of the split design: a 256-thread block per row.  The four waves pull the row's tiles of G
// (+ FF) from the scratch slot into LDS with 32 independent loads in flight each -- the phase is pure
// latency otherwise, LDS holds two rows per CU at vdim 128 -- mirror the missing triangle, share the
// M p0 matvec, then wave 0 runs als_dense_solve.
__global__ __launch_bounds__(256) void als_solve_kernel(AlsParams p, const AlsHeavy* __restrict__ heavy, int n_heavy, const float* __restrict__ scratch,
                                                        int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int vdim = p.vdim, D = p.d, ld = vdim + ALS_LD_PAD, T = vdim / 32;
    float* M = lds;
    float* gv = lds + vdim * ld;
    float* pl = gv + vdim;
    float* w0 = pl + vdim; float* w1 = w0 + vdim; float* w2 = w1 + vdim; float* w3 = w2 + vdim; float* w4 = w3 + vdim;
    float* p0 = w4 + vdim; float* f0 = p0 + vdim;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, col = lane & 31;
    const bool lossk = p.compute_loss && p.axis == 1;
    double nume = 0.0, deno = 0.0;
    for (int h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        const AlsHeavy hv = heavy[h];
        __syncthreads();
        const float* S = scratch + static_cast<size_t>(hv.slot) * als_slot_floats(vdim);
        float* Pu = p.P + static_cast<size_t>(hv.row) * vdim;
        int k = 0;
        for (int ta = 0; ta < T; ++ta)
            for (int tb = 0; tb < T; ++tb) {
                const int j = tb - ta;
                if (j < 0) continue;                     // the slot holds the upper triangle of tiles
                if ((k++ & 3) != wv) continue;   // tiles are dealt round-robin to the four waves
                const size_t base = static_cast<size_t>(ta * 32 + half) * vdim + tb * 32 + col;
                float sv[16], fv[16];
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    sv[it] = S[base + static_cast<size_t>(2 * it) * vdim];
                    fv[it] = p.ff_scale * p.FF[base + static_cast<size_t>(2 * it) * vdim];
                }
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int rr = ta * 32 + 2 * it + half, cc = tb * 32 + col;
                    const float m = sv[it] + fv[it];
                    M[rr * ld + cc] = m;
                    if (j != 0) M[cc * ld + rr] = m;   // mirrored tile (FF is symmetric)
                }
            }
        for (int e = threadIdx.x; e < vdim; e += blockDim.x) {
            gv[e] = S[vdim * vdim + e];   // iALS++: h = sum alpha v (q.p0 - 1) q (als_gram_kernel);  dense solvers: y
            if (lossk && mode == 8) w1[e] = S[vdim * vdim + vdim + e];
            pl[e] = Pu[e];
            p0[e] = Pu[e];
            f0[e] = 0.f;
        }
        __syncthreads();
        if (mode == 8 || lossk) {   // f0 = M p0: the row's 32-row blocks are dealt to the waves, the half-waves split the columns
            for (int blk = wv; blk < T; blk += 4) {
                const float* Mi = M + (blk * 32 + col) * ld + half * (vdim / 2);
                const float* pp = p0 + half * (vdim / 2);
                float sum = 0.f;
#pragma unroll 8
                for (int j = 0; j < vdim / 2; ++j) sum += Mi[j] * pp[j];
                sum += __shfl_xor(sum, 32, 64);
                if (half == 0) f0[blk * 32 + col] = sum;
            }
            __syncthreads();
        }
        if (mode == 8) {
            // iALS++: the gradient wants FF p0 (als_dense_solve: b = M[blk,:] delta + f0 + gv + reg p), the loss M p0 (kept in w2) and
            // g_w + g_1 with g_w = G p0 - h = (M - FF) p0 - h (w0).  FF p0 straight from the 64 KB of FF in L2.
            for (int blk = wv; blk < T; blk += 4) {
                const float* Fi = p.FF + static_cast<size_t>(blk * 32 + col) * vdim + half * (vdim / 2);
                const float* pp = p0 + half * (vdim / 2);
                float sum = 0.f;
#pragma unroll 8
                for (int j = 0; j < vdim / 2; ++j) sum += Fi[j] * pp[j];
                sum += __shfl_xor(sum, 32, 64);
                if (half == 0) {
                    const int e = blk * 32 + col;
                    const float mp0 = f0[e], fp0 = p.ff_scale * sum;
                    w2[e] = mp0;
                    f0[e] = fp0;
                    w0[e] = (mp0 - fp0) - gv[e] + (lossk ? w1[e] : 0.f);
                }
            }
            __syncthreads();
        }
        if (wv == 0) {
            const float ada = p.adaptive_reg ? static_cast<float>(hv.n) : 1.0f;
            if (p.compute_loss) als_row_loss(p, pl, mode == 8 ? w2 : f0, mode == 8 ? w0 : gv, nullptr, ada, lane, nume, deno);
            if (!(p.debug & 1)) als_dense_solve(M, gv, pl, p0, f0, w0, w1, w2, w3, w4, p, lane, p.reg * ada, mode);
            for (int i = lane; i < D; i += 64) Pu[i] = pl[i];
        }
    }
    if (p.compute_loss && threadIdx.x == 0) {
        if (nume != 0.0) atomicAdd(p.loss, nume);
        if (deno != 0.0) atomicAdd(p.loss + 1, deno);
    }
}

}  // namespace bfh
#include "als_pc.hpp"
namespace bfh {

// ------------------------------------------------------------------------------------------------
#ifndef BFH_ALS_KERNELS_ONLY   // scripts/als_asm_stats.sh compiles single kernels out of this header
class AlsHandle : public HandleBase {
 public:
    struct WorkList;
    ~AlsHandle() override {
        unpin_host();
        if (stream) (void)hipStreamDestroy(stream);
    }

    bool init(const char* opt_path) {
        std::string err;
        if (!opt_.load(opt_path ? opt_path : "", &err)) {
            last_error = err;
            return false;
        }
        BFH_HIP(hipSetDevice(device));
        if (!stream) BFH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        hipDeviceProp_t prop;
        BFH_HIP(hipGetDeviceProperties(&prop, device));
        num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        d_ = opt_.integer("d");
        BFH_REQUIRE(d_ > 0, "option d must be positive");
        vdim_ = vdim_of(d_);
        BFH_REQUIRE(vdim_ <= 1024, "d > 1024 is not supported by the gfx950 kernels yet");
        alpha_ = static_cast<float>(opt_.num("alpha"));
        reg_u_ = static_cast<float>(opt_.num("reg_u"));
        reg_i_ = static_cast<float>(opt_.num("reg_i"));
        adaptive_reg_ = opt_.boolean_or("adaptive_reg", false);
        compute_loss_ = opt_.boolean_or("compute_loss_on_training", false);
        eps_ = static_cast<float>(opt_.num_or("eps", 1e-10));
        cg_tol_ = static_cast<float>(opt_.num_or("cg_tolerance", 1e-10));
        num_cg_max_iters_ = static_cast<int>(opt_.num_or("num_cg_max_iters", 3));
        block_size_ = static_cast<int>(opt_.num_or("block_size", 32));
        BFH_REQUIRE(block_size_ > 0, "block_size must be positive");
        std::string optimizer = opt_.str("optimizer");
        if (d_ >= 128) optimizer = "ialspp";  // als.cc:46 (Q-13)
        if (optimizer == "llt") code_ = 0;
        else if (optimizer == "ldlt") code_ = 1;
        else if (optimizer == "manual_cg") code_ = 2;
        else if (optimizer == "ialspp") code_ = 8;
        else throw Error(BFH_ERR_UNSUPPORTED, "optimizer '" + optimizer + "' is not implemented on gfx950 (supported: llt, ldlt, manual_cg, ialspp)");
        FF_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        FF64_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        loss_.resize(2, true, stream);
        ticket_.resize(1, true, stream);
        inited_ = true;
        BFH_HIP(hipStreamSynchronize(stream));
        return true;
    }

    void initialize_model(float* P, int P_rows, float* Q, int Q_rows) {
        BFH_REQUIRE(inited_, "initialize_model called before init");
        BFH_REQUIRE(P && Q && P_rows > 0 && Q_rows > 0, "initialize_model: null factors or empty shapes");
        hostP_ = P; hostQ_ = Q; P_rows_ = P_rows; Q_rows_ = Q_rows;
        const size_t np = static_cast<size_t>(P_rows) * vdim_, nq = static_cast<size_t>(Q_rows) * vdim_;
        unpin_host();
        if (pin_host_) {   // opt-in ("pin_host" = 1): page-lock the caller's arrays; the default goes through the library's own pinned ring
            for (auto pr : {std::make_pair(static_cast<void*>(P), np * sizeof(float)), std::make_pair(static_cast<void*>(Q), nq * sizeof(float))}) {
                if (pr.second < (size_t(1) << 20)) continue;   // small arrays share heap pages with other objects: see SgdHandle::initialize_model
                if (hipHostRegister(pr.first, pr.second, hipHostRegisterDefault) == hipSuccess) pinned_.push_back(pr.first);
                else (void)hipGetLastError();
            }
        }
        P_.resize(np); Q_.resize(nq);
        BFH_HIP(hipMemcpyAsync(P_.get(), P, np * sizeof(float), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(Q_.get(), Q, nq * sizeof(float), hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += static_cast<double>((np + nq) * sizeof(float));
        BFH_HIP(hipStreamSynchronize(stream));
        ++fver_[0]; ++fver_[1];
        model_ = true;
    }

    void set_placeholder(const int64_t* lindptr, const int64_t* rindptr, size_t batch_size) {
        BFH_REQUIRE(model_, "set_placeholder called before initialize_model");
        BFH_REQUIRE(lindptr && rindptr, "set_placeholder: null indptr");
        const int64_t* ip[2] = {lindptr, rindptr};
        const int rows[2] = {P_rows_, Q_rows_};
        for (int a = 0; a < 2; ++a) {
            ax_[a].indptr_host.assign(ip[a], ip[a] + rows[a]);
            ax_[a].indptr.resize(rows[a]);
            BFH_HIP(hipMemcpyAsync(ax_[a].indptr.get(), ip[a], rows[a] * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        }
        keys_.resize(batch_size);
        vals_.resize(batch_size);
        yui_.resize(batch_size);
        ax_[0].chunks.clear();
        ax_[1].chunks.clear();
        BFH_HIP(hipStreamSynchronize(stream));
        work_cache_.clear();
        placeholder_ = true;
    }

    void set_resident_csr(int axis, const int64_t* indptr, const int32_t* keys, const float* vals, int64_t nnz) {
        BFH_REQUIRE(model_, "set_resident_csr called before initialize_model");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        BFH_REQUIRE(indptr && keys && vals, "set_resident_csr: null arrays");
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        BFH_REQUIRE(indptr[rows - 1] == nnz, "set_resident_csr: indptr[-1] != nnz");
        Axis& A = ax_[axis];
        A.indptr_host.assign(indptr, indptr + rows);
        A.indptr.resize(rows);
        A.keys.resize(static_cast<size_t>(nnz));
        A.vals.resize(static_cast<size_t>(nnz));
        BFH_HIP(hipMemcpyAsync(A.indptr.get(), indptr, rows * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(A.keys.get(), keys, nnz * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(A.vals.get(), vals, nnz * sizeof(float), hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += static_cast<double>(rows * sizeof(int64_t) + nnz * 8);
        if (yui_.size() < static_cast<size_t>(nnz)) yui_.resize(static_cast<size_t>(nnz));
        BFH_HIP(hipStreamSynchronize(stream));
        work_cache_.clear();
        ++vals_ver_;
        A.resident = true;
    }

    void precompute(int axis) {
        BFH_REQUIRE(model_, "precompute before initialize_model");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        gramian_of(axis == 0 ? Q_.get() : P_.get(), axis == 0 ? Q_rows_ : P_rows_);
        // a new half-epoch: whatever was derived from the other factor (its interleaved copy, the split scale) is rebuilt by the next
        // partial_update -- the factor may have been written through a device pointer handed out earlier (row exchange of a sharded
        // run), which no version counter of this handle sees; the chunks of ONE half-epoch still share the copy
        qi_side_ = -1;
    }
    // FF = F^T F for a device matrix [rows, vdim]
    void gramian_of(const float* F, int rows) {
        BFH_HIP(hipMemsetAsync(FF64_.get(), 0, FF64_.bytes(), stream));
        const int T = vdim_ / 32;
        constexpr int NT = 4;
        const int TG = (T + NT - 1) / NT;
        // waves per CU: 4 at vdim 128 (configs[2]: 0.157 instead of 0.229 ms per epoch, every d = 128 parity case unchanged), 8 elsewhere (see gram_waves_per_cu_)
        const int wpc = gram_waves_per_cu_ > 0 ? gram_waves_per_cu_ : (vdim_ == 128 ? 4 : 8);
        int slices = (num_cus_ * wpc) / (T * TG);
        if (slices < 1) slices = 1;
        int rps = (rows + slices - 1) / slices;
        rps = (rps + 1) & ~1;  // even: row pairs never straddle slices
        if (rps < 2) rps = 2;
        slices = (rows + rps - 1) / rps;
        const int slot = t_aux_.begin(stream);
        if (gram_upg_ == 4) hipLaunchKernelGGL((als_gramian_kernel<NT, 4>), dim3(slices, T, TG), dim3(64), 0, stream, F, rows, vdim_, rps, FF64_.get());
        else hipLaunchKernelGGL((als_gramian_kernel<NT, 8>), dim3(slices, T, TG), dim3(64), 0, stream, F, rows, vdim_, rps, FF64_.get());
        BFH_HIP(hipGetLastError());
        const int nff = vdim_ * vdim_;
        hipLaunchKernelGGL(als_gramian_round_kernel, dim3((nff + 255) / 256), dim3(256), 0, stream, FF64_.get(), FF_.get(), nff);
        BFH_HIP(hipGetLastError());
        t_aux_.end(slot, stream);
        // (no synchronisation here since round 6: nothing of the Gramian is read by the host, the next call on the stream waits for it anyway, and a
        //  blocking call per precompute was ~30 us of the epoch; the timer is drained where the stream is idle next -- partial_update, get_stats)
    }
    // stream idle: account what the aux timer holds
    void drain_aux() { stats.aux_ms += t_aux_.drain(); }
    // bfh_*_get_stats: everything queued so far is part of the numbers
    void flush_timers() {
        if (stream) BFH_HIP(hipStreamSynchronize(stream));
        drain_aux();
    }

    void partial_update(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals, int axis,
                        double* nume, double* deno) {
        BFH_REQUIRE(model_, "partial_update before initialize_model");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        Axis& A = ax_[axis];
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        BFH_REQUIRE(A.resident || placeholder_, "partial_update before set_placeholder");
        BFH_REQUIRE(0 <= start_x && start_x <= next_x && next_x <= rows, "partial_update: bad row range");
        *nume = 0.0;
        *deno = 0.0;
        if (next_x == start_x) return;  // als.cc:219-222
        const int64_t* ip = indptr ? indptr : A.indptr_host.data();
        const int64_t beg = start_x == 0 ? 0 : ip[start_x - 1];
        const int64_t end = ip[next_x - 1];
        const int64_t n = end - beg;
        AlsParams p{};
        p.P = axis == 0 ? P_.get() : Q_.get();
        p.Q = axis == 0 ? Q_.get() : P_.get();
        p.FF = FF_.get();
        p.indptr = A.indptr.get();
        p.shift = beg;
        p.start_x = start_x; p.next_x = next_x;
        p.d = d_; p.vdim = vdim_;
        p.op_rows = axis == 0 ? Q_rows_ : P_rows_;
        p.block_size = block_size_;
        p.alpha = alpha_;
        p.reg = axis == 0 ? reg_u_ : reg_i_;
        p.eps = eps_; p.cg_tol = cg_tol_;
        p.adaptive_reg = adaptive_reg_; p.compute_loss = compute_loss_; p.axis = axis;
        p.num_cg_max_iters = num_cg_max_iters_;
        p.loss = loss_.get();
        p.ticket = ticket_.get();
        p.debug = debug_;
        p.solver = static_cast<int>(code_);
        p.out_scale = 1.0f;
        p.ff_scale = 1.0f;
        if (A.resident) {
            p.keys = A.keys.get() + beg;
            p.vals = A.vals.get() + beg;
            p.yui = yui_.get();
        } else if (auto_resident_ && keys && vals && !A.indptr_host.empty()) {
            // the reference hands keys / vals over on every call (cuda/_als.pyx:52-67): a chunk seen before -- same row range,
            // same length, same 64-bit hash over both host buffers -- is served from its place in a full-size device copy
            const int64_t total = A.indptr_host.back();
            BFH_REQUIRE(end <= total, "partial_update: indptr disagrees with the placeholder's");
            if (A.keys.size() < static_cast<size_t>(total)) {
                A.keys.resize(static_cast<size_t>(total));
                A.vals.resize(static_cast<size_t>(total));
                A.chunks.clear();
            }
            if (yui_.size() < static_cast<size_t>(n)) yui_.resize(static_cast<size_t>(n));
            const uint64_t sig = content_signature(keys, n) * 31u + content_signature(reinterpret_cast<const int32_t*>(vals), n);
            auto it = A.chunks.find({start_x, next_x});
            if (it == A.chunks.end() || it->second.first != n || it->second.second != sig) {
                if (n) {
                    BFH_HIP(hipMemcpyAsync(A.keys.get() + beg, keys, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                    BFH_HIP(hipMemcpyAsync(A.vals.get() + beg, vals, n * sizeof(float), hipMemcpyHostToDevice, stream));
                    stats.h2d_bytes += static_cast<double>(n * 8);
                    ++vals_ver_;
                }
                A.chunks[{start_x, next_x}] = {n, sig};
            }
            p.keys = A.keys.get() + beg;
            p.vals = A.vals.get() + beg;
            p.yui = yui_.get();
        } else {
            BFH_REQUIRE(keys && vals, "partial_update: keys/vals == NULL needs bfh_als_set_resident_csr first");
            BFH_REQUIRE(static_cast<size_t>(n) <= keys_.size(), "partial_update: chunk larger than the placeholder batch_size");
            if (n) {
                BFH_HIP(hipMemcpyAsync(keys_.get(), keys, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                BFH_HIP(hipMemcpyAsync(vals_.get(), vals, n * sizeof(float), hipMemcpyHostToDevice, stream));
                stats.h2d_bytes += static_cast<double>(n * 8);
            }
            ++vals_ver_;
            p.keys = keys_.get();
            p.vals = vals_.get();
            p.yui = yui_.get();
        }
        BFH_HIP(hipMemsetAsync(loss_.get(), 0, 2 * sizeof(double), stream));
        BFH_HIP(hipMemsetAsync(ticket_.get(), 0, sizeof(int), stream));
        const int nrows = next_x - start_x;
        const int K = (vdim_ + 63) / 64;
        const bool gram_path = vdim_ <= 128 && !force_v1_;
        const bool big = static_cast<uint64_t>(p.op_rows) * vdim_ * 4 >= (1ull << 32);   // 64-bit gather offsets into the other factor
        // 128 < vdim <= 256: block-per-row kernel with the tiles spread over ceil(T/2) waves (als_wide_kernel)
        const bool wide_path = !gram_path && !force_v1_ && vdim_ > 128 && vdim_ <= 256 && code_ == 8 && block_size_ == 32 && d_ == vdim_;
        WorkList* wl = nullptr;
        bool pc_launched = false;
        if (gram_path || wide_path) {
            wl = &work_list(axis, start_x, next_x, ip, beg);
            if (wl->n_heavy) BFH_HIP(hipMemsetAsync(scratch_.get(), 0, static_cast<size_t>(wl->n_heavy) * als_slot_floats(vdim_) * sizeof(float), stream));
        }
        const int slot = t_main_.begin(stream);
        if (gram_path) {
            // wave-per-row Gramian pass; rows are solved from the accumulators (iALS++, block_size 32, d == vdim) or
            // go through an HBM scratch slot to the dense-solve kernel (see als_gram_kernel)
            const int T = vdim_ / 32;
            bool inreg = code_ == 8 && block_size_ == 32 && d_ == vdim_ && !no_inreg_;
            const size_t per_row = als_slot_floats(vdim_);
            const size_t lds_h = als_gs_lds_bytes(vdim_);
            BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(als_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(lds_h)));
            const int items = wl->n_work;
            int blocks = (items + 3) / 4;                           // 4 independent waves per block, one work item each
            if (blocks > num_cus_ * 4) blocks = num_cus_ * 4;       // persistent: residency is set by the kernel's VGPR count
            // producer / consumer pairs (als_pc.hpp): the default for the in-place iALS++ rows at d = 64 / 96 / 128
            // (measured on the ML-20M shape, profiles/r04_als_pc_steps.txt: d = 128 4.53 vs 5.08 ms, d = 96 3.47 vs 3.95, d = 64 2.47 vs 2.06 --
            //  at T = 2 round 3's kernel already runs two waves per SIMD, and the pairs only add their hand-off: "als_pc" = 2 forces them)
            bool use_pc = items > 0 && inreg && split_f16_ && T <= 4 && (pc_ >= 2 ? T >= 2 : (pc_ == 1 && T >= 3));
            if (use_pc) {
                float* const before = scratch_.get();
                scan_deferred(*wl, p, items);
                // the scan may GROW scratch_ (a new buffer, neither copied nor zeroed): the heavy rows' slots zeroed above are then gone
                if (scratch_.get() != before && wl->n_heavy)
                    BFH_HIP(hipMemsetAsync(scratch_.get(), 0, static_cast<size_t>(wl->n_heavy) * als_slot_floats(vdim_) * sizeof(float), stream));
                if (wl->n_def_rows > 4096 || wl->n_def * 4 > items) {
                    // weights mostly outside the f16 path (negative confidences, ...): every row of the call takes the route the flagged
                    // ones would take -- fp32 instruction, scratch slot, dense-solve kernel (the branch below)
                    use_pc = false;
                    inreg = false;
                }
            }
            if (items > 0 && inreg && split_f16_ && T >= 2) {   // the scale of the split pass, decided on the device (no host round trip)
                const int oside = axis == 0 ? 1 : 0;   // which factor is "the other side"
                if (split_out_.size() < 4) { split_part_.resize(ALS_STAT_BLOCKS); split_out_.resize(4); }
                if (use_pc) {
                    // ... together with the block-interleaved copy of the other factor the producers gather from; both are kept while that
                    // factor does not change (the chunks of one half-epoch share them)
                    const size_t nq = static_cast<size_t>(p.op_rows) * vdim_;
                    if (qi_.size() < nq) { qi_.resize(nq); qi_side_ = -1; }
                    if (qi_side_ != oside || qi_ver_ != fver_[oside] || qi_wcut_ != split_wcut_) {
                        if (T == 2) hipLaunchKernelGGL(als_interleave_stats_kernel<2>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                        else if (T == 3) hipLaunchKernelGGL(als_interleave_stats_kernel<3>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                        else hipLaunchKernelGGL(als_interleave_stats_kernel<4>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                        hipLaunchKernelGGL(als_split_scale_kernel, dim3(1), dim3(64), 0, stream, split_part_.get(), ALS_STAT_BLOCKS, split_wcut_, split_out_.get());
                        BFH_HIP(hipGetLastError());
                        qi_side_ = oside; qi_ver_ = fver_[oside]; qi_wcut_ = split_wcut_;
                    }
                } else {
                    hipLaunchKernelGGL(als_split_stats_kernel, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows) * vdim_,
                                       split_part_.get());
                    hipLaunchKernelGGL(als_split_scale_kernel, dim3(1), dim3(64), 0, stream, split_part_.get(), ALS_STAT_BLOCKS, split_wcut_, split_out_.get());
                    BFH_HIP(hipGetLastError());
                    qi_side_ = -1;   // split_out_ was rewritten for another matrix
                }
                p.split = split_out_.get();
                {   // FF p0 for every row of the call
                    const size_t need0 = static_cast<size_t>(nrows) * vdim_;
                    if (rowff_.size() < need0) rowff_.resize(need0);
                    const int quads = (nrows + 3) / 4;
                    const int rb = std::max(1, std::min((quads + 3) / 4, num_cus_ * 8));
                    if (T == 2) hipLaunchKernelGGL(als_rowff_kernel<2>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                    else if (T == 3) hipLaunchKernelGGL(als_rowff_kernel<3>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                    else hipLaunchKernelGGL(als_rowff_kernel<4>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                    BFH_HIP(hipGetLastError());
                    p.F0 = rowff_.get();
                }
                p.batch = 16;   // rows per ticket at most (fewer where the rows are long)
            }
            if (use_pc) {
                if (pc_err_.size() < 8) pc_err_.resize(8);   // [0] error bits, [1] placement statistic, [2..5] the clock probe of workgroup 0 (als_debug bit 1024)
                BFH_HIP(hipMemsetAsync(pc_err_.get(), 0, 8 * sizeof(int), stream));
                const int nslots = wl->n_heavy + wl->n_def_rows;
                if (wl->n_def_rows)   // (the heavy rows' slots were zeroed above)
                    BFH_HIP(hipMemsetAsync(scratch_.get() + static_cast<size_t>(wl->n_heavy) * per_row, 0, static_cast<size_t>(wl->n_def_rows) * per_row * sizeof(float), stream));
                (void)nslots;
                const int pblocks = std::max(1, std::min((items + 3) / 4, num_cus_));
                const bool lk = compute_loss_ && axis == 1;
#define BFH_PC(TT, BG, LS)                                                                                                          \
    do {                                                                                                                            \
        BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(als_pc_kernel<TT, BG, LS>), hipFuncAttributeMaxDynamicSharedMemorySize, AlsPc<TT>::LDS_B)); \
        hipLaunchKernelGGL((als_pc_kernel<TT, BG, LS>), dim3(pblocks), dim3(512), AlsPc<TT>::LDS_B, stream, p, wl->work.get(), items, scratch_.get(), \
                           qi_.get(), wl->defer.get(), pc_err_.get());                                                              \
    } while (0)
#define BFH_PC_T(TT)                                 \
    do {                                             \
        if (big) { if (lk) BFH_PC(TT, true, true); else BFH_PC(TT, true, false); }     \
        else { if (lk) BFH_PC(TT, false, true); else BFH_PC(TT, false, false); }       \
    } while (0)
                if (T == 2) BFH_PC_T(2);
                else if (T == 3) BFH_PC_T(3);
                else BFH_PC_T(4);
#undef BFH_PC_T
#undef BFH_PC
                BFH_HIP(hipGetLastError());
                pc_launched = true;
                if (wl->n_def > 0) {   // items with weights outside the f16 path: fp32 instruction, tiles into their scratch slots
                    BFH_HIP(hipMemsetAsync(ticket_.get(), 0, sizeof(int), stream));
                    const int dblocks = std::max(1, std::min((wl->n_def + 3) / 4, num_cus_ * 4));
#define BFH_GD(TT)                                                                                                                  \
    do {                                                                                                                            \
        if (big) hipLaunchKernelGGL((als_gram_kernel<TT, true, false, true>), dim3(dblocks), dim3(256), 0, stream, p, wl->dlist.get(), wl->n_def, scratch_.get(), 0); \
        else hipLaunchKernelGGL((als_gram_kernel<TT, true, false, false>), dim3(dblocks), dim3(256), 0, stream, p, wl->dlist.get(), wl->n_def, scratch_.get(), 0);   \
    } while (0)
                    if (T == 2) BFH_GD(2);
                    else if (T == 3) BFH_GD(3);
                    else BFH_GD(4);
#undef BFH_GD
                    BFH_HIP(hipGetLastError());
                }
                if (wl->n_heavy)
                    hipLaunchKernelGGL(als_solve_kernel, dim3(wl->n_heavy), dim3(256), lds_h, stream, p, wl->heavy.get(), wl->n_heavy, scratch_.get(),
                                       static_cast<int>(code_));
                if (wl->n_def_rows)
                    hipLaunchKernelGGL(als_solve_kernel, dim3(wl->n_def_rows), dim3(256), lds_h, stream, p, wl->dsolve.get(), wl->n_def_rows, scratch_.get(),
                                       static_cast<int>(code_));
                BFH_HIP(hipGetLastError());
            } else
            if (items > 0 && inreg) {
#define BFH_GK(TT, SP)                                                                                                              \
    do {                                                                                                                            \
        if (big) hipLaunchKernelGGL((als_gram_kernel<TT, true, true, true, true, SP>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, scratch_.get(), 0); \
        else if (compute_loss_ && axis == 1) hipLaunchKernelGGL((als_gram_kernel<TT, true, true, false, true, SP>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, scratch_.get(), 0);   \
        else hipLaunchKernelGGL((als_gram_kernel<TT, true, true, false, false, SP>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, scratch_.get(), 0);   \
    } while (0)
                // als_split_f16 (default on, d >= 64): the Gramian through the f16 matrix cores at fp32 accuracy (see als_gram_kernel)
                if (T <= 1) BFH_GK(1, false);
                else if (T <= 2) { if (split_f16_) BFH_GK(2, true); else BFH_GK(2, false); }
                else if (T <= 3) { if (split_f16_) BFH_GK(3, true); else BFH_GK(3, false); }
                else { if (split_f16_) BFH_GK(4, true); else BFH_GK(4, false); }
#undef BFH_GK
                BFH_HIP(hipGetLastError());
                if (wl->n_heavy)   // heavy rows: chunk partials were summed in scratch_ (zeroed above)
                    hipLaunchKernelGGL(als_solve_kernel, dim3(wl->n_heavy), dim3(256), lds_h, stream, p, wl->heavy.get(), wl->n_heavy, scratch_.get(),
                                       static_cast<int>(code_));
                BFH_HIP(hipGetLastError());
            } else if (items > 0) {
                // scratch: one slot per row of the chunk, then one (zeroed) accumulation slot per heavy row
                const size_t need = (static_cast<size_t>(nrows) + wl->n_heavy) * per_row;
                if (gscratch_.size() < need) gscratch_.resize(need);
                if (wl->n_heavy)
                    BFH_HIP(hipMemsetAsync(gscratch_.get() + static_cast<size_t>(nrows) * per_row, 0, wl->n_heavy * per_row * sizeof(float), stream));
#define BFH_GK(TT)                                                                                                                  \
    do {                                                                                                                            \
        if (big) {                                                                                                                  \
            if (code_ == 8) hipLaunchKernelGGL((als_gram_kernel<TT, true, false, true>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, gscratch_.get(), nrows); \
            else hipLaunchKernelGGL((als_gram_kernel<TT, false, false, true>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, gscratch_.get(), nrows);          \
        } else if (code_ == 8) hipLaunchKernelGGL((als_gram_kernel<TT, true, false, false>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, gscratch_.get(), nrows); \
        else hipLaunchKernelGGL((als_gram_kernel<TT, false, false, false>), dim3(blocks), dim3(256), 0, stream, p, wl->work.get(), items, gscratch_.get(), nrows);          \
    } while (0)
                if (T <= 1) BFH_GK(1);
                else if (T <= 2) BFH_GK(2);
                else if (T <= 3) BFH_GK(3);
                else BFH_GK(4);
#undef BFH_GK
                BFH_HIP(hipGetLastError());
                int sblocks = num_cus_ * static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_h)));
                if (sblocks > wl->n_solve) sblocks = wl->n_solve;
                hipLaunchKernelGGL(als_solve_kernel, dim3(sblocks), dim3(256), lds_h, stream, p, wl->solve.get(), wl->n_solve, gscratch_.get(),
                                   static_cast<int>(code_));
                BFH_HIP(hipGetLastError());
            }
        } else if (wide_path) {
            const int T = vdim_ / 32;
            const size_t lds = als_wide_lds_bytes(vdim_);
            int blocks = std::min(wl->n_work, num_cus_ * 2);
            // "als_wide_split" (default on; vdim 160 .. 224, "als_wide_split_max_t"): the Gramian through the f16 matrix cores at fp32 accuracy, rows gathered once
            // per block by a producer wave (als_wide_item<SPLIT>); from T = 6 up with the fourth product l l (round 5 kept d = 192 on the fp32 form because
            // the three-product form put one ill-conditioned tiny case at 5.9x the oracle's distance from float64: with l l it lands at 3.8x, bound 4x).
            // For calls whose weights all fit the f16 path; als_defer_scan_kernel says so (cached per chunk while the values do not change)
            bool wsplit = wide_split_ && split_f16_ && T >= 5 && T <= wide_split_max_t_ && wl->n_work > 0;
            if (wsplit) {
                float* const before = scratch_.get();
                scan_deferred(*wl, p, wl->n_work);
                // the scan may GROW scratch_ (a new buffer, neither copied nor zeroed): the heavy rows' slots zeroed above are then gone -- whichever
                // instantiation runs below (round 5 re-zeroed them only when no item was deferred: a call with heavy rows AND weights outside the
                // f16 path summed its chunk tiles into uninitialised slots on the first growth)
                if (scratch_.get() != before && wl->n_heavy)
                    BFH_HIP(hipMemsetAsync(scratch_.get(), 0, static_cast<size_t>(wl->n_heavy) * als_slot_floats(vdim_) * sizeof(float), stream));
                if (wl->n_def > 0) wsplit = false;
            }
            if (wsplit) {   // the scale of the split pass, decided on the device (als_gram_kernel's rule), together with the block-interleaved copy of
                // the other factor the producers gather from; both are kept while that factor does not change (the chunks of one half-epoch share them)
                if (split_out_.size() < 4) { split_part_.resize(ALS_STAT_BLOCKS); split_out_.resize(4); }
                const int oside = axis == 0 ? 1 : 0;
                const size_t nq = static_cast<size_t>(p.op_rows) * vdim_;
                if (qi_.size() < nq) { qi_.resize(nq); qi_side_ = -1; }
                if (qi_side_ != oside || qi_ver_ != fver_[oside] || qi_wcut_ != split_wcut_) {
                    if (T == 5) hipLaunchKernelGGL(als_interleave_stats_kernel<5>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                    else if (T == 6) hipLaunchKernelGGL(als_interleave_stats_kernel<6>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                    else if (T == 7) hipLaunchKernelGGL(als_interleave_stats_kernel<7>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                    else hipLaunchKernelGGL(als_interleave_stats_kernel<8>, dim3(ALS_STAT_BLOCKS), dim3(256), 0, stream, p.Q, static_cast<size_t>(p.op_rows), qi_.get(), split_part_.get());
                    hipLaunchKernelGGL(als_split_scale_kernel, dim3(1), dim3(64), 0, stream, split_part_.get(), ALS_STAT_BLOCKS, split_wcut_, split_out_.get());
                    BFH_HIP(hipGetLastError());
                    qi_side_ = oside; qi_ver_ = fver_[oside]; qi_wcut_ = split_wcut_;
                }
                p.split = split_out_.get();
                p.Qi = qi_.get();
            }
#define BFH_WIDE_L(TT, BG, SP, ITEMS, N, FIN)                                                                                    \
    hipLaunchKernelGGL((als_wide_kernel<TT, BG, SP>), dim3(std::max(1, std::min(N, num_cus_ * als_wide_blocks_per_cu(TT, SP)))), dim3(64 * ((TT + 1) / 2 + (SP ? 1 : 0))), als_wide_lds_bytes(vdim_, SP), stream, p, ITEMS, N, scratch_.get(), FIN)
#define BFH_WIDE(TT, ITEMS, N, FIN)                                                                                              \
    do {                                                                                                                         \
        if (big) BFH_WIDE_L(TT, true, false, ITEMS, N, FIN);                                                                     \
        else BFH_WIDE_L(TT, false, false, ITEMS, N, FIN);                                                                        \
    } while (0)
#define BFH_WIDE_S(TT, ITEMS, N, FIN)                                                                                            \
    do {                                                                                                                         \
        if (big) BFH_WIDE_L(TT, true, true, ITEMS, N, FIN);                                                                      \
        else BFH_WIDE_L(TT, false, true, ITEMS, N, FIN);                                                                         \
    } while (0)
#define BFH_WIDE_T(ITEMS, N, FIN)                  \
    do {                                           \
        if (T == 5) { if (wsplit) BFH_WIDE_S(5, ITEMS, N, FIN); else BFH_WIDE(5, ITEMS, N, FIN); }    \
        else if (T == 6) { if (wsplit) BFH_WIDE_S(6, ITEMS, N, FIN); else BFH_WIDE(6, ITEMS, N, FIN); } \
        else if (T == 7) { if (wsplit) BFH_WIDE_S(7, ITEMS, N, FIN); else BFH_WIDE(7, ITEMS, N, FIN); } \
        else { if (wsplit) BFH_WIDE_S(8, ITEMS, N, FIN); else BFH_WIDE(8, ITEMS, N, FIN); }           \
    } while (0)
            (void)blocks;
            {   // FF p0 for every row of the call (the residual-first gradient starts from it, als_wide_item)
                const size_t need0 = static_cast<size_t>(nrows) * vdim_;
                if (rowff_.size() < need0) rowff_.resize(need0);
                const int quads = (nrows + 3) / 4;
                const int rb = std::max(1, std::min((quads + 3) / 4, num_cus_ * 8));
                if (T == 5) hipLaunchKernelGGL(als_rowff_kernel<5>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                else if (T == 6) hipLaunchKernelGGL(als_rowff_kernel<6>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                else if (T == 7) hipLaunchKernelGGL(als_rowff_kernel<7>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                else hipLaunchKernelGGL(als_rowff_kernel<8>, dim3(rb), dim3(256), 0, stream, p.P, start_x, nrows, FF_.get(), rowff_.get());
                BFH_HIP(hipGetLastError());
                p.F0 = rowff_.get();
            }
            if (wl->n_work > 0) BFH_WIDE_T(wl->work.get(), wl->n_work, 0);
            BFH_HIP(hipGetLastError());
            if (wl->n_heavy) {   // heavy rows: FF + summed chunk tiles -> solve
                BFH_HIP(hipMemsetAsync(ticket_.get(), 0, sizeof(int), stream));
                BFH_WIDE_T(wl->heavy_work.get(), wl->n_heavy, 1);
                BFH_HIP(hipGetLastError());
            }
#undef BFH_WIDE_T
#undef BFH_WIDE_S
#undef BFH_WIDE
#undef BFH_WIDE_L
        } else if (code_ == 8) {
            const int bs = block_size_ < d_ ? block_size_ : d_;
            const int KB = (bs + 63) / 64;
            int waves = num_cus_ * 16;
            if (waves > nrows) waves = nrows;
            dim3 grid((waves + 3) / 4), block(256);
            launch_ialspp(K, KB, grid, block, p);
        } else if (code_ == 2) {
            int waves = num_cus_ * 16;
            if (waves > nrows) waves = nrows;
            dim3 grid((waves + 3) / 4), block(256);
            if (K <= 1) hipLaunchKernelGGL(als_cg_kernel<1>, grid, block, 0, stream, p);
            else if (K <= 2) hipLaunchKernelGGL(als_cg_kernel<2>, grid, block, 0, stream, p);
            else throw Error(BFH_ERR_UNSUPPORTED, "manual_cg path expects d < 128");
        } else {
            int blocks = num_cus_ * 4;
            if (blocks > nrows) blocks = nrows;
            const size_t lds = (static_cast<size_t>(vdim_) * (vdim_ + 1) + vdim_) * sizeof(float);
            if (K <= 1) hipLaunchKernelGGL(als_chol_kernel<1>, dim3(blocks), dim3(64), lds, stream, p);
            else if (K <= 2) hipLaunchKernelGGL(als_chol_kernel<2>, dim3(blocks), dim3(64), lds, stream, p);
            else throw Error(BFH_ERR_UNSUPPORTED, "llt/ldlt path expects d < 128");
        }
        BFH_HIP(hipGetLastError());
        t_main_.end(slot, stream);
        double l[2] = {0, 0};
        if (compute_loss_) BFH_HIP(hipMemcpyAsync(l, loss_.get(), 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
        int pe[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (pc_launched) BFH_HIP(hipMemcpyAsync(pe, pc_err_.get(), 8 * sizeof(int), hipMemcpyDeviceToHost, stream));
        if (writeback_) {  // als.cu:403: updated rows go back to the caller's array
            float* hostF = axis == 0 ? hostP_ : hostQ_;
            const size_t off = static_cast<size_t>(start_x) * vdim_, cnt = static_cast<size_t>(nrows) * vdim_;
            copy_out(hostF + off, p.P + off, cnt * sizeof(float));
        }
        BFH_HIP(hipStreamSynchronize(stream));
        ++fver_[axis];   // the side just solved changed
        drain_aux();
        stats.kernel_ms += t_main_.drain();
        stats.launches += 1;
        stats.samples += n;
        if (pc_launched) {
            pc_same_simd_ = pe[1];
            {   // als_debug bit 1024: shader clock of workgroup 0 over the kernel = s_memtime ticks per 100 MHz s_memrealtime tick
                unsigned long long core = 0, real = 0;
                std::memcpy(&core, pe + 2, 8);
                std::memcpy(&real, pe + 4, 8);
                pc_clock_mhz_ = real ? static_cast<int>(100.0 * static_cast<double>(core) / static_cast<double>(real)) : 0;
            }
            if (pe[0] & 1) throw Error(BFH_ERR_HIP, "als_pc_kernel: a producer / consumer hand-off timed out (results of this call are invalid)");
            if (pe[0] & 2) throw Error(BFH_ERR_HIP, "als_pc_kernel: a weight outside the f16 path reached the kernel (stale weight scan)");
        }
        *nume = l[0];
        *deno = l[1];
    }

    // Which work items hold weights the split pass cannot carry (als_defer_scan_kernel)?  Depends on the chunk's values only, so it is
    // kept with the work list and redone when values were uploaded since (or the cut moved); the one host round trip it costs buys
    // launch shapes the host knows.
    void scan_deferred(WorkList& wl, const AlsParams& p, int items) {
        if (wl.scan_ver == vals_ver_ && wl.scan_wcut == split_wcut_ && wl.scan_vals == p.vals) return;
        if (wl.defer.size() < static_cast<size_t>(items)) {
            wl.defer.resize(items);
            wl.dlist.resize(items);
            wl.dsolve.resize(items);
            wl.dcount.resize(2);
        }
        BFH_HIP(hipMemsetAsync(wl.dcount.get(), 0, 2 * sizeof(int), stream));
        hipLaunchKernelGGL(als_defer_scan_kernel, dim3((items + 3) / 4), dim3(256), 0, stream, wl.work.get(), items, p.vals, p.alpha, split_wcut_, wl.n_heavy,
                           wl.defer.get(), wl.dlist.get(), wl.dsolve.get(), wl.dcount.get());
        BFH_HIP(hipGetLastError());
        int c[2] = {0, 0};
        BFH_HIP(hipMemcpyAsync(c, wl.dcount.get(), 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        wl.n_def = c[0];
        wl.n_def_rows = c[1];
        wl.scan_ver = vals_ver_;
        wl.scan_wcut = split_wcut_;
        wl.scan_vals = p.vals;
        const size_t need = std::max<size_t>(1, static_cast<size_t>(wl.n_heavy) + (wl.n_def_rows <= 4096 ? wl.n_def_rows : 0)) * als_slot_floats(vdim_);
        if (scratch_.size() < need) {   // (grow-only; the heavy rows' slots are zeroed by the caller before every use)
            scratch_.resize(need);
        }
    }

    struct WorkList {
        DevBuf<AlsWork> work;
        DevBuf<AlsHeavy> heavy;   // fused kernels: heavy rows only (slot = scratch slot)
        DevBuf<AlsHeavy> solve;   // split design: every non-empty row, longest first (slot = row - start_x)
        DevBuf<AlsWork> heavy_work;   // wide kernel's finalize launch: one item per heavy row (kend - kbeg = its nnz)
        int n_work = 0, n_heavy = 0, n_solve = 0;
        // als_pc_kernel: items whose weights need the fp32 instruction (scan_deferred)
        DevBuf<int> defer;             // per work item
        DevBuf<AlsWork> dlist;         // the flagged items (whole rows with their scratch slot = n_heavy + j)
        DevBuf<AlsHeavy> dsolve;       // the flagged whole rows, for als_solve_kernel
        DevBuf<int> dcount;
        int n_def = 0, n_def_rows = 0;
        uint64_t scan_ver = ~uint64_t(0);
        float scan_wcut = -1.f;
        const float* scan_vals = nullptr;
    };
    // Work items of one partial_update call: one per non-empty row, rows above HEAVY nnz cut into
    // chunks; longest first (dynamic ticket order) so the tail is short.  Cached per (axis, range).
    WorkList& work_list(int axis, int start_x, int next_x, const int64_t* ip, int64_t shift) {
        const auto key = std::make_tuple(axis, start_x, next_x);
        auto it = work_cache_.find(key);
        if (it != work_cache_.end()) return *it->second;
        constexpr int64_t HEAVY = 4096;
        std::vector<AlsWork> w;
        std::vector<AlsHeavy> h, sv;
        w.reserve(next_x - start_x);
        sv.reserve(next_x - start_x);
        int64_t prev = start_x == 0 ? 0 : ip[start_x - 1];
        for (int x = start_x; x < next_x; ++x) {
            const int64_t e = ip[x], n = e - prev;
            if (n > 0) {  // Q-16: empty rows are left untouched
                const int64_t kb = prev - shift;
                if (n <= HEAVY) {
                    w.push_back({x, static_cast<int>(kb), static_cast<int>(kb + n), -1});
                    sv.push_back({x, x - start_x, n});
                } else {
                    const int slot = static_cast<int>(h.size());
                    sv.push_back({x, (next_x - start_x) + slot, n});   // split design: heavy slots follow the per-row slots
                    h.push_back({x, slot, n});
                    const int64_t nch = (n + HEAVY - 1) / HEAVY, per = ((n + nch - 1) / nch + 1) & ~int64_t(1);
                    for (int64_t c0 = 0; c0 < n; c0 += per)
                        w.push_back({x, static_cast<int>(kb + c0), static_cast<int>(kb + std::min(n, c0 + per)), slot});
                }
            }
            prev = e;
        }
        std::stable_sort(w.begin(), w.end(), [](const AlsWork& a, const AlsWork& b) { return (a.kend - a.kbeg) > (b.kend - b.kbeg); });
        std::stable_sort(sv.begin(), sv.end(), [](const AlsHeavy& a, const AlsHeavy& b) { return a.n > b.n; });
        auto wl = std::make_unique<WorkList>();
        wl->n_work = static_cast<int>(w.size());
        wl->n_heavy = static_cast<int>(h.size());
        wl->n_solve = static_cast<int>(sv.size());
        wl->solve.resize(std::max<size_t>(1, sv.size()));
        if (!sv.empty()) BFH_HIP(hipMemcpyAsync(wl->solve.get(), sv.data(), sv.size() * sizeof(AlsHeavy), hipMemcpyHostToDevice, stream));
        wl->work.resize(std::max<size_t>(1, w.size()));
        wl->heavy.resize(std::max<size_t>(1, h.size()));
        {
            std::vector<AlsWork> hw;
            for (const auto& hh : h) hw.push_back({hh.row, 0, static_cast<int>(hh.n), hh.slot});
            wl->heavy_work.resize(std::max<size_t>(1, hw.size()));
            if (!hw.empty()) BFH_HIP(hipMemcpyAsync(wl->heavy_work.get(), hw.data(), hw.size() * sizeof(AlsWork), hipMemcpyHostToDevice, stream));
        }
        if (!w.empty()) BFH_HIP(hipMemcpyAsync(wl->work.get(), w.data(), w.size() * sizeof(AlsWork), hipMemcpyHostToDevice, stream));
        if (!h.empty()) BFH_HIP(hipMemcpyAsync(wl->heavy.get(), h.data(), h.size() * sizeof(AlsHeavy), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        const size_t need = std::max<size_t>(1, h.size()) * als_slot_floats(vdim_);
        if (scratch_.size() < need) scratch_.resize(need);
        if (work_cache_.size() > 64) work_cache_.clear();
        return *(work_cache_[key] = std::move(wl));
    }

    void launch_ialspp(int K, int KB, dim3 grid, dim3 block, const AlsParams& p) {
#define BFH_IALS(KK, KKB) hipLaunchKernelGGL((als_ialspp_kernel<KK, KKB>), grid, block, 0, stream, p)
        if (KB <= 1) {
            if (K <= 1) BFH_IALS(1, 1);
            else if (K <= 2) BFH_IALS(2, 1);
            else if (K <= 4) BFH_IALS(4, 1);
            else if (K <= 8) BFH_IALS(8, 1);
            else BFH_IALS(16, 1);
        } else if (KB <= 2) {
            if (K <= 2) BFH_IALS(2, 2);
            else if (K <= 4) BFH_IALS(4, 2);
            else if (K <= 8) BFH_IALS(8, 2);
            else BFH_IALS(16, 2);
        } else {
            throw Error(BFH_ERR_UNSUPPORTED, "block_size > 128 is not implemented on gfx950");
        }
#undef BFH_IALS
    }

    void synchronize(bool d2h) {
        BFH_REQUIRE(model_, "synchronize before initialize_model");
        const size_t np = static_cast<size_t>(P_rows_) * vdim_, nq = static_cast<size_t>(Q_rows_) * vdim_;
        if (d2h) {
            copy_out(hostP_, P_.get(), np * sizeof(float));
            copy_out(hostQ_, Q_.get(), nq * sizeof(float));
        } else {
            BFH_HIP(hipMemcpyAsync(P_.get(), hostP_, np * sizeof(float), hipMemcpyHostToDevice, stream));
            BFH_HIP(hipMemcpyAsync(Q_.get(), hostQ_, nq * sizeof(float), hipMemcpyHostToDevice, stream));
            stats.h2d_bytes += static_cast<double>((np + nq) * sizeof(float));
            ++fver_[0]; ++fver_[1];
        }
        BFH_HIP(hipStreamSynchronize(stream));
    }

    // Multi-GPU: rows of the side being solved are sharded, both factor matrices replicated.
    // After a half-epoch in which rank r solved rows [bounds[r], bounds[r+1]) every rank receives every block: the uneven
    // all-gather as one group of ncclBroadcast calls (direct xGMI copies).  Every row is solved by exactly one rank from
    // identical inputs, so the replicas stay bit-identical to the single-GPU run.
    void publish_rows(int axis, const int* bounds, int n_bounds) {
        BFH_REQUIRE(model_, "publish_rows before initialize_model");
        BFH_REQUIRE(comm_, "publish_rows before bfh_als_set_comm");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        BFH_REQUIRE(bounds && n_bounds == comm_->size() + 1, "publish_rows: need world_size + 1 row boundaries");
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        BFH_REQUIRE(bounds[0] == 0 && bounds[n_bounds - 1] == rows, "publish_rows: boundaries must cover [0, rows)");
        for (int r = 0; r + 1 < n_bounds; ++r) BFH_REQUIRE(bounds[r] <= bounds[r + 1], "publish_rows: boundaries must ascend");   // before the group opens
        float* F = axis == 0 ? P_.get() : Q_.get();
        comm_->group_start();
        for (int r = 0; r + 1 < n_bounds; ++r) {
            const size_t cnt = static_cast<size_t>(bounds[r + 1] - bounds[r]) * vdim_;
            comm_->broadcast_bytes(F + static_cast<size_t>(bounds[r]) * vdim_, cnt * sizeof(float), r, stream);
        }
        comm_->group_end();
        BFH_HIP(hipStreamSynchronize(stream));
        ++fver_[axis];
        stats.exchanges += 1;
    }
    void set_comm(Comm* c) {
        BFH_REQUIRE(!c || c->device == device, "set_comm: the communicator lives on another device than this handle");
        comm_ = c;
    }

    void set_mode(const std::string& name, int64_t v) {
        if (name == "als_writeback") writeback_ = v != 0;
        else if (name == "auto_resident") auto_resident_ = v != 0;
        else if (name == "pin_host") pin_host_ = v != 0;
        else if (name == "als_v1") force_v1_ = v != 0;
        else if (name == "als_wide_split") wide_split_ = v != 0;         // 128 < vdim <= 192: 1 = split-f16 Gramian in als_wide_kernel (default), 0 = the fp32 instruction
        else if (name == "als_debug") debug_ = static_cast<int>(v);
        else if (name == "als_split_wcut") split_wcut_ = static_cast<float>(v);   // weights above this take the fp32 side path (default 2^15; tests lower it)
        else if (name == "als_split_f16") split_f16_ = v != 0;             // 0: the in-place iALS++ rows keep the fp32 matrix instruction
        else if (name == "als_gram_waves") gram_waves_per_cu_ = static_cast<int>(v);   // als_gramian_kernel: waves per CU (slices of the rows x tile rows)
        else if (name == "als_gram_upg") gram_upg_ = static_cast<int>(v);             // ... row pairs per trip (4 | 8)
        else if (name == "als_wide_split_max_t") wide_split_max_t_ = static_cast<int>(v);   // the split-f16 wide kernel up to vdim 32 * this (5 .. 8)
        else if (name == "als_pc") {
            BFH_REQUIRE(v >= 0 && v <= 2, "als_pc must be 0, 1 or 2");
            pc_ = static_cast<int>(v);
        }   // 0: round 3's wave-per-row split kernel; 1: producer / consumer pairs where they win (d = 96, 128); 2: also at d = 64
        else if (name == "als_inreg") no_inreg_ = v == 0;                 // 0: iALS++ rows go through the scratch + solve kernel instead of the in-register solve
        else if (name == "timing") timing = v != 0;
        else throw Error(BFH_ERR_INVALID, "unknown mode '" + name + "'");
    }

    void device_buffer(const std::string& name, void** p, size_t* bytes) {
        ++fver_[0]; ++fver_[1];   // whoever holds a raw pointer may write through it: cached views of the factors are dropped
        if (name == "als_pc_clock_mhz") { *p = nullptr; *bytes = static_cast<size_t>(pc_clock_mhz_); return; }   // als_debug bit 1024: shader clock during the last als_pc_kernel launch
        if (name == "als_pc_same_simd") { *p = nullptr; *bytes = static_cast<size_t>(pc_same_simd_); return; }   // placement statistic of the last als_pc_kernel launch
        // whoever takes a raw pointer reads it on ANOTHER stream (torch's): everything this handle has queued is finished first
        // (precompute no longer blocks: round 6)
        if (stream) BFH_HIP(hipStreamSynchronize(stream));
        if (name == "P") { *p = P_.get(); *bytes = P_.bytes(); }
        else if (name == "Q") { *p = Q_.get(); *bytes = Q_.bytes(); }
        else if (name == "FF") { *p = FF_.get(); *bytes = FF_.bytes(); }
        else throw Error(BFH_ERR_INVALID, "unknown device buffer '" + name + "'");
    }

    struct Axis {
        std::vector<int64_t> indptr_host;
        DevBuf<int64_t> indptr;
        DevBuf<int32_t> keys;
        DevBuf<float> vals;
        bool resident = false;
        std::map<std::pair<int, int>, std::pair<int64_t, uint64_t>> chunks;   // auto-residency: row range -> (length, checksum)
    };
    // device -> the caller's array: through the library's own pinned ring (HostStager, common.hpp) unless the caller asked for its arrays
    // to be registered ("pin_host" = 1); an array that is no longer mapped is an error, not a fault
    void copy_out(void* dst, const void* src_dev, size_t bytes) {
        if (!host_range_mapped(dst, bytes)) throw Error(BFH_ERR_INVALID, "the caller's factor array is no longer mapped (freed while the model still writes to it?)");
        if (!pinned_.empty()) BFH_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream));
        else stager_.d2h(dst, src_dev, bytes, stream, device);
        stats.d2h_bytes += static_cast<double>(bytes);
    }
    HostStager stager_;
    void unpin_host() {
        if (!pinned_.empty() && stream) (void)hipStreamSynchronize(stream);   // (see SgdHandle::unpin_host)
        for (void* q : pinned_) (void)hipHostUnregister(q);
        if (!pinned_.empty()) (void)hipGetLastError();
        pinned_.clear();
    }
    std::vector<void*> pinned_;
    bool auto_resident_ = true, pin_host_ = false;   // pin_host: opt-in since round 5 (HostStager, common.hpp)

    Options opt_;
    bool inited_ = false, model_ = false, placeholder_ = false, writeback_ = true;
    int d_ = 0, vdim_ = 0, P_rows_ = 0, Q_rows_ = 0, code_ = 2, num_cg_max_iters_ = 3, block_size_ = 32, num_cus_ = 256;
    float alpha_ = 0, reg_u_ = 0, reg_i_ = 0, eps_ = 1e-10f, cg_tol_ = 1e-10f;
    bool adaptive_reg_ = false, compute_loss_ = false;
    float *hostP_ = nullptr, *hostQ_ = nullptr;
    DevBuf<float> P_, Q_, FF_, vals_, yui_;
    DevBuf<double> FF64_;   // fp64 accumulator of the Gramian slices (see als_gramian_kernel)
    DevBuf<int32_t> keys_;
    DevBuf<double> loss_;
    DevBuf<int> ticket_;
    Axis ax_[2];
    bool force_v1_ = false;
    bool wide_split_ = true;
    int debug_ = 0;
    bool no_inreg_ = false;
    bool split_f16_ = true;
    int pc_ = 1;
    // the split-f16 wide kernel up to vdim 32 * this.  Measured (profiles/r06_als_wide_split_192_256.txt, ML-20M, ms per epoch of row kernels, split | fp32):
    // d = 192 11.8 | 23.1, d = 224 23.3 | 26.0, d = 256 37.0 | 30.6 -- above T = 6 a CU holds ONE workgroup (128 / 144 accumulators want 256 registers) and
    // the producer's three row sets spill (110 registers at T = 7, 600 at T = 8): T = 8 stays on the fp32 instruction
    int wide_split_max_t_ = 7;
    // als_gramian_kernel: measured on ML-20M at d = 128 (profiles/r06_als_gramian.txt, ms for the items / the users): 4 waves per CU 0.048 / 0.109, 8: 0.086 / 0.143,
    // 12: 0.118 / 0.163 -- every slice ends in 64 fp64 atomics per lane on the same 16 K addresses, so fewer, longer slices are faster.  Outside vdim 128 the default STAYS at 8:
    // the slice boundaries decide FF's last bits, and with 4 the one matrix-free tiny case at d = 160 / block_size 64 -- three CG steps on systems conditioned
    // beyond fp32 -- lands at 20x the oracle's distance from float64 instead of 0.2x (deterministically; every other case unchanged: GPU call 14).  A re-roll of
    // the rounding, not an error of either FF (both 7e-8 from float64) -- but the parity suite is held as it is: 0 = 4 waves per CU at vdim 128 only, 8 elsewhere.
    int gram_waves_per_cu_ = 0, gram_upg_ = 8;
    float split_wcut_ = 32768.0f;
    uint64_t fver_[2] = {1, 1};     // bumped whenever P (0) / Q (1) may have changed on the device
    uint64_t vals_ver_ = 1;         // bumped whenever confidence values were uploaded
    DevBuf<float> qi_;              // block-interleaved copy of the other factor (als_interleave_stats_kernel)
    int qi_side_ = -1;
    uint64_t qi_ver_ = 0;
    float qi_wcut_ = -1.f;
    DevBuf<int> pc_err_;
    int pc_same_simd_ = 0;
    int pc_clock_mhz_ = 0;
    DevBuf<float> split_part_;
    DevBuf<float> split_out_;
    DevBuf<float> rowff_;
    DevBuf<float> gscratch_;
    DevBuf<float> scratch_;
    std::map<std::tuple<int, int, int>, std::unique_ptr<WorkList>> work_cache_;
    EventTimer t_main_, t_aux_;
    Comm* comm_ = nullptr;   // not owned
};
#endif   // BFH_ALS_KERNELS_ONLY

}  // namespace bfh
