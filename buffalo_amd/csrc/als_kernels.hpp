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

#include "common.hpp"

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
template <int NT>
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
    for (int r = r0; r < r1; r += 2) {
        const int row = r + half;
        const bool ok = row < r1;
        const float* fr = F + static_cast<size_t>(ok ? row : r0) * vdim;
        const float a = ok ? fr[bi * 32 + col] : 0.f;
#pragma unroll
        for (int g = 0; g < NT; ++g) {
            const float b = (ok && bj0 + g < T) ? fr[(bj0 + g) * 32 + col] : 0.f;
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
        }
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
// One workgroup (4 waves) owns a row (or a 4096-nnz chunk of a heavy row).  ONE pass over the row's
// nnz builds, on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32),
//     G = alpha * sum_k v_k q_k q_k^T   (vdim x vdim; wave w accumulates tile-row w: T tiles x 16 regs)
//     g = sum_k coef_k q_k              (coef = alpha*v for iALS++, 1 + alpha*v for the dense solvers)
// with q rows streamed straight from HBM/L2 into the MFMA operand layout (lane = (k&1)*32 + i reads
// q_k[32*bj + i]: a 128-byte segment per half-wave, no LDS staging, no transposition).  M = FF + G then
// lands in LDS and wave 0 runs the small dense algebra:
//   * iALS++ (als.cc:269-352): the reference tracks Yui_k = p.q_k incrementally; with the explicit
//     Gramian, sum_k alpha v_k (Yui_k - 1) q_k == r0 + G (p - p0) where p0 is the row at entry and
//     r0 = sum_k alpha v_k (p0.q_k - 1) q_k is gathered in the same pass, so the block gradient is
//     b = M[blk,:](p - p0) + (FF p0)_blk + reg p_blk + r0_blk and the CG matrix M[blk,blk] + reg I:
//     the same recurrence without the reference's 5 passes over the nnz per block, and without the
//     cancellation a naive G p - g would introduce in fp32;
//   * manual_cg / llt / ldlt (als.cc:180-204 + algo.cc:52-82): A = M + reg*ada*I explicitly.
// Heavy rows (item "Star Wars" has 10^5 users) are cut into chunks whose partial G/g are combined
// with fp32 atomics in a scratch slot and solved by a second, tiny launch.
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

// ------------------------------------------------------------------------------------------------
// Gramian pass of ONE wave over the nnz of one work item: acc[g] += (alpha v q[tw-block])^T q[g-block]
// on the matrix cores (v_mfma_f32_32x32x2_f32: one instruction eats TWO nnz, k-index = lane>>5), and
// gpart += c q[tw*32+col] with c = alpha v (iALS++: g = sum alpha v q) or 1 + alpha v (dense solvers: y).
// The 64 keys/vals of a chunk sit one-per-lane; a pair's two row ids are wave-uniform so they come out
// with v_readlane (SALU), the operand rows are plain dword loads (half-wave = one 128-B line per tile),
// and the loop is register double-buffered: the rows of group j+1 (UP pairs) are in flight while group j
// feeds the MFMAs.  Steady state per pair: 4 readlane, 2 cndmask, 1 address mad, T+1 loads, 2 mul, 1 fma.
// ------------------------------------------------------------------------------------------------
template <int T, bool IALS>
__device__ __forceinline__ void als_gram_rowpass(const AlsParams& p, const AlsWork& wk, int tw, int lane, int half, int col, bool lossk,
                                                 const float (&pcol)[T], f32x16 (&acc)[T], float& gpart, double& nume_k, double& deno_k) {
    const int vdim = p.vdim;
    const int64_t n = wk.kend - wk.kbeg;
    constexpr int UP = 4;
    const int64_t nchunks = (n + 63) / 64;
    auto fetch_keys = [&](int64_t chunk, int& cc, float& vvv) {
        const int64_t kk = chunk * 64 + lane;
        cc = 0;        // padding lanes: row 0 of the other factor with weight 0
        vvv = 0.f;
        if (kk < n) {
            cc = p.keys[wk.kbeg + kk];
            vvv = p.vals[wk.kbeg + kk];
        }
    };
    auto load_pair = [&](int myc, float myv, int pr, float (&q)[T], float& qs, float& v) {
        const int c0 = __builtin_amdgcn_readlane(myc, 2 * pr), c1 = __builtin_amdgcn_readlane(myc, 2 * pr + 1);
        const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), 2 * pr));
        const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myv), 2 * pr + 1));
        v = half ? v1 : v0;
        const float* q_ = p.Q + static_cast<size_t>(half ? c1 : c0) * vdim + col;
#pragma unroll
        for (int g = 0; g < T; ++g) q[g] = q_[g * 32];
        qs = q_[tw * 32];   // A-operand column block (same cache line as q[tw]; tw is wave-uniform but not a compile-time index)
    };
    auto consume = [&](const float (&q)[T], float qs, float v, float one) {
        const float wgt = p.alpha * v;
        const float a = wgt * qs;
#pragma unroll
        for (int g = 0; g < T; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, q[g], acc[g], 0, 0, 0);
        // als.cc:184: float(1.0 + double(v*alpha)) == 1.0f + v*alpha (the exact sum rounded once either way)
        gpart += (IALS ? wgt : (one + wgt)) * qs;
        if (lossk) {  // als.cc:187-192 / 298-303 on the ORIGINAL row
            float dp = 0.f;
#pragma unroll
            for (int g = 0; g < T; ++g) dp += q[g] * pcol[g];
            for (int sft = 1; sft < 32; sft <<= 1) dp += __shfl_xor(dp, sft, 64);
            if (col == 0 && one != 0.f) {
                nume_k -= static_cast<double>(dp * dp);
                nume_k += static_cast<double>((dp - 1) * (dp - 1)) * (1.0 + static_cast<double>(wgt));
                deno_k += static_cast<double>(wgt);
            }
        }
    };
    auto nnz_of = [&](int64_t ch) { return ch < nchunks ? static_cast<int>((n - ch * 64) < 64 ? (n - ch * 64) : 64) : 0; };
    auto groups_of = [&](int64_t ch) { return (nnz_of(ch) >> 1) / UP; };   // groups hold COMPLETE pairs only (no padding lane)
    int myc, myc_n;
    float myv, myv_n;
    fetch_keys(0, myc, myv);
    fetch_keys(1, myc_n, myv_n);
    float qa[UP][T], qsa[UP], va[UP];
    if (groups_of(0) > 0) {
#pragma unroll
        for (int uu = 0; uu < UP; ++uu) load_pair(myc, myv, uu, qa[uu], qsa[uu], va[uu]);
    }
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int npairs = (nnz_of(ch) + 1) >> 1;
        const int ngroups = groups_of(ch);
        const bool next_has_group = groups_of(ch + 1) > 0;
        for (int gidx = 0; gidx < ngroups; ++gidx) {
            float qb[UP][T], qsb[UP], vb[UP];
            const bool here = gidx + 1 < ngroups;
            if (here) {
#pragma unroll
                for (int uu = 0; uu < UP; ++uu) load_pair(myc, myv, (gidx + 1) * UP + uu, qb[uu], qsb[uu], vb[uu]);
            } else if (next_has_group) {   // first group of the next chunk
#pragma unroll
                for (int uu = 0; uu < UP; ++uu) load_pair(myc_n, myv_n, uu, qb[uu], qsb[uu], vb[uu]);
            }
#pragma unroll
            for (int uu = 0; uu < UP; ++uu) consume(qa[uu], qsa[uu], va[uu], 1.0f);
            if (here || next_has_group) {
#pragma unroll
                for (int uu = 0; uu < UP; ++uu) {
                    qsa[uu] = qsb[uu];
                    va[uu] = vb[uu];
#pragma unroll
                    for (int g = 0; g < T; ++g) qa[uu][g] = qb[uu][g];
                }
            }
        }
        // <= UP leftover pairs: only the last chunk of the item has them; its last pair may be half padding
        for (int pr = ngroups * UP; pr < npairs; ++pr) {
            float q1[T], qs1, v1;
            load_pair(myc, myv, pr, q1, qs1, v1);
            consume(q1, qs1, v1, (ch * 64 + 2 * pr + half < n) ? 1.0f : 0.f);
        }
        myc = myc_n;
        myv = myv_n;
        fetch_keys(ch + 2, myc_n, myv_n);
    }
    gpart += __shfl_xor(gpart, 32, 64);   // the two halves hold the k-parities of the same element
}

// Everything after the Gramian pass of one work item: heavy-row partial -> scratch, or M = FF + G into LDS
// and the dense solve by wave 0.  Shared by the register-operand and the LDS-tile kernels.
template <int T>
__device__ __forceinline__ void als_finish_item(const AlsParams& p, const AlsWork& wk, f32x16 (&acc)[T], float gpart, float* __restrict__ scratch,
                                                float* M, float* gv, float* pl, float* p0, float* f0, float* w0, float* w1, float* w2, float* w3,
                                                float* w4, float* Pu, int64_t n, int mode, int lane, int wv, int half, int col, double& nume,
                                                double& deno) {
    const int vdim = p.vdim, D = p.d, ld = vdim + ALS_LD_PAD;
    if (wk.slot >= 0) {
        // heavy row: add this chunk's partial into the scratch slot, solved by als_solve_kernel
        float* S = scratch + static_cast<size_t>(wk.slot) * (vdim * vdim + vdim);
#pragma unroll
        for (int g = 0; g < T; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
                atomic_add_f32(S + static_cast<size_t>(wv * 32 + i) * vdim + g * 32 + col, acc[g][e]);
            }
        if (half == 0) atomic_add_f32(S + vdim * vdim + wv * 32 + col, gpart);
        return;
    }
    // M = FF + G -> LDS; iALS++ keeps -g (see als_dense_solve: b = M[blk,:] delta + f0 + gv + reg p with f0 = M p0)
    if (!(p.debug & 4)) {
#pragma unroll
        for (int g = 0; g < T; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
                const int rr = wv * 32 + i, cc = g * 32 + col;
                M[rr * ld + cc] = acc[g][e] + p.FF[static_cast<size_t>(rr) * vdim + cc];
            }
        if (half == 0) gv[wv * 32 + col] = mode == 8 ? -gpart : gpart;
    }
    __syncthreads();
    if (mode == 8 && !(p.debug & 4)) {   // f0 = M p0: wave w rows 32w..32w+31, the half-waves split the columns
        const float* Mi = M + (wv * 32 + col) * ld + half * (vdim / 2);
        const float* pp = p0 + half * (vdim / 2);
        float sum = 0.f;
#pragma unroll 8
        for (int j = 0; j < vdim / 2; ++j) sum += Mi[j] * pp[j];
        sum += __shfl_xor(sum, 32, 64);
        if (half == 0) f0[wv * 32 + col] = sum;
        __syncthreads();
    }
    if (wv == 0) {
        const float ada = p.adaptive_reg ? static_cast<float>(n) : 1.0f;
        if (p.compute_loss) {
            float pp = 0.f, pfp = 0.f;
            for (int i = lane; i < D; i += 64) {
                pp += pl[i] * pl[i];
                if (p.axis == 1) {
                    float sum = 0.f;
                    for (int j = 0; j < D; ++j) sum += p.FF[static_cast<size_t>(i) * vdim + j] * pl[j];
                    pfp += pl[i] * sum;
                }
            }
            pp = wave_sum(pp);
            nume += static_cast<double>(ada * p.reg * pp);
            if (p.axis == 1) {
                pfp = wave_sum(pfp);
                nume += static_cast<double>(pfp);
                deno += static_cast<double>(p.op_rows);
            }
        }
        if (!(p.debug & 1)) als_dense_solve(M, gv, pl, p0, f0, w0, w1, w2, w3, w4, p, lane, p.reg * ada, mode);
        for (int i = lane; i < D; i += 64) Pu[i] = pl[i];
    }
}

// Fused design: one 64*T-thread block per row; wave w accumulates tile-row w of G in registers, the block
// assembles M = FF + G in LDS and wave 0 runs the dense phase in place -- G never touches HBM.
template <int T, bool IALS>
__global__ __launch_bounds__(64 * T, 2) void als_gram_solve_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_work, float* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int vdim = p.vdim, ld = vdim + ALS_LD_PAD;
    constexpr int mode_ials = 8;
    float* M = lds;
    float* gv = lds + vdim * ld;
    float* pl = gv + vdim;
    float* w0 = pl + vdim; float* w1 = w0 + vdim; float* w2 = w1 + vdim; float* w3 = w2 + vdim; float* w4 = w3 + vdim;
    float* p0 = w4 + vdim; float* f0 = p0 + vdim;
    int* s_item = reinterpret_cast<int*>(f0 + vdim);   // all LDS lives in the dynamic region (16-B aligned base)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // tile row of this wave
    const int half = lane >> 5, col = lane & 31;
    const int mode = IALS ? mode_ials : p.solver;
    double nume = 0.0, deno = 0.0;      // row-level terms (identical in every lane of wave 0)
    double nume_k = 0.0, deno_k = 0.0;  // per-nnz terms (lanes 0 and 32 of wave 0 hold the two k-parities)

    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) *s_item = atomicAdd(p.ticket, 1);
        __syncthreads();
        const int item = *s_item;
        if (item >= n_work) break;
        const AlsWork wk = work[item];
        float* Pu = p.P + static_cast<size_t>(wk.row) * vdim;
        // current row -> LDS (the loss terms and the solve read it)
        for (int e = threadIdx.x; e < vdim; e += blockDim.x) { pl[e] = Pu[e]; p0[e] = Pu[e]; }
        f32x16 acc[T];
#pragma unroll
        for (int g = 0; g < T; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
        float gpart = 0.f;
        const bool lossk = p.compute_loss && p.axis == 1 && wv == 0;
        float pcol[T];
#pragma unroll
        for (int g = 0; g < T; ++g) pcol[g] = lossk ? Pu[g * 32 + col] : 0.f;
        als_gram_rowpass<T, IALS>(p, wk, wv, lane, half, col, lossk, pcol, acc, gpart, nume_k, deno_k);
        als_finish_item<T>(p, wk, acc, gpart, scratch, M, gv, pl, p0, f0, w0, w1, w2, w3, w4, Pu, wk.kend - wk.kbeg, mode, lane, wv, half, col, nume,
                           deno);
    }
    if (p.compute_loss && wv == 0) {
        nume_k += __shfl_xor(nume_k, 32, 64);
        deno_k += __shfl_xor(deno_k, 32, 64);
        if (lane == 0) {
            if (nume + nume_k != 0.0) atomicAdd(p.loss, nume + nume_k);
            if (deno + deno_k != 0.0) atomicAdd(p.loss + 1, deno + deno_k);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split design: Gramian pass and dense solve as two launches.
//   als_gram_kernel   -- no LDS, <= 168 VGPRs: 3 waves per SIMD stream q rows into the matrix cores; a
//                        64*T-thread block shares a row so its q rows reach the CU's L1 once.  G and g go
//                        to an HBM scratch slot per row (row-major [vdim][vdim] + [vdim]); chunks of heavy
//                        rows add with fp32 atomics into the (zeroed) slot.
//   als_solve_kernel  -- one wave per row: M = FF + G from the scratch into LDS, then als_dense_solve.
// Costs 2 * vdim^2 * 4 B of HBM traffic per row, which is nothing at vdim <= 96 (where the fused kernel's
// low occupancy hurts most) and ~5 ms per ML-20M epoch at vdim = 128 -- hence the per-vdim default.
// ------------------------------------------------------------------------------------------------
template <int T, bool IALS>
__global__ __launch_bounds__(64 * T, 3) void als_gram_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_items, float* __restrict__ scratch) {
    const int vdim = p.vdim;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31;
    double nume_k = 0.0, deno_k = 0.0;
    __shared__ int s_item;
    const int tw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // tile-row of G this wave accumulates
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = atomicAdd(p.ticket, 1);
        __syncthreads();
        const int item = s_item;
        if (item >= n_items) break;
        const AlsWork wk = work[item];
        const int u = wk.row;
        f32x16 acc[T];
#pragma unroll
        for (int g = 0; g < T; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
        float gpart = 0.f;
        const bool lossk = p.compute_loss && p.axis == 1 && tw == 0;
        float pcol[T];
#pragma unroll
        for (int g = 0; g < T; ++g) pcol[g] = lossk ? p.P[static_cast<size_t>(u) * vdim + g * 32 + col] : 0.f;
        als_gram_rowpass<T, IALS>(p, wk, tw, lane, half, col, lossk, pcol, acc, gpart, nume_k, deno_k);
        float* S = scratch + static_cast<size_t>(u - p.start_x) * (static_cast<size_t>(vdim) * vdim + vdim);
        if (wk.slot >= 0) {   // chunk of a heavy row: partials are summed (the slot was zeroed by the host)
#pragma unroll
            for (int g = 0; g < T; ++g)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
                    atomic_add_f32(S + static_cast<size_t>(tw * 32 + i) * vdim + g * 32 + col, acc[g][e]);
                }
            if (half == 0) atomic_add_f32(S + static_cast<size_t>(vdim) * vdim + tw * 32 + col, gpart);
        } else {
#pragma unroll
            for (int g = 0; g < T; ++g)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
                    S[static_cast<size_t>(tw * 32 + i) * vdim + g * 32 + col] = acc[g][e];
                }
            if (half == 0) S[static_cast<size_t>(vdim) * vdim + tw * 32 + col] = gpart;
        }
    }
    if (p.compute_loss && p.axis == 1) {
        nume_k += __shfl_xor(nume_k, 32, 64);
        deno_k += __shfl_xor(deno_k, 32, 64);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}

struct AlsHeavy {
    int row, slot;
    int64_t n;
};

__global__ __launch_bounds__(64) void als_solve_kernel(AlsParams p, const AlsHeavy* __restrict__ heavy, int n_heavy, const float* __restrict__ scratch,
                                                             int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int vdim = p.vdim, D = p.d, ld = vdim + ALS_LD_PAD;
    float* M = lds;
    float* gv = lds + vdim * ld;
    float* pl = gv + vdim;
    float* w0 = pl + vdim; float* w1 = w0 + vdim; float* w2 = w1 + vdim; float* w3 = w2 + vdim; float* w4 = w3 + vdim;
    float* p0 = w4 + vdim; float* f0 = p0 + vdim;
    const int lane = threadIdx.x;
    for (int h = blockIdx.x; h < n_heavy; h += gridDim.x) {
    const AlsHeavy hv = heavy[h];
    __syncthreads();
    const float* S = scratch + static_cast<size_t>(hv.slot) * (static_cast<size_t>(vdim) * vdim + vdim);
    float* Pu = p.P + static_cast<size_t>(hv.row) * vdim;
    for (int e = lane; e < vdim * vdim; e += 64) {
        const int rr = e / vdim, cc = e % vdim;
        M[rr * ld + cc] = S[e] + p.FF[e];
    }
    for (int e = lane; e < vdim; e += 64) {
        gv[e] = mode == 8 ? -S[vdim * vdim + e] : S[vdim * vdim + e];   // iALS++: -g, see als_finish_item
        pl[e] = Pu[e];
        p0[e] = Pu[e];
        f0[e] = 0.f;
    }
    __syncthreads();
    if (mode == 8) {   // f0 = M p0 (row i of M per lane: stride ld = vdim+1 words, conflict-free)
        for (int i = lane; i < vdim; i += 64) {
            const float* Mi = M + i * ld;
            float sum = 0.f;
#pragma unroll 8
            for (int j = 0; j < vdim; ++j) sum += Mi[j] * p0[j];
            f0[i] = sum;
        }
    }
    if (p.compute_loss && p.axis == 1) {
        for (int i = lane; i < vdim; i += 64) {   // FF p0 for the p FF p loss term (FF symmetric: column i read as FF[j][i])
            float sum = 0.f;
#pragma unroll 8
            for (int j = 0; j < vdim; ++j) sum += p.FF[static_cast<size_t>(j) * vdim + i] * p0[j];
            w1[i] = sum;
        }
    }
    __syncthreads();
    const float ada = p.adaptive_reg ? static_cast<float>(hv.n) : 1.0f;
    double nume = 0.0, deno = 0.0;
    if (p.compute_loss) {
        float pp = 0.f, pfp = 0.f;
        for (int i = lane; i < D; i += 64) {
            pp += pl[i] * pl[i];
            if (p.axis == 1) pfp += pl[i] * w1[i];     // p (FF p)
        }
        pp = wave_sum(pp);
        nume += static_cast<double>(ada * p.reg * pp);
        if (p.axis == 1) {
            pfp = wave_sum(pfp);
            nume += static_cast<double>(pfp);
            deno += static_cast<double>(p.op_rows);
        }
    }
    if (!(p.debug & 1)) als_dense_solve(M, gv, pl, p0, f0, w0, w1, w2, w3, w4, p, lane, p.reg * ada, mode);
    for (int i = lane; i < D; i += 64) Pu[i] = pl[i];
    if (p.compute_loss && lane == 0) {
        if (nume != 0.0) atomicAdd(p.loss, nume);
        if (deno != 0.0) atomicAdd(p.loss + 1, deno);
    }
    }  // rows
}

// ------------------------------------------------------------------------------------------------
class AlsHandle : public HandleBase {
 public:
    ~AlsHandle() override {
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
        P_.resize(np); Q_.resize(nq);
        BFH_HIP(hipMemcpyAsync(P_.get(), P, np * sizeof(float), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(Q_.get(), Q, nq * sizeof(float), hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += static_cast<double>((np + nq) * sizeof(float));
        BFH_HIP(hipStreamSynchronize(stream));
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
        A.resident = true;
    }

    void precompute(int axis) {
        BFH_REQUIRE(model_, "precompute before initialize_model");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        const float* F = axis == 0 ? Q_.get() : P_.get();
        const int rows = axis == 0 ? Q_rows_ : P_rows_;
        BFH_HIP(hipMemsetAsync(FF64_.get(), 0, FF64_.bytes(), stream));
        const int T = vdim_ / 32;
        constexpr int NT = 4;
        const int TG = (T + NT - 1) / NT;
        int slices = (num_cus_ * 8) / (T * TG);
        if (slices < 1) slices = 1;
        int rps = (rows + slices - 1) / slices;
        rps = (rps + 1) & ~1;  // even: row pairs never straddle slices
        if (rps < 2) rps = 2;
        slices = (rows + rps - 1) / rps;
        const int slot = t_aux_.begin(stream);
        hipLaunchKernelGGL(als_gramian_kernel<NT>, dim3(slices, T, TG), dim3(64), 0, stream, F, rows, vdim_, rps, FF64_.get());
        BFH_HIP(hipGetLastError());
        const int nff = vdim_ * vdim_;
        hipLaunchKernelGGL(als_gramian_round_kernel, dim3((nff + 255) / 256), dim3(256), 0, stream, FF64_.get(), FF_.get(), nff);
        BFH_HIP(hipGetLastError());
        t_aux_.end(slot, stream);
        BFH_HIP(hipStreamSynchronize(stream));
        stats.aux_ms += t_aux_.drain();
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
        if (A.resident) {
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
            p.keys = keys_.get();
            p.vals = vals_.get();
            p.yui = yui_.get();
        }
        BFH_HIP(hipMemsetAsync(loss_.get(), 0, 2 * sizeof(double), stream));
        BFH_HIP(hipMemsetAsync(ticket_.get(), 0, sizeof(int), stream));
        const int nrows = next_x - start_x;
        const int K = (vdim_ + 63) / 64;
        const bool gram_path = vdim_ <= 128 && !force_v1_;
        const WorkList* wl = nullptr;
        if (gram_path) {
            wl = &work_list(axis, start_x, next_x, ip, beg);
            if (wl->n_heavy) BFH_HIP(hipMemsetAsync(scratch_.get(), 0, static_cast<size_t>(wl->n_heavy) * (vdim_ * vdim_ + vdim_) * sizeof(float), stream));
        }
        const int slot = t_main_.begin(stream);
        const bool split = design_ < 0 ? vdim_ < 128 : design_ == 0;   // see als_gram_kernel: G round trip through HBM only pays below vdim 128
        if (gram_path && split) {
            // split design: Gramian pass at full occupancy -> HBM scratch -> dense solve (see als_gram_kernel)
            const int T = vdim_ / 32;
            const size_t per_row = static_cast<size_t>(vdim_) * vdim_ + vdim_;
            const size_t need = static_cast<size_t>(nrows) * per_row;
            if (gscratch_.size() < need) gscratch_.resize(need);
            for (int hr : wl->heavy_rows)
                BFH_HIP(hipMemsetAsync(gscratch_.get() + static_cast<size_t>(hr - start_x) * per_row, 0, per_row * sizeof(float), stream));
            const int items = wl->n_work;
            if (items > 0) {
                int blocks = items;
                const int per_cu = (12 / T) > 0 ? (12 / T) : 1;       // <= 166 VGPRs -> 3 waves per SIMD = 12 waves per CU
                if (blocks > num_cus_ * per_cu) blocks = num_cus_ * per_cu;
#define BFH_GK(TT)                                                                                                                  \
    do {                                                                                                                            \
        if (code_ == 8) hipLaunchKernelGGL((als_gram_kernel<TT, true>), dim3(blocks), dim3(64 * TT), 0, stream, p, wl->work.get(), items, gscratch_.get()); \
        else hipLaunchKernelGGL((als_gram_kernel<TT, false>), dim3(blocks), dim3(64 * TT), 0, stream, p, wl->work.get(), items, gscratch_.get());          \
    } while (0)
                if (T <= 1) BFH_GK(1);
                else if (T <= 2) BFH_GK(2);
                else if (T <= 3) BFH_GK(3);
                else BFH_GK(4);
#undef BFH_GK
                BFH_HIP(hipGetLastError());
                const size_t lds_h = als_gs_lds_bytes(vdim_);
                int sblocks = num_cus_ * static_cast<int>(std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / lds_h)));
                if (sblocks > wl->n_solve) sblocks = wl->n_solve;
                BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(als_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(lds_h)));
                hipLaunchKernelGGL(als_solve_kernel, dim3(sblocks), dim3(64), lds_h, stream, p, wl->solve.get(), wl->n_solve,
                                   gscratch_.get(), static_cast<int>(code_));
                BFH_HIP(hipGetLastError());
            }
        } else if (gram_path) {
            // fused design: Gramian on the matrix cores + dense LDS solve in one launch
            const int T = vdim_ / 32;
            const size_t lds = als_gs_lds_bytes(vdim_);
            int blocks = num_cus_ * static_cast<int>(std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds)));
            if (blocks > wl->n_work) blocks = wl->n_work;
            if (blocks > 0) {
#define BFH_GS1(TT, II)                                                                                                        \
    do {                                                                                                                      \
        BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(als_gram_solve_kernel<TT, II>),                            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                      \
        hipLaunchKernelGGL((als_gram_solve_kernel<TT, II>), dim3(blocks), dim3(64 * TT), lds, stream, p, wl->work.get(),      \
                           wl->n_work, scratch_.get());                                                                       \
    } while (0)
#define BFH_GS(TT)                      \
    do {                                \
        if (code_ == 8) BFH_GS1(TT, true); \
        else BFH_GS1(TT, false);        \
    } while (0)
                if (T <= 1) BFH_GS(1);
                else if (T <= 2) BFH_GS(2);
                else if (T <= 3) BFH_GS(3);
                else BFH_GS(4);
#undef BFH_GS
#undef BFH_GS1
                BFH_HIP(hipGetLastError());
            }
            if (wl->n_heavy) {
                const size_t lds_h = als_gs_lds_bytes(vdim_);
                BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(als_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(lds_h)));
                hipLaunchKernelGGL(als_solve_kernel, dim3(wl->n_heavy), dim3(64), lds_h, stream, p, wl->heavy.get(), wl->n_heavy,
                                   scratch_.get(), static_cast<int>(code_));
            }
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
        if (writeback_) {  // als.cu:403: updated rows go back to the caller's array
            float* hostF = axis == 0 ? hostP_ : hostQ_;
            const size_t off = static_cast<size_t>(start_x) * vdim_, cnt = static_cast<size_t>(nrows) * vdim_;
            BFH_HIP(hipMemcpyAsync(hostF + off, p.P + off, cnt * sizeof(float), hipMemcpyDeviceToHost, stream));
            stats.d2h_bytes += static_cast<double>(cnt * sizeof(float));
        }
        BFH_HIP(hipStreamSynchronize(stream));
        stats.kernel_ms += t_main_.drain();
        stats.launches += 1;
        stats.samples += n;
        *nume = l[0];
        *deno = l[1];
    }

    struct WorkList {
        DevBuf<AlsWork> work;
        DevBuf<AlsHeavy> heavy;   // fused kernels: heavy rows only (slot = scratch slot)
        DevBuf<AlsHeavy> solve;   // split design: every non-empty row, longest first (slot = row - start_x)
        std::vector<int> heavy_rows;
        int n_work = 0, n_heavy = 0, n_solve = 0;
    };
    // Work items of one partial_update call: one per non-empty row, rows above HEAVY nnz cut into
    // chunks; longest first (dynamic ticket order) so the tail is short.  Cached per (axis, range).
    const WorkList& work_list(int axis, int start_x, int next_x, const int64_t* ip, int64_t shift) {
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
                sv.push_back({x, x - start_x, n});
                const int64_t kb = prev - shift;
                if (n <= HEAVY) {
                    w.push_back({x, static_cast<int>(kb), static_cast<int>(kb + n), -1});
                } else {
                    const int slot = static_cast<int>(h.size());
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
        for (const auto& hh : h) wl->heavy_rows.push_back(hh.row);
        wl->solve.resize(std::max<size_t>(1, sv.size()));
        if (!sv.empty()) BFH_HIP(hipMemcpyAsync(wl->solve.get(), sv.data(), sv.size() * sizeof(AlsHeavy), hipMemcpyHostToDevice, stream));
        wl->work.resize(std::max<size_t>(1, w.size()));
        wl->heavy.resize(std::max<size_t>(1, h.size()));
        if (!w.empty()) BFH_HIP(hipMemcpyAsync(wl->work.get(), w.data(), w.size() * sizeof(AlsWork), hipMemcpyHostToDevice, stream));
        if (!h.empty()) BFH_HIP(hipMemcpyAsync(wl->heavy.get(), h.data(), h.size() * sizeof(AlsHeavy), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        const size_t need = std::max<size_t>(1, h.size()) * (static_cast<size_t>(vdim_) * vdim_ + vdim_);
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
            BFH_HIP(hipMemcpyAsync(hostP_, P_.get(), np * sizeof(float), hipMemcpyDeviceToHost, stream));
            BFH_HIP(hipMemcpyAsync(hostQ_, Q_.get(), nq * sizeof(float), hipMemcpyDeviceToHost, stream));
            stats.d2h_bytes += static_cast<double>((np + nq) * sizeof(float));
        } else {
            BFH_HIP(hipMemcpyAsync(P_.get(), hostP_, np * sizeof(float), hipMemcpyHostToDevice, stream));
            BFH_HIP(hipMemcpyAsync(Q_.get(), hostQ_, nq * sizeof(float), hipMemcpyHostToDevice, stream));
            stats.h2d_bytes += static_cast<double>((np + nq) * sizeof(float));
        }
        BFH_HIP(hipStreamSynchronize(stream));
    }

    void set_mode(const std::string& name, int64_t v) {
        if (name == "als_writeback") writeback_ = v != 0;
        else if (name == "als_v1") force_v1_ = v != 0;
        else if (name == "als_debug") debug_ = static_cast<int>(v);
        else if (name == "als_fused") design_ = v < 0 ? -1 : (v != 0);   // 1 fused Gramian+solve kernel, 0 split design, -1 per-vdim default
        else if (name == "timing") timing = v != 0;
        else throw Error(BFH_ERR_INVALID, "unknown mode '" + name + "'");
    }

    void device_buffer(const std::string& name, void** p, size_t* bytes) {
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
    };

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
    int debug_ = 0;
    int design_ = -1;
    DevBuf<float> gscratch_;
    DevBuf<float> scratch_;
    std::map<std::tuple<int, int, int>, std::unique_ptr<WorkList>> work_cache_;
    EventTimer t_main_, t_aux_;
};

}  // namespace bfh
