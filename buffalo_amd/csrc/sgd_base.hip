// SgdHandle: device-resident model, placeholders, lr schedule and the epoch-end optimizer pass
// shared by the BPRMF and WARP backends.
#include "sgd_base.hpp"

#include <algorithm>

#include "comm.hpp"

namespace bfh {

// ------------------------------------------------------------------------------------------------
// fill_rows: row id of every nnz of a chunk (cf. fill_rows_kernel, /root/reference/lib/cuda/bpr/
// bpr.cu:22-33, which runs one thread per block).  One thread per nnz, binary search in indptr.
// ------------------------------------------------------------------------------------------------
__global__ void fill_rows_kernel(const int64_t* __restrict__ indptr, int start_x, int next_x, int64_t shift, int64_t n,
                                 int32_t* __restrict__ rows) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t g = shift + t;
    // first row u in [start_x, next_x) with indptr[u] > g
    int lo = start_x, hi = next_x;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (indptr[mid] <= g) lo = mid + 1;
        else hi = mid;
    }
    rows[t] = lo;
}

void launch_fill_rows(const int64_t* indptr, int start_x, int next_x, int64_t shift, int64_t n, int32_t* rows, hipStream_t s) {
    if (n <= 0) return;
    const int bs = 256;
    const int64_t grid = (n + bs - 1) / bs;
    hipLaunchKernelGGL(fill_rows_kernel, dim3(static_cast<unsigned>(grid)), dim3(bs), 0, s, indptr, start_x, next_x, shift, n, rows);
    BFH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Epoch-end optimizer pass: SGDAlgorithm::update_parameters / update_adam / update_adagrad
// (/root/reference/lib/algo.cc:365-465) fused with CWARP's unit-ball projection
// (/root/reference/lib/algo_impl/warp/warp.cc:192-201).  Pure streaming: (3 or 4) arrays read +
// written once.  Quirks kept: grad keeps the transformed step (Q-6), FEPS outside the sqrt (Q-5).
// A row is handled by G = 64/RPW lanes, float4 per lane.
// ------------------------------------------------------------------------------------------------
struct OptConsts {
    float reg2, lr, b1, omb1, b2, omb2, c1, c2;
};

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int s = 1; s < G; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}

template <bool ADAM>
__device__ __forceinline__ void opt_step4(float4& x, float4& g, float4& m, float4& v, bool has_cnt, float cntf,
                                          const OptConsts& k, float& nrm) {
    const float FEPS = 1e-10f;
    float* xp = reinterpret_cast<float*>(&x);
    float* gp = reinterpret_cast<float*>(&g);
    float* vp = reinterpret_cast<float*>(&v);
    float* mp = reinterpret_cast<float*>(&m);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float ge = gp[e];
        if (has_cnt) ge = ge / cntf;
        ge = ge - xp[e] * k.reg2;
        if (ADAM) {
            mp[e] = k.b1 * mp[e] + k.omb1 * ge;
            vp[e] = k.b2 * vp[e] + k.omb2 * (ge * ge);
            const float m_hat = mp[e] / k.c1;
            const float v_hat = vp[e] / k.c2;
            ge = m_hat / (sqrtf(v_hat) + FEPS);
        } else {
            vp[e] = vp[e] + ge * ge;
            ge = ge / (sqrtf(vp[e]) + FEPS);
        }
        gp[e] = ge;
        xp[e] = xp[e] + k.lr * ge;
        nrm += xp[e] * xp[e];
    }
}

template <bool ADAM, bool PROJECT>
__global__ __launch_bounds__(256) void sgd_update_rows_kernel(float* __restrict__ X, float* __restrict__ grad,
                                                               float* __restrict__ mom, float* __restrict__ vel,
                                                               const int* __restrict__ cnt, int rows, int vdim, int G,
                                                               OptConsts k) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int rpw = 64 / G;
    const int row = wave * rpw + lane / G;
    const int gl = lane % G;
    const bool active = row < rows;
    float nrm = 0.f;
    float cntf = 1.f;
    bool has_cnt = false;
    if (active && cnt) {
        const int c = cnt[row];
        has_cnt = c != 0;
        cntf = static_cast<float>(c);
    }
    if (vdim <= G * 4) {
        // one float4 per lane: the row stays in registers between the step and the projection
        const bool has = active && gl * 4 < vdim;
        const size_t o = static_cast<size_t>(active ? row : 0) * vdim + gl * 4;
        float4 x = make_float4(0, 0, 0, 0), g = x, v = x, m = x;
        if (has) {
            x = *reinterpret_cast<float4*>(X + o);
            g = *reinterpret_cast<float4*>(grad + o);
            v = *reinterpret_cast<float4*>(vel + o);
            if (ADAM) m = *reinterpret_cast<float4*>(mom + o);
            opt_step4<ADAM>(x, g, m, v, has_cnt, cntf, k, nrm);
            *reinterpret_cast<float4*>(grad + o) = g;
            *reinterpret_cast<float4*>(vel + o) = v;
            if (ADAM) *reinterpret_cast<float4*>(mom + o) = m;
        }
        if (PROJECT) {
            nrm = group_sum(nrm, G);
            const float dn = fmaxf(1.0f, sqrtf(nrm));
            x.x /= dn; x.y /= dn; x.z /= dn; x.w /= dn;
        }
        if (has) *reinterpret_cast<float4*>(X + o) = x;
        return;
    }
    // vdim > 256: G == 64, several float4 per lane; projection needs a second pass over the row
    if (active) {
        for (int c = gl * 4; c < vdim; c += G * 4) {
            const size_t o = static_cast<size_t>(row) * vdim + c;
            float4 x = *reinterpret_cast<float4*>(X + o);
            float4 g = *reinterpret_cast<float4*>(grad + o);
            float4 v = *reinterpret_cast<float4*>(vel + o);
            float4 m = make_float4(0, 0, 0, 0);
            if (ADAM) m = *reinterpret_cast<float4*>(mom + o);
            opt_step4<ADAM>(x, g, m, v, has_cnt, cntf, k, nrm);
            *reinterpret_cast<float4*>(grad + o) = g;
            *reinterpret_cast<float4*>(vel + o) = v;
            if (ADAM) *reinterpret_cast<float4*>(mom + o) = m;
            *reinterpret_cast<float4*>(X + o) = x;
        }
    }
    if (PROJECT) {
        nrm = group_sum(nrm, G);
        if (active) {
            const float dn = fmaxf(1.0f, sqrtf(nrm));
            for (int c = gl * 4; c < vdim; c += G * 4) {
                const size_t o = static_cast<size_t>(row) * vdim + c;
                float4 x = *reinterpret_cast<float4*>(X + o);
                x.x /= dn; x.y /= dn; x.z /= dn; x.w /= dn;
                *reinterpret_cast<float4*>(X + o) = x;
            }
        }
    }
}

// bias column Qb[I,1]: thread per item (lib/algo.cc:415-419, 448-452)
template <bool ADAM>
__global__ void sgd_update_bias_kernel(float* __restrict__ X, float* __restrict__ grad, float* __restrict__ mom,
                                       float* __restrict__ vel, const int* __restrict__ cnt, int rows, bool use_bias,
                                       OptConsts k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float FEPS = 1e-10f;
    float g = grad[i];
    if (cnt && cnt[i]) g = g / static_cast<float>(cnt[i]);  // Q-9: divided even without use_bias
    if (use_bias) {
        g = g - X[i] * k.reg2;
        if (ADAM) {
            const float m = k.b1 * mom[i] + k.omb1 * g;
            const float v = k.b2 * vel[i] + k.omb2 * (g * g);
            mom[i] = m;
            vel[i] = v;
            g = (m / k.c1) / (sqrtf(v / k.c2) + FEPS);
        } else {
            const float v = vel[i] + g * g;
            vel[i] = v;
            g = g / (sqrtf(v) + FEPS);
        }
        X[i] = X[i] + k.lr * g;
    }
    grad[i] = g;
}

// ------------------------------------------------------------------------------------------------
// grad_gather_kernel: the item side of gradient accumulation (see GatherParams in sgd_base.hpp).
// A wave owns kGatherChunk consecutive incidences of the item-sorted list.  Per 64 incidences the lanes
// fetch (item, index) coalesced and the (user, coefficient) record of the triple with one 8-byte gather;
// the wave then walks them with UN user rows in flight (dword per lane: element k*64+lane, so every load and
// every flushed atomic covers whole 128-B lines), sums  c * P[u]  in registers while the item stays the same
// and pushes the run with one atomic row add.  Bound: HBM / Infinity-Cache gathers of 4*vdim bytes per incidence.
// ------------------------------------------------------------------------------------------------
constexpr int kGatherChunk = 256;

template <int K, int UN, bool QTERM>
__global__ __launch_bounds__(256) void grad_gather_kernel(GatherParams g) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * wpb;
    const int vdim = g.vdim;
    const int64_t n_work = (g.n + kGatherChunk - 1) / kGatherChunk;
    for (int64_t w = wave0; w < n_work; w += nwaves) {
        const int64_t k_beg = w * kGatherChunk;
        const int64_t k_end = (k_beg + kGatherChunk < g.n) ? k_beg + kGatherChunk : g.n;
        int cur = -1, cnt = 0;
        float csum = 0.f;
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
        auto flush = [&]() {
            if (cur >= 0 && cnt > 0) {
                float* dst = g.gradQ + static_cast<size_t>(cur) * vdim;
                const float qs = g.a * csum + g.b * static_cast<float>(cnt);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int e = k * 64 + lane;
                    if (e < vdim) {
                        float v = g.sign * acc[k];
                        if (QTERM) v += qs * g.Q[static_cast<size_t>(cur) * vdim + e];
                        atomic_add_f32(dst + e, v);
                    }
                }
                if (lane == 0) {
                    if (g.gradQb) atomic_add_f32(g.gradQb + cur, g.sign * csum);
                    if (g.cntQ) atomicAdd(g.cntQ + cur, cnt);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = 0.f;
            csum = 0.f;
            cnt = 0;
        };
        for (int64_t k0 = k_beg; k0 < k_end; k0 += 64) {
            const int64_t kk = k0 + lane;
            int my_item = -1, my_u = -1;
            float my_c = 0.f;
            if (kk < k_end) {
                const uint32_t it = g.inc_key[kk];
                if (it < static_cast<uint32_t>(g.Q_rows)) {
                    const int32_t idx = g.inc_idx[kk];
                    bool any = false;
                    // one 8-byte gather carries the user and the coefficient (negative = rejected: BPR / WARP coefficients are >= 0 by
                    // construction -- sigmoid / log of a count; a NaN of a diverged model is NOT negative and propagates like in the reference)
                    const int64_t base = g.pos_list ? static_cast<int64_t>(idx) * g.num_neg : static_cast<int64_t>(idx);
                    const int slots = g.pos_list ? g.num_neg : 1;
                    for (int sl = 0; sl < slots; ++sl) {
                        const float2 v = g.uc[base + sl];
                        if (!(v.y < 0.f)) {
                            my_c += v.y;
                            my_u = __builtin_bit_cast(int, v.x);
                            any = true;
                        }
                    }
                    if (any) my_item = static_cast<int>(it);
                }
            }
            const int n_here = static_cast<int>((k_end - k0) < 64 ? (k_end - k0) : 64);
            for (int j0 = 0; j0 < n_here; j0 += UN) {
                float r[UN][K];
#pragma unroll
                for (int s = 0; s < UN; ++s) {
                    const int u = __builtin_amdgcn_readlane(my_u, (j0 + s) & 63);
                    if (u >= 0) {
                        const float* src = g.P + static_cast<size_t>(u) * vdim;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const int e = k * 64 + lane;
                            r[s][k] = e < vdim ? src[e] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < UN; ++s) {
                    const int u = __builtin_amdgcn_readlane(my_u, (j0 + s) & 63);
                    if (u >= 0) {
                        const int item = __builtin_amdgcn_readlane(my_item, (j0 + s) & 63);
                        if (item != cur) {
                            flush();
                            cur = item;
                        }
                        const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), (j0 + s) & 63));
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[k] += c * r[s][k];
                        csum += c;
                        cnt += 1;
                    }
                }
            }
        }
        flush();
    }
}

__global__ __launch_bounds__(256) void incidence_iota_kernel(const int32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ key_out,
                                                             int32_t* __restrict__ idx_out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= n) return;
    if (key_out) key_out[t] = static_cast<uint32_t>(keys[t]);
    idx_out[t] = static_cast<int32_t>(t);
}

// per_coordinate_normalize counts of a list that is not gathered (update_i / update_j == false still count: bpr.cc:139-143, 175-181)
__global__ __launch_bounds__(256) void incidence_count_kernel(const uint32_t* __restrict__ item, int64_t n, int q_rows, int* __restrict__ cnt) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t < n && item[t] < static_cast<uint32_t>(q_rows)) atomicAdd(cnt + item[t], 1);
}

void launch_incidence_iota(const int32_t* keys, int64_t n, uint32_t* key_out, int32_t* idx_out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(incidence_iota_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, keys, n, key_out, idx_out);
    BFH_HIP(hipGetLastError());
}

template <int K, int UN>
static void launch_grad_gather_k(const GatherParams& g, dim3 grid, hipStream_t s) {
    if (g.a != 0.f || g.b != 0.f) hipLaunchKernelGGL((grad_gather_kernel<K, UN, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((grad_gather_kernel<K, UN, false>), grid, dim3(256), 0, s, g);
}

void launch_grad_gather(const GatherParams& g, int64_t max_waves, hipStream_t s) {
    if (g.n <= 0) return;
    const int64_t n_work = (g.n + kGatherChunk - 1) / kGatherChunk;
    const int64_t waves = std::max<int64_t>(1, std::min(max_waves, n_work));
    const dim3 grid(static_cast<unsigned>((waves + 3) / 4));
    const int K = (g.vdim + 63) / 64;
    if (K <= 1) launch_grad_gather_k<1, 8>(g, grid, s);
    else if (K <= 2) launch_grad_gather_k<2, 8>(g, grid, s);
    else if (K <= 4) launch_grad_gather_k<4, 8>(g, grid, s);
    else if (K <= 8) launch_grad_gather_k<8, 4>(g, grid, s);
    else launch_grad_gather_k<16, 2>(g, grid, s);
    BFH_HIP(hipGetLastError());
}

static int acc_bits_for(int64_t range) {
    int b = 1;
    while ((int64_t(1) << b) < range) ++b;
    return b;
}

void SgdHandle::acc_prepare(int64_t triples) {
    const size_t n = static_cast<size_t>(triples);
    if (acc_uc_.size() < n) acc_uc_.resize(n);
    if (acc_neg_.size() < n) acc_neg_.resize(n);
    if (acc_key_b_.size() < n) acc_key_b_.resize(n);
    if (acc_idx_b_.size() < n) acc_idx_b_.resize(n);
    if (acc_iota_n_ < triples) {
        acc_iota_.resize(n);
        launch_incidence_iota(nullptr, triples, nullptr, acc_iota_.get(), stream);
        acc_iota_n_ = triples;
    }
}

void SgdHandle::acc_build_positive_list(const SgdParams& p, int start_x, int next_x) {
    const int64_t n = p.chunk_nnz;
    BFH_REQUIRE(n < (int64_t(1) << 31), "gradient gather: chunk of 2^31 or more interactions");
    const bool keeps = resident_ || (auto_resident_ && !chunks_.empty());
    if (keeps && acc_pos_gen_ == csr_generation_ && acc_pos_start_ == start_x && acc_pos_next_ == next_x && acc_pos_n_ == n) return;
    if (acc_pkey_.size() < static_cast<size_t>(n)) acc_pkey_.resize(static_cast<size_t>(n));
    if (acc_pidx_.size() < static_cast<size_t>(n)) acc_pidx_.resize(static_cast<size_t>(n));
    // the chunk's keys are the sort keys as they are (item ids are non-negative int32)
    device_sort_pairs_u32(reinterpret_cast<const uint32_t*>(p.keys), acc_pkey_.get(), acc_iota_.get(), acc_pidx_.get(), n, acc_bits_for(Q_rows_),
                          acc_tmp_, stream);
    acc_pos_gen_ = keeps ? csr_generation_ : -1;
    acc_pos_start_ = start_x; acc_pos_next_ = next_x; acc_pos_n_ = n;
}

void SgdHandle::acc_gather(const SgdParams& p, int num_neg, bool do_pos, bool do_neg, const float sab_pos[3], const float sab_neg[3],
                           bool with_bias) {
    const int64_t n = p.chunk_nnz, triples = n * num_neg;
    GatherParams g{};
    g.uc = acc_uc_.get();
    g.P = p.P; g.Q = p.Q;
    g.gradQ = p.gradQ;
    g.gradQb = with_bias ? p.gradQb : nullptr;
    g.cntQ = pcn_ ? p.cntQ : nullptr;
    g.num_neg = num_neg; g.vdim = vdim_; g.Q_rows = Q_rows_;
    // a streaming gather: 26-67 VGPRs leave room for 8 waves per SIMD, and latency hiding is all this kernel needs
    const int64_t waves = static_cast<int64_t>(num_cus_) * (gather_waves_per_cu_ > 0 ? gather_waves_per_cu_ : 32);
    if (do_pos) {
        g.inc_key = acc_pkey_.get(); g.inc_idx = acc_pidx_.get(); g.n = n; g.pos_list = 1;
        g.sign = sab_pos[0]; g.a = sab_pos[1]; g.b = sab_pos[2];
        launch_grad_gather(g, waves, stream);
    } else if (pcn_ && n > 0) {
        hipLaunchKernelGGL(incidence_count_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const uint32_t*>(p.keys), n, Q_rows_, p.cntQ);
        BFH_HIP(hipGetLastError());
    }
    if (!do_neg && pcn_ && triples > 0) {
        hipLaunchKernelGGL(incidence_count_kernel, dim3(static_cast<unsigned>((triples + 255) / 256)), dim3(256), 0, stream, acc_neg_.get(), triples,
                           Q_rows_, p.cntQ);
        BFH_HIP(hipGetLastError());
    }
    if (do_neg) {
        device_sort_pairs_u32(acc_neg_.get(), acc_key_b_.get(), acc_iota_.get(), acc_idx_b_.get(), triples, acc_bits_for(static_cast<int64_t>(Q_rows_) + 1),
                              acc_tmp_, stream);
        g.inc_key = acc_key_b_.get(); g.inc_idx = acc_idx_b_.get(); g.n = triples; g.pos_list = 0;
        g.sign = sab_neg[0]; g.a = sab_neg[1]; g.b = sab_neg[2];
        launch_grad_gather(g, waves, stream);
    }
}

// ------------------------------------------------------------------------------------------------
// Delta exchange between ranks (see sgd_base.hpp).  Elementwise streaming kernels over Q | Qb.
// ------------------------------------------------------------------------------------------------
// S = X - Z  (what this rank changed since the state Z every rank agrees on)
__global__ __launch_bounds__(256) void delta_begin_kernel(const float* __restrict__ X, const float* __restrict__ Z, float* __restrict__ S, int64_t n) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) S[i] = X[i] - Z[i];
}
// Z <- Z + w R: the agreed state advances by the combined deltas -- the same arithmetic on every rank, so Z stays
// bit-identical.  w (per row, exchange_weight_kernel) interpolates between the SUM of the ranks' deltas (rows that
// received few updates in the interval: the deltas are independent steps) and their MEAN (rows every rank has driven to
// its local equilibrium: the deltas are N estimates of the same move).  X: if this rank kept working since `begin`
// (progressed) its own delta is replaced by the combination, X += w R - S; otherwise X <- Z exactly, which makes the
// replicas bit-identical after a flush.
__global__ __launch_bounds__(256) void delta_finish_kernel(float* __restrict__ X, float* __restrict__ Z, const float* __restrict__ S,
                                                           const float* __restrict__ R, const float* __restrict__ W, int row_len, int64_t n,
                                                           int progressed) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
        const float r = W[i / row_len] * R[i];
        const float zn = Z[i] + r;
        X[i] = progressed ? X[i] + (r - S[i]) : zn;
        Z[i] = zn;
    }
}
// Row i receives m = upd[i] * share updates per rank in one exchange interval (upd = updates of the row per epoch over all
// ranks, share = this rank's part of the epoch inside the interval).  Under SGD a coordinate with curvature k contracts
// towards its equilibrium by exp(-x), x = lr * k * m; n deltas that started from the same state therefore combine like ONE
// run of n m updates when scaled by  w = (1 - exp(-n x)) / (n (1 - exp(-x))):  w -> 1 for cold rows (the deltas are
// independent steps: SUM), w -> 1/n for saturated ones (n estimates of the same move: MEAN).  Two curvatures: the item
// bias sees the logistic loss itself (k_b = 1/4 at most), a factor row sees reg + sigma' |p|^2 (k_q ~ the regulariser
// while the factors are small).  Measured against the single-process run at BASELINE scale, 8 ranks
// (profiles/r02_local_sgd_study_*): the plain sum leaves |Qb| 4.5x and |Q| 2x too large (lr 0.002) or diverges (lr 0.05);
// with the weights and 4 exchange points per epoch loss, |P|, |Q|, |Qb| land within 0.01 / 0.1 / 3.5 / 1.5 %.
__global__ void exchange_weight_kernel(const int* __restrict__ gcnt, const int64_t* __restrict__ cum, int64_t cum_total, int rows, double pos_scale,
                                       double neg_total, double neg_uniform, const float* __restrict__ summed, double stiff_q, double stiff_b,
                                       int n_ranks, float* __restrict__ W, float* __restrict__ Wb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    // the ranks' interval sizes and learning rates, summed by the same all-reduce as the deltas: their means are the same
    // numbers on every rank, so W -- and with it Z -- stays bit-identical across the ranks
    const double share = neg_total > 0 ? static_cast<double>(summed[0]) / n_ranks / neg_total : 0.0;
    const double lr = static_cast<double>(summed[1]) / n_ranks;
    double pneg = neg_uniform;
    if (cum) pneg = static_cast<double>(cum[i] - (i ? cum[i - 1] : 0)) / static_cast<double>(cum_total);
    const double m = (gcnt[i] * pos_scale + neg_total * pneg) * share;
    auto weight = [&](double x) { return (n_ranks > 1 && x > 1e-9) ? -expm1(-n_ranks * x) / (n_ranks * -expm1(-x)) : 1.0; };
    W[i] = static_cast<float>(weight(lr * stiff_q * m));
    Wb[i] = static_cast<float>(weight(lr * stiff_b * m));
}
__global__ void exchange_scalars_kernel(float* __restrict__ S, float interval, float lr) {
    S[0] = interval; S[1] = lr; S[2] = 0.f; S[3] = 0.f;
}
// X = Z + R  (gradients: state after the last optimizer step + every rank's accumulation since)
__global__ __launch_bounds__(256) void delta_apply_kernel(float* __restrict__ X, const float* __restrict__ Z, const float* __restrict__ R, int64_t n) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) X[i] = Z[i] + R[i];
}

static inline dim3 stream_grid(int64_t n) { return dim3(static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)))); }

void SgdHandle::set_comm(Comm* c) {
    if (x_pending_) exchange_finish();
    comm_ = c;
    x_inited_ = false;
    if (c) {
        BFH_REQUIRE(c->device == device, "set_comm: the communicator lives on another device than this handle");
        if (!x_ready_) BFH_HIP(hipEventCreateWithFlags(&x_ready_, hipEventDisableTiming));
        if (!x_done_) BFH_HIP(hipEventCreateWithFlags(&x_done_, hipEventDisableTiming));
    }
}

// Z <- the replicated state as it is now.  Called before the first local change of a model (initialize_model uploaded the
// same Q / Qb on every rank; the gradient buffers start at zero), so Z is identical on every rank.
void SgdHandle::exchange_arm() {
    if (!comm_ || x_inited_ || !model_on_gpu_) return;
    const size_t n = x_count(), nq = static_cast<size_t>(Q_rows_) * vdim_;
    if (xZ_.size() < n) { xZ_.resize(n, true, stream); xS_.resize(n, true, stream); xR_.resize(n, true, stream); }
    const bool grad = optimizer_ != "sgd";
    BFH_HIP(hipMemcpyAsync(xZ_.get(), grad ? gradQ_.get() : Q_.get(), nq * sizeof(float), hipMemcpyDeviceToDevice, stream));
    BFH_HIP(hipMemcpyAsync(xZ_.get() + nq, grad ? gradQb_.get() : Qb_.get(), static_cast<size_t>(Q_rows_) * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (!grad) {
        // per-row combination weights (1 until exchange_weights computes them) and the popularity of every item over ALL ranks
        xW_.resize(static_cast<size_t>(Q_rows_));
        xWb_.resize(static_cast<size_t>(Q_rows_));
        std::vector<float> ones(static_cast<size_t>(Q_rows_), 1.0f);
        BFH_HIP(hipMemcpyAsync(xW_.get(), ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(xWb_.get(), ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        sync_stream();
        x_gcnt_ready_ = false;
    }
    x_inited_ = true;
    x_pending_ = false;
}

// Global item popularity (positives per item over every rank's shard) from this rank's keys: one int all-reduce, once per model.
void SgdHandle::exchange_histogram(const int32_t* keys, int64_t n) {
    if (!comm_ || x_gcnt_ready_) return;
    x_gcnt_.resize(static_cast<size_t>(Q_rows_), true, stream);
    if (n > 0) {
        hipLaunchKernelGGL(incidence_count_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const uint32_t*>(keys), n, Q_rows_, x_gcnt_.get());
        BFH_HIP(hipGetLastError());
    }
    comm_->all_reduce_i32(x_gcnt_.get(), x_gcnt_.get(), static_cast<size_t>(Q_rows_), stream);
    sync_stream();
    double tot[1] = {static_cast<double>(n)};
    if (comm_->size() > 1) {
        DevBuf<double> d;
        d.resize(1);
        BFH_HIP(hipMemcpyAsync(d.get(), tot, sizeof(double), hipMemcpyHostToDevice, stream));
        comm_->all_reduce_f64(d.get(), d.get(), 1, stream);
        BFH_HIP(hipMemcpyAsync(tot, d.get(), sizeof(double), hipMemcpyDeviceToHost, stream));
        sync_stream();
    }
    x_gcnt_total_ = tot[0];
    x_gcnt_ready_ = true;
}

// what the exchange that begins next covers: `interval_triples` = this rank's triples inside it, `lr` = its learning rate.
// The weights themselves are computed in exchange_finish from the sums over the ranks (exchange_weight_kernel).
void SgdHandle::exchange_weights(double interval_triples, double lr, int num_neg, bool uniform) {
    x_w_interval_ = interval_triples;
    x_w_lr_ = lr;
    x_w_num_neg_ = num_neg;
    x_w_uniform_ = uniform;
}

// Q | Qb -> S (and Z), one all-reduce on the communicator's stream behind everything issued on `stream` so far
void SgdHandle::exchange_begin() {
    if (!comm_ || comm_->size() < 1 || !model_on_gpu_) return;
    if (x_pending_) exchange_finish(true);
    const int64_t nq = static_cast<int64_t>(Q_rows_) * vdim_;
    const bool grad = optimizer_ != "sgd";
    BFH_REQUIRE(!grad, "exchange_begin is the Hogwild (sgd) exchange");
    BFH_REQUIRE(x_inited_, "exchange_begin before exchange_arm");
    float* S = xS_.get();
    const int xk = t_xk_.begin(stream);
    hipLaunchKernelGGL(delta_begin_kernel, stream_grid(nq), dim3(256), 0, stream, static_cast<const float*>(Q_.get()),
                       static_cast<const float*>(xZ_.get()), S, nq);
    hipLaunchKernelGGL(delta_begin_kernel, stream_grid(Q_rows_), dim3(256), 0, stream, static_cast<const float*>(Qb_.get()),
                       static_cast<const float*>(xZ_.get() + nq), S + nq, static_cast<int64_t>(Q_rows_));
    hipLaunchKernelGGL(exchange_scalars_kernel, dim3(1), dim3(1), 0, stream, S + x_scalars(), static_cast<float>(x_w_interval_), static_cast<float>(x_w_lr_));
    BFH_HIP(hipGetLastError());
    t_xk_.end(xk, stream);
    BFH_HIP(hipEventRecord(x_ready_, stream));
    BFH_HIP(hipStreamWaitEvent(comm_->comm_stream(), x_ready_, 0));
    const int ar = t_ar_.begin(comm_->comm_stream());
    comm_->all_reduce_f32(S, xR_.get(), x_count(), comm_->comm_stream());
    t_ar_.end(ar, comm_->comm_stream());
    BFH_HIP(hipEventRecord(x_done_, comm_->comm_stream()));
    x_p_num_neg_ = x_w_num_neg_;   // what exchange_finish weighs THIS exchange with (exchange_weights may be called for a later one meanwhile)
    x_p_uniform_ = x_w_uniform_;
    x_pending_ = true;
    stats.exchanges += 1;
}

void SgdHandle::exchange_finish(bool progressed) {
    if (!x_pending_) return;
    const int64_t nq = static_cast<int64_t>(Q_rows_) * vdim_;
    BFH_HIP(hipStreamWaitEvent(stream, x_done_, 0));
    const int xk = t_xk_.begin(stream);
    if (x_gcnt_ready_) {
        const double glob_triples = static_cast<double>(num_nnz_) * x_p_num_neg_;          // one epoch over all ranks
        const double pos_scale = x_gcnt_total_ > 0 ? static_cast<double>(num_nnz_) / x_gcnt_total_ * x_p_num_neg_ : 0.0;   // counted keys -> the whole matrix
        hipLaunchKernelGGL(exchange_weight_kernel, dim3((Q_rows_ + 255) / 256), dim3(256), 0, stream, static_cast<const int*>(x_gcnt_.get()),
                           x_p_uniform_ ? nullptr : static_cast<const int64_t*>(cum_.get()), cum_total_, Q_rows_, pos_scale, glob_triples,
                           x_p_uniform_ ? 1.0 / Q_rows_ : 0.0, static_cast<const float*>(xR_.get() + x_scalars()), comm_stiffness_q_milli_ * 1e-3,
                           comm_stiffness_milli_ * 1e-3, comm_->size(), xW_.get(), xWb_.get());
    }
    hipLaunchKernelGGL(delta_finish_kernel, stream_grid(nq), dim3(256), 0, stream, Q_.get(), xZ_.get(), static_cast<const float*>(xS_.get()),
                       static_cast<const float*>(xR_.get()), static_cast<const float*>(xW_.get()), vdim_, nq, progressed ? 1 : 0);
    hipLaunchKernelGGL(delta_finish_kernel, stream_grid(Q_rows_), dim3(256), 0, stream, Qb_.get(), xZ_.get() + nq, static_cast<const float*>(xS_.get() + nq),
                       static_cast<const float*>(xR_.get() + nq), static_cast<const float*>(xWb_.get()), 1, static_cast<int64_t>(Q_rows_), progressed ? 1 : 0);
    BFH_HIP(hipGetLastError());
    t_xk_.end(xk, stream);
    x_pending_ = false;
}

// adam / adagrad / WARP: sum what every rank accumulated into gradQ | gradQb (| counts) since the last optimizer step
void SgdHandle::exchange_gradients() {
    if (!comm_ || optimizer_ == "sgd") return;
    const int64_t nq = static_cast<int64_t>(Q_rows_) * vdim_;
    float* S = xS_.get();
    int xk = t_xk_.begin(stream);
    // Z holds the residue the gradient buffers kept after the last step (Q-6: they are never re-zeroed); zero before the first
    hipLaunchKernelGGL(delta_begin_kernel, stream_grid(nq), dim3(256), 0, stream, static_cast<const float*>(gradQ_.get()),
                       static_cast<const float*>(xZ_.get()), S, nq);
    hipLaunchKernelGGL(delta_begin_kernel, stream_grid(Q_rows_), dim3(256), 0, stream, static_cast<const float*>(gradQb_.get()),
                       static_cast<const float*>(xZ_.get() + nq), S + nq, static_cast<int64_t>(Q_rows_));
    BFH_HIP(hipGetLastError());
    t_xk_.end(xk, stream);
    const int ar = t_ar_.begin(stream);
    comm_->group_start();
    comm_->all_reduce_f32(S, xR_.get(), static_cast<size_t>(nq) + static_cast<size_t>(Q_rows_), stream);
    if (pcn_) comm_->all_reduce_i32(cntQ_.get(), cntQ_.get(), static_cast<size_t>(Q_rows_), stream);
    comm_->group_end();
    t_ar_.end(ar, stream);
    xk = t_xk_.begin(stream);
    hipLaunchKernelGGL(delta_apply_kernel, stream_grid(nq), dim3(256), 0, stream, gradQ_.get(), static_cast<const float*>(xZ_.get()),
                       static_cast<const float*>(xR_.get()), nq);
    hipLaunchKernelGGL(delta_apply_kernel, stream_grid(Q_rows_), dim3(256), 0, stream, gradQb_.get(), static_cast<const float*>(xZ_.get() + nq),
                       static_cast<const float*>(xR_.get() + nq), static_cast<int64_t>(Q_rows_));
    BFH_HIP(hipGetLastError());
    t_xk_.end(xk, stream);
    stats.exchanges += 1;
}

// ------------------------------------------------------------------------------------------------
void SgdHandle::unpin_host() {
    // nothing of this handle may still be writing into a page-locked array when its registration goes (every entry point synchronises
    // before it returns, so this costs nothing; it is the belt to that pair of braces)
    if (!pinned_.empty() && stream) (void)hipStreamSynchronize(stream);
    for (auto& p : pinned_) (void)hipHostUnregister(p.first);
    if (!pinned_.empty()) (void)hipGetLastError();   // an array that was freed while pinned must not leave a sticky error behind
    pinned_.clear();
}

SgdHandle::~SgdHandle() {
    if (host_stale_ && model_on_gpu_ && hostP_) {   // lazy_sync: the deferred copy-back happens at the latest here
        try { synchronize(true, true); } catch (...) {}
    }
    unpin_host();
    if (x_pending_ && x_done_) (void)hipEventSynchronize(x_done_);
    if (x_ready_) (void)hipEventDestroy(x_ready_);
    if (x_done_) (void)hipEventDestroy(x_done_);
    if (stream) (void)hipStreamDestroy(stream);
}

bool SgdHandle::init(const char* opt_path) {
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
    num_iters_ = opt_.integer("num_iters");
    optimizer_ = opt_.str("optimizer");
    BFH_REQUIRE(optimizer_ == "sgd" || optimizer_ == "adam" || optimizer_ == "adagrad",
                "optimizer must be one of sgd, adam, adagrad");
    lr_ = opt_.num("lr");
    min_lr_ = opt_.num_or("min_lr", lr_);
    beta1_ = opt_.num_or("beta1", 0.9);
    reg_u_ = static_cast<float>(opt_.num("reg_u"));
    reg_i_ = static_cast<float>(opt_.num("reg_i"));
    reg_j_ = static_cast<float>(opt_.num("reg_j"));
    reg_b_d_ = opt_.num_or("reg_b", 0.0);
    reg_b_ = static_cast<float>(reg_b_d_);
    update_i_ = opt_.boolean_or("update_i", true);
    update_j_ = opt_.boolean_or("update_j", true);
    use_bias_ = opt_.boolean_or("use_bias", false);
    pcn_ = opt_.boolean_or("per_coordinate_normalize", false);
    compute_loss_ = opt_.boolean_or("compute_loss_on_training", false);
    // the reference's CUDA backend reads a non-existent key "rand_seed" (bpr.cu:262, Q-19);
    // this backend follows the CPU path and uses "random_seed" (bpr.cc:83).
    seed_ = static_cast<uint32_t>(opt_.num_or("random_seed", 0));
    parse_specific();
    scratch_.resize(8, true, stream);
    inited_ = true;
    return true;
}

void SgdHandle::initialize_model(float* P, int P_rows, float* Q, float* Qb, int Q_rows, int64_t num_nnz, bool set_gpu) {
    BFH_REQUIRE(inited_, "initialize_model called before init");
    BFH_REQUIRE(P && Q && Qb && P_rows > 0 && Q_rows > 0, "initialize_model: null factors or empty shapes");
    if (host_stale_ && model_on_gpu_ && hostP_) synchronize(true, true);   // lazy_sync: the previous model's arrays are still owed
    hostP_ = P; hostQ_ = Q; hostQb_ = Qb;
    P_rows_ = P_rows; Q_rows_ = Q_rows;
    num_nnz_ = num_nnz;
    if (!set_gpu) return;  // bpr.cu:293-298: only record host pointers
    const size_t np = static_cast<size_t>(P_rows) * vdim_, nq = static_cast<size_t>(Q_rows) * vdim_;
    unpin_host();
    if (pin_host_) {   // opt-in ("pin_host" = 1; the default copies back through the library's own pinned ring, HostStager).  Best effort: a refusal just leaves the copies pageable
        // only arrays of a MiB or more: those sit in pages of their own (malloc hands them out by mmap); a small array shares its
        // pages with whatever else lives on the heap -- including arrays another handle has registered -- and overlapping
        // registrations have aborted inside the runtime (seen once in tests/test_errors_gpu.py, on 1.5 KB factors)
        auto pin = [&](void* p, size_t bytes) {
            if (bytes < (size_t(1) << 20)) return;
            if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) pinned_.emplace_back(p, bytes);
            else (void)hipGetLastError();
        };
        pin(P, np * sizeof(float));
        pin(Q, nq * sizeof(float));
        pin(Qb, static_cast<size_t>(Q_rows) * sizeof(float));
    }
    P_.resize(np); Q_.resize(nq); Qb_.resize(Q_rows);
    BFH_HIP(hipMemcpyAsync(P_.get(), P, np * sizeof(float), hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(Q_.get(), Q, nq * sizeof(float), hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(Qb_.get(), Qb, Q_rows * sizeof(float), hipMemcpyHostToDevice, stream));
    stats.h2d_bytes += static_cast<double>((np + nq + Q_rows) * sizeof(float));
    if (optimizer_ != "sgd") {  // lib/algo.cc:221-255
        gradP_.resize(np, true, stream); gradQ_.resize(nq, true, stream); gradQb_.resize(Q_rows, true, stream);
        velP_.resize(np, true, stream); velQ_.resize(nq, true, stream); velQb_.resize(Q_rows, true, stream);
        if (optimizer_ == "adam") {
            momP_.resize(np, true, stream); momQ_.resize(nq, true, stream); momQb_.resize(Q_rows, true, stream);
        }
        if (pcn_) {
            cntP_.resize(P_rows, true, stream);
            cntQ_.resize(Q_rows, true, stream);
        }
    }
    iters_ = 0;
    epoch_ = 0;
    processed_ = 0;
    model_on_gpu_ = true;
    if (x_pending_ && x_done_) (void)hipEventSynchronize(x_done_);
    x_pending_ = false;
    x_inited_ = false;
    sync_stream();
}

void SgdHandle::set_placeholder(const int64_t* indptr, size_t batch_size) {
    BFH_REQUIRE(P_rows_ > 0, "set_placeholder called before initialize_model");
    BFH_REQUIRE(indptr, "set_placeholder: null indptr");
    indptr_host_.assign(indptr, indptr + P_rows_);
    indptr_.resize(P_rows_);
    BFH_HIP(hipMemcpyAsync(indptr_.get(), indptr, P_rows_ * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    stats.h2d_bytes += static_cast<double>(P_rows_ * sizeof(int64_t));
    if (!resident_) {
        keys_.resize(batch_size);
        rows_.resize(batch_size);
        chunks_.clear();
    }
    placeholder_set_ = true;
    sync_stream();
}

void SgdHandle::set_resident_csr(const int64_t* indptr, const int32_t* keys, int64_t nnz) {
    BFH_REQUIRE(P_rows_ > 0, "set_resident_csr called before initialize_model");
    BFH_REQUIRE(indptr && keys && nnz >= 0, "set_resident_csr: null arrays");
    BFH_REQUIRE(indptr[P_rows_ - 1] == nnz, "set_resident_csr: indptr[-1] != nnz");
    indptr_host_.assign(indptr, indptr + P_rows_);
    indptr_.resize(P_rows_);
    keys_.resize(static_cast<size_t>(nnz));
    rows_.resize(static_cast<size_t>(nnz));
    BFH_HIP(hipMemcpyAsync(indptr_.get(), indptr, P_rows_ * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    BFH_HIP(hipMemcpyAsync(keys_.get(), keys, nnz * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    stats.h2d_bytes += static_cast<double>(P_rows_ * sizeof(int64_t) + nnz * sizeof(int32_t));
    launch_fill_rows(indptr_.get(), 0, P_rows_, 0, nnz, rows_.get(), stream);
    resident_ = true;
    csr_generation_ += 1;
    resident_nnz_ = nnz;
    placeholder_set_ = true;
    sync_stream();
}

void SgdHandle::set_cumulative_table(const int64_t* table) {
    BFH_REQUIRE(Q_rows_ > 0, "set_cumulative_table called before initialize_model");
    BFH_REQUIRE(table, "set_cumulative_table: null table");
    cum_.resize(Q_rows_);
    BFH_HIP(hipMemcpyAsync(cum_.get(), table, Q_rows_ * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    cum_total_ = table[Q_rows_ - 1];
    have_cum_ = true;
    sync_stream();
}

int64_t SgdHandle::stage_chunk(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, SgdParams* pr) {
    BFH_REQUIRE(model_on_gpu_, "partial_update before initialize_model(..., set_gpu=True)");
    BFH_REQUIRE(placeholder_set_, "partial_update before set_placeholder");
    BFH_REQUIRE(0 <= start_x && start_x <= next_x && next_x <= P_rows_, "partial_update: bad row range");
    const int64_t* ip = indptr ? indptr : indptr_host_.data();
    const int64_t beg = start_x == 0 ? 0 : ip[start_x - 1];
    const int64_t end = next_x == 0 ? 0 : ip[next_x - 1];
    const int64_t n = end - beg;
    std::memset(pr, 0, sizeof(*pr));
    pr->P = P_.get(); pr->Q = Q_.get(); pr->Qb = Qb_.get();
    pr->gradP = gradP_.get(); pr->gradQ = gradQ_.get(); pr->gradQb = gradQb_.get();
    pr->cntP = cntP_.get(); pr->cntQ = cntQ_.get();
    pr->indptr = indptr_.get();
    pr->cum_table = have_cum_ ? cum_.get() : nullptr;
    pr->chunk_nnz = n;
    pr->shift = beg;
    pr->nnz_offset = nnz_offset_;
    pr->P_rows = P_rows_; pr->Q_rows = Q_rows_; pr->d = d_; pr->vdim = vdim_;
    pr->seed = seed_; pr->epoch = epoch_;
    if (n == 0) return 0;
    if (keys) {
        if (resident_) {
            // caller insists on passing keys although a resident CSR exists: honour the resident copy
            pr->keys = keys_.get() + beg;
            pr->rows = rows_.get() + beg;
        } else if (auto_resident_) {
            // every chunk keeps its place in a full-size device copy of the matrix; it is uploaded when its row range is new or
            // the 64-bit hash over the whole host buffer differs from the one it was uploaded with
            const int64_t total = indptr_host_.empty() ? n : indptr_host_.back();
            BFH_REQUIRE(end <= total, "partial_update: indptr disagrees with the placeholder's");
            if (keys_.size() < static_cast<size_t>(total)) {
                keys_.resize(static_cast<size_t>(total));
                rows_.resize(static_cast<size_t>(total));
                chunks_.clear();
            }
            const uint64_t sig = content_signature(keys, n);
            auto it = chunks_.find({start_x, next_x});
            if (it == chunks_.end() || it->second.n != n || it->second.sig != sig) {
                BFH_HIP(hipMemcpyAsync(keys_.get() + beg, keys, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                stats.h2d_bytes += static_cast<double>(n * sizeof(int32_t));
                launch_fill_rows(indptr_.get(), start_x, next_x, beg, n, rows_.get() + beg, stream);
                chunks_[{start_x, next_x}] = ChunkSig{n, sig};
                csr_generation_ += 1;   // whatever was derived from the old content (item-major regrouping, incidence lists) is stale
            }
            pr->keys = keys_.get() + beg;
            pr->rows = rows_.get() + beg;
        } else {
            BFH_REQUIRE(static_cast<size_t>(n) <= keys_.size(), "partial_update: chunk larger than the placeholder batch_size");
            BFH_HIP(hipMemcpyAsync(keys_.get(), keys, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
            stats.h2d_bytes += static_cast<double>(n * sizeof(int32_t));
            launch_fill_rows(indptr_.get(), start_x, next_x, beg, n, rows_.get(), stream);
            pr->keys = keys_.get();
            pr->rows = rows_.get();
        }
    } else {
        BFH_REQUIRE(resident_, "partial_update: keys == NULL needs bfh_*_set_resident_csr first");
        pr->keys = keys_.get() + beg;
        pr->rows = rows_.get() + beg;
    }
    return n;
}

// Deterministic form of the reference's lr decay (lib/algo.cc:280-287, Q-8): the lr of a call is
// the one the progress thread would publish after all previously submitted jobs completed.  Same
// rule as CuBPR's host-side decay (bpr.cu:399-401) but with the CPU path's progress accounting.
double SgdHandle::current_lr() const {
    const double total = static_cast<double>(num_nnz_) * static_cast<double>(num_iters_);
    const double progress = total > 0 ? processed_ / total : 0.0;
    double a = lr_ - (lr_ - min_lr_) * progress;
    return a > min_lr_ ? a : min_lr_;
}

void SgdHandle::advance_progress(int start_x, int next_x, const int64_t* ip_arg) {
    const int64_t* ip = ip_arg ? ip_arg : indptr_host_.data();
    // job.size = 1 (user slot) + n_pos for every non-empty user (include/buffalo/algo.hpp:38-42)
    int64_t sz = 0;
    int64_t prev = start_x == 0 ? 0 : ip[start_x - 1];
    for (int x = start_x; x < next_x; ++x) {
        const int64_t e = ip[x];
        if (e > prev) sz += 1 + (e - prev);
        prev = e;
    }
    processed_ += static_cast<double>(sz) * num_shards_;
}

void SgdHandle::harvest_timers() {
    stats.kernel_ms += t_main_.drain();
    stats.optimizer_ms += t_opt_.drain();
    stats.aux_ms += t_aux_.drain();
    stats.exchange_kernel_ms += t_xk_.drain();
    stats.allreduce_ms += t_ar_.drain();
}

void SgdHandle::synchronize(bool device_to_host, bool force) {
    BFH_REQUIRE(hostP_ && model_on_gpu_, "synchronize before initialize_model(..., set_gpu=True)");
    exchange_finish();
    if (device_to_host && lazy_sync_ && !force) {   // the copy is owed until synchronize(2) / destroy / the next initialize_model
        host_stale_ = true;
        sync_stream();
        return;
    }
    if (device_to_host) host_stale_ = false;
    const size_t np = static_cast<size_t>(P_rows_) * vdim_, nq = static_cast<size_t>(Q_rows_) * vdim_;
    const hipMemcpyKind kind = device_to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
    if (device_to_host) {
        // through the library's own pinned ring (HostStager, common.hpp) unless the caller asked for its arrays to be registered
        // ("pin_host" = 1); an array that is no longer mapped is an error, not a fault
        for (auto& c : {std::make_pair(static_cast<void*>(hostP_), np * sizeof(float)), std::make_pair(static_cast<void*>(hostQ_), nq * sizeof(float)),
                        std::make_pair(static_cast<void*>(hostQb_), static_cast<size_t>(Q_rows_) * sizeof(float))})
            if (!host_range_mapped(c.first, c.second))
                throw Error(BFH_ERR_INVALID, "the caller's factor array is no longer mapped (freed while the model still owes it a copy?)");
        if (!pinned_.empty()) {
            BFH_HIP(hipMemcpyAsync(hostP_, P_.get(), np * sizeof(float), kind, stream));
            BFH_HIP(hipMemcpyAsync(hostQ_, Q_.get(), nq * sizeof(float), kind, stream));
        } else {
            stager_.d2h(hostP_, P_.get(), np * sizeof(float), stream, device);
            stager_.d2h(hostQ_, Q_.get(), nq * sizeof(float), stream, device);
        }
        BFH_HIP(hipMemcpyAsync(hostQb_, Qb_.get(), Q_rows_ * sizeof(float), kind, stream));
        stats.d2h_bytes += static_cast<double>((np + nq + Q_rows_) * sizeof(float));
    } else {
        BFH_HIP(hipMemcpyAsync(P_.get(), hostP_, np * sizeof(float), kind, stream));
        BFH_HIP(hipMemcpyAsync(Q_.get(), hostQ_, nq * sizeof(float), kind, stream));
        BFH_HIP(hipMemcpyAsync(Qb_.get(), hostQb_, Q_rows_ * sizeof(float), kind, stream));
        stats.h2d_bytes += static_cast<double>((np + nq + Q_rows_) * sizeof(float));
    }
    sync_stream();
}

void SgdHandle::update_parameters() {
    BFH_REQUIRE(model_on_gpu_, "update_parameters before initialize_model(..., set_gpu=True)");
    if (optimizer_ != "sgd") {
        if (comm_) {
            exchange_arm();
            exchange_gradients();
        }
        const bool adam = optimizer_ == "adam";
        const double beta1 = beta1_, beta2 = beta1_;  // Q-5: beta2 is read from "beta1" (lib/algo.cc:396)
        OptConsts k;
        k.lr = static_cast<float>(lr_);
        k.b1 = static_cast<float>(beta1); k.omb1 = static_cast<float>(1.0 - beta1);
        k.b2 = static_cast<float>(beta2); k.omb2 = static_cast<float>(1.0 - beta2);
        k.c1 = static_cast<float>(1.0 - std::pow(beta1, iters_ + 1));
        k.c2 = static_cast<float>(1.0 - std::pow(beta2, iters_ + 1));
        int G = 8;
        while (G < 64 && G * 4 < vdim_) G <<= 1;
        const int rpw = 64 / G;
        const bool proj = project_unit_ball();
        auto run = [&](float* X, float* g, float* m, float* v, const int* cnt, int rows, float reg) {
            k.reg2 = static_cast<float>(2.0 * static_cast<double>(reg));
            const int waves = (rows + rpw - 1) / rpw;
            const int blocks = (waves + 3) / 4;
            if (adam && proj) hipLaunchKernelGGL((sgd_update_rows_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, X, g, m, v, cnt, rows, vdim_, G, k);
            else if (adam) hipLaunchKernelGGL((sgd_update_rows_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, X, g, m, v, cnt, rows, vdim_, G, k);
            else if (proj) hipLaunchKernelGGL((sgd_update_rows_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, X, g, m, v, cnt, rows, vdim_, G, k);
            else hipLaunchKernelGGL((sgd_update_rows_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, X, g, m, v, cnt, rows, vdim_, G, k);
            BFH_HIP(hipGetLastError());
        };
        const int slot = t_opt_.begin(stream);
        run(P_.get(), gradP_.get(), momP_.get(), velP_.get(), pcn_ ? cntP_.get() : nullptr, P_rows_, reg_u_);
        // reference order: Q rows, with the bias handled inside the same loop (algo.cc:407-420)
        {
            k.reg2 = static_cast<float>(2.0 * static_cast<double>(reg_b_));
            const int blocks = (Q_rows_ + 255) / 256;
            if (adam) hipLaunchKernelGGL((sgd_update_bias_kernel<true>), dim3(blocks), dim3(256), 0, stream, Qb_.get(), gradQb_.get(), momQb_.get(), velQb_.get(), pcn_ ? cntQ_.get() : nullptr, Q_rows_, use_bias_, k);
            else hipLaunchKernelGGL((sgd_update_bias_kernel<false>), dim3(blocks), dim3(256), 0, stream, Qb_.get(), gradQb_.get(), momQb_.get(), velQb_.get(), pcn_ ? cntQ_.get() : nullptr, Q_rows_, use_bias_, k);
            BFH_HIP(hipGetLastError());
        }
        run(Q_.get(), gradQ_.get(), momQ_.get(), velQ_.get(), pcn_ ? cntQ_.get() : nullptr, Q_rows_, reg_i_);
        if (pcn_) {
            BFH_HIP(hipMemsetAsync(cntP_.get(), 0, cntP_.bytes(), stream));
            BFH_HIP(hipMemsetAsync(cntQ_.get(), 0, cntQ_.bytes(), stream));
        }
        if (comm_ && x_inited_) {   // what the gradient buffers keep after the step (Q-6) is the base of the next delta
            const size_t nq = static_cast<size_t>(Q_rows_) * vdim_;
            BFH_HIP(hipMemcpyAsync(xZ_.get(), gradQ_.get(), nq * sizeof(float), hipMemcpyDeviceToDevice, stream));
            BFH_HIP(hipMemcpyAsync(xZ_.get() + nq, gradQb_.get(), static_cast<size_t>(Q_rows_) * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
        t_opt_.end(slot, stream);
    }
    iters_ += 1;
    epoch_ += 1;
    sync_stream();
    harvest_timers();
}

void SgdHandle::set_mode(const std::string& name, int64_t v) {
    if (name == "sequential") sequential_ = static_cast<int>(v);
    else if (name == "hogwild_atomic") {
        BFH_REQUIRE(v == 0 || v == 1 || ((v == 2 || v == 3) && kind_ == 0),
                    "hogwild_atomic must be 0 (write-through stores), 1 (fp32 atomics) or, for BPRMF, 2 (per-XCD replicas) / 3 (item-major)");
        hogwild_atomic_ = static_cast<int>(v);
    }
    else if (name == "xcd_sync_updates") { BFH_REQUIRE(v >= 1, "xcd_sync_updates must be positive"); xcd_sync_updates_ = v; }
    else if (name == "xcd_merge_mean") xcd_merge_mean_ = v != 0;
    else if (name == "xcd_stiff_q" || name == "xcd_stiff_b" || name == "xcd_stiff_p") {
        BFH_REQUIRE(v >= 0 && v <= 100000, name + " is a curvature in permille, 0 (plain sum) .. 100000");
        (name == "xcd_stiff_q" ? xcd_stiff_q_milli_ : name == "xcd_stiff_b" ? xcd_stiff_b_milli_ : xcd_stiff_p_milli_) = static_cast<int>(v);
    }
    else if (name == "im_user_lr_max") { BFH_REQUIRE(v >= 0, "im_user_lr_max is a learning rate in permille >= 0"); im_user_lr_max_milli_ = static_cast<int>(v); }
    else if (name == "xcd_fresh") xcd_fresh_ = v != 0 ? 1 : 0;
    else if (name == "im_drift_budget") { BFH_REQUIRE(v >= 0, "im_drift_budget is a permille value >= 0"); im_drift_budget_milli_ = static_cast<int>(v); }
    else if (name == "im_blocks") { BFH_REQUIRE(v >= 0 && v <= 64, "im_blocks must be in [0,64] (0 = choose from the learning rate)"); im_blocks_ = static_cast<int>(v); }
    else if (name == "im_presample") im_presample_ = v != 0;
    else if (name == "im_presample_ahead") im_presample_ahead_ = v != 0;
    else if (name == "im_drain_only") im_drain_only_ = v != 0;
    else if (name == "im_single_wave") im_single_wave_ = v != 0;
    else if (name == "im_trace") { BFH_REQUIRE(v >= 0, "im_trace is a capacity in triples"); im_trace_.resize(static_cast<size_t>(v), true, stream); sync_stream(); }
    else if (name == "im_force_queues") { BFH_REQUIRE(v >= 0 && v <= 8, "im_force_queues must be in [0,8]"); im_force_queues_ = static_cast<int>(v); }
    else if (name == "im_p_nt") im_p_nt_ = v != 0;
    else if (name == "im_study") im_study_ = static_cast<int>(v);
    else if (name == "im_dual_generic") im_dual_generic_ = v != 0;
    else if (name == "xcd_stiff_lr_ref") xcd_stiff_lr_ref_micro_ = static_cast<int>(v);
    else if (name == "im_dual") { BFH_REQUIRE(v >= -1 && v <= 1, "im_dual must be -1 (by the call's size), 0 or 1"); im_dual_ = static_cast<int>(v); }
    else if (name == "im_neg_limit") { BFH_REQUIRE(v >= 0, "im_neg_limit must be >= 0"); im_neg_limit_ = static_cast<int>(v); }
    else if (name == "im_user_hybrid") { BFH_REQUIRE(v >= 0 && v <= 2, "im_user_hybrid must be 0 (off), 1 (heavy users over all queues) or 2 (over as few as needed)"); im_user_hybrid_ = static_cast<int>(v); }
    else if (name == "im_user_replicas") { BFH_REQUIRE(v >= -1 && v <= 1, "im_user_replicas must be -1 (by shard size), 0 or 1"); im_user_replicas_ = static_cast<int>(v); }
    else if (name == "im_max_stale") { BFH_REQUIRE(v >= 1, "im_max_stale must be positive"); im_max_stale_ = static_cast<int>(v); }
    else if (name == "xcd_v4") xcd_v4_ = v != 0;
    else if (name == "xcd_hot_tau") { BFH_REQUIRE(v >= 0, "xcd_hot_tau is a permille value >= 0"); xcd_hot_tau_ = static_cast<int>(v); }
    else if (name == "accum_two_pass") accum_two_pass_ = v != 0;
    else if (name == "gather_waves_per_cu") gather_waves_per_cu_ = static_cast<int>(v);
    else if (name == "auto_resident") auto_resident_ = v != 0;
    else if (name == "lazy_sync") lazy_sync_ = v != 0;
    else if (name == "pin_host") pin_host_ = v != 0;
    else if (name == "comm_overlap") comm_overlap_ = v != 0;
    else if (name == "comm_stiffness") { BFH_REQUIRE(v >= 0, "comm_stiffness is a permille value >= 0 (0: plain sum of the deltas)"); comm_stiffness_milli_ = static_cast<int>(v); }
    else if (name == "comm_stiffness_q") { BFH_REQUIRE(v >= 0, "comm_stiffness_q is a permille value >= 0"); comm_stiffness_q_milli_ = static_cast<int>(v); }
    else if (name == "comm_segments") { BFH_REQUIRE(v >= 0 && v <= 64, "comm_segments must be in [0,64] (0 = one blocking exchange per call)"); comm_segments_ = static_cast<int>(v); }
    else if (name == "prefetch") prefetch_ = static_cast<int>(v);
    else if (name == "waves_per_cu") waves_per_cu_ = static_cast<int>(v);
    else if (name == "chunk") { BFH_REQUIRE(v >= 64 && v % 64 == 0, "chunk must be a positive multiple of 64"); chunk_ = static_cast<int>(v); chunk_set_ = true; }
    else if (name == "timing") timing = v != 0;
    else if (name == "epoch") epoch_ = static_cast<uint32_t>(v);
    else throw Error(BFH_ERR_INVALID, "unknown mode '" + name + "'");
}

void SgdHandle::device_buffer(const std::string& name, void** p, size_t* bytes) {
    if (x_pending_) {   // the caller is about to read (or all-reduce) the replicated tensors itself
        exchange_finish();
        sync_stream();
    }
    struct { const char* n; void* ptr; size_t b; } tab[] = {
        {"P", P_.get(), P_.bytes()}, {"Q", Q_.get(), Q_.bytes()}, {"Qb", Qb_.get(), Qb_.bytes()},
        {"gradP", gradP_.get(), gradP_.bytes()}, {"gradQ", gradQ_.get(), gradQ_.bytes()}, {"gradQb", gradQb_.get(), gradQb_.bytes()},
        {"countP", cntP_.get(), cntP_.bytes()}, {"countQ", cntQ_.get(), cntQ_.bytes()},
        {"velP", velP_.get(), velP_.bytes()}, {"velQ", velQ_.get(), velQ_.bytes()},
        {"momP", momP_.get(), momP_.bytes()}, {"momQ", momQ_.get(), momQ_.bytes()},
        {"im_trace", im_trace_.get(), im_trace_.bytes()},
    };
    for (auto& t : tab)
        if (name == t.n) {
            *p = t.ptr;
            *bytes = t.b;
            return;
        }
    throw Error(BFH_ERR_INVALID, "unknown device buffer '" + name + "'");
}

}  // namespace bfh
