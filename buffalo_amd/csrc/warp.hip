// WARP on gfx950: rejection-sampling trial loop + gradient accumulation kernel, loss kernel, C ABI.
//
// Reference semantics: CWARP::worker / update_parameters / compute_loss
// (/root/reference/lib/algo_impl/warp/warp.cc:103-226).  The reference has no GPU WARP
// (/root/reference/buffalo/algo/warp.py:31-32); the object surface is the one warp.py:212-234
// expects from an accelerator (same as CuBPR).
//
// Kernel shape (wave64): a wave owns `chunk` consecutive nnz positions, keeps P[u] in registers
// (K = vdim/64 dwords per lane) across the user's run, and for every positive
//   * takes the first S unseen candidate negatives from a pre-pass (warp_presample_kernel: one thread per positive
//     in CSR order draws Philox candidates and tests them against the user's sorted key run -- "seen" negatives are
//     not counted, warp.cc:137-138 -- so the dependent binary-search chain is off the wave's critical path and runs in
//     cached key runs); beyond S it draws 64 candidates at once (one Philox draw per lane) and tests them in parallel,
//   * scores unseen candidates in draw order, 1/2/4 rows per step (speculation grows when early
//     candidates do not violate the margin), emulating the reference's trial counter exactly
//     (Q-10: the k-th counted candidate is scored at trial = 2k),
//   * accumulates gradP in registers per run; the item-side rows go through the sorted gather of sgd_base.hpp
//     (or, accum_two_pass = 0, fp32 atomics).
// P and Q are frozen during the epoch (only gradients change), so results are independent of the
// order in which waves run, up to fp32 summation order.
#include "sgd_base.hpp"

#include "comm.hpp"

namespace bfh {

struct WarpConsts {
    float reg_u, reg_i, reg_j;
    double threshold;
    int max_trial, l2, pcn;
    int chunk;
    int64_t total;
    double* loss_out;        // [0] partial loss sum
    unsigned long long* cnt; // [0] scored negatives, [1] accepted positives, [2] candidate rows fetched (scored + speculated)
    // two-pass accumulation (sgd_base.hpp GatherParams): record Phi and the violating negative of every accepted
    // positive (Q_rows: rejected); the two item-side gradient rows are summed by grad_gather_kernel
    int two_pass;
    float2* uc_out;          // [total] (user as int bits, Phi; Phi = -1: rejected) -- the fused list of sgd_base.hpp GatherParams::uc
    uint32_t* neg_out;       // [total]
    // pre-drawn unseen candidates (warp_presample_kernel): cand[t * S + k] = k-th unseen draw of positive t (-1: the attempt
    // cap was reached first), next_attempt[t] = the attempt the in-kernel sampler continues from
    const int32_t* cand;
    const int32_t* next_attempt;
};

constexpr uint32_t kWarpAttemptCap = 64 * 64;   // draws per positive before giving up on "seen" candidates (the reference never does)

template <int S>
__global__ __launch_bounds__(256) void warp_presample_kernel(SgdParams p, int64_t total, int32_t* __restrict__ cand, int32_t* __restrict__ next_attempt) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= total) return;
    const int u = p.rows[t];
    const int64_t ubeg = (u == 0 ? 0 : p.indptr[u - 1]) - p.shift;
    const int64_t uend = p.indptr[u] - p.shift;
    const uint64_t gpos = static_cast<uint64_t>(p.nnz_offset + p.shift + t);
    int k = 0;
    uint32_t a = 0;
    for (; a < kWarpAttemptCap && k < S; ++a) {
        uint32_t o0, o1;
        counter_draw(p.seed, 1u, gpos, 0u, p.epoch, a, o0, o1);
        const int c = static_cast<int>((static_cast<uint64_t>(o0) * static_cast<uint32_t>(p.Q_rows)) >> 32);
        if (!sorted_contains(p.keys, ubeg, uend, c)) cand[t * S + k++] = c;
    }
    for (; k < S; ++k) cand[t * S + k] = -1;
    next_attempt[t] = static_cast<int32_t>(a);
}

template <int K>
struct WRow {
    float v[K];
};

template <int K>
__device__ __forceinline__ void wload(WRow<K>& r, const float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        r.v[k] = (e < vdim) ? base[e] : 0.0f;
    }
}
template <int K>
__device__ __forceinline__ void watomic(const WRow<K>& r, float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        if (e < vdim) atomic_add_f32(base + e, r.v[k]);
    }
}
// warp.cc:21-28
template <int K>
__device__ __forceinline__ float wscore_part(const WRow<K>& u, const WRow<K>& i, bool l2) {
    float s = 0.f;
    if (!l2) {
#pragma unroll
        for (int k = 0; k < K; ++k) s += u.v[k] * i.v[k];
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float df = u.v[k] - i.v[k];
            s -= df * df;
        }
    }
    return s;
}

constexpr int WARP_SPEC = 4;  // max candidate rows scored per step

// S: candidates per positive taken from the pre-pass (0: none, every candidate is drawn in the kernel)
template <int K, int S>
__global__ __launch_bounds__(256) void warp_update_kernel(SgdParams p, WarpConsts c) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * wpb;
    const int vdim = p.vdim;
    const int64_t n_work = (c.total + c.chunk - 1) / c.chunk;
    const bool l2 = c.l2 != 0;

    int cur_u = -1;
    WRow<K> pu, gacc;
    double loss = 0.0;
    unsigned long long scored = 0, accepted = 0, loaded = 0;

    auto flush_user = [&]() {
        if (cur_u < 0) return;
        watomic<K>(gacc, p.gradP + static_cast<size_t>(cur_u) * vdim, lane, vdim);
        cur_u = -1;
    };

    for (int64_t w = wave0; w < n_work; w += nwaves) {
        const int64_t t_beg = w * c.chunk;
        const int64_t t_end = (t_beg + c.chunk < c.total) ? t_beg + c.chunk : c.total;
        for (int64_t t0 = t_beg; t0 < t_end; t0 += 64) {
            const int64_t t = t0 + lane;
            int my_u = 0, my_pos = 0;
            if (t < t_end) {
                my_u = p.rows[t];
                my_pos = p.keys[t];
            }
            const int n_here = static_cast<int>((t_end - t0) < 64 ? (t_end - t0) : 64);
            float my_phi = -1.f;                                    // two-pass: lane j keeps positive j's Phi (-1: rejected) / negative,
            uint32_t my_nego = static_cast<uint32_t>(p.Q_rows);    // stored coalesced after the walk
            int my_c[S > 0 ? S : 1];
            int my_next = 0;
            if constexpr (S > 0) {
#pragma unroll
                for (int s4 = 0; s4 < S; s4 += 4) {
                    int4 v = make_int4(-1, -1, -1, -1);
                    if (t < t_end) v = *reinterpret_cast<const int4*>(c.cand + t * S + s4);
                    my_c[s4] = v.x; my_c[s4 + 1] = v.y; my_c[s4 + 2] = v.z; my_c[s4 + 3] = v.w;
                }
                if (t < t_end) my_next = c.next_attempt[t];
            }
            for (int j = 0; j < n_here; ++j) {
                const int u = __builtin_amdgcn_readlane(my_u, j);
                const int pos = __builtin_amdgcn_readlane(my_pos, j);
                if (u != cur_u) {
                    flush_user();
                    cur_u = u;
                    wload<K>(pu, p.P + static_cast<size_t>(u) * vdim, lane, vdim);
#pragma unroll
                    for (int k = 0; k < K; ++k) gacc.v[k] = 0.f;
                }
                const int64_t ubeg = (u == 0 ? 0 : p.indptr[u - 1]) - p.shift;
                const int64_t uend = p.indptr[u] - p.shift;
                const uint64_t gpos = static_cast<uint64_t>(p.nnz_offset + p.shift + t0 + j);
                WRow<K> qi;
                wload<K>(qi, p.Q + static_cast<size_t>(pos) * vdim, lane, vdim);
                const float ui = wave_sum(wscore_part<K>(pu, qi, l2));

                // ---------------- trial loop (warp.cc:133-147, Q-10) ----------------
                int kcount = 0;        // counted (unseen) candidates scored so far
                bool found = false, done = false;
                int neg = 0;
                float uj = 0.f;
                WRow<K> qj;
                int bsz = 1;
                // score up to nb candidates (in draw order): load the rows together, then decide one after the other
                auto score = [&](const int (&cl)[WARP_SPEC], int nb) {
                    WRow<K> cr[WARP_SPEC];
                    float sc[WARP_SPEC];
#pragma unroll
                    for (int s = 0; s < WARP_SPEC; ++s)
                        if (s < nb) wload<K>(cr[s], p.Q + static_cast<size_t>(cl[s]) * vdim, lane, vdim);
#pragma unroll
                    for (int s = 0; s < WARP_SPEC; ++s)
                        if (s < nb) sc[s] = wave_sum(wscore_part<K>(pu, cr[s], l2));
#pragma unroll
                    for (int s = 0; s < WARP_SPEC; ++s) {
                        if (s < nb && !done) {
                            if (1 + 2 * kcount > c.max_trial) {  // `while (trial <= max_trial)` fails
                                done = true;
                            } else {
                                kcount += 1;
                                scored += 1;
                                if (static_cast<double>(ui - sc[s]) < c.threshold) {
                                    found = true;
                                    done = true;
                                    neg = cl[s];
                                    uj = sc[s];
                                    qj = cr[s];
                                }
                            }
                        }
                    }
                    loaded += nb;
                    if (bsz < WARP_SPEC) bsz <<= 1;
                };
                uint32_t a0 = 0;       // attempt the in-kernel sampler starts from
                if constexpr (S > 0) {
                    // pre-drawn candidates: batches of 1, 2, 4, 4 ... at compile-time register indices
                    a0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(my_next, j));
#define BFH_WARP_BATCH(B0, BS)                                                                   \
                    if (!done && (B0) < S) {                                                     \
                        int cl[WARP_SPEC];                                                       \
                        int nb = 0;                                                              \
                        _Pragma("unroll") for (int s = 0; s < WARP_SPEC; ++s) {                  \
                            cl[s] = 0;                                                           \
                            if (s < (BS) && (B0) + s < S) {                                      \
                                const int cc = __builtin_amdgcn_readlane(my_c[((B0) + s) < S ? ((B0) + s) : 0], j); \
                                if (cc >= 0 && nb == s) { cl[s] = cc; nb = s + 1; }              \
                            }                                                                    \
                        }                                                                        \
                        if (nb > 0) score(cl, nb);                                               \
                    }
                    BFH_WARP_BATCH(0, 1)
                    BFH_WARP_BATCH(1, 2)
                    BFH_WARP_BATCH(3, 4)
                    BFH_WARP_BATCH(7, 4)
#undef BFH_WARP_BATCH
                    if (!done && 1 + 2 * kcount > c.max_trial) done = true;
                }
                for (; a0 < kWarpAttemptCap && !done; a0 += 64) {  // the reference never gives up on "seen" draws
                    uint32_t o0, o1;
                    const uint32_t attempt = a0 + lane;
                    counter_draw(p.seed, 1u, gpos, 0u, p.epoch, attempt, o0, o1);
                    const int cand = static_cast<int>((static_cast<uint64_t>(o0) * static_cast<uint32_t>(p.Q_rows)) >> 32);
                    const bool seen = attempt >= kWarpAttemptCap || sorted_contains(p.keys, ubeg, uend, cand);
                    unsigned long long unseen = __ballot(!seen);
                    while (unseen != 0ull && !done) {
                        // up to bsz next unseen candidates, in draw order
                        int cl[WARP_SPEC];
                        int nb = 0;
                        unsigned long long m = unseen;
#pragma unroll
                        for (int s = 0; s < WARP_SPEC; ++s) {
                            cl[s] = 0;
                            if (s < bsz && m != 0ull) {
                                const int l = __builtin_ctzll(m);
                                m &= m - 1ull;
                                cl[s] = __builtin_amdgcn_readlane(cand, l);
                                nb = s + 1;
                            }
                        }
                        unseen = m;
                        score(cl, nb);
                    }
                    if (!done && 1 + 2 * kcount > c.max_trial) done = true;
                }
                const int trial = found ? 2 * kcount : 2 * kcount + 1;
                if (!found || trial >= c.max_trial) continue;  // warp.cc:148-149

                // Phi = log(max(1, int((Q_rows - |seen| - 1) / trial)))  (warp.cc:151)
                const int64_t nseen = uend - ubeg;
                const int64_t ratio = (static_cast<int64_t>(p.Q_rows) - nseen - 1) / trial;
                const float Phi = static_cast<float>(log(static_cast<double>(ratio > 1 ? ratio : 1)));
                WRow<K> gi, gj;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float ud, idv, jdv;
                    if (!l2) {  // warp.cc:30-40
                        ud = Phi * (qi.v[k] - qj.v[k]);
                        idv = Phi * pu.v[k];
                        jdv = -idv;
                    } else {    // warp.cc:42-52 (Q-11)
                        ud = Phi * 2 * (qi.v[k] - qj.v[k]);
                        idv = Phi * (pu.v[k] - qi.v[k]);
                        jdv = -Phi * (pu.v[k] - qj.v[k]);
                    }
                    gacc.v[k] += ud - c.reg_u * pu.v[k];
                    gi.v[k] = idv - c.reg_i * qi.v[k];
                    gj.v[k] = jdv - c.reg_j * qj.v[k];
                }
                if (c.two_pass) {
                    if (lane == j) {
                        my_phi = Phi;
                        my_nego = static_cast<uint32_t>(neg);
                    }
                    if (c.pcn && lane == 0) atomicAdd(p.cntP + u, 1);
                } else {
                    watomic<K>(gi, p.gradQ + static_cast<size_t>(pos) * vdim, lane, vdim);
                    watomic<K>(gj, p.gradQ + static_cast<size_t>(neg) * vdim, lane, vdim);
                    if (c.pcn && lane == 0) {
                        atomicAdd(p.cntP + u, 1);
                        atomicAdd(p.cntQ + pos, 1);
                        atomicAdd(p.cntQ + neg, 1);
                    }
                }
                loss += static_cast<double>(uj - ui) + c.threshold;
                accepted += 1;
            }
            if (c.two_pass && t < t_end) {
                c.uc_out[t] = make_float2(__builtin_bit_cast(float, my_u), my_phi);
                c.neg_out[t] = my_nego;
            }
        }
        flush_user();
    }
    flush_user();
    if (lane == 0) {
        if (loss != 0.0) atomicAdd(c.loss_out, loss);
        if (scored) atomicAdd(c.cnt, scored);
        if (accepted) atomicAdd(c.cnt + 1, accepted);
        if (loaded) atomicAdd(c.cnt + 2, loaded);
    }
}

// CWARP::compute_loss warp.cc:205-226: fraction of sampled triples that violate the margin.
__global__ void warp_loss_kernel(const float* __restrict__ P, const float* __restrict__ Q, const int32_t* __restrict__ users,
                                 const int32_t* __restrict__ pos, const int32_t* __restrict__ neg, int n, int vdim, int l2,
                                 double threshold, unsigned long long* out) {
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    const float* pu = P + static_cast<size_t>(users[w]) * vdim;
    const float* qi = Q + static_cast<size_t>(pos[w]) * vdim;
    const float* qj = Q + static_cast<size_t>(neg[w]) * vdim;
    float a = 0.f, b = 0.f;
    for (int e = lane; e < vdim; e += 64) {
        if (!l2) {
            a += pu[e] * qi[e];
            b += pu[e] * qj[e];
        } else {
            const float d1 = pu[e] - qi[e], d2 = pu[e] - qj[e];
            a -= d1 * d1;
            b -= d2 * d2;
        }
    }
    const float xi = wave_sum(a), xj = wave_sum(b);
    if (lane == 0 && (static_cast<double>(xi) - static_cast<double>(xj)) < threshold) atomicAdd(out, 1ull);
}

// ------------------------------------------------------------------------------------------------
class WarpHandle : public SgdHandle {
 public:
    WarpHandle() : SgdHandle(1) {}
    bool project_unit_ball() const override { return true; }  // warp.cc:194-200
    void parse_specific() override {
        BFH_REQUIRE(optimizer_ != "sgd", "WARP accumulates gradients: optimizer must be adagrad or adam");
        max_trial_ = opt_.integer("max_trials");
        threshold_ = opt_.num("threshold");
        // warp.cc:76-83: only the exact string "l2" selects the L2 score (Q-23)
        l2_ = opt_.str("score_func") == "l2";
        cnt_.resize(3, true, stream);
    }

    void partial_update(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, double* loss_sum, double* n_samples) {
        SgdParams p;
        const int64_t n = stage_chunk(start_x, next_x, indptr, keys, &p);
        *loss_sum = 0.0;
        *n_samples = static_cast<double>(n);
        if (comm_) exchange_arm();   // Z = the gradient buffers BEFORE this model accumulates anything (they are exchanged as deltas)
        if (n == 0) return;          // no collective in WARP's partial_update: an empty shard chunk may return
        WarpConsts c{};
        c.reg_u = reg_u_; c.reg_i = reg_i_; c.reg_j = reg_j_;
        c.threshold = threshold_;
        c.max_trial = max_trial_; c.l2 = l2_; c.pcn = pcn_;
        c.chunk = chunk_;
        c.total = n;
        c.loss_out = scratch_.get();
        c.cnt = cnt_.get();
        BFH_HIP(hipMemsetAsync(scratch_.get(), 0, sizeof(double), stream));
        BFH_HIP(hipMemsetAsync(cnt_.get(), 0, 3 * sizeof(unsigned long long), stream));
        const bool two_pass = accum_two_pass_ != 0;
        // candidates from the pre-pass: 4 while most positives accept one of their first draws, 8 once they do not
        const int S = presample_ < 0 ? (last_T_ > 2.0 ? 8 : 4) : presample_;
        if (S > 0) {
            if (cand_.size() < static_cast<size_t>(n) * S) cand_.resize(static_cast<size_t>(n) * S);
            if (next_.size() < static_cast<size_t>(n)) next_.resize(static_cast<size_t>(n));
            const int slot0 = t_aux_.begin(stream);
            const dim3 g0(static_cast<unsigned>((n + 255) / 256)), b0(256);
            if (S == 4) hipLaunchKernelGGL(warp_presample_kernel<4>, g0, b0, 0, stream, p, n, cand_.get(), next_.get());
            else hipLaunchKernelGGL(warp_presample_kernel<8>, g0, b0, 0, stream, p, n, cand_.get(), next_.get());
            BFH_HIP(hipGetLastError());
            t_aux_.end(slot0, stream);
            c.cand = cand_.get();
            c.next_attempt = next_.get();
        }
        if (two_pass) {
            acc_prepare(n);
            c.two_pass = 1;
            c.uc_out = acc_uc_.get();
            c.neg_out = acc_neg_.get();
        }
        const int64_t n_work = (c.total + c.chunk - 1) / c.chunk;
        dim3 block(256), grid(1);
        if (sequential_) {
            block = dim3(64);
        } else {
            const int wpc = waves_per_cu_ > 0 ? waves_per_cu_ : (vdim_ <= 256 ? 24 : 16);   // 24: what 75-79 VGPRs / 106 SGPRs admit
            int64_t waves = static_cast<int64_t>(num_cus_) * wpc;
            if (waves > n_work) waves = n_work;
            grid = dim3(static_cast<unsigned>((waves + 3) / 4));
        }
        const int slot = t_main_.begin(stream);
        const int K = (vdim_ + 63) / 64;
#define BFH_WARP_LAUNCH(SS)                                                                                         \
        do {                                                                                                        \
            if (K <= 1) hipLaunchKernelGGL((warp_update_kernel<1, SS>), grid, block, 0, stream, p, c);               \
            else if (K <= 2) hipLaunchKernelGGL((warp_update_kernel<2, SS>), grid, block, 0, stream, p, c);          \
            else if (K <= 4) hipLaunchKernelGGL((warp_update_kernel<4, SS>), grid, block, 0, stream, p, c);          \
            else if (K <= 8) hipLaunchKernelGGL((warp_update_kernel<8, SS>), grid, block, 0, stream, p, c);          \
            else hipLaunchKernelGGL((warp_update_kernel<16, SS>), grid, block, 0, stream, p, c);                     \
        } while (0)
        if (S == 0) BFH_WARP_LAUNCH(0);
        else if (S == 4) BFH_WARP_LAUNCH(4);
        else BFH_WARP_LAUNCH(8);
#undef BFH_WARP_LAUNCH
        BFH_HIP(hipGetLastError());
        t_main_.end(slot, stream);
        if (two_pass) {
            // gi = Phi * p_u - reg_i q_i, gj = -Phi * p_u - reg_j q_j (warp.cc:30-40); L2 score (Q-11, warp.cc:42-52):
            // gi = Phi * (p_u - q_i) - reg_i q_i, gj = -Phi * (p_u - q_j) - reg_j q_j
            const int slot2 = t_aux_.begin(stream);
            acc_build_positive_list(p, start_x, next_x);
            const float sab_pos[3] = {1.f, l2_ ? -1.f : 0.f, -reg_i_}, sab_neg[3] = {-1.f, l2_ ? 1.f : 0.f, -reg_j_};
            acc_gather(p, 1, true, true, sab_pos, sab_neg, false);
            t_aux_.end(slot2, stream);
        }
        unsigned long long cnt[3] = {0, 0, 0};
        BFH_HIP(hipMemcpyAsync(loss_sum, scratch_.get(), sizeof(double), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipMemcpyAsync(cnt, cnt_.get(), sizeof(cnt), hipMemcpyDeviceToHost, stream));
        sync_stream();
        harvest_timers();
        stats.launches += 1;
        stats.samples += n;
        stats.scored_negatives += static_cast<int64_t>(cnt[0]);
        stats.accepted += static_cast<int64_t>(cnt[1]);
        stats.loaded_rows += static_cast<int64_t>(cnt[2]);
        last_T_ = static_cast<double>(cnt[0]) / static_cast<double>(n);
        advance_progress(start_x, next_x, indptr);
    }

    double compute_loss(int n, const int32_t* users, const int32_t* pos, const int32_t* neg) {
        BFH_REQUIRE(model_on_gpu_, "compute_loss before initialize_model(..., set_gpu=True)");
        if (n <= 0) return 0.0;
        inj_.resize(static_cast<size_t>(3) * n);
        BFH_HIP(hipMemcpyAsync(inj_.get(), users, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(inj_.get() + n, pos, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(inj_.get() + 2 * n, neg, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemsetAsync(cnt_.get(), 0, sizeof(unsigned long long), stream));
        const int slot = t_aux_.begin(stream);
        hipLaunchKernelGGL(warp_loss_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, P_.get(), Q_.get(), inj_.get(), inj_.get() + n,
                           inj_.get() + 2 * n, n, vdim_, static_cast<int>(l2_), threshold_, cnt_.get());
        BFH_HIP(hipGetLastError());
        t_aux_.end(slot, stream);
        unsigned long long v = 0;
        BFH_HIP(hipMemcpyAsync(&v, cnt_.get(), sizeof(v), hipMemcpyDeviceToHost, stream));
        sync_stream();
        harvest_timers();
        return static_cast<double>(v) / static_cast<double>(n);
    }

    int max_trial_ = 0;
    int presample_ = -1;      // "warp_presample": candidates per positive from the pre-pass (0, 4, 8; -1: from the last call's T)
    double last_T_ = 0.0;     // scored negatives per positive of the previous call
    DevBuf<int32_t> cand_, next_;
    double threshold_ = 0;
    bool l2_ = false;
    DevBuf<unsigned long long> cnt_;
    DevBuf<int32_t> inj_;
};

}  // namespace bfh

using bfh::guarded;
using bfh::WarpHandle;

extern "C" {

void* bfh_warp_create(void) {
    try {
        WarpHandle* h = new WarpHandle();
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
void bfh_warp_destroy(void* h) { delete static_cast<WarpHandle*>(h); }
int bfh_warp_set_device(void* h, int device) {
    return guarded(h, [&] {
        BFH_REQUIRE(!static_cast<WarpHandle*>(h)->stream || static_cast<WarpHandle*>(h)->device == device,
                    "set_device after init: the handle's stream and buffers live on the device it was initialised on");
        static_cast<WarpHandle*>(h)->device = device;
        BFH_HIP(hipSetDevice(device));
        return BFH_OK;
    });
}
int bfh_warp_init(void* h, const char* opt_json_path) {
    int ok = 0;
    int rc = guarded(h, [&] { ok = static_cast<WarpHandle*>(h)->init(opt_json_path) ? 1 : 0; return BFH_OK; });
    return rc == BFH_OK ? ok : rc;
}
int bfh_warp_get_vdim(void* h) { return h ? static_cast<WarpHandle*>(h)->get_vdim() : BFH_ERR_INVALID; }
int bfh_warp_initialize_model(void* h, float* P, int P_rows, float* Q, float* Qb, int Q_rows, int64_t num_nnz, int set_gpu) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->initialize_model(P, P_rows, Q, Qb, Q_rows, num_nnz, set_gpu != 0); return BFH_OK; });
}
int bfh_warp_set_placeholder(void* h, const int64_t* indptr, size_t batch_size) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->set_placeholder(indptr, batch_size); return BFH_OK; });
}
int bfh_warp_set_cumulative_table(void* h, const int64_t* table) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->set_cumulative_table(table); return BFH_OK; });
}
int bfh_warp_partial_update(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys, double* loss_sum, double* n_samples) {
    return guarded(h, [&] {
        double l = 0, n = 0;
        static_cast<WarpHandle*>(h)->partial_update(start_x, next_x, indptr, keys, &l, &n);
        if (loss_sum) *loss_sum = l;
        if (n_samples) *n_samples = n;
        return BFH_OK;
    });
}
int bfh_warp_update_parameters(void* h) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->update_parameters(); return BFH_OK; });
}
int bfh_warp_synchronize(void* h, int device_to_host) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->synchronize(device_to_host != 0, device_to_host == 2); return BFH_OK; });
}
int bfh_warp_compute_loss(void* h, int n, const int32_t* users, const int32_t* positives, const int32_t* negatives, double* loss) {
    return guarded(h, [&] { *loss = static_cast<WarpHandle*>(h)->compute_loss(n, users, positives, negatives); return BFH_OK; });
}
int bfh_warp_set_resident_csr(void* h, const int64_t* indptr, const int32_t* keys, int64_t nnz) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->set_resident_csr(indptr, keys, nnz); return BFH_OK; });
}
int bfh_warp_set_mode(void* h, const char* name, int64_t value) {
    return guarded(h, [&] {
        WarpHandle* w = static_cast<WarpHandle*>(h);
        if (name && std::string(name) == "warp_presample") {
            BFH_REQUIRE(value == -1 || value == 0 || value == 4 || value == 8, "warp_presample must be -1 (auto), 0, 4 or 8");
            w->presample_ = static_cast<int>(value);
        } else {
            w->set_mode(name ? name : "", value);
        }
        return BFH_OK;
    });
}
int bfh_warp_set_shard(void* h, int64_t nnz_offset, int num_shards) {
    return guarded(h, [&] {
        BFH_REQUIRE(num_shards >= 1 && nnz_offset >= 0, "set_shard: bad arguments");
        static_cast<WarpHandle*>(h)->nnz_offset_ = nnz_offset;
        static_cast<WarpHandle*>(h)->num_shards_ = num_shards;
        return BFH_OK;
    });
}
int bfh_warp_device_buffer(void* h, const char* name, void** dptr, size_t* bytes) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->device_buffer(name ? name : "", dptr, bytes); return BFH_OK; });
}
void* bfh_warp_stream(void* h) { return h ? static_cast<void*>(static_cast<WarpHandle*>(h)->stream) : nullptr; }
int bfh_warp_set_comm(void* h, void* comm) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->set_comm(static_cast<bfh::Comm*>(comm)); return BFH_OK; });
}
int bfh_warp_comm_flush(void* h) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->exchange_finish(); static_cast<WarpHandle*>(h)->sync_stream(); return BFH_OK; });
}
int bfh_warp_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { *out = static_cast<WarpHandle*>(h)->stats; return BFH_OK; });
}
int bfh_warp_reset_stats(void* h) {
    return guarded(h, [&] { static_cast<WarpHandle*>(h)->stats = bfh_stats{}; return BFH_OK; });
}

}  // extern "C"
