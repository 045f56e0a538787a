// BPRMF on gfx950: update kernels, loss kernel, C ABI.
//
// Two walks over the same (u, i, j) triples (the sampler is a pure function of the triple's position, so both see
// the same triples):
//   * bpr_item_major_kernel (bpr_item_major.hpp): the default for optimizer = sgd ("hogwild_atomic" = 3) -- the
//     positive item's row in registers, users owned by XCDs, negatives in per-XCD replicas;
//   * bpr_update_kernel (below): user-major -- P[u] in registers.  The deterministic parity mode ("sequential"),
//     adam / adagrad gradient accumulation, injected triples and the policies 0 / 1 / 2.
// The rest of this comment describes the user-major kernel.
//
// Reference semantics: CBPRMF::worker (/root/reference/lib/algo_impl/bpr/bpr.cc:72-188) -- the CPU
// path, as BASELINE.json's north_star asks -- behind CuBPR's object surface
// (/root/reference/include/buffalo/cuda/bpr/bpr.hpp:29-45).  Not derived from lib/cuda/bpr/bpr.cu:
// that backend materialises (user,pos,neg) arrays in HBM, spends one 128-thread block and two
// block-wide barriers per sample and uses expf instead of the CPU's sigmoid table.
//
// Kernel shape (wave64):
//   * the chunk's nnz positions are cut into work items of `chunk` consecutive positions; a wave
//     owns a work item (perfect load balance on heavy-tailed degrees, coalesced key/row-id loads);
//   * per 64 positions the lanes sample negatives in parallel (Philox counter draws, verify_neg by
//     binary search in the user's sorted key run);
//   * the wave then walks the 64 triples with the next triple's item rows prefetched; a dot product
//     is a few FMAs per lane + a DPP row reduction -- no LDS, no barrier;
//   * P[u] lives in registers across the user's run (one load + one store per run instead of per
//     triple);
//   * item rows are shared by every wave on the chip.  The 8 XCDs have private, mutually
//     non-coherent L2s, so "just store the row" silently forks Q into 8 copies (measured: 14
//     concurrent waves on a 400-item table lose 9/10 of the learning signal).  Two coherent forms:
//       - write-through Hogwild (policy 0): rows are float4 per lane, read with `buffer_load_dwordx4
//         sc1` and written with `buffer_store_dwordx4 sc1` (device scope: bypass L1, write through
//         L2), i.e. lock-free racy read-modify-write like the CPU reference, visible chip-wide;
//       - fp32 hardware atomics (policy 1): K = vdim/64 dwords per lane (element k*64+lane) so one
//         `global_atomic_add_f32` covers two full cache lines; no update is ever lost.
//     Measured on MI355X (scripts/micro/atomics.hip): uniform 512-B row atomics run at
//     2.6 G rows/s and a single hot row at 24 ns per update, write-through rows at ~2.3 G rows/s;
//     on a small catalogue write-through loses most colliding updates (NDCG 0.04 vs 0.27 on the
//     400-item planted test), so atomics are this walk's default.  Policy 2 runs it on per-XCD replicas of Q
//     (plain stores through the XCD's own L2, merged by the delta rule) with the popular rows on atomics.
#include "sgd_base.hpp"

#include <algorithm>

#include "comm.hpp"

namespace bfh {

struct BprConsts {
    float lr, reg_u, reg_i, reg_j, reg_b;
    double lr_d, reg_b_d;   // the bias statements of the reference are scalar C++ in double (bpr.cc:81, 99, 162, 168)
    int use_bias, update_i, update_j, verify_neg, uniform, num_neg, pcn, compute_loss, atomic, sequential;
    int64_t cum_total;
    const float* exp_table;
    double* loss_out;
    int chunk;
    int neg_limit;        // study knob: uniform negatives are folded into [0, neg_limit) (0: off) -- what the walk does when the negatives' rows fit an L2
    // injected triples (bfh_bpr_update_triples)
    const int32_t* inj_u;
    const int32_t* inj_p;
    const int32_t* inj_n;
    int64_t total;  // number of (position, slot) items
    int64_t work_begin, work_end;  // work items [begin, end) of this launch (a segment of the call)
    // policy 2: one private copy of the item factors per XCD (see xcd_* kernels below)
    float* rep_Q;
    float* rep_Qb;
    int64_t rep_stride, rep_bstride;
    const uint8_t* hot;   // [Q_rows] 1 = row stays in the chip-wide matrix and is updated with atomics
    int fresh;            // re-read replica rows right before storing them
    // adam / adagrad, two-pass accumulation (sgd_base.hpp GatherParams): this kernel only records the logit and the
    // negative of every triple; the item-side gradient rows are summed by grad_gather_kernel
    int two_pass;
    float2* uc_out;       // [total] (user as int bits, logit): the fused list of sgd_base.hpp GatherParams::uc
    uint32_t* neg_out;    // [total]
};

// new bias = (float)(b + alpha * (+-logit - reg_b * b)) with the product and sums in double, as the reference's scalar statement rounds
__device__ __forceinline__ float bias_step(float b, float signed_logit, double lr, double reg_b) {
    return static_cast<float>(static_cast<double>(b) + lr * (static_cast<double>(signed_logit) - reg_b * static_cast<double>(b)));
}


// The XCD this wave runs on (0..7), from the hardware register: the address of a wave's item-factor
// replica depends on it, so it must be the truth, not a guess from blockIdx.
__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return static_cast<int>(x & 7u);
}

// CBPRMF::build_exp_table bpr.cc:57-63 + lookup bpr.cc:124-131 (Q-2: integer 1000/6/2 == 83)
__device__ __forceinline__ float bpr_logit(float x, const float* __restrict__ table) {
    if (6.0f < x) return 0.0f;
    if (x < -6.0f) return 1.0f;
    const int idx = __builtin_amdgcn_readfirstlane(static_cast<int>((x + 6.0f) * 83.0f));
    return table[idx];
}

__device__ __forceinline__ int bpr_sample_negative(const SgdParams& p, const BprConsts& c, uint64_t gpos, uint32_t slot,
                                                   int64_t ubeg, int64_t uend) {
    int neg = 0;
    for (uint32_t attempt = 0; attempt < (1u << 20); ++attempt) {  // the reference loops forever (bpr.cc:106-117)
        uint32_t o0, o1;
        counter_draw(p.seed, 0u, gpos, slot, p.epoch, attempt, o0, o1);
        if (c.uniform) {
            neg = static_cast<int>((static_cast<uint64_t>(o0) * static_cast<uint32_t>(p.Q_rows)) >> 32);
            if (c.neg_limit > 0) neg %= c.neg_limit;
        } else {
            const uint64_t r64 = (static_cast<uint64_t>(o1) << 32) | o0;
            const int64_t r = static_cast<int64_t>(__umul64hi(r64, static_cast<uint64_t>(c.cum_total)));
            neg = static_cast<int>(lower_bound_dev<int64_t>(p.cum_table, p.Q_rows, r));  // Q-4: lower_bound
        }
        if (!c.verify_neg || !sorted_contains(p.keys, ubeg, uend, neg)) break;
    }
    return neg;
}

template <int K>
struct Row {
    float v[K];
};

template <int K>
__device__ __forceinline__ void load_row(Row<K>& r, const float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        r.v[k] = (e < vdim) ? base[e] : 0.0f;
    }
}
// same map, every dword loaded past the CU's L1 (global_load_dword sc1): what another CU of this
// XCD stored is in the L2, not in this CU's L1
template <int K>
__device__ __forceinline__ void load_row_coh(Row<K>& r, const float* base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        r.v[k] = (e < vdim) ? __hip_atomic_load(base + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
    }
}
template <int K>
__device__ __forceinline__ void store_row(const Row<K>& r, float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        if (e < vdim) base[e] = r.v[k];
    }
}
template <int K>
__device__ __forceinline__ void atomic_add_row(const Row<K>& r, float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = k * 64 + lane;
        if (e < vdim) atomic_add_f32(base + e, r.v[k]);
    }
}

// the b128 buffer builtins traffic in their own 128-bit type: always go through bit_cast (an
// implicit conversion to an ext_vector splats the low dword!)
using b128_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t(), 0, 0, 0));
struct f32q { float v[4]; };

// Row I/O.  V4 == false: element k*64+lane (dword per lane).  V4 == true: K = 4*KV, lane holds the
// 4 consecutive floats (kv*64+lane)*4 .. +3, moved with 16-byte buffer instructions whose bounds
// check (num_records = row bytes) masks the lanes beyond vdim; COH selects sc1 (device-coherent).
template <int K, bool V4, bool COH>
__device__ __forceinline__ void row_load(Row<K>& r, const float* __restrict__ base, int lane, int vdim) {
    if constexpr (V4) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, vdim * 4, 0x00020000);
#pragma unroll
        for (int kv = 0; kv < K / 4; ++kv) {
            const b128_t raw = COH ? __builtin_amdgcn_raw_buffer_load_b128(rs, (kv * 64 + lane) * 16, 0, 16)
                                   : __builtin_amdgcn_raw_buffer_load_b128(rs, (kv * 64 + lane) * 16, 0, 0);
            const f32q v = __builtin_bit_cast(f32q, raw);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) r.v[kv * 4 + c4] = v.v[c4];
        }
    } else {
        load_row<K>(r, base, lane, vdim);
    }
}
template <int K, bool V4, bool COH>
__device__ __forceinline__ void row_store(const Row<K>& r, float* __restrict__ base, int lane, int vdim) {
    if constexpr (V4) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, vdim * 4, 0x00020000);
#pragma unroll
        for (int kv = 0; kv < K / 4; ++kv) {
            f32q v;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) v.v[c4] = r.v[kv * 4 + c4];
            const b128_t raw = __builtin_bit_cast(b128_t, v);
            if (COH) __builtin_amdgcn_raw_buffer_store_b128(raw, rs, (kv * 64 + lane) * 16, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(raw, rs, (kv * 64 + lane) * 16, 0, 0);
        }
    } else {
        store_row<K>(r, base, lane, vdim);
    }
}
template <int K, bool V4>
__device__ __forceinline__ void row_atomic_add(const Row<K>& r, float* __restrict__ base, int lane, int vdim) {
    if constexpr (V4) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = ((k >> 2) * 64 + lane) * 4 + (k & 3);
            if (e < vdim) atomic_add_f32(base + e, r.v[k]);
        }
    } else {
        atomic_add_row<K>(r, base, lane, vdim);
    }
}
// float4 rows with the non-temporal hint on top (aux bit 1): load past the L1 (sc1) as row_load<.., COH>, store plain
template <int K>
__device__ __forceinline__ void row_load_nt(Row<K>& r, const float* __restrict__ base, int lane, int vdim) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, vdim * 4, 0x00020000);
#pragma unroll
    for (int kv = 0; kv < K / 4; ++kv) {
        const f32q v = __builtin_bit_cast(f32q, __builtin_amdgcn_raw_buffer_load_b128(rs, (kv * 64 + lane) * 16, 0, 18));
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) r.v[kv * 4 + c4] = v.v[c4];
    }
}
template <int K>
__device__ __forceinline__ void row_store_nt(const Row<K>& r, float* __restrict__ base, int lane, int vdim) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, vdim * 4, 0x00020000);
#pragma unroll
    for (int kv = 0; kv < K / 4; ++kv) {
        f32q v;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) v.v[c4] = r.v[kv * 4 + c4];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(b128_t, v), rs, (kv * 64 + lane) * 16, 0, 2);
    }
}
// device-coherent scalar (bias) access for the write-through policy
__device__ __forceinline__ float coh_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coh_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// SGD: true -> Hogwild SGD branch (bpr.cc:157-172); false -> gradient accumulation branch for
// adam/adagrad (bpr.cc:138-156,175-181).  PIPE: prefetch the next triple's item rows.
// INJECT: triples come from arrays instead of CSR + sampler.  V4: float4 layout + sc1 item-row I/O.
template <int K, bool SGD, bool PIPE, bool INJECT, bool V4>
__global__ __launch_bounds__(256) void bpr_update_kernel(SgdParams p, BprConsts c) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * wpb;
    const int vdim = p.vdim;
    // item factors this wave works on: the chip-wide matrix, or (policy 2) the replica owned by
    // the wave's XCD -- only waves of that XCD ever touch it, so its L2 is the point of coherence
    // and plain stores are visible to every other wave that can read the row
    float* Qbase = p.Q;
    float* Qbbase = p.Qb;
    bool rep = false;
    if constexpr (SGD) {
        if (c.atomic == 2) {
            const int x = xcc_id();
            Qbase = c.rep_Q + static_cast<size_t>(x) * c.rep_stride;
            Qbbase = c.rep_Qb + static_cast<size_t>(x) * c.rep_bstride;
            rep = true;
        }
    }

    // policy 2 keeps the popular ("hot") rows in the chip-wide matrix: (pol bit set) <=> atomics on p.Q
    auto q_of = [&](int item, bool hot) -> float* { return (hot ? p.Q : Qbase) + static_cast<size_t>(item) * vdim; };
    auto qb_of = [&](int item, bool hot) -> float* { return (hot ? p.Qb : Qbbase) + item; };
    // item rows and biases are read past the L1 whenever another CU may have plain-stored them
    auto qload = [&](Row<K>& r, const float* base) {
        if constexpr (V4) row_load<K, true, true>(r, base, lane, vdim);
        else if (rep) load_row_coh<K>(r, base, lane, vdim);
        else load_row<K>(r, base, lane, vdim);
    };
    auto bload = [&](const float* ptr) -> float { return (V4 || rep) ? coh_load(ptr) : *ptr; };

    int cur_u = -1;
    bool cur_excl = true;
    Row<K> pu, p0;     // current / as-loaded user row (SGD) or accumulated / unused (accumulate)
    Row<K> gacc;       // accumulate mode: gradient of P[u] gathered over the run
    double loss = 0.0;

    auto flush_user = [&]() {
        if (cur_u < 0) return;
        if (SGD) {
            float* Pu = p.P + static_cast<size_t>(cur_u) * vdim;
            if (cur_excl) {
                row_store<K, V4, false>(pu, Pu, lane, vdim);   // the run is owned by this wave
            } else {
                Row<K> dlt;
#pragma unroll
                for (int k = 0; k < K; ++k) dlt.v[k] = pu.v[k] - p0.v[k];
                row_atomic_add<K, V4>(dlt, Pu, lane, vdim);
            }
        } else {
            row_atomic_add<K, V4>(gacc, p.gradP + static_cast<size_t>(cur_u) * vdim, lane, vdim);
        }
        cur_u = -1;
    };

    for (int64_t w = c.work_begin + wave0; w < c.work_end; w += nwaves) {
        const int64_t t_beg = w * c.chunk;
        const int64_t t_end = (t_beg + c.chunk < c.total) ? t_beg + c.chunk : c.total;
        for (int64_t t0 = t_beg; t0 < t_end; t0 += 64) {
            // ---------------- lane-parallel: fetch (u,pos) and sample the negative ----------------
            const int64_t t = t0 + lane;
            const bool valid = t < t_end;
            int my_u = 0, my_pos = 0, my_neg = 0, my_excl = 1, my_pol = 3;  // bit0: pos row atomic, bit1: neg row atomic
            if (valid) {
                if (INJECT) {
                    my_u = c.inj_u[t];
                    my_pos = c.inj_p[t];
                    my_neg = c.inj_n[t];
                    my_excl = c.sequential;
                } else {
                    const int64_t pos_idx = t / c.num_neg;           // chunk-local nnz position
                    const uint32_t slot = static_cast<uint32_t>(t % c.num_neg);
                    my_u = p.rows[pos_idx];
                    my_pos = p.keys[pos_idx];
                    const int64_t ubeg = (my_u == 0 ? 0 : p.indptr[my_u - 1]) - p.shift;
                    const int64_t uend = p.indptr[my_u] - p.shift;
                    my_neg = bpr_sample_negative(p, c, static_cast<uint64_t>(p.nnz_offset + p.shift + pos_idx), slot, ubeg, uend);
                    // does this wave own the user's whole run?  (then P[u] needs no atomics)
                    my_excl = c.sequential || (ubeg * c.num_neg >= t_beg && uend * c.num_neg <= t_end);
                }
                if (c.sequential || c.atomic != 1) my_pol = 0;   // sequential: one wave, plain stores are exact
                if (rep && c.hot) my_pol = (c.hot[my_pos] ? 1 : 0) | (c.hot[my_neg] ? 2 : 0);
            }
            const int n_here = static_cast<int>((t_end - t0) < 64 ? (t_end - t0) : 64);
            float my_coef = 0.f;   // two-pass accumulation: lane j keeps triple j's logit, stored coalesced after the walk
            if (!SGD && !INJECT && c.two_pass && valid) c.neg_out[t] = static_cast<uint32_t>(my_neg);

            Row<K> qi, qj, qi_n, qj_n;
            float bi = 0.f, bj = 0.f, bi_n = 0.f, bj_n = 0.f;
            int pos = __builtin_amdgcn_readlane(my_pos, 0);
            int neg = __builtin_amdgcn_readlane(my_neg, 0);
            if (PIPE) {
                const int pol0 = __builtin_amdgcn_readlane(my_pol, 0);
                const bool h_i = rep && (pol0 & 1), h_j = rep && (pol0 & 2);
                qload(qi, q_of(pos, h_i));
                qload(qj, q_of(neg, h_j));
                if (c.use_bias) { bi = bload(qb_of(pos, h_i)); bj = bload(qb_of(neg, h_j)); }
            }
            for (int j = 0; j < n_here; ++j) {
                const int u = __builtin_amdgcn_readlane(my_u, j);
                const int excl = __builtin_amdgcn_readlane(my_excl, j);
                const int pol = __builtin_amdgcn_readlane(my_pol, j);
                const bool at_i = (pol & 1) != 0, at_j = (pol & 2) != 0;
                pos = __builtin_amdgcn_readlane(my_pos, j);
                neg = __builtin_amdgcn_readlane(my_neg, j);
                float* Qi = q_of(pos, rep && at_i);
                float* Qj = q_of(neg, rep && at_j);
                float* Bi = qb_of(pos, rep && at_i);
                float* Bj = qb_of(neg, rep && at_j);
                int pos_n = 0, neg_n = 0;
                bool hn_i = false, hn_j = false;
                if (PIPE) {
                    if (j + 1 < n_here) {
                        pos_n = __builtin_amdgcn_readlane(my_pos, j + 1);
                        neg_n = __builtin_amdgcn_readlane(my_neg, j + 1);
                        const int pol_n = __builtin_amdgcn_readlane(my_pol, j + 1);
                        hn_i = rep && (pol_n & 1);
                        hn_j = rep && (pol_n & 2);
                        qload(qi_n, q_of(pos_n, hn_i));
                        qload(qj_n, q_of(neg_n, hn_j));
                        if (c.use_bias) {
                            bi_n = bload(qb_of(pos_n, hn_i));
                            bj_n = bload(qb_of(neg_n, hn_j));
                        }
                    }
                } else {
                    qload(qi, Qi);
                    qload(qj, Qj);
                    if (c.use_bias) { bi = bload(Bi); bj = bload(Bj); }
                }
                if (u != cur_u) {
                    flush_user();
                    cur_u = u;
                    cur_excl = excl != 0;
                    row_load<K, V4, false>(pu, p.P + static_cast<size_t>(u) * vdim, lane, vdim);
                    if (SGD) {
                        p0 = pu;
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) gacc.v[k] = 0.f;
                    }
                }
                // ---------------- score + sigmoid table (bpr.cc:119-131) ----------------
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) part += pu.v[k] * (qi.v[k] - qj.v[k]);
                float x = wave_sum(part);
                if (c.use_bias) x += (bi - bj);
                const float logit = bpr_logit(x, c.exp_table);
                if (c.compute_loss) loss += static_cast<double>(log1pf(__expf(-fminf(fmaxf(x, -6.f), 6.f))));

                if (SGD) {
                    // bpr.cc:157-171 incl. Q-1: the user step sees the already-updated item rows
                    Row<K> di, dj;
                    // pos == neg can only happen with verify_neg=false or injected triples; the
                    // reference then updates the one row twice in sequence (bpr.cc:159-169)
                    const bool same = pos == neg;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float idv = logit * pu.v[k];
                        di.v[k] = c.update_i ? c.lr * (idv - c.reg_i * qi.v[k]) : 0.f;
                        qi.v[k] += di.v[k];
                        if (same) qj.v[k] = qi.v[k];
                        dj.v[k] = c.update_j ? c.lr * (-idv - c.reg_j * qj.v[k]) : 0.f;
                        qj.v[k] += dj.v[k];
                        if (same) qi.v[k] = qj.v[k];
                        pu.v[k] += c.lr * (logit * (qi.v[k] - qj.v[k]) - c.reg_u * pu.v[k]);
                    }
                    // replica rows: optionally re-read the row right before the store, so that the window in
                    // which another wave's update of the same row can be overwritten is one L2 round trip
                    // instead of the prefetch distance (the step itself was computed from the prefetched row)
                    const bool fr_i = rep && c.fresh && c.update_i && !at_i, fr_j = rep && c.fresh && c.update_j && !at_j && !same;
                    Row<K> fi, fj;
                    if (fr_i) qload(fi, Qi);
                    if (fr_j) qload(fj, Qj);
                    if (c.update_i) {
                        if (at_i) row_atomic_add<K, V4>(di, Qi, lane, vdim);
                        else if (fr_i) {
#pragma unroll
                            for (int k = 0; k < K; ++k) fi.v[k] += di.v[k];
                            row_store<K, V4, false>(fi, Qi, lane, vdim);
                        }
                        else if (rep) row_store<K, V4, false>(qi, Qi, lane, vdim);
                        else row_store<K, V4, true>(qi, Qi, lane, vdim);
                    }
                    if (c.update_j) {
                        if (at_j) row_atomic_add<K, V4>(dj, Qj, lane, vdim);
                        else if (fr_j) {
#pragma unroll
                            for (int k = 0; k < K; ++k) fj.v[k] += dj.v[k];
                            row_store<K, V4, false>(fj, Qj, lane, vdim);
                        }
                        else if (rep) row_store<K, V4, false>(qj, Qj, lane, vdim);
                        else row_store<K, V4, true>(qj, Qj, lane, vdim);
                    }
                    if (c.use_bias && lane == 0) {
                        // bpr.cc:162, 168 are scalar statements with `double alpha`, `double reg_b`: evaluated in double, stored as float
                        const float bi_new = c.update_i ? bias_step(bi, logit, c.lr_d, c.reg_b_d) : bi;
                        if (same) bj = bi_new;
                        const float bj_new = bias_step(bj, -logit, c.lr_d, c.reg_b_d);
                        if (c.update_i) {
                            if (at_i) atomic_add_f32(Bi, bi_new - bi);
                            else if (V4 && !rep) coh_store(Bi, bi_new);
                            else *Bi = bi_new;
                        }
                        if (c.update_j) {
                            if (at_j) atomic_add_f32(Bj, bj_new - bj);
                            else if (V4 && !rep) coh_store(Bj, bj_new);
                            else *Bj = bj_new;
                        }
                    }
                } else {
                    // bpr.cc:138-156: P, Q are frozen during the epoch; gradients are summed
                    Row<K> gi, gj;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float idv = logit * pu.v[k];
                        gacc.v[k] += logit * (qi.v[k] - qj.v[k]);
                        gi.v[k] = idv;
                        gj.v[k] = -idv;
                    }
                    if (!INJECT && c.two_pass) {
                        if (lane == j) my_coef = logit;
                        if (c.pcn && lane == 0 && ((t0 + j) % c.num_neg) == c.num_neg - 1) atomicAdd(p.cntP + u, 1);
                    } else {
                    if (c.update_i) row_atomic_add<K, V4>(gi, p.gradQ + static_cast<size_t>(pos) * vdim, lane, vdim);
                    if (c.update_j) row_atomic_add<K, V4>(gj, p.gradQ + static_cast<size_t>(neg) * vdim, lane, vdim);
                    if (lane == 0) {
                        if (c.use_bias) {
                            if (c.update_i) atomic_add_f32(p.gradQb + pos, logit);
                            if (c.update_j) atomic_add_f32(p.gradQb + neg, -logit);
                        }
                        if (c.pcn) {  // Q-9 counting rules (bpr.cc:139-143, 175-181)
                            atomicAdd(p.cntQ + neg, 1);
                            const bool last_slot = INJECT ? true : (((t0 + j) % c.num_neg) == c.num_neg - 1);
                            if (last_slot) {
                                atomicAdd(p.cntP + u, 1);
                                atomicAdd(p.cntQ + pos, 1);
                            }
                        }
                    }
                    }
                }
                if (PIPE) {
                    if (j + 1 < n_here) {
                        qi = qi_n; qj = qj_n; bi = bi_n; bj = bj_n;
                        if (SGD && (!at_i || !at_j) && (pos_n == pos || pos_n == neg || neg_n == pos || neg_n == neg)) {
                            // plain-store rows: the prefetch raced with this wave's own stores -> reload
                            qload(qi, q_of(pos_n, hn_i));
                            qload(qj, q_of(neg_n, hn_j));
                            if (c.use_bias) {
                                bi = bload(qb_of(pos_n, hn_i));
                                bj = bload(qb_of(neg_n, hn_j));
                            }
                        }
                    }
                }
            }
            if (!SGD && !INJECT && c.two_pass && valid) c.uc_out[t] = make_float2(__builtin_bit_cast(float, my_u), my_coef);   // logit >= 0: never "rejected"
        }
        if (!c.sequential) flush_user();
    }
    flush_user();
    if (c.compute_loss && lane == 0 && loss != 0.0) atomicAdd(c.loss_out, loss);
}

// CBPRMF::compute_loss bpr.cc:227-244: mean log(1+exp(-(x_ui - x_uj))) in double; a wave per sample.
__global__ void bpr_loss_kernel(const float* __restrict__ P, const float* __restrict__ Q, const float* __restrict__ Qb,
                                const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
                                const int32_t* __restrict__ neg, int n, int vdim, int use_bias, double* out) {
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    const float* pu = P + static_cast<size_t>(users[w]) * vdim;
    const float* qi = Q + static_cast<size_t>(pos[w]) * vdim;
    const float* qj = Q + static_cast<size_t>(neg[w]) * vdim;
    float a = 0.f, b = 0.f;
    for (int e = lane; e < vdim; e += 64) {
        a += pu[e] * qi[e];
        b += pu[e] * qj[e];
    }
    float xi = wave_sum(a), xj = wave_sum(b);  // CBPRMF::distance returns float precision
    if (use_bias) { xi += Qb[pos[w]]; xj += Qb[neg[w]]; }
    if (lane == 0) {
        const double x = static_cast<double>(xi) - static_cast<double>(xj);
        atomicAdd(out, log(1.0 + exp(-x)));
    }
}

// ------------------------------------------------------------------------------------------------
// Policy 2: per-XCD replicas of the item factors.
//
// The 8 XCDs' L2s are not coherent with each other, and the only chip-wide coherent update of a
// shared row -- an fp32 atomic per dword, executed one dword per clock per channel -- caps the
// update kernel at half the HBM roofline.  Inside ONE XCD the L2 is the point of coherence: plain
// stores are visible to every wave of that XCD (item rows are loaded with sc1, i.e. past the CU's
// L1).  So every XCD trains on its own copy of Q/Qb with the CPU reference's literal Hogwild
// read-modify-write (bpr.cc:157-172), and the copies are reconciled every `xcd_sync_updates`
// updates with the rule buffalo_amd/dist.py applies between GPUs: Q <- S + sum_x (Q_x - S), where
// S is the state at the previous reconciliation.  A launch boundary writes the L2s back, so the
// merge kernel sees every replica's final state.  An update is therefore never lost between XCDs
// (it arrives at the next merge); within an XCD two waves racing on one row behave like two CPU
// threads racing on it.
// ------------------------------------------------------------------------------------------------
constexpr int kXcdReplicas = 8;

template <typename T>
__global__ __launch_bounds__(256) void xcd_broadcast_kernel(const T* __restrict__ S, T* __restrict__ rep, int64_t n, int64_t stride, int copies) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const T v = S[i];
        for (int x = 0; x < copies; ++x) rep[x * stride + i] = v;
    }
}

// the same for the rows whose flag equals `only` (P: the users that have replicas)
template <typename T>
__global__ __launch_bounds__(256) void xcd_broadcast_rows_kernel(const T* __restrict__ S, T* __restrict__ rep, int64_t n, int64_t stride, int copies,
                                                                  const uint8_t* __restrict__ flag, int row_len, int only) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        if (flag[i / row_len] != only) continue;
        const T v = S[i];
        for (int x = 0; x < copies; ++x) rep[x * stride + i] = v;
    }
}

__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_fma(float sc, float4 a, float4 b) { return make_float4(sc * a.x + b.x, sc * a.y + b.y, sc * a.z + b.z, sc * a.w + b.w); }
__device__ __forceinline__ float f4_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f4_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f4_fma(float sc, float a, float b) { return sc * a + b; }

// S <- S + scale * sum_x (rep_x - base); the replicas are refreshed unless this was the last segment.
// Hot rows live in S itself (updated there with atomics) and are skipped; `row_len` = elements per row.
// `base` is what the replicas started the segment from: S itself (policy 2, null), or a ninth copy when S
// also receives atomic steps during the segment (policy 3: the register-resident rows are flushed into S).
// `hot` (per row, or null): with `only` == 0 rows whose flag is non-zero are skipped (the item rows that live chip-wide); with
// `only` != 0 exactly the rows whose flag equals it are merged (P: the users that have replicas, ImQueues::hot_user == 2).
template <typename T>
__global__ __launch_bounds__(256) void xcd_merge_kernel(T* __restrict__ S, T* __restrict__ rep, int64_t n, int64_t stride, float scale,
                                                         int write_replicas, const uint8_t* __restrict__ hot, int row_len, T* __restrict__ base,
                                                         int only = 0, const float* __restrict__ W = nullptr) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        if (hot && (only ? hot[i / row_len] != only : hot[i / row_len] != 0)) continue;
        const T s_now = S[i];
        const T s0 = base ? base[i] : s_now;
        T r[kXcdReplicas];
#pragma unroll
        for (int x = 0; x < kXcdReplicas; ++x) r[x] = rep[x * stride + i];
        T acc = f4_sub(r[0], s0);
#pragma unroll
        for (int x = 1; x < kXcdReplicas; ++x) acc = f4_add(acc, f4_sub(r[x], s0));
        const T out = f4_fma(W ? scale * W[i / row_len] : scale, acc, s_now);
        S[i] = out;
        if (write_replicas) {
#pragma unroll
            for (int x = 0; x < kXcdReplicas; ++x) rep[x * stride + i] = out;
            if (base) base[i] = out;
        }
    }
}

// Per-row weight of the merge's sum (the rule of exchange_weight_kernel, sgd_base.hip, applied between the XCDs of one GPU): a
// replica row receives m steps between two merges; with curvature k each contracts the row towards its local equilibrium by
// exp(-lr k), so n replicas that started from the same state combine like ONE run of n m steps when their summed deltas are scaled
// by w = (1 - exp(-n x)) / (n (1 - exp(-x))), x = lr k m  (w -> 1: independent steps, SUM; w -> 1/n: n estimates of one move, MEAN).
// Items: only the NEGATIVE steps of a row land in its replicas (the positive item lives in registers and is flushed into S).
// m is an expectation; steps are whole: a row that saw at most one step over all replicas (n m <= 1) cannot have overshot, so the
// saturation is counted from the second step on, x = lr k (m - 1/n) -- w = 1 exactly for the cold tail (and for conflict-free tests).
__device__ __forceinline__ double xcd_sat_weight(double a, double m, int n) {
    const double x = a * (m - 1.0 / n);
    return (n > 1 && x > 1e-9) ? -expm1(-n * x) / (n * -expm1(-x)) : 1.0;
}
__global__ void xcd_item_weight_kernel(const int64_t* __restrict__ cum, int64_t cum_total, int rows, double neg_steps, double neg_uniform, double lr,
                                       double kq, double kb, int n, float* __restrict__ Wq, float* __restrict__ Wb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double pneg = neg_uniform;
    if (cum) pneg = static_cast<double>(cum[i] - (i ? cum[i - 1] : 0)) / static_cast<double>(cum_total);
    const double m = neg_steps * pneg / n;      // negative steps of row i per replica between two merges
    Wq[i] = static_cast<float>(xcd_sat_weight(lr * kq, m, n));
    Wb[i] = static_cast<float>(xcd_sat_weight(lr * kb, m, n));
}
// Users that have replicas: every step of the user lands in them, spread over n queues (im_keys_kernel's rule: all nq, or for
// spread mode 3 the smallest power of two r with deg < heavy_deg * r).
__global__ void xcd_user_weight_kernel(const int64_t* __restrict__ indptr, int first_row, int rows, double steps_per_entry, double lr, double kp, int nq,
                                       int spread_mode, int64_t heavy_deg, float* __restrict__ Wp) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= rows) return;
    const int g = first_row + u;
    const int64_t deg = indptr[g] - (g ? indptr[g - 1] : 0);
    int n = nq;
    if (spread_mode == 3 && heavy_deg > 0) {
        n = 2;
        while (n < nq && deg >= heavy_deg * n) n <<= 1;
        if (n > nq) n = nq;
    }
    Wp[g] = static_cast<float>(xcd_sat_weight(lr * kp, static_cast<double>(deg) * steps_per_entry / n, n));
}

// How often is every item row updated?  One int atomic per index (once per resident CSR).
__global__ void item_count_kernel(const int32_t* __restrict__ idx, int64_t n, int* __restrict__ cnt) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        atomicAdd(cnt + idx[i], 1);
}

// A row is hot when the expected number of OTHER waves of the same XCD holding it between their load
// and their store -- updates_i / updates_total * (item rows in flight per XCD) -- reaches tau: those
// rows would lose that fraction of their updates to racing plain stores (a CPU Hogwild thread pool
// sits at a few percent on the head items).  updates_i = positives_i * pos_mult + triples * P(neg = i).
__global__ void xcd_hot_kernel(const int* __restrict__ cnt, const int64_t* __restrict__ cum, int64_t cum_total, int rows, double pos_mult,
                               double triples, double neg_uniform, double inflight, double tau, uint8_t* __restrict__ hot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double pneg = neg_uniform;
    if (cum) pneg = static_cast<double>(cum[i] - (i ? cum[i - 1] : 0)) / static_cast<double>(cum_total);
    const double upd = cnt[i] * pos_mult + triples * pneg;
    hot[i] = (upd * inflight >= tau * 2.0 * triples) ? 1 : 0;
}

}  // namespace bfh

#include "bpr_item_major.hpp"

namespace bfh {

// ------------------------------------------------------------------------------------------------
class BprHandle : public SgdHandle {
 public:
    BprHandle() : SgdHandle(0) { hogwild_atomic_ = 3; }   // sgd default: the item-major walk (bpr_item_major.hpp)
    ~BprHandle() override {
        if (pre_stream_) {
            (void)hipStreamSynchronize(pre_stream_);
            (void)hipStreamDestroy(pre_stream_);
            (void)hipEventDestroy(pre_done_);
            (void)hipEventDestroy(pre_ready_);
        }
    }
    void parse_specific() override {
        num_neg_ = opt_.integer("num_negative_samples");
        BFH_REQUIRE(num_neg_ >= 1 && num_neg_ <= 255, "num_negative_samples must be in [1,255]");
        verify_neg_ = opt_.boolean_or("verify_neg", true);
        uniform_ = opt_.num_or("sampling_power", 0.0) == 0.0;
        // CBPRMF::build_exp_table bpr.cc:57-63, evaluated on the host exactly like the CPU path
        std::vector<float> t(1000);
        for (int i = 0; i < 1000; ++i) {
            float e = static_cast<float>(std::exp((i / static_cast<float>(1000) * 2 - 1) * 6));
            t[i] = static_cast<float>(1.0 / (e + 1));
        }
        exp_table_.resize(1000);
        BFH_HIP(hipMemcpyAsync(exp_table_.get(), t.data(), 1000 * sizeof(float), hipMemcpyHostToDevice, stream));
        sync_stream();
    }

    BprConsts consts(double lr) {
        BprConsts c{};
        c.lr = static_cast<float>(lr);
        c.lr_d = lr;
        c.reg_b_d = reg_b_d_;
        c.reg_u = reg_u_; c.reg_i = reg_i_; c.reg_j = reg_j_; c.reg_b = reg_b_;
        c.use_bias = use_bias_; c.update_i = update_i_; c.update_j = update_j_;
        c.verify_neg = verify_neg_; c.uniform = uniform_; c.num_neg = num_neg_; c.neg_limit = im_neg_limit_;
        c.pcn = pcn_; c.compute_loss = compute_loss_;
        c.atomic = hogwild_atomic_; c.sequential = sequential_;
        c.cum_total = cum_total_;
        c.exp_table = exp_table_.get();
        c.loss_out = scratch_.get();
        c.chunk = chunk_;
        if (xcd_replicas()) {
            if (!chunk_set_) c.chunk = 64;   // short work items: a segment ends when its slowest wave does
            c.atomic = 2;
            c.fresh = xcd_fresh_ > 0;
        } else if (item_major()) {
            c.atomic = 3;
            c.fresh = xcd_fresh_ != 0;       // default (-1): on
        } else if (c.atomic >= 2) {
            c.atomic = 1;                    // adam/adagrad accumulate exact sums: atomics
        }
        return c;
    }

    using KernelFn = void (*)(SgdParams, BprConsts);

    template <int K, bool INJECT, bool V4>
    KernelFn pick_k() const {
        const bool sgd = optimizer_ == "sgd";
        const bool pipe = prefetch_ != 0 && !sequential_;
        if (sgd && pipe) return bpr_update_kernel<K, true, true, INJECT, V4>;
        if (sgd) return bpr_update_kernel<K, true, false, INJECT, V4>;
        if (pipe) return bpr_update_kernel<K, false, true, INJECT, V4>;
        return bpr_update_kernel<K, false, false, INJECT, V4>;
    }
    // the instantiation for this handle's vdim / optimizer / policy
    template <bool INJECT>
    KernelFn pick(const BprConsts& c) const {
        // write-through Hogwild (policy 0; policy 2 on request) moves item rows as float4 with sc1; the
        // atomic, the replica and the deterministic sequential paths keep the dword-per-lane layout
        const bool v4 = optimizer_ == "sgd" && !sequential_ && (c.atomic == 0 || (c.atomic == 2 && xcd_v4_));
        if (v4) {
            const int KV = (vdim_ + 255) / 256;
            if (KV <= 1) return pick_k<4, INJECT, true>();
            if (KV <= 2) return pick_k<8, INJECT, true>();
            return pick_k<16, INJECT, true>();
        }
        const int K = (vdim_ + 63) / 64;
        if (K <= 1) return pick_k<1, INJECT, false>();
        if (K <= 2) return pick_k<2, INJECT, false>();
        if (K <= 4) return pick_k<4, INJECT, false>();
        if (K <= 8) return pick_k<8, INJECT, false>();
        return pick_k<16, INJECT, false>();
    }
    // waves of `fn` the chip keeps resident at once (256-thread blocks); "waves_per_cu" overrides
    int64_t resident_waves(KernelFn fn) {
        if (waves_per_cu_ > 0) return static_cast<int64_t>(num_cus_) * waves_per_cu_;
        auto it = occupancy_.find(reinterpret_cast<const void*>(fn));
        if (it == occupancy_.end()) {
            int blocks = 0;
            BFH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(fn), 256, 0));
            it = occupancy_.emplace(reinterpret_cast<const void*>(fn), std::max(1, std::min(blocks, 8))).first;
        }
        return static_cast<int64_t>(num_cus_) * it->second * 4;
    }

    bool xcd_replicas() const { return hogwild_atomic_ == 2 && optimizer_ == "sgd" && !sequential_; }
    bool item_major() const { return hogwild_atomic_ == 3 && optimizer_ == "sgd" && !sequential_; }
    // item-major default: no prefetch, rows are read where they are used (narrowest race window, least traffic,
    // 7 waves per SIMD cover the latency); "prefetch" = 1 selects the two-triples-ahead slots
    bool im_prefetch() const { return prefetch_ > 0; }

    // ---------------------------------------------------------------------------------------------
    // policy 3 (bpr_item_major.hpp)
    // ---------------------------------------------------------------------------------------------
    void im_probe() {   // which XCC ids do workgroups of this device report?
        if (im_nq_ > 0) return;
        DevBuf<int> seen;
        seen.resize(16, true, stream);
        hipLaunchKernelGGL(xcd_probe_kernel, dim3(4096), dim3(64), 0, stream, seen.get());
        BFH_HIP(hipGetLastError());
        int host[16];
        BFH_HIP(hipMemcpyAsync(host, seen.get(), sizeof(host), hipMemcpyDeviceToHost, stream));
        sync_stream();
        int n = 0;
        for (int i = 0; i < 16; ++i) im_xcd_queue_[i] = host[i] ? n++ : -1;
        BFH_REQUIRE(n >= 1 && n <= kImMaxQueues, "hogwild_atomic=3: unexpected number of XCDs reported by HW_REG_XCC_ID");
        im_nq_ = n;
    }

    template <int K>
    void im_launch_k(const SgdParams& p, const BprConsts& c, const ImQueues& q, int64_t waves, bool drain) {
        // "im_single_wave" (test hook): one wave drains every queue in ticket order -- a deterministic sequential run
        const dim3 grid(im_single_wave_ ? 1u : static_cast<unsigned>((waves + 3) / 4)), block(im_single_wave_ ? 64 : 256);
        if (drain) hipLaunchKernelGGL((bpr_item_major_kernel<K, false, true>), grid, block, 0, stream, p, c, q);
        else if (im_prefetch()) hipLaunchKernelGGL((bpr_item_major_kernel<K, true, false>), grid, block, 0, stream, p, c, q);
        else hipLaunchKernelGGL((bpr_item_major_kernel<K, false, false>), grid, block, 0, stream, p, c, q);
        BFH_HIP(hipGetLastError());
    }
    // two triples per wave (bpr_item_major_dual_kernel): vdim <= 128, rows read where they are used, not a test-hook run
    // Measured (profiles/r02_dual_triples_per_wave.txt, same box): 4.95 -> 4.56 ms per launch (4.20 without the hot-user atomics, whose
    // share grows because twice as many rows are held per queue); 16, 20 and 24 waves per CU give the same time -- the walk is at the
    // fabric's ceiling there, so the kernel is built for 5 waves per SIMD (81 VGPRs, no scratch).  "im_dual" = 0 keeps the one-triple walk.
    // Rounds 2-5: on small shards it lost (per-rank epoch at 4 shards 2.42 -> 2.54 ms, at 8 shards 1.31 -> 1.37: twice the rows held per queue
    // on few users turns more of them hot) and was used from 6144 users per queue up.  After round 6's diet of the kernel it wins there too
    // (profiles/r06_walk_variance.txt, call 32: 4 shards 2.29 -> 1.94 ms, 8 shards 1.42 -> 1.29), so the default is now 1024 users per queue.
    bool im_dual() const { return im_dual_call_; }
    // whole 32-element groups per row: the instantiation without per-lane guards ("im_dual_generic" = 1 keeps the guarded one: A/B)
    int im_dual_nk() const { return (vdim_ % 32 == 0 && vdim_ <= 128 && !im_dual_generic_) ? vdim_ / 32 : 0; }
    void im_choose_dual(int64_t users_here, int nq) {
        im_dual_call_ = im_dual_ != 0 && vdim_ <= 128 && !im_prefetch() && !im_single_wave_ && !im_drain_only_ &&
                        (im_dual_ > 0 || users_here >= static_cast<int64_t>(nq) * 1024);
    }
    void im_launch(const SgdParams& p, const BprConsts& c, const ImQueues& q, int64_t waves, bool drain) {
        if (!drain && im_dual()) {
            const dim3 grid(static_cast<unsigned>((waves + 3) / 4)), block(256);
            switch (im_dual_nk()) {
                case 1: hipLaunchKernelGGL(bpr_item_major_dual_kernel<1>, grid, block, 0, stream, p, c, q); break;
                case 2: hipLaunchKernelGGL(bpr_item_major_dual_kernel<2>, grid, block, 0, stream, p, c, q); break;
                case 3: hipLaunchKernelGGL(bpr_item_major_dual_kernel<3>, grid, block, 0, stream, p, c, q); break;
                case 4: hipLaunchKernelGGL(bpr_item_major_dual_kernel<4>, grid, block, 0, stream, p, c, q); break;
                default: hipLaunchKernelGGL(bpr_item_major_dual_kernel<0>, grid, block, 0, stream, p, c, q); break;
            }
            BFH_HIP(hipGetLastError());
            return;
        }
        const int KV = (vdim_ + 255) / 256;
        if (KV <= 1) im_launch_k<4>(p, c, q, waves, drain);
        else if (KV <= 2) im_launch_k<8>(p, c, q, waves, drain);
        else im_launch_k<16>(p, c, q, waves, drain);
    }
    int64_t im_resident_waves() {
        if (waves_per_cu_ > 0) return static_cast<int64_t>(num_cus_) * waves_per_cu_;
        const int KV = (vdim_ + 255) / 256;
        const void* fn = nullptr;
        if (im_dual()) {
            switch (im_dual_nk()) {
                case 1: fn = reinterpret_cast<const void*>(bpr_item_major_dual_kernel<1>); break;
                case 2: fn = reinterpret_cast<const void*>(bpr_item_major_dual_kernel<2>); break;
                case 3: fn = reinterpret_cast<const void*>(bpr_item_major_dual_kernel<3>); break;
                case 4: fn = reinterpret_cast<const void*>(bpr_item_major_dual_kernel<4>); break;
                default: fn = reinterpret_cast<const void*>(bpr_item_major_dual_kernel<0>); break;
            }
        }
        else if (im_prefetch())
            fn = KV <= 1 ? reinterpret_cast<const void*>(bpr_item_major_kernel<4, true, false>)
                         : (KV <= 2 ? reinterpret_cast<const void*>(bpr_item_major_kernel<8, true, false>)
                                    : reinterpret_cast<const void*>(bpr_item_major_kernel<16, true, false>));
        else
            fn = KV <= 1 ? reinterpret_cast<const void*>(bpr_item_major_kernel<4, false, false>)
                         : (KV <= 2 ? reinterpret_cast<const void*>(bpr_item_major_kernel<8, false, false>)
                                    : reinterpret_cast<const void*>(bpr_item_major_kernel<16, false, false>));
        auto it = occupancy_.find(fn);
        if (it == occupancy_.end()) {
            int blocks = 0;
            BFH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 256, 0));
            it = occupancy_.emplace(fn, std::max(1, std::min(blocks, 8))).first;
        }
        return static_cast<int64_t>(num_cus_) * it->second * 4;
    }

    // the item-major regrouping sorts 32-bit keys (queue, block, item) with 32-bit entry indices: catalogues / chunks beyond
    // that fall back to policy 1 instead of failing (about 33 M items at lr >= 0.1, or 2^31 interactions per call)
    bool im_fits(const BprConsts& c, int64_t n) {
        im_probe();
        const int64_t blocks = im_blocks_ > 0 ? im_blocks_ : std::min<int64_t>(16, std::max<int64_t>(1, static_cast<int64_t>(std::ceil(c.lr * 160.0))));
        return n < (int64_t(1) << 31) && static_cast<int64_t>(im_nq_) * blocks * Q_rows_ < (int64_t(1) << 32);
    }

    // one call of the item-major path over the staged chunk [start_x, next_x)
    void launch_item_major(const SgdParams& p, BprConsts c, int start_x, int next_x) {
        im_probe();
        const int64_t n = p.chunk_nnz;
        BFH_REQUIRE(n < (int64_t(1) << 31), "hogwild_atomic=3: chunk of 2^31 or more interactions");
        // runs an item's entries are cut into per queue: what a run of consecutive positive steps does to a row grows
        // with lr x run length (DESIGN.md "burst length": invisible at lr 0.002, a 7 % worse sampled loss at lr 0.05 with
        // one run, gone with 8), so the default follows the call's learning rate; "im_blocks" pins it
        const int64_t blocks = im_blocks_ > 0 ? im_blocks_ : std::min<int64_t>(16, std::max<int64_t>(1, static_cast<int64_t>(std::ceil(c.lr * 160.0))));
        BFH_REQUIRE(static_cast<int64_t>(im_nq_) * blocks * Q_rows_ < (int64_t(1) << 32), "hogwild_atomic=3: too many items for the 32-bit sort key");
        const int nq = (im_single_wave_ && im_force_queues_ > 0) ? std::min(im_force_queues_, kImMaxQueues) : im_nq_;
        im_choose_dual(next_x - start_x, nq);
        int slot = t_aux_.begin(stream);
        // ---- entries grouped by (owner queue of the user, item); cached for a resident matrix ----
        const bool keeps = resident_ || (auto_resident_ && !chunks_.empty());   // the staged chunk lives on in HBM under csr_generation_
        // Small shards (multi-GPU): the same waves work on an N-times smaller user set, and with one owner XCD per user most
        // triples fall under the collision rule and pay a user-row atomic (8 shards of ML-20M: 3/4 of them, 1.44 vs 1.15 ms).
        // There the users get what the negatives have: per-XCD replicas of P, entries spread over the queues by position (a
        // user's share of one queue is nq times smaller), plain stores through the XCD's own L2, the delta rule at the merges.
        const int64_t users_here = next_x - start_x;
        // Measured (profiles/r02_shard_times_user_replicas.txt, ML-20M / d=128, per-rank epoch): 8 shards 1.73 -> 1.31 ms (walk 1.52 -> 1.07);
        // 4 shards 2.43 -> 2.65, 2 shards 4.32 -> 5.45, whole matrix 9.2 -> 10.2: eight copies of a big P fall out of the Infinity Cache,
        // so the rule is "fewer than 3072 users per queue".  Statistics (profiles/r02_gate_study_user_replicas.txt, whole matrix, 8 copies):
        // at the reference's lr the gate metrics stay inside the oracle pair's spread; at lr 0.05 the sum of eight deltas of a heavy
        // user overshoots (|P| 390 vs 430, one run in three diverging), so above lr 0.01 the owner form stays.
        const float user_lr_max = im_user_lr_max_milli_ * 1e-3f;
        const bool p_rep = im_user_replicas_ > 0 ||
                           (im_user_replicas_ < 0 && !im_single_wave_ && users_here < static_cast<int64_t>(nq) * 3072 && c.lr <= user_lr_max);
        // Whole matrices keep one owner XCD per user -- except for the HEAVY users, the ones the collision rule below would put on
        // fp32 atomics (ML-20M shape: degree >= ~780, 2 % of the users, 17 % of the triples; `xcd_hot_tau = 0` showed those atomics
        // cost 8 % of the walk).  They alone get the replica treatment: their entries are spread over the queues, so a heavy user's
        // share of one queue is nq times smaller and its row is updated with plain stores on the XCD's replica; 13 MB of replicas
        // instead of 640, merged by the delta rule with the item replicas.  Same lr bound as the all-user form.
        const int64_t waves0 = im_resident_waves();
        const double inflight0 = (!im_prefetch() ? 0.25 : (c.fresh ? 0.5 : 2.0)) * (static_cast<double>(waves0) / nq) * (im_dual() ? 2.0 : 1.0);
        const double tau0 = xcd_hot_tau_ * 1e-3;
        const bool p_hyb = !p_rep && im_user_hybrid_ && !im_single_wave_ && c.lr <= user_lr_max && tau0 > 0.0 && inflight0 > 0.0 && nq > 1;
        // degree from which the owner-share rule fires: deg * num_neg / (triples / nq) * inflight >= tau
        const int64_t heavy_deg = p_hyb ? std::max<int64_t>(1, static_cast<int64_t>(std::ceil(tau0 * (static_cast<double>(c.total) / nq) / (inflight0 * num_neg_)))) : 0;
        const int spread_mode = p_rep ? 1 : (p_hyb ? (im_user_hybrid_ >= 2 ? 3 : 2) : 0);
        const bool cached = keeps && im_gen_ == csr_generation_ && im_start_ == start_x && im_next_ == next_x && im_n_ == n && im_built_blocks_ == blocks && im_built_nq_ == nq &&
                            im_built_spread_mode_ == spread_mode && im_built_heavy_deg_ == heavy_deg;
        if (!cached) {
            im_key_a_.resize(static_cast<size_t>(n)); im_key_b_.resize(static_cast<size_t>(n));
            im_pos_a_.resize(static_cast<size_t>(n)); im_pos_b_.resize(static_cast<size_t>(n));
            im_qbeg_dev_.resize(kImMaxQueues + 1);
            hipLaunchKernelGGL(im_keys_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, p.rows, p.keys, n, nq,
                               static_cast<uint32_t>(blocks), static_cast<uint32_t>(Q_rows_), spread_mode, p.indptr, heavy_deg, im_key_a_.get(), im_pos_a_.get());
            BFH_HIP(hipGetLastError());
            int bits = 1;
            while ((int64_t(1) << bits) < static_cast<int64_t>(nq) * blocks * Q_rows_) ++bits;
            device_sort_pairs_u32(im_key_a_.get(), im_key_b_.get(), im_pos_a_.get(), im_pos_b_.get(), n, bits, im_tmp_, stream);
            hipLaunchKernelGGL(im_bounds_kernel, dim3(1), dim3(64), 0, stream, im_key_b_.get(), n, nq, static_cast<uint32_t>(blocks * Q_rows_),
                               im_qbeg_dev_.get());
            BFH_HIP(hipGetLastError());
            BFH_HIP(hipMemcpyAsync(im_qbeg_, im_qbeg_dev_.get(), (nq + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
            sync_stream();
            im_gen_ = keeps ? csr_generation_ : -1;
            im_start_ = start_x; im_next_ = next_x; im_n_ = n; im_built_blocks_ = blocks; im_built_nq_ = nq; im_built_spread_mode_ = spread_mode; im_built_heavy_deg_ = heavy_deg;
        }
        // ---- per-row policy flags ----
        const int64_t waves = im_resident_waves();
        const double triples = static_cast<double>(c.total);
        itemcnt_.resize(static_cast<size_t>(Q_rows_));
        hot_.resize(static_cast<size_t>(Q_rows_));
        im_flush_.resize(static_cast<size_t>(Q_rows_));
        im_hot_user_.resize(static_cast<size_t>(P_rows_));
        double cnt_triples = triples;   // what the item counts are normalised by
        if (resident_) {
            if (itemcnt_gen_ != csr_generation_) {
                BFH_HIP(hipMemsetAsync(itemcnt_.get(), 0, itemcnt_.bytes(), stream));
                hipLaunchKernelGGL(item_count_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((resident_nnz_ + 255) / 256, 4096))), dim3(256), 0,
                                   stream, keys_.get(), resident_nnz_, itemcnt_.get());
                itemcnt_gen_ = csr_generation_;
            }
            cnt_triples = static_cast<double>(resident_nnz_) * num_neg_;
        } else if (keeps && itemcnt_gen_ == csr_generation_ && itemcnt_start_ == start_x && itemcnt_next_ == next_x) {
            // the histogram of this very chunk (auto-resident) is still there
        } else {
            BFH_HIP(hipMemsetAsync(itemcnt_.get(), 0, itemcnt_.bytes(), stream));
            hipLaunchKernelGGL(item_count_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 4096))), dim3(256), 0, stream, p.keys, n,
                               itemcnt_.get());
            itemcnt_gen_ = keeps ? csr_generation_ : -1;
            itemcnt_start_ = start_x; itemcnt_next_ = next_x;
        }
        const double queue_waves = static_cast<double>(waves) / nq;
        // rows a queue's waves hold between the load and the store of one update: the current and the prefetched
        // triple's, or -- when the row is re-read right before the store -- one L2 round trip out of a triple's time
        const double inflight = (!im_prefetch() ? 0.25 : (c.fresh ? 0.5 : 2.0)) * queue_waves * (im_dual() ? 2.0 : 1.0);
        const double tau = xcd_hot_tau_ * 1e-3;
        // the staleness budgets are stated for lr = 0.05 and scale with 1 / lr: what matters is how far a row moves
        int64_t sync_updates = xcd_sync_updates_ > 0 ? xcd_sync_updates_ : int64_t(1) << 23;
        if (comm_) {
            // multi-GPU: every merge segment is an exchange point.  By default a call is ONE segment whose exchange is finished
            // before it returns (blocking).  "comm_segments" = k cuts the call into k segments whose exchanges travel behind the
            // NEXT segment's walk (one segment late; the last one stays in flight until the next exchange point or reader).
            // Measured / simulated (profiles/r02_shard_times_*, r02_local_sgd_study_*): a segment costs ~0.2 ms of merge / drain /
            // launch-tail work on top of its walk -- as much as the 14 MB all-reduce it hides -- so on ML-20M pipelining loses at
            // every N (N = 2: 5.1 ms per epoch with four segments against 4.2 + ~0.2 blocking); and a delta that lands one
            // interval late needs >= 4 exchange points per epoch to match the blocking exchange's statistics (8 ranks, lr 0.002:
            // the popular items' biases end 40 % off with one delayed exchange, 14 % with two, 1.5 % with four; blocking: 6 %).
            // It pays only where a walk is long against the fixed cost (large d, WARP-sized shards).
            // Every rank must run the SAME number of exchange points per call (each is a collective): the knob, not local sizes.
            const int64_t segs = comm_segments_ > 0 ? comm_segments_ : 1;
            comm_blocking_call_ = segs == 1;
            comm_forced_segments_ = segs;
            sync_updates = std::min<int64_t>(sync_updates, std::max<int64_t>(1, (c.total + segs - 1) / segs));
        }
        int64_t q_entries[kImMaxQueues] = {0};
        for (int x = 0; x < nq; ++x) q_entries[x] = im_qbeg_[x + 1] - im_qbeg_[x];
        ImPlan plan = im_make_plan(nq, q_entries, num_neg_, sync_updates);
        if (comm_) plan.segments = comm_forced_segments_;   // not left to the rounding of local sizes
        const int64_t segments = plan.segments;
        const double lr_scale = c.lr > 0.f ? 0.05 / static_cast<double>(c.lr) : 1e9;
        const double max_stale = std::min(1e9, static_cast<double>(im_max_stale_) * lr_scale);
        // positive steps of a row between two merges, in units of lr: counts (of `cnt_triples` triples) -> this call's share
        const double lr_steps_per_count = static_cast<double>(num_neg_) * (triples / cnt_triples) / static_cast<double>(segments) * c.lr;
        hipLaunchKernelGGL(im_item_flags_kernel, dim3((Q_rows_ + 255) / 256), dim3(256), 0, stream, itemcnt_.get(),
                           uniform_ ? nullptr : p.cum_table, cum_total_, Q_rows_, static_cast<double>(num_neg_), cnt_triples,
                           uniform_ ? 1.0 / Q_rows_ : 0.0, inflight, tau, static_cast<double>(waves) * (im_dual() ? 2.0 : 1.0), max_stale, lr_steps_per_count,
                           im_drift_budget_milli_ * 1e-3, hot_.get(), im_flush_.get());
        BFH_HIP(hipMemsetAsync(im_hot_user_.get(), 0, im_hot_user_.bytes(), stream));
        hipLaunchKernelGGL(im_user_flags_kernel, dim3((next_x - start_x + 255) / 256), dim3(256), 0, stream, p.indptr, start_x, next_x - start_x,
                           static_cast<double>(num_neg_), triples / nq, triples, inflight, tau, spread_mode >= 2 ? 2 : spread_mode, heavy_deg, im_hot_user_.get());
        BFH_HIP(hipGetLastError());
        // ---- weights of the merges' sums (0 = plain sum) ----
        // (not for the single-wave test hook: one wave takes every step there, in ONE replica, and the sum is already the sequential result)
        // ... nor with "xcd_merge_mean": the mean already scales the sum by 1 / n, and a saturation weight (between 1 / n and 1) on top of it
        // would damp the rows twice.  The weight of n replicas assumes the merge sums exactly those: nq <= kXcdReplicas
        BFH_REQUIRE(nq <= kXcdReplicas, "hogwild_atomic=3: more queues than per-XCD replicas");
        // The stiffness constants model a row that contracts towards a local equilibrium by exp(-lr k) per step with k = the curvature of the
        // sigmoid (<= 1/4).  They were calibrated at the reference's default lr (0.002).  A row whose steps are large -- a higher lr -- sits in
        // the flat part of the sigmoid for most of them (the reference path's own biases at lr 0.05: logits of a few per cent), its curvature is a
        // fraction of 1/4, and the constant over-damps: measured on the reference benchmark's schedule (lr 0.05 -> 0.0001, 10 epochs) the merged
        // negative steps of the biases were scaled by ~0.36 in the middle epochs and |Qb| ended at 147.7 against 183.0 for the reference path at
        // EVERY pool width and for this library's own all-atomic kernel (profiles/r06_bpr_lr005_width_and_knobs.txt, r06_bpr_lr005_bias_rows.txt).
        // Above the calibration lr the constants therefore shrink like lr_ref / lr: the argument x = lr k m of the saturation weight stays what it
        // is at lr_ref.  ("xcd_stiff_lr_ref" = 0: the constants at every lr, the form up to round 5.)
        const double lr_ref = xcd_stiff_lr_ref_micro_ * 1e-6;
        const double stiff_scale = (lr_ref > 0.0 && static_cast<double>(c.lr) > lr_ref) ? lr_ref / static_cast<double>(c.lr) : 1.0;
        const bool w_items = (xcd_stiff_q_milli_ > 0 || xcd_stiff_b_milli_ > 0) && !im_single_wave_ && !xcd_merge_mean_;
        const bool w_users = xcd_stiff_p_milli_ > 0 && spread_mode != 0 && !im_single_wave_ && !xcd_merge_mean_;
        if (w_items) {
            xcd_wq_.resize(static_cast<size_t>(Q_rows_));
            xcd_wb_.resize(static_cast<size_t>(Q_rows_));
            hipLaunchKernelGGL(xcd_item_weight_kernel, dim3((Q_rows_ + 255) / 256), dim3(256), 0, stream, uniform_ ? nullptr : p.cum_table, cum_total_, Q_rows_,
                               triples / static_cast<double>(segments), uniform_ ? 1.0 / Q_rows_ : 0.0, static_cast<double>(c.lr), xcd_stiff_q_milli_ * 1e-3 * stiff_scale,
                               xcd_stiff_b_milli_ * 1e-3 * stiff_scale, nq, xcd_wq_.get(), xcd_wb_.get());
        }
        if (w_users) {
            xcd_wp_.resize(static_cast<size_t>(P_rows_));
            hipLaunchKernelGGL(xcd_user_weight_kernel, dim3((next_x - start_x + 255) / 256), dim3(256), 0, stream, p.indptr, start_x, next_x - start_x,
                               static_cast<double>(num_neg_) / static_cast<double>(segments), static_cast<double>(c.lr), xcd_stiff_p_milli_ * 1e-3 * stiff_scale, nq, spread_mode,
                               heavy_deg, xcd_wp_.get());
        }
        BFH_HIP(hipGetLastError());
        // ---- replicas of the item factors (+ the copy they started from) ----
        xcd_alloc(true);
        c.rep_Q = repQ_.get();
        c.rep_Qb = repQb_.get();
        c.rep_stride = static_cast<int64_t>(Q_rows_) * vdim_;
        c.rep_bstride = rep_bstride();
        c.hot = hot_.get();
        xcd_broadcast(true);
        const int64_t np4 = static_cast<int64_t>(P_rows_) * vdim_ / 4;          // replica stride (float4s)
        const int64_t up4 = users_here * vdim_ / 4, uoff4 = static_cast<int64_t>(start_x) * vdim_ / 4;   // this call's rows
        const bool p_any = p_rep || p_hyb;
        if (p_any) {
            // eight replicas + the copy they started from, addressed like P (only the rows of users with flag 2 are ever touched);
            // P itself receives the hot users' atomics and whatever the drain launch does
            if (repP_.size() < static_cast<size_t>(kXcdReplicas + 1) * P_rows_ * vdim_) repP_.resize(static_cast<size_t>(kXcdReplicas + 1) * P_rows_ * vdim_);
            hipLaunchKernelGGL((xcd_broadcast_rows_kernel<float4>), dim3(static_cast<unsigned>(std::min<int64_t>((up4 + 255) / 256, 8192))), dim3(256), 0, stream,
                               reinterpret_cast<const float4*>(P_.get()) + uoff4, reinterpret_cast<float4*>(repP_.get()) + uoff4, up4, np4, kXcdReplicas + 1,
                               static_cast<const uint8_t*>(im_hot_user_.get()) + start_x, vdim_ / 4, 2);
            BFH_HIP(hipGetLastError());
        }
        // ---- queues, slice order, segments ----
        ImQueues q{};
        q.ent_key = im_key_b_.get();
        q.ent_pos = im_pos_b_.get();
        q.nq = nq;
        for (int i = 0; i < 16; ++i) q.xcd_queue[i] = im_xcd_queue_[i];
        q.p_nt = im_p_nt_;
        q.study = im_study_;
        q.hot_user = im_hot_user_.get();
        q.rep_P = p_any ? repP_.get() : nullptr;
        q.rep_pstride = static_cast<int64_t>(P_rows_) * vdim_;
        q.flush_every = im_flush_.get();
        q.strict = im_single_wave_;
        q.trace = (im_single_wave_ && im_trace_.size() >= static_cast<size_t>(c.total)) ? im_trace_.get() : nullptr;
        q.done = reinterpret_cast<unsigned long long*>(scratch_.get() + 1);
        if (im_presample_) {
            // A draw is a pure function of (seed, nnz position, slot, epoch, attempt) -- not of the model -- so the negatives of
            // the NEXT epoch over this same chunk can be drawn on a side stream while this epoch's walk runs (0.35 ms per
            // ML-20M epoch off the critical path).  The speculation is keyed on everything the draws depend on; a call it does
            // not match (another chunk, the same epoch again, changed keys) draws its own on the main stream as before.
            const dim3 pgrid(static_cast<unsigned>((c.total + 255) / 256)), pblock(256);
            const PreKey want{static_cast<int64_t>(p.epoch), start_x, next_x, c.total, csr_generation_, p.nnz_offset, p.shift, static_cast<int64_t>(p.seed),
                              (c.uniform ? 1 : 0) | (c.verify_neg ? 2 : 0) | (c.num_neg << 2), c.cum_total};
            int buf = 0;
            if (pre_valid_ && pre_key_ == want) {
                buf = pre_buf_;
                BFH_HIP(hipStreamWaitEvent(stream, pre_done_, 0));
            } else {
                if (pre_valid_) BFH_HIP(hipStreamSynchronize(pre_stream_));   // a stale speculation may still be writing the other buffer
                if (im_neg_[0].size() < static_cast<size_t>(c.total)) im_neg_[0].resize(static_cast<size_t>(c.total));
                hipLaunchKernelGGL(bpr_presample_kernel, pgrid, pblock, 0, stream, p, c, im_neg_[0].get());
                BFH_HIP(hipGetLastError());
            }
            pre_valid_ = false;
            q.neg_pre = im_neg_[buf].get();
            if (im_presample_ahead_ && keeps && !im_single_wave_) {
                if (!pre_stream_) {
                    BFH_HIP(hipStreamCreateWithFlags(&pre_stream_, hipStreamNonBlocking));
                    BFH_HIP(hipEventCreateWithFlags(&pre_done_, hipEventDisableTiming));
                    BFH_HIP(hipEventCreateWithFlags(&pre_ready_, hipEventDisableTiming));
                }
                const int other = 1 - buf;
                if (im_neg_[other].size() < static_cast<size_t>(c.total)) im_neg_[other].resize(static_cast<size_t>(c.total));
                SgdParams p2 = p;
                p2.epoch = p.epoch + 1;
                BFH_HIP(hipEventRecord(pre_ready_, stream));                   // the staged chunk (keys, row ids) is in place behind this point
                BFH_HIP(hipStreamWaitEvent(pre_stream_, pre_ready_, 0));
                hipLaunchKernelGGL(bpr_presample_kernel, pgrid, pblock, 0, pre_stream_, p2, c, im_neg_[other].get());
                BFH_HIP(hipGetLastError());
                BFH_HIP(hipEventRecord(pre_done_, pre_stream_));
                pre_key_ = want;
                pre_key_.epoch = static_cast<int64_t>(p.epoch) + 1;
                pre_buf_ = other;
                pre_valid_ = true;
            }
        }
        BFH_HIP(hipMemsetAsync(scratch_.get() + 1, 0, sizeof(double), stream));
        q.slice_len = plan.slice_len;
        for (int x = 0; x < nq; ++x) {
            q.q_beg[x] = im_qbeg_[x];
            q.q_triples[x] = plan.q_triples[x];
            q.q_slices[x] = plan.q_slices[x];
            q.q_stride[x] = plan.q_stride[x];
        }
        im_tickets_.resize(static_cast<size_t>(segments) * kImMaxQueues);
        BFH_HIP(hipMemsetAsync(im_tickets_.get(), 0, im_tickets_.bytes(), stream));
        t_aux_.end(slot, stream);
        for (int64_t sgm = 0; sgm < segments; ++sgm) {
            int64_t seg_slices = 0;
            for (int x = 0; x < nq; ++x) {
                im_segment_tickets(plan, x, sgm, &q.t_beg[x], &q.t_end[x]);
                seg_slices += q.t_end[x] - q.t_beg[x];
            }
            q.tickets = im_tickets_.get() + sgm * kImMaxQueues;
            const int64_t grid_waves = std::max<int64_t>(4, std::min(waves, seg_slices));
            slot = t_main_.begin(stream);
            if (!im_drain_only_ && !im_single_wave_) im_launch(p, c, q, grid_waves, false);
            t_main_.end(slot, stream);
            stats.launches += 1;
            slot = t_aux_.begin(stream);
            im_launch(p, c, q, grid_waves, true);
            // multi-GPU: the other ranks' deltas of the previous exchange point land in Q before the replicas are folded in and
            // refreshed; this segment's own delta goes out behind the merge and travels while the next walk runs
            exchange_finish(true);
            xcd_merge(sgm + 1 < segments, c.hot, true, w_items);
            if (p_any) {   // P <- P + sum_x (P_x - B) for the users that have replicas (flag 2); the others were updated in P itself
                hipLaunchKernelGGL((xcd_merge_kernel<float4>), dim3(static_cast<unsigned>(std::min<int64_t>((up4 + 255) / 256, 8192))), dim3(256), 0, stream,
                                   reinterpret_cast<float4*>(P_.get()) + uoff4, reinterpret_cast<float4*>(repP_.get()) + uoff4, up4, np4, 1.0f,
                                   sgm + 1 < segments ? 1 : 0, static_cast<const uint8_t*>(im_hot_user_.get()) + start_x, vdim_ / 4,
                                   reinterpret_cast<float4*>(repP_.get()) + kXcdReplicas * np4 + uoff4, 2, w_users ? xcd_wp_.get() + start_x : nullptr);
                BFH_HIP(hipGetLastError());
            }
            t_aux_.end(slot, stream);
            stats.merges += 1;
            if (comm_) {
                exchange_weights(static_cast<double>(seg_slices) * plan.slice_len, c.lr, num_neg_, uniform_);
                exchange_begin();
            }
        }
        im_expect_done_ = c.total;
    }
    // after the stream was synchronised: every triple must have been processed exactly once
    void im_check_done() {
        if (im_expect_done_ < 0) return;
        unsigned long long done = 0;
        BFH_HIP(hipMemcpy(&done, scratch_.get() + 1, sizeof(done), hipMemcpyDeviceToHost));
        const int64_t want = im_expect_done_;
        im_expect_done_ = -1;
        if (static_cast<int64_t>(done) != want)
            throw Error(BFH_ERR_HIP, "hogwild_atomic=3: " + std::to_string(done) + " of " + std::to_string(want) + " updates were processed");
    }

    // stream-ordered helpers of policy 2
    // `with_base`: a ninth copy keeps what the replicas started from (policy 3)
    void xcd_alloc(bool with_base) {
        const size_t copies = kXcdReplicas + (with_base ? 1 : 0);
        if (repQ_.size() < copies * Q_rows_ * vdim_) repQ_.resize(copies * Q_rows_ * vdim_);
        if (repQb_.size() < copies * rep_bstride()) repQb_.resize(copies * rep_bstride());
    }
    void xcd_broadcast(bool with_base) {
        const int64_t nq4 = static_cast<int64_t>(Q_rows_) * vdim_ / 4;
        const int copies = kXcdReplicas + (with_base ? 1 : 0);
        hipLaunchKernelGGL((xcd_broadcast_kernel<float4>), dim3(static_cast<unsigned>(std::min<int64_t>((nq4 + 255) / 256, 8192))), dim3(256), 0, stream,
                           reinterpret_cast<const float4*>(Q_.get()), reinterpret_cast<float4*>(repQ_.get()), nq4, nq4, copies);
        hipLaunchKernelGGL((xcd_broadcast_kernel<float>), dim3((Q_rows_ + 255) / 256), dim3(256), 0, stream,
                           static_cast<const float*>(Qb_.get()), repQb_.get(), static_cast<int64_t>(Q_rows_), rep_bstride(), copies);
        BFH_HIP(hipGetLastError());
    }
    void xcd_merge(bool write_replicas, const uint8_t* hot, bool with_base, bool weighted = false) {
        const int64_t nq4 = static_cast<int64_t>(Q_rows_) * vdim_ / 4;
        const float scale = xcd_merge_mean_ ? 1.0f / kXcdReplicas : 1.0f;
        float4* base4 = with_base ? reinterpret_cast<float4*>(repQ_.get()) + kXcdReplicas * nq4 : nullptr;
        float* baseb = with_base ? repQb_.get() + kXcdReplicas * rep_bstride() : nullptr;
        hipLaunchKernelGGL((xcd_merge_kernel<float4>), dim3(static_cast<unsigned>(std::min<int64_t>((nq4 + 255) / 256, 8192))), dim3(256), 0, stream,
                           reinterpret_cast<float4*>(Q_.get()), reinterpret_cast<float4*>(repQ_.get()), nq4, nq4, scale, write_replicas ? 1 : 0,
                           hot, vdim_ / 4, base4, 0, weighted ? xcd_wq_.get() : nullptr);
        hipLaunchKernelGGL((xcd_merge_kernel<float>), dim3((Q_rows_ + 255) / 256), dim3(256), 0, stream, Qb_.get(), repQb_.get(),
                           static_cast<int64_t>(Q_rows_), rep_bstride(), scale, write_replicas ? 1 : 0, hot, 1, baseb, 0, weighted ? xcd_wb_.get() : nullptr);
        BFH_HIP(hipGetLastError());
    }
    // hot-row flags for this call (policy 2); returns null when the split is disabled
    template <bool INJECT>
    const uint8_t* xcd_hot_rows(const SgdParams& p, const BprConsts& c, int64_t seg_work) {
        if (xcd_hot_tau_ <= 0) return nullptr;
        itemcnt_.resize(static_cast<size_t>(Q_rows_));
        hot_.resize(static_cast<size_t>(Q_rows_));
        double pos_mult = num_neg_, triples = static_cast<double>(c.total), neg_uniform = uniform_ ? 1.0 / Q_rows_ : 0.0;
        const int64_t* cum = (!INJECT && !uniform_) ? p.cum_table : nullptr;
        auto count = [&](const int32_t* idx, int64_t n) {
            hipLaunchKernelGGL(item_count_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 4096))), dim3(256), 0, stream, idx, n,
                               itemcnt_.get());
        };
        if (INJECT) {
            BFH_HIP(hipMemsetAsync(itemcnt_.get(), 0, itemcnt_.bytes(), stream));
            count(c.inj_p, c.total);
            count(c.inj_n, c.total);
            pos_mult = 1.0;
            neg_uniform = 0.0;
            itemcnt_gen_ = -1;
        } else if (resident_) {
            if (itemcnt_gen_ != csr_generation_) {   // popularity of the whole resident matrix, counted once
                BFH_HIP(hipMemsetAsync(itemcnt_.get(), 0, itemcnt_.bytes(), stream));
                count(keys_.get(), resident_nnz_);
                itemcnt_gen_ = csr_generation_;
            }
            triples = static_cast<double>(resident_nnz_) * num_neg_;
        } else {
            BFH_HIP(hipMemsetAsync(itemcnt_.get(), 0, itemcnt_.bytes(), stream));
            count(p.keys, p.chunk_nnz);
            itemcnt_gen_ = -1;
        }
        const double waves = static_cast<double>(std::min<int64_t>(resident_waves(pick<INJECT>(c)), seg_work));
        const double inflight = 2.0 * waves / kXcdReplicas;     // a wave holds the two item rows of its next triple
        hipLaunchKernelGGL(xcd_hot_kernel, dim3((Q_rows_ + 255) / 256), dim3(256), 0, stream, itemcnt_.get(), cum, cum_total_, Q_rows_, pos_mult,
                           triples, neg_uniform, inflight, xcd_hot_tau_ * 1e-3, hot_.get());
        BFH_HIP(hipGetLastError());
        return hot_.get();
    }
    int64_t rep_bstride() const { return (static_cast<int64_t>(Q_rows_) + 63) / 64 * 64; }

    template <bool INJECT>
    void launch(const SgdParams& p, const BprConsts& c_in, int start_x = 0, int next_x = 0) {
        BprConsts c = c_in;
        const int64_t n_work = (c.total + c.chunk - 1) / c.chunk;
        const bool reps = c.atomic == 2;
        // adam / adagrad: P, Q are frozen, so the item-side gradients are summed by the sorted gather (no per-triple atomics)
        const bool two_pass = !INJECT && optimizer_ != "sgd" && accum_two_pass_ != 0;
        if (two_pass) {
            acc_prepare(c.total);
            c.two_pass = 1;
            c.uc_out = acc_uc_.get();
            c.neg_out = acc_neg_.get();
        }
        int64_t seg_work = n_work;          // work items per launch
        if (reps) {
            xcd_alloc(false);
            c.rep_Q = repQ_.get();
            c.rep_Qb = repQb_.get();
            c.rep_stride = static_cast<int64_t>(Q_rows_) * vdim_;
            c.rep_bstride = rep_bstride();
            // a segment is a whole number of work items per resident wave: it ends when its slowest wave does
            const int64_t waves = resident_waves(pick<INJECT>(c));
            const int64_t sync_updates = xcd_sync_updates_ > 0 ? xcd_sync_updates_ : int64_t(1) << 21;
            seg_work = std::max<int64_t>(1, (sync_updates / c.chunk + waves / 2) / waves) * waves;
            const int slot = t_aux_.begin(stream);
            c.hot = xcd_hot_rows<INJECT>(p, c, std::min(seg_work, n_work));
            xcd_broadcast(false);
            t_aux_.end(slot, stream);
        }
        for (int64_t w0 = 0; w0 < n_work; w0 += seg_work) {
            c.work_begin = w0;
            c.work_end = std::min(n_work, w0 + seg_work);
            launch_segment<INJECT>(p, c);
            if (reps) {
                const int slot = t_aux_.begin(stream);
                xcd_merge(c.work_end < n_work, c.hot, false);
                t_aux_.end(slot, stream);
                stats.merges += 1;
            }
        }
        if (two_pass) {
            const int slot = t_aux_.begin(stream);
            acc_build_positive_list(p, start_x, next_x);
            const float sab_pos[3] = {1.f, 0.f, 0.f}, sab_neg[3] = {-1.f, 0.f, 0.f};
            acc_gather(p, num_neg_, update_i_, update_j_, sab_pos, sab_neg, use_bias_);
            t_aux_.end(slot, stream);
        }
    }

    template <bool INJECT>
    void launch_segment(const SgdParams& p, const BprConsts& c) {
        const int64_t n_work = c.work_end - c.work_begin;
        const KernelFn fn = pick<INJECT>(c);
        dim3 block(256), grid(1);
        if (sequential_) {
            block = dim3(64);
        } else {
            int64_t waves = resident_waves(fn);
            if (waves > n_work) waves = n_work;
            grid = dim3(static_cast<unsigned>((waves + 3) / 4));
        }
        const int slot = t_main_.begin(stream);
        hipLaunchKernelGGL(fn, grid, block, 0, stream, p, c);
        BFH_HIP(hipGetLastError());
        t_main_.end(slot, stream);
        stats.launches += 1;
    }

    void partial_update(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, double* loss_sum, double* n_samples) {
        SgdParams p;
        const int64_t n = stage_chunk(start_x, next_x, indptr, keys, &p);
        *loss_sum = 0.0;
        *n_samples = static_cast<double>(n) * num_neg_;
        if (comm_) exchange_arm();   // Z = the replicated state (sgd: Q | Qb, else the gradient buffers) before this rank changes it
        if (n == 0) {
            // an empty chunk of this rank's shard: nothing to walk, but every exchange point of the call is a collective the
            // other ranks enter -- take part with a zero delta
            if (comm_ && optimizer_ == "sgd") {
                exchange_histogram(resident_ ? keys_.get() : p.keys, resident_ ? resident_nnz_ : 0);
                const int64_t points = comm_segments_ > 0 ? comm_segments_ : 1;
                for (int64_t k = 0; k < points; ++k) {
                    exchange_finish();   // no local work since the last begin: Q <- Z exactly
                    exchange_weights(0.0, current_lr(), num_neg_, uniform_);
                    exchange_begin();
                }
                if (!comm_overlap_ || points == 1) exchange_finish();
                sync_stream();
            }
            return;
        }
        BFH_REQUIRE(uniform_ || have_cum_, "sampling_power != 0 needs set_cumulative_table");
        BFH_REQUIRE(uniform_ || cum_total_ > 0, "cumulative table is empty");
        BprConsts c = consts(current_lr());
        c.total = n * num_neg_;
        if (compute_loss_) BFH_HIP(hipMemsetAsync(scratch_.get(), 0, sizeof(double), stream));
        if (comm_ && optimizer_ == "sgd") {
            // the popularity every rank weighs its rows by: the resident matrix when there is one, else this call's chunk
            exchange_histogram(resident_ ? keys_.get() : p.keys, resident_ ? resident_nnz_ : n);
        }
        if (c.atomic == 3 && !im_fits(c, n)) c.atomic = 1;   // 32-bit sort key / entry index exhausted: the user-major atomic walk
        if (c.atomic == 3) {
            launch_item_major(p, c, start_x, next_x);
        } else {
            if (optimizer_ == "sgd") exchange_finish();
            launch<false>(p, c, start_x, next_x);
            if (comm_ && optimizer_ == "sgd") {
                // the user-major walks make one exchange point per call; with "comm_segments" = k every rank must still enter k
                // collectives (a rank whose chunk does not fit the item-major plan lands here while the others cut theirs)
                const int64_t points = comm_segments_ > 0 ? comm_segments_ : 1;
                comm_blocking_call_ = points == 1;
                exchange_weights(static_cast<double>(c.total), c.lr, num_neg_, uniform_);
                exchange_begin();
                for (int64_t k = 1; k < points; ++k) {
                    exchange_finish(false);
                    exchange_weights(0.0, c.lr, num_neg_, uniform_);
                    exchange_begin();
                }
            }
        }
        if (comm_ && (!comm_overlap_ || comm_blocking_call_)) exchange_finish();
        if (compute_loss_) BFH_HIP(hipMemcpyAsync(loss_sum, scratch_.get(), sizeof(double), hipMemcpyDeviceToHost, stream));
        sync_stream();
        im_check_done();
        harvest_timers();
        stats.samples += c.total;
        advance_progress(start_x, next_x, indptr);
    }

    void update_triples(int64_t n, const int32_t* users, const int32_t* pos, const int32_t* neg, double lr) {
        BFH_REQUIRE(model_on_gpu_, "update_triples before initialize_model(..., set_gpu=True)");
        if (n <= 0) return;
        exchange_finish();
        inj_.resize(static_cast<size_t>(3 * n));
        BFH_HIP(hipMemcpyAsync(inj_.get(), users, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(inj_.get() + n, pos, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(inj_.get() + 2 * n, neg, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        SgdParams p{};
        p.P = P_.get(); p.Q = Q_.get(); p.Qb = Qb_.get();
        p.gradP = gradP_.get(); p.gradQ = gradQ_.get(); p.gradQb = gradQb_.get();
        p.cntP = cntP_.get(); p.cntQ = cntQ_.get();
        p.P_rows = P_rows_; p.Q_rows = Q_rows_; p.d = d_; p.vdim = vdim_;
        BprConsts c = consts(lr);
        if (c.atomic == 3) c.atomic = 1;   // injected triples have no CSR to regroup: atomics
        c.compute_loss = 0;
        c.num_neg = 1;
        c.total = n;
        c.inj_u = inj_.get(); c.inj_p = inj_.get() + n; c.inj_n = inj_.get() + 2 * n;
        launch<true>(p, c);
        sync_stream();
        harvest_timers();
        stats.samples += n;
    }

    double compute_loss(int n, const int32_t* users, const int32_t* pos, const int32_t* neg) {
        BFH_REQUIRE(model_on_gpu_, "compute_loss before initialize_model(..., set_gpu=True)");
        if (n <= 0) return 0.0;  // the reference divides by zero here (bpr.cc:243); callers guard (bpr.py:141)
        exchange_finish();
        inj_.resize(static_cast<size_t>(3) * n);
        BFH_HIP(hipMemcpyAsync(inj_.get(), users, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(inj_.get() + n, pos, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpyAsync(inj_.get() + 2 * n, neg, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemsetAsync(scratch_.get(), 0, sizeof(double), stream));
        const int slot = t_aux_.begin(stream);
        hipLaunchKernelGGL(bpr_loss_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, P_.get(), Q_.get(), Qb_.get(), inj_.get(),
                           inj_.get() + n, inj_.get() + 2 * n, n, vdim_, static_cast<int>(use_bias_), scratch_.get());
        BFH_HIP(hipGetLastError());
        t_aux_.end(slot, stream);
        double l = 0.0;
        BFH_HIP(hipMemcpyAsync(&l, scratch_.get(), sizeof(double), hipMemcpyDeviceToHost, stream));
        sync_stream();
        harvest_timers();
        return l / static_cast<double>(n);
    }

    int num_neg_ = 1;
    bool verify_neg_ = true, uniform_ = true;
    DevBuf<float> exp_table_;
    DevBuf<float> repQ_, repQb_;   // policy 2: [8][Q_rows][vdim], [8][ceil64(Q_rows)]
    DevBuf<float> repP_;           // policy 3 on small shards: [8 + 1][P_rows][vdim]
    DevBuf<int> itemcnt_;          // policy 2: updates per item row (popularity)
    DevBuf<uint8_t> hot_;
    int64_t itemcnt_gen_ = -1;
    int itemcnt_start_ = -1, itemcnt_next_ = -1;
    std::map<const void*, int> occupancy_;   // kernel -> resident 256-thread blocks per CU
    // policy 3
    int im_nq_ = 0, im_xcd_queue_[16];
    DevBuf<uint32_t> im_key_a_, im_key_b_;
    DevBuf<int32_t> im_pos_a_, im_pos_b_;
    DevBuf<char> im_tmp_;
    DevBuf<int64_t> im_qbeg_dev_;
    DevBuf<uint8_t> im_flush_, im_hot_user_;
    DevBuf<int> im_tickets_;
    DevBuf<int32_t> im_neg_[2];    // pre-drawn negatives: this call's, and the speculation for the next epoch
    struct PreKey {
        int64_t epoch;
        int start_x, next_x;
        int64_t total, gen, nnz_offset, shift, seed;
        int flags;
        int64_t cum_total;
        bool operator==(const PreKey& o) const {
            return epoch == o.epoch && start_x == o.start_x && next_x == o.next_x && total == o.total && gen == o.gen && nnz_offset == o.nnz_offset &&
                   shift == o.shift && seed == o.seed && flags == o.flags && cum_total == o.cum_total;
        }
    };
    PreKey pre_key_{};
    bool pre_valid_ = false;
    int pre_buf_ = 0;
    hipStream_t pre_stream_ = nullptr;
    hipEvent_t pre_done_ = nullptr, pre_ready_ = nullptr;
    int64_t im_qbeg_[kImMaxQueues + 1] = {0};
    int64_t im_gen_ = -1, im_n_ = -1, im_expect_done_ = -1, im_built_blocks_ = -1;
    int im_built_nq_ = -1;
    int im_start_ = -1, im_next_ = -1;
    DevBuf<int32_t> inj_;
};

}  // namespace bfh

using bfh::BprHandle;
using bfh::guarded;

extern "C" {

void* bfh_bpr_create(void) {
    try {
        BprHandle* h = new BprHandle();
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
void bfh_bpr_destroy(void* h) { delete static_cast<BprHandle*>(h); }
int bfh_bpr_set_device(void* h, int device) {
    return guarded(h, [&] {
        BFH_REQUIRE(!static_cast<BprHandle*>(h)->stream || static_cast<BprHandle*>(h)->device == device,
                    "set_device after init: the handle's stream and buffers live on the device it was initialised on");
        static_cast<BprHandle*>(h)->device = device;
        BFH_HIP(hipSetDevice(device));
        return BFH_OK;
    });
}
int bfh_bpr_init(void* h, const char* opt_json_path) {
    int ok = 0;
    int rc = guarded(h, [&] { ok = static_cast<BprHandle*>(h)->init(opt_json_path) ? 1 : 0; return BFH_OK; });
    return rc == BFH_OK ? ok : rc;
}
int bfh_bpr_get_vdim(void* h) { return h ? static_cast<BprHandle*>(h)->get_vdim() : BFH_ERR_INVALID; }
int bfh_bpr_initialize_model(void* h, float* P, int P_rows, float* Q, float* Qb, int Q_rows, int64_t num_nnz, int set_gpu) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->initialize_model(P, P_rows, Q, Qb, Q_rows, num_nnz, set_gpu != 0); return BFH_OK; });
}
int bfh_bpr_set_placeholder(void* h, const int64_t* indptr, size_t batch_size) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->set_placeholder(indptr, batch_size); return BFH_OK; });
}
int bfh_bpr_set_cumulative_table(void* h, const int64_t* table) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->set_cumulative_table(table); return BFH_OK; });
}
int bfh_bpr_partial_update(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys, double* loss_sum, double* n_samples) {
    return guarded(h, [&] {
        double l = 0, n = 0;
        static_cast<BprHandle*>(h)->partial_update(start_x, next_x, indptr, keys, &l, &n);
        if (loss_sum) *loss_sum = l;
        if (n_samples) *n_samples = n;
        return BFH_OK;
    });
}
int bfh_bpr_update_parameters(void* h) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->update_parameters(); return BFH_OK; });
}
int bfh_bpr_synchronize(void* h, int device_to_host) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->synchronize(device_to_host != 0, device_to_host == 2); return BFH_OK; });
}
int bfh_bpr_compute_loss(void* h, int n, const int32_t* users, const int32_t* positives, const int32_t* negatives, double* loss) {
    return guarded(h, [&] { *loss = static_cast<BprHandle*>(h)->compute_loss(n, users, positives, negatives); return BFH_OK; });
}
int bfh_bpr_set_resident_csr(void* h, const int64_t* indptr, const int32_t* keys, int64_t nnz) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->set_resident_csr(indptr, keys, nnz); return BFH_OK; });
}
int bfh_bpr_set_mode(void* h, const char* name, int64_t value) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->set_mode(name ? name : "", value); return BFH_OK; });
}
int bfh_bpr_set_shard(void* h, int64_t nnz_offset, int num_shards) {
    return guarded(h, [&] {
        BFH_REQUIRE(num_shards >= 1 && nnz_offset >= 0, "set_shard: bad arguments");
        static_cast<BprHandle*>(h)->nnz_offset_ = nnz_offset;
        static_cast<BprHandle*>(h)->num_shards_ = num_shards;
        return BFH_OK;
    });
}
int bfh_bpr_update_triples(void* h, int64_t n, const int32_t* users, const int32_t* positives, const int32_t* negatives, double lr) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->update_triples(n, users, positives, negatives, lr); return BFH_OK; });
}
int bfh_bpr_device_buffer(void* h, const char* name, void** dptr, size_t* bytes) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->device_buffer(name ? name : "", dptr, bytes); return BFH_OK; });
}
void* bfh_bpr_stream(void* h) { return h ? static_cast<void*>(static_cast<BprHandle*>(h)->stream) : nullptr; }
int bfh_bpr_item_major_plan(int num_queues, const int64_t* queue_entries, int num_negative_samples, int64_t sync_updates, int* slice_len,
                            int64_t* segments, int64_t* queue_slices, int64_t* queue_stride) {
    if (num_queues < 1 || num_queues > bfh::kImMaxQueues || !queue_entries || num_negative_samples < 1 || !slice_len || !segments ||
        !queue_slices || !queue_stride)
        return BFH_ERR_INVALID;
    const bfh::ImPlan pl = bfh::im_make_plan(num_queues, queue_entries, num_negative_samples, sync_updates);
    *slice_len = pl.slice_len;
    *segments = pl.segments;
    for (int x = 0; x < num_queues; ++x) {
        queue_slices[x] = pl.q_slices[x];
        queue_stride[x] = pl.q_stride[x];
    }
    return BFH_OK;
}
int bfh_bpr_set_comm(void* h, void* comm) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->set_comm(static_cast<bfh::Comm*>(comm)); return BFH_OK; });
}
int bfh_bpr_comm_flush(void* h) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->exchange_finish(); static_cast<BprHandle*>(h)->sync_stream(); return BFH_OK; });
}
int bfh_bpr_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { *out = static_cast<BprHandle*>(h)->stats; return BFH_OK; });
}
int bfh_bpr_reset_stats(void* h) {
    return guarded(h, [&] { static_cast<BprHandle*>(h)->stats = bfh_stats{}; return BFH_OK; });
}

}  // extern "C"
