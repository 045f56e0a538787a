// BPRMF policy 3: the Hogwild SGD epoch walked ITEM-major, users owned by XCDs.
// (included by bpr.hip after the row helpers; same reference semantics: CBPRMF::worker,
// /root/reference/lib/algo_impl/bpr/bpr.cc:72-188)
//
// Why a second formulation.  The user-major kernel keeps P[u] in registers and pays for BOTH item
// rows of every triple in shared memory.  The positive item follows the popularity law: the 300
// most popular rows of the ML-20M-shaped matrix take a third of all positive updates, a single row
// 130 K per epoch.  Chip-wide fp32 atomics on such rows serialise (measured ~1.1 ns per row update
// on the head, 24 ns back to back on one row), XCD-private replicas of them diverge (thousands of
// independent updates summed at every merge) and racy stores lose most of the colliding updates
// (profiles/r01_xcd_study_pass*.json).  Walking the same triples grouped by their POSITIVE item
// turns the hot row into the register-resident one:
//
//   * entries (nnz positions) are split into one queue per XCD by `user % n_xcd` and sorted by item
//     inside a queue; a wave only pulls work from the queue of the XCD it runs on (HW_REG_XCC_ID),
//     so a user row is only ever touched through ONE L2: plain loads (sc1: past the L1) and plain
//     stores of P[u] are coherent without atomics;
//   * Q[i] of the slice's item lives in registers; its accumulated step goes to the chip-wide matrix
//     with one atomic row add every `flush_every[i]` triples (1 for the hottest rows .. 64), after
//     which the row is re-read -- the number of updates of a row that are in flight unseen by the
//     other waves stays below `im_max_stale` whatever the row's popularity;
//   * Q[j] (uniformly drawn negatives: no popular rows unless the catalogue is tiny) uses the per-XCD
//     replicas of policy 2: plain read-modify-write through the XCD's L2, reconciled by the delta
//     rule at segment boundaries; rows the collision rule marks hot -- and heavy users' P rows -- are
//     updated with atomics on the chip-wide copy instead;
//   * slices are handed out through a per-queue ticket in a golden-ratio order, so the waves that run
//     at the same time work on different items.
//
// A launch is followed by a `drain` launch in which any wave may take any ticket that is left (a
// queue whose XCD received no workgroups, e.g. a tiny grid) and performs every update with atomics
// on the chip-wide copies: completeness never depends on where the hardware places a workgroup.
#pragma once

namespace bfh {

constexpr int kImMaxQueues = 8;

struct ImQueues {
    const uint32_t* ent_key;   // [n] queue * Q_rows + item, sorted
    const int32_t* ent_pos;    // [n] chunk-local nnz position of the entry
    int nq;                    // number of queues (= XCDs seen by the probe)
    int xcd_queue[16];         // HW_REG_XCC_ID -> queue (-1: not seen by the probe)
    int64_t q_beg[kImMaxQueues];      // first entry of a queue
    int64_t q_triples[kImMaxQueues];  // (entries of the queue) * num_neg
    int slice_len;                    // triples per slice: the largest multiple of num_neg <= 64, so that the slots of one
                                      // entry (which share the user row) are never split between two waves
    int64_t q_slices[kImMaxQueues];   // ceil(q_triples / slice_len)
    int64_t q_stride[kImMaxQueues];   // slice = (ticket * stride) % q_slices, gcd(stride, q_slices) == 1
    int64_t t_beg[kImMaxQueues], t_end[kImMaxQueues];   // ticket range of this launch
    int* tickets;              // [nq] tickets handed out so far in this launch's range
    unsigned long long* done;  // triples processed (checked by the host)
    const uint8_t* hot_user;   // [P_rows] 0: plain loads / stores on P[u] (one owner XCD); 1: fp32 atomics on P[u]; 2: plain on this XCD's replica of P[u]
    const uint8_t* flush_every;  // [Q_rows] triples between two flushes of the register-resident item row (1..64)
    const int32_t* neg_pre;    // [chunk nnz * num_neg] negatives drawn by bpr_presample_kernel, or null: draw in the walk
    float* rep_P;              // null: a user's entries all sit in the queue of ONE XCD, which alone touches P[u].  Otherwise
                               // [nq][P_rows * vdim] per-XCD replicas of P: entries are spread over the queues by position and
                               // a wave works on its XCD's copy (small shards: see launch_item_major)
    int64_t rep_pstride;
    int p_nt;                  // P rows are read / written with the non-temporal hint (they are streamed once per triple; study knob)
    int study;                 // measurement knob: bit 0 = the chip-wide atomics of the negatives' rows are NOT issued (wrong results: timing only)
    int strict;                // test hook: wait for every memory operation of a triple before the next one starts
    int32_t* trace;            // test hook (single-wave runs): sigmoid-table index of every triple in processing order, or null
};

// ------------------------------------------------------------------------------------------------
// The slice schedule (host side; also exported as bfh_bpr_item_major_plan so that it can be checked
// without a GPU): slice length, merge segments, and per queue the number of slices and the stride of
// the visiting order.  Ticket t of queue x works on slice (t * stride[x]) mod slices[x] -- a permutation
// because gcd(stride, slices) == 1; segment s of S takes the tickets [slices*s/S, slices*(s+1)/S).
// ------------------------------------------------------------------------------------------------
struct ImPlan {
    int nq = 0, slice_len = 64;
    int64_t segments = 1;
    int64_t q_triples[kImMaxQueues] = {0}, q_slices[kImMaxQueues] = {0}, q_stride[kImMaxQueues] = {0};
};

static inline int64_t im_gcd(int64_t a, int64_t b) {
    while (b) { const int64_t t = a % b; a = b; b = t; }
    return a;
}

static inline ImPlan im_make_plan(int nq, const int64_t* q_entries, int num_neg, int64_t sync_updates) {
    ImPlan pl;
    pl.nq = nq;
    // the largest multiple of num_neg that fits a wave: the slots of one entry share the user row
    pl.slice_len = num_neg <= 64 ? (64 / num_neg) * num_neg : 64;
    int64_t total = 0;
    for (int x = 0; x < nq; ++x) {
        pl.q_triples[x] = q_entries[x] * num_neg;
        total += pl.q_triples[x];
        pl.q_slices[x] = (pl.q_triples[x] + pl.slice_len - 1) / pl.slice_len;
        // golden-ratio order: consecutive tickets land ~0.618 of the queue apart, so the waves that run at the same
        // time work on different items; the stride is bumped until it is coprime with the number of slices
        int64_t st = static_cast<int64_t>(static_cast<double>(pl.q_slices[x]) * 0.6180339887498949) | 1;
        while (pl.q_slices[x] > 1 && im_gcd(st, pl.q_slices[x]) != 1) st += 2;
        pl.q_stride[x] = pl.q_slices[x] > 1 ? st % pl.q_slices[x] : 1;
        if (pl.q_stride[x] == 0) pl.q_stride[x] = 1;
    }
    pl.segments = sync_updates > 0 ? (total + sync_updates / 2) / sync_updates : 1;
    if (pl.segments < 1) pl.segments = 1;
    return pl;
}
static inline void im_segment_tickets(const ImPlan& pl, int x, int64_t sgm, int64_t* t_beg, int64_t* t_end) {
    *t_beg = pl.q_slices[x] * sgm / pl.segments;
    *t_end = pl.q_slices[x] * (sgm + 1) / pl.segments;
}

__device__ __forceinline__ int xcc_id_raw() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return static_cast<int>(x & 15u);
}

__global__ void xcd_probe_kernel(int* seen) {
    if (threadIdx.x == 0) atomicOr(seen + xcc_id_raw(), 1);
}

// sort key of every entry: ((owner queue of the user) * blocks + block) * Q_rows + item.  `blocks` > 1 cuts an
// item's entries inside a queue into that many runs (by a hash of the nnz position), visited at different times.
// `spread`: the queue is a hash of the position instead of the user's owner (per-XCD replicas of P, ImQueues::rep_P).
// `spread` = 2: only the entries of HEAVY users (degree >= heavy_deg: the users the collision rule would put on atomics) are spread
// over the queues -- they get per-XCD replicas of their row, everybody else keeps one owner XCD.
__global__ __launch_bounds__(256) void im_keys_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ keys, int64_t n, int nq,
                                                      uint32_t blocks, uint32_t q_rows, int spread, const int64_t* __restrict__ indptr,
                                                      int64_t heavy_deg, uint32_t* __restrict__ kout, int32_t* __restrict__ vout) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t blk = blocks > 1 ? ((static_cast<uint32_t>(t) * 2654435761u) >> 16) % blocks : 0u;
    const int u = rows[t];
    const uint32_t h = (static_cast<uint32_t>(t) * 0x85EBCA6Bu) >> 11;
    uint32_t queue = static_cast<uint32_t>(u % nq);
    if (spread == 1) {
        queue = h % static_cast<uint32_t>(nq);
    } else if (spread >= 2) {
        const int64_t deg = indptr[u] - (u ? indptr[u - 1] : 0);
        if (deg >= heavy_deg) {
            // spread == 3: only over as many queues (the owner's and its neighbours) as bring the user's share of one queue under
            // the threshold -- a user just above it gets two replicas in use, not eight, and the rows an XCD keeps hot stay few
            uint32_t r = static_cast<uint32_t>(nq);
            if (spread == 3) {
                r = 2;
                while (r < static_cast<uint32_t>(nq) && deg >= heavy_deg * static_cast<int64_t>(r)) r <<= 1;
                if (r > static_cast<uint32_t>(nq)) r = static_cast<uint32_t>(nq);
            }
            queue = (queue + h % r) % static_cast<uint32_t>(nq);
        }
    }
    kout[t] = (queue * blocks + blk) * q_rows + static_cast<uint32_t>(keys[t]);
    vout[t] = static_cast<int32_t>(t);
}

__global__ void im_bounds_kernel(const uint32_t* __restrict__ sorted, int64_t n, int nq, uint32_t queue_span, int64_t* __restrict__ q_beg) {
    const int x = threadIdx.x;
    if (x > nq) return;
    q_beg[x] = x == nq ? n : lower_bound_dev<uint32_t>(sorted, n, static_cast<uint32_t>(x) * queue_span);
}

// per-row policy flags.  `inflight` = item (or user) rows a queue's waves hold between load and store.
//   hot_item[i]    : the NEGATIVE updates of row i go to the chip-wide row with atomics (and read it) when
//                    (a) P(negative == i) * inflight >= tau: racing plain stores would lose that share of them, or
//                    (b) the row's positive steps between two merges, weighted by the learning rate, reach
//                        `drift_budget`: a replica does not see the flushes of the register-resident copies until the
//                        next merge, and a negative step computed against a row that has since moved by O(1) closes
//                        the feedback loop one merge late (measured: a popular catalogue head diverges at lr 0.05);
//   flush_every[i] : max_stale / (waves working on item i at once) clamped to [1, 64]
//   hot_user[u]    : (share of the queue's triples with user u) * inflight >= tau
__global__ void im_item_flags_kernel(const int* __restrict__ cnt, const int64_t* __restrict__ cum, int64_t cum_total, int rows, double pos_triples,
                                     double total_triples, double neg_uniform, double inflight, double tau, double waves, double max_stale,
                                     double lr_steps_per_count, double drift_budget, uint8_t* __restrict__ hot_item,
                                     uint8_t* __restrict__ flush_every) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double pneg = neg_uniform;
    if (cum) pneg = static_cast<double>(cum[i] - (i ? cum[i - 1] : 0)) / static_cast<double>(cum_total);
    const bool collide = tau > 0.0 && pneg * inflight >= tau;
    const bool drift = drift_budget > 0.0 && cnt[i] * lr_steps_per_count >= drift_budget;
    hot_item[i] = (collide || drift) ? 1 : 0;
    const double conc = cnt[i] * pos_triples / total_triples * waves;   // waves inside item i's entries at any time
    double f = conc > 0.0 ? max_stale / conc : 64.0;
    f = f < 1.0 ? 1.0 : (f > 64.0 ? 64.0 : f);
    flush_every[i] = static_cast<uint8_t>(f);
}

// mode 0: every user has one owner XCD -- flag 1 (atomics) where the collision rule fires at the owner queue's share, else 0;
// mode 1: every user is replicated per XCD (its entries are spread over the queues: 1/nq of the share) -- 1 where the rule still
//         fires, else 2 (replica);
// mode 2: only the users with degree >= heavy_deg (= where the owner-share rule fires; the threshold the keys were built with) are
//         replicated: they get 1 / 2 by the spread-share rule, everybody else 0.
__global__ void im_user_flags_kernel(const int64_t* __restrict__ indptr, int first_row, int rows, double num_neg, double owner_queue_triples,
                                     double all_triples, double inflight, double tau, int mode, int64_t heavy_deg, uint8_t* __restrict__ hot_user) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= rows) return;
    const int g = first_row + u;
    const int64_t degi = indptr[g] - (g ? indptr[g - 1] : 0);
    const double deg = static_cast<double>(degi);
    const bool hot_owner = tau > 0.0 && deg * num_neg / owner_queue_triples * inflight >= tau;
    const bool hot_spread = tau > 0.0 && deg * num_neg / all_triples * inflight >= tau;
    uint8_t f = 0;
    if (mode == 0) f = hot_owner ? 1 : 0;
    else if (mode == 1) f = hot_spread ? 1 : 2;
    else f = degi >= heavy_deg ? (hot_spread ? 1 : 2) : 0;
    hot_user[g] = f;
}

// float4-per-lane registers -> dword-per-lane order (element k*64 + lane), so that one atomic
// instruction covers whole 128-B lines: the atomic units charge per line touched, and a strided
// float4 row would touch every line of the row four times.
// The negatives of a whole call, drawn in CSR order before the item-major walk: neighbouring threads share a
// user, so the rejection test's binary search runs in cached key runs -- inside the item-major kernel every lane
// of a slice has a different user and the same search costs ~6 GB of scattered sector reads per ML-20M epoch.
// Same draw as bpr_update_kernel: a pure function of (seed, global nnz position, slot, epoch, attempt).
__global__ __launch_bounds__(256) void bpr_presample_kernel(SgdParams p, BprConsts c, int32_t* __restrict__ neg_out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= c.total) return;
    const int64_t pos_idx = t / c.num_neg;
    const uint32_t slot = static_cast<uint32_t>(t % c.num_neg);
    const int u = p.rows[pos_idx];
    const int64_t ubeg = (u == 0 ? 0 : p.indptr[u - 1]) - p.shift;
    const int64_t uend = p.indptr[u] - p.shift;
    neg_out[t] = bpr_sample_negative(p, c, static_cast<uint64_t>(p.nnz_offset + p.shift + pos_idx), slot, ubeg, uend);
}

template <int K>
__device__ __forceinline__ void row_atomic_add_full_lines(const Row<K>& r, float* __restrict__ base, int lane, int vdim) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k * 64 < vdim) {
            const int src = ((k & 3) * 16 + (lane >> 2)) * 4;   // byte address of the source lane
            const int kv = (k >> 2) * 4;
            const int c0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 0]));
            const int c1 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 1]));
            const int c2 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 2]));
            const int c3 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 3]));
            const int sel = lane & 3;
            const int v = sel == 0 ? c0 : (sel == 1 ? c1 : (sel == 2 ? c2 : c3));
            const int e = k * 64 + lane;
            if (e < vdim) atomic_add_f32(base + e, __builtin_bit_cast(float, v));
        }
    }
}

// The same add with the pre-add values returned: `now` = the row right after this wave's add, in the caller's
// float4-per-lane layout -- one instruction instead of an add and a re-read, and no reliance on the order in which
// the memory side performs an atomic and a later load of other lanes (scripts/micro/atomic_then_load.hip shows the
// same-lane case is ordered on gfx950; the row's elements change lanes between the two layouts).
template <int K>
__device__ __forceinline__ void row_atomic_add_full_lines_fetch(const Row<K>& r, float* __restrict__ base, int lane, int vdim, Row<K>& now) {
    float nw[K];   // dword-per-lane order
#pragma unroll
    for (int k = 0; k < K; ++k) {
        nw[k] = 0.f;
        if (k * 64 < vdim) {
            const int src = ((k & 3) * 16 + (lane >> 2)) * 4;
            const int kv = (k >> 2) * 4;
            const int c0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 0]));
            const int c1 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 1]));
            const int c2 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 2]));
            const int c3 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r.v[kv + 3]));
            const int sel = lane & 3;
            const float v = __builtin_bit_cast(float, sel == 0 ? c0 : (sel == 1 ? c1 : (sel == 2 ? c2 : c3)));
            const int e = k * 64 + lane;
            if (e < vdim) nw[k] = __hip_atomic_fetch_add(base + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + v;
        }
    }
    // back to float4-per-lane: register kv*4+c of lane l holds element (kv*64 + l)*4 + c
    //   = dword register kv*4 + (l >> 4), lane (l & 15)*4 + c
#pragma unroll
    for (int rr = 0; rr < K; ++rr) {
        const int kv = rr >> 2, cc = rr & 3;
        const int src = ((lane & 15) * 4 + cc) * 4;
        const int ks = lane >> 4;
        int got = 0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = kv * 4 + s4;
            if (k < K && k * 64 < vdim) {
                const int g = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, nw[k]));
                if (ks == s4) got = g;
            }
        }
        const int e = (kv * 64 + lane) * 4 + cc;
        now.v[rr] = e < vdim ? __builtin_bit_cast(float, got) : 0.f;
    }
}

// DRAIN: the clean-up launch (any wave takes any ticket that is left, every update an atomic on the chip-wide copies)
template <int K, bool PIPE, bool DRAIN>
__global__ __launch_bounds__(256, K <= 4 ? (PIPE ? 5 : 6) : 1) void bpr_item_major_kernel(SgdParams p, BprConsts c, ImQueues q) {
    const int lane = threadIdx.x & 63;
    const int vdim = p.vdim;
    const int my_queue = q.xcd_queue[xcc_id_raw()];
    constexpr bool drain = DRAIN;
    float* const Qrep = drain ? p.Q : c.rep_Q + static_cast<size_t>(my_queue < 0 ? 0 : my_queue) * c.rep_stride;
    float* const Qbrep = drain ? p.Qb : c.rep_Qb + static_cast<size_t>(my_queue < 0 ? 0 : my_queue) * c.rep_bstride;
    float* const Prep = (drain || !q.rep_P) ? p.P : q.rep_P + static_cast<size_t>(my_queue < 0 ? 0 : my_queue) * q.rep_pstride;
    auto rload = [&](Row<K>& r, const float* base) { row_load<K, true, true>(r, base, lane, vdim); };
    auto rstore = [&](const Row<K>& r, float* base) { row_store<K, true, false>(r, base, lane, vdim); };
    auto pload = [&](Row<K>& r, const float* base) {
        if (q.p_nt) row_load_nt<K>(r, base, lane, vdim);
        else row_load<K, true, true>(r, base, lane, vdim);
    };
    auto pstore = [&](const Row<K>& r, float* base) {
        if (q.p_nt) row_store_nt<K>(r, base, lane, vdim);
        else row_store<K, true, false>(r, base, lane, vdim);
    };

    int cur_i = -1, since_flush = 0, flush_n = 64;
    Row<K> qi, dqi, qi_re;   // the slice's item row, its step since the last flush, the row as re-read after a flush
    float bi = 0.f, dbi_acc = 0.f, bi_re = 0.f;
    bool re_pending = false;
#pragma unroll
    for (int k = 0; k < K; ++k) { qi.v[k] = 0.f; dqi.v[k] = 0.f; qi_re.v[k] = 0.f; }
    double loss = 0.0;
    unsigned long long processed = 0;

    // push the accumulated step of the register-resident row to the chip-wide matrix.  `reload` (mid-run): the add
    // returns what the row held, so `qi_re` = the row with every wave's steps up to and including this flush; nobody
    // waits for it -- the next triple runs on the local copy and folds `qi_re` in when it has arrived.
    auto flush_item = [&](bool reload) {
        if (cur_i < 0) return;
        float* Qi = p.Q + static_cast<size_t>(cur_i) * vdim;
        re_pending = false;
        if (reload && c.update_i) {
            row_atomic_add_full_lines_fetch<K>(dqi, Qi, lane, vdim, qi_re);
            if (c.use_bias) {
                float now = 0.f;
                if (lane == 0) now = __hip_atomic_fetch_add(p.Qb + cur_i, dbi_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + dbi_acc;
                bi_re = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, now)));
            }
            re_pending = true;
        } else if (c.update_i) {
            row_atomic_add_full_lines<K>(dqi, Qi, lane, vdim);
            if (c.use_bias && lane == 0) atomic_add_f32(p.Qb + cur_i, dbi_acc);
        } else if (reload) {   // frozen positives (update_i = false): nothing to push, just refresh the row
            rload(qi_re, Qi);
            if (c.use_bias) bi_re = coh_load(p.Qb + cur_i);
            re_pending = true;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) dqi.v[k] = 0.f;
        dbi_acc = 0.f;
        since_flush = 0;
    };

    for (int qq = 0; qq < q.nq; ++qq) {
        if (!drain && qq != my_queue) continue;
        const int64_t n_tickets = q.t_end[qq] - q.t_beg[qq];
        if (n_tickets <= 0) continue;
        for (;;) {
            int64_t tk = 0;
            if (lane == 0) {
                // drain: look before taking a ticket (thousands of waves find nothing left)
                const int seen = drain ? __hip_atomic_load(q.tickets + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                tk = seen >= n_tickets ? n_tickets : static_cast<int64_t>(atomicAdd(q.tickets + qq, 1));
            }
            tk = __builtin_amdgcn_readfirstlane(static_cast<int>(tk));
            if (tk >= n_tickets) break;
            const int64_t slice = static_cast<int64_t>((static_cast<unsigned long long>(q.t_beg[qq] + tk) * static_cast<unsigned long long>(q.q_stride[qq])) %
                                                       static_cast<unsigned long long>(q.q_slices[qq]));
            const int64_t t0 = slice * q.slice_len;
            const int n_here = static_cast<int>((q.q_triples[qq] - t0) < q.slice_len ? (q.q_triples[qq] - t0) : q.slice_len);
            // ---------------- lane-parallel: entry -> (item, user), sample the negative ----------------
            int my_u = -1, my_item = -1, my_neg = -1, my_pol = 0;   // bit0: P[u] atomic, bit1: Q[neg] atomic on the chip-wide row
            if (lane < n_here) {
                const int64_t t = t0 + lane;
                const int64_t e = q.q_beg[qq] + t / c.num_neg;
                const uint32_t slot = static_cast<uint32_t>(t % c.num_neg);
                my_item = static_cast<int>(q.ent_key[e] % static_cast<uint32_t>(p.Q_rows));
                const int64_t pos_idx = q.ent_pos[e];
                my_u = p.rows[pos_idx];
                if (q.neg_pre) {
                    my_neg = q.neg_pre[pos_idx * c.num_neg + slot];
                } else {
                    const int64_t ubeg = (my_u == 0 ? 0 : p.indptr[my_u - 1]) - p.shift;
                    const int64_t uend = p.indptr[my_u] - p.shift;
                    my_neg = bpr_sample_negative(p, c, static_cast<uint64_t>(p.nnz_offset + p.shift + pos_idx), slot, ubeg, uend);
                }
                const int fu = q.hot_user[my_u];   // bit 0: atomics on P[u]; bit 2: this XCD's replica of P[u]; bit 1: the negative's row is chip-wide
                my_pol = drain ? 3 : ((fu == 1 ? 1 : 0) | (c.hot[my_neg] ? 2 : 0) | (fu == 2 ? 4 : 0));
            }
            auto pu_ptr = [&](int u, int pol) -> float* { return ((pol & 4) ? Prep : p.P) + static_cast<size_t>(u) * vdim; };
            auto qj_ptr = [&](int j, bool hot) -> float* { return (hot ? p.Q : Qrep) + static_cast<size_t>(j) * vdim; };
            auto bj_ptr = [&](int j, bool hot) -> float* { return (hot ? p.Qb : Qbrep) + j; };

            // the two per-triple rows are fetched two triples ahead into the slots A (even triples) and B (odd):
            // a wave keeps four rows in flight, which is what random 512-B rows need to cover the HBM latency
            struct Slot { Row<K> pu, qj; float bj; };
            Slot A, B;
            auto fetch = [&](Slot& s, int j) {
                if (j < n_here) {
                    const int u = __builtin_amdgcn_readlane(my_u, j), ng = __builtin_amdgcn_readlane(my_neg, j);
                    const int pl = __builtin_amdgcn_readlane(my_pol, j);
                    const bool hj = (pl & 2) != 0;
                    pload(s.pu, pu_ptr(u, pl));
                    rload(s.qj, qj_ptr(ng, hj));
                    s.bj = c.use_bias ? coh_load(bj_ptr(ng, hj)) : 0.f;
                }
            };
            Row<K> pu, qj;
            float bj = 0.f;
            int prev_u = -1, prev_neg = -1, prev2_u = -1, prev2_neg = -1;
            bool prev_hj = false;

            auto step = [&](Slot& s, int j) {
                const int item = __builtin_amdgcn_readlane(my_item, j);
                const int u = __builtin_amdgcn_readlane(my_u, j);
                const int neg = __builtin_amdgcn_readlane(my_neg, j);
                const int pol = __builtin_amdgcn_readlane(my_pol, j);
                const bool at_u = (pol & 1) != 0, at_j = (pol & 2) != 0;
                // take the prefetched rows -- unless this wave itself changed the row after the prefetch was issued.
                // !PIPE: no prefetch at all, the rows are read here and written back a few hundred cycles later (the
                // narrowest window another wave's update of the same row can fall into; latency is covered by occupancy)
                if (u == prev_u) {
                    // consecutive slots of one entry (num_negative_samples > 1): carry the updated row
                } else if (!PIPE || u == prev2_u) {
                    pload(pu, pu_ptr(u, pol));
                } else {
                    pu = s.pu;
                }
                if (neg == prev_neg && at_j == prev_hj) {
                    // the same negative twice in a row: carry
                } else if (!PIPE || neg == prev2_neg) {
                    rload(qj, qj_ptr(neg, at_j));
                    if (c.use_bias) bj = coh_load(bj_ptr(neg, at_j));
                } else {
                    qj = s.qj;
                    bj = s.bj;
                }
                if (PIPE) fetch(s, j + 2);
                if (item != cur_i) {
                    flush_item(false);
                    cur_i = item;
                    flush_n = drain ? 1 : q.flush_every[item];
                    rload(qi, p.Q + static_cast<size_t>(item) * vdim);
                    if (c.use_bias) bi = coh_load(p.Qb + item);
                }
                // ---------------- score + sigmoid table (bpr.cc:119-131) ----------------
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) part += pu.v[k] * (qi.v[k] - qj.v[k]);
                float x = wave_sum(part);
                if (c.use_bias) x += (bi - bj);
                const float logit = bpr_logit(x, c.exp_table);
                if (c.compute_loss) loss += static_cast<double>(log1pf(__expf(-fminf(fmaxf(x, -6.f), 6.f))));
                if (q.trace && lane == 0) q.trace[processed] = 6.0f < x ? 1000 : (x < -6.0f ? -1 : static_cast<int>((x + 6.0f) * 83.0f));
                // ---------------- bpr.cc:157-171 (Q-1: the user step sees the updated item rows) ----------------
                const bool same = item == neg;   // verify_neg == false only: the one row takes both steps in turn
                Row<K> dj, dpu;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float idv = logit * pu.v[k];
                    const float di = c.update_i ? c.lr * (idv - c.reg_i * qi.v[k]) : 0.f;
                    qi.v[k] += di;
                    dqi.v[k] += di;
                    if (same) qj.v[k] = qi.v[k];
                    dj.v[k] = c.update_j ? c.lr * (-idv - c.reg_j * qj.v[k]) : 0.f;
                    qj.v[k] += dj.v[k];
                    if (same) { qi.v[k] = qj.v[k]; dqi.v[k] += dj.v[k]; }
                    dpu.v[k] = c.lr * (logit * (qi.v[k] - qj.v[k]) - c.reg_u * pu.v[k]);
                    pu.v[k] += dpu.v[k];
                }
                float dbj = 0.f;
                if (c.use_bias) {
                    const float bi_new = c.update_i ? bias_step(bi, logit, c.lr_d, c.reg_b_d) : bi;   // bpr.cc:162 in double
                    dbi_acc += bi_new - bi;
                    bi = bi_new;
                    if (same) bj = bi;
                    const float bj_new = c.update_j ? bias_step(bj, -logit, c.lr_d, c.reg_b_d) : bj;  // bpr.cc:168
                    dbj = bj_new - bj;
                    bj = bj_new;
                    if (same) { bi = bj; dbi_acc += dbj; }
                }
                // ---------------- write the two per-triple rows back ----------------
                float* Pu = pu_ptr(u, pol);
                float* Qj = qj_ptr(neg, at_j);
                const bool fr_u = PIPE && c.fresh && !at_u, fr_j = PIPE && c.fresh && !at_j && !same && c.update_j;
                Row<K> fu, fj;
                if (fr_u) pload(fu, Pu);
                if (fr_j) rload(fj, Qj);
                if (at_u) {
                    row_atomic_add_full_lines<K>(dpu, Pu, lane, vdim);
                } else if (fr_u) {
#pragma unroll
                    for (int k = 0; k < K; ++k) fu.v[k] += dpu.v[k];
                    pstore(fu, Pu);
                } else {
                    pstore(pu, Pu);
                }
                if (c.update_j && !same) {
                    if (at_j) {
                        row_atomic_add_full_lines<K>(dj, Qj, lane, vdim);
                    } else if (fr_j) {
#pragma unroll
                        for (int k = 0; k < K; ++k) fj.v[k] += dj.v[k];
                        rstore(fj, Qj);
                    } else {
                        rstore(qj, Qj);
                    }
                    if (c.use_bias && lane == 0) {
                        float* Bj = bj_ptr(neg, at_j);
                        if (at_j) atomic_add_f32(Bj, dbj);
                        else *Bj = bj;
                    }
                }
                // test hook (one wave, deterministic order): every store / atomic of this triple is acknowledged before
                // the next triple reads anything, so the run is sequential in the strict sense
                if (q.strict) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                processed += 1;
                since_flush += 1;
                // a row re-read behind the previous flush has arrived by now: it carries the other waves' steps
                if (re_pending) {
#pragma unroll
                    for (int k = 0; k < K; ++k) qi.v[k] = qi_re.v[k] + dqi.v[k];
                    if (c.use_bias) bi = bi_re + dbi_acc;
                    re_pending = false;
                }
                const bool item_goes_on = j + 1 < n_here && __builtin_amdgcn_readlane(my_item, j + 1) == item;
                if (since_flush >= flush_n && item_goes_on) flush_item(true);
                prev2_u = prev_u; prev2_neg = prev_neg;
                prev_u = u; prev_neg = neg; prev_hj = at_j;
            };

            if (PIPE) {
                fetch(A, 0);
                fetch(B, 1);
            }
            for (int j = 0; j < n_here; j += 2) {
                step(A, j);
                if (j + 1 < n_here) step(B, j + 1);
            }
            flush_item(false);
            cur_i = -1;
        }
    }
    if (lane == 0 && processed) atomicAdd(q.done, processed);
    if (c.compute_loss && lane == 0 && loss != 0.0) atomicAdd(c.loss_out, loss);
}

// ------------------------------------------------------------------------------------------------
// Two triples per wave (vdim <= 128).  The walk above keeps one triple in flight per wave and spends ~2.5 us on each
// (a dependent chain of one fabric round trip, a wave reduction, a table lookup and the stores) with half of the lanes
// idle at d = 128.  Here the two half-waves walk two different slices side by side: a row is one dword per lane of the
// half (element k * 32 + lane32, k < 4: every load / store / atomic instruction covers one full 128-B line per half, so
// the flush needs no lane permutation), everything that is wave-uniform above becomes half-uniform (held per lane), the
// control flow diverges by half where the two slices differ (new item, hot row, flush), and the only cross-lane steps --
// the dot product's reduction and the metadata broadcast -- run for both halves at once.  Same schedule, same flush /
// replica / hot-row rules, same sampler; the drain launch and the test hooks stay with the kernel above.
// ------------------------------------------------------------------------------------------------
struct HRow { float v[4]; };

__device__ __forceinline__ float half_sum(float v, int half) {   // sum over the 32 lanes of each half; every lane gets its half's total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    // every lane holds its 16-lane row's sum; the half's total = own row + the other row of the half (lane ^ 16: ds_swizzle, bit mode,
    // xor mask 0x10 -- no memory access).  r0 + r1 in row 0, r1 + r0 in row 1: the same number (round 5 used four readlanes and a select)
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
    (void)half;
    return v + o;
}
__device__ __forceinline__ void hrow_load(HRow& r, const float* rp, int nk) {   // rp = row + lane32; past the L1 (sc1)
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = k < nk ? __hip_atomic_load(rp + k * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
}
__device__ __forceinline__ void hrow_store(const HRow& r, float* rp, int nk) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nk) rp[k * 32] = r.v[k];
}
__device__ __forceinline__ void hrow_atomic_add(const HRow& r, float* rp, int nk) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nk) atomic_add_f32(rp + k * 32, r.v[k]);
}

// NK > 0: vdim == 32 * NK -- every lane holds NK elements of a row and no load / store / atomic needs a per-lane guard (the guards cost an
// exec-mask branch per memory instruction: round 6's instruction diet); NK == 0: any vdim <= 128.
template <int NK>
__global__ __launch_bounds__(256, 5) void bpr_item_major_dual_kernel(SgdParams p, BprConsts c, ImQueues q) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int vdim = p.vdim;
    // elements this lane holds: k * 32 + l32 < vdim
    const int nk = NK > 0 ? NK : (l32 < vdim ? (vdim - l32 + 31) / 32 : 0);
    // the sigmoid table (bpr.cc:119-131) in LDS: its lookup sits in every triple's dependent chain
    __shared__ float s_exp[1000];
    for (int t = threadIdx.x; t < 1000; t += 256) s_exp[t] = c.exp_table[t];
    __syncthreads();
    const int qq = q.xcd_queue[xcc_id_raw()];
    if (qq < 0) return;   // a workgroup on an XCD the probe did not see: the drain launch covers whatever is left
    float* const Qrep = c.rep_Q + static_cast<size_t>(qq) * c.rep_stride;
    float* const Qbrep = c.rep_Qb + static_cast<size_t>(qq) * c.rep_bstride;
    float* const Prep = q.rep_P ? q.rep_P + static_cast<size_t>(qq) * q.rep_pstride : p.P;
    const int64_t n_tickets = q.t_end[qq] - q.t_beg[qq];
    if (n_tickets <= 0) return;

    // per half (the same value in its 32 lanes)
    int cur_i = -1, since_flush = 0, flush_n = 64;
    HRow qi, dqi, qi_re, pu, qj;
    float bi = 0.f, dbi_acc = 0.f, bi_re = 0.f, bj = 0.f;
    bool re_pending = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { qi.v[k] = 0.f; dqi.v[k] = 0.f; qi_re.v[k] = 0.f; pu.v[k] = 0.f; qj.v[k] = 0.f; }
    double loss = 0.0;
    unsigned long long processed = 0;

    auto flush_item = [&](bool reload) {   // see bpr_item_major_kernel: one atomic row add, `reload` = the add returns the row
        if (cur_i < 0) return;
        float* Qi = p.Q + static_cast<size_t>(cur_i) * vdim + l32;
        re_pending = false;
        if (reload && c.update_i) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                qi_re.v[k] = k < nk ? __hip_atomic_fetch_add(Qi + k * 32, dqi.v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + dqi.v[k] : 0.f;
            if (c.use_bias) {
                // lane32 0 adds; every lane of the half needs the result: all of them read the row's bias after the add is
                // issued would race with it, so the adding lane's value is spread with a half-wide readlane below
                float now = 0.f;
                if (l32 == 0) now = __hip_atomic_fetch_add(p.Qb + cur_i, dbi_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + dbi_acc;
                bi_re = now;   // valid in lane32 0 only; broadcast by the caller (uniform control flow)
            }
            re_pending = true;
        } else if (c.update_i) {
            hrow_atomic_add(dqi, Qi, nk);
            if (c.use_bias && l32 == 0) atomic_add_f32(p.Qb + cur_i, dbi_acc);
        } else if (reload) {   // frozen positives: nothing to push, just refresh the row
            hrow_load(qi_re, Qi, nk);
            if (c.use_bias) bi_re = coh_load(p.Qb + cur_i);
            re_pending = true;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dqi.v[k] = 0.f;
        dbi_acc = 0.f;
        since_flush = 0;
    };

    for (;;) {
        // ---- two tickets: slice A for half 0, slice B for half 1 ----
        int64_t tk = 0;
        if (lane == 0) tk = static_cast<int64_t>(atomicAdd(q.tickets + qq, 2));
        tk = __builtin_amdgcn_readfirstlane(static_cast<int>(tk));
        if (tk >= n_tickets) break;
        int n_sl[2];
        int it_[2], u_[2], ng_[2], pl_[2];   // metadata of triple `lane` of slice A / B
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            it_[sidx] = -1; u_[sidx] = -1; ng_[sidx] = -1; pl_[sidx] = 0;
            n_sl[sidx] = 0;
            if (tk + sidx >= n_tickets) continue;
            const int64_t slice = static_cast<int64_t>((static_cast<unsigned long long>(q.t_beg[qq] + tk + sidx) * static_cast<unsigned long long>(q.q_stride[qq])) %
                                                       static_cast<unsigned long long>(q.q_slices[qq]));
            const int64_t t0 = slice * q.slice_len;
            const int n_here = static_cast<int>((q.q_triples[qq] - t0) < q.slice_len ? (q.q_triples[qq] - t0) : q.slice_len);
            n_sl[sidx] = n_here;
            if (lane < n_here) {
                const int64_t t = t0 + lane;
                const int64_t e = q.q_beg[qq] + t / c.num_neg;
                const uint32_t slot = static_cast<uint32_t>(t % c.num_neg);
                it_[sidx] = static_cast<int>(q.ent_key[e] % static_cast<uint32_t>(p.Q_rows));
                const int64_t pos_idx = q.ent_pos[e];
                u_[sidx] = p.rows[pos_idx];
                if (q.neg_pre) {
                    ng_[sidx] = q.neg_pre[pos_idx * c.num_neg + slot];
                } else {
                    const int64_t ubeg = (u_[sidx] == 0 ? 0 : p.indptr[u_[sidx] - 1]) - p.shift;
                    const int64_t uend = p.indptr[u_[sidx]] - p.shift;
                    ng_[sidx] = bpr_sample_negative(p, c, static_cast<uint64_t>(p.nnz_offset + p.shift + pos_idx), slot, ubeg, uend);
                }
                const int fu = q.hot_user[u_[sidx]];
                pl_[sidx] = (fu == 1 ? 1 : 0) | (c.hot[ng_[sidx]] ? 2 : 0) | (fu == 2 ? 4 : 0);
            }
        }
        const int n_mine = half ? n_sl[1] : n_sl[0];
        const int n_max = n_sl[0] > n_sl[1] ? n_sl[0] : n_sl[1];
        int prev_u = -1, prev_neg = -1;
        bool prev_hj = false;

        for (int j = 0; j < n_max; ++j) {
            // half-uniform metadata of this step (readlanes for both halves, one select)
            // (both halves' values are read into scalar registers first: written as `half ? readlane(b) : readlane(a)` the compiler
            // turns every field into a divergent if / else around the two readlanes -- 15 instructions and two branches per field)
            const int jn = j + 1 < 64 ? j + 1 : 63;
            const int it0 = __builtin_amdgcn_readlane(it_[0], j), it1 = __builtin_amdgcn_readlane(it_[1], j);
            const int us0 = __builtin_amdgcn_readlane(u_[0], j), us1 = __builtin_amdgcn_readlane(u_[1], j);
            const int ng0 = __builtin_amdgcn_readlane(ng_[0], j), ng1 = __builtin_amdgcn_readlane(ng_[1], j);
            const int pl0 = __builtin_amdgcn_readlane(pl_[0], j), pl1 = __builtin_amdgcn_readlane(pl_[1], j);
            const int in0 = __builtin_amdgcn_readlane(it_[0], jn), in1 = __builtin_amdgcn_readlane(it_[1], jn);
            const int item = half ? it1 : it0;
            const int u = half ? us1 : us0;
            const int neg = half ? ng1 : ng0;
            const int pol = half ? pl1 : pl0;
            const int item_next = half ? in1 : in0;
            const bool act = j < n_mine;
            const bool at_u = (pol & 1) != 0, at_j = (pol & 2) != 0;
            const bool same = item == neg;
            float* const Pu = ((pol & 4) ? Prep : p.P) + static_cast<size_t>(act ? u : 0) * vdim + l32;
            float* const Qj = (at_j ? p.Q : Qrep) + static_cast<size_t>(act ? neg : 0) * vdim + l32;
            float* const Bj = (at_j ? p.Qb : Qbrep) + (act ? neg : 0);
            if (act) {
                if (u != prev_u) hrow_load(pu, Pu, nk);                       // else: consecutive slots of one entry carry the row
                if (!(neg == prev_neg && at_j == prev_hj)) {
                    hrow_load(qj, Qj, nk);
                    if (c.use_bias) bj = coh_load(Bj);
                }
                if (item != cur_i) {
                    flush_item(false);
                    cur_i = item;
                    flush_n = q.flush_every[item];
                    hrow_load(qi, p.Q + static_cast<size_t>(item) * vdim + l32, nk);
                    if (c.use_bias) bi = coh_load(p.Qb + item);
                }
            }
            // ---- score + sigmoid table (bpr.cc:119-131) ----
            float part = 0.f;
            if (act) {
#pragma unroll
                for (int k = 0; k < 4; ++k) part += pu.v[k] * (qi.v[k] - qj.v[k]);
            }
            float x = half_sum(part, half);
            if (c.use_bias) x += (bi - bj);
            float logit = 0.f;
            if (act) {
                if (6.0f < x) logit = 0.0f;
                else if (x < -6.0f) logit = 1.0f;
                else logit = s_exp[static_cast<int>((x + 6.0f) * 83.0f)];
                if (c.compute_loss) loss += static_cast<double>(log1pf(__expf(-fminf(fmaxf(x, -6.f), 6.f))));
            }
            bool want_flush = false;
            if (act) {
                // ---- bpr.cc:157-171 (Q-1: the user step sees the updated item rows) ----
                HRow dj, dpu;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float idv = logit * pu.v[k];
                    const float di = c.update_i ? c.lr * (idv - c.reg_i * qi.v[k]) : 0.f;
                    qi.v[k] += di;
                    dqi.v[k] += di;
                    if (same) qj.v[k] = qi.v[k];
                    dj.v[k] = c.update_j ? c.lr * (-idv - c.reg_j * qj.v[k]) : 0.f;
                    qj.v[k] += dj.v[k];
                    if (same) { qi.v[k] = qj.v[k]; dqi.v[k] += dj.v[k]; }
                    dpu.v[k] = c.lr * (logit * (qi.v[k] - qj.v[k]) - c.reg_u * pu.v[k]);
                    pu.v[k] += dpu.v[k];
                }
                float dbj = 0.f;
                if (c.use_bias) {
                    const float bi_new = c.update_i ? bias_step(bi, logit, c.lr_d, c.reg_b_d) : bi;   // bpr.cc:162 in double
                    dbi_acc += bi_new - bi;
                    bi = bi_new;
                    if (same) bj = bi;
                    const float bj_new = c.update_j ? bias_step(bj, -logit, c.lr_d, c.reg_b_d) : bj;  // bpr.cc:168
                    dbj = bj_new - bj;
                    bj = bj_new;
                    if (same) { bi = bj; dbi_acc += dbj; }
                }
                // ---- write the two per-triple rows back ----
                if (at_u) hrow_atomic_add(dpu, Pu, nk);
                else hrow_store(pu, Pu, nk);
                if (c.update_j && !same) {
                    if (at_j) { if (!(q.study & 1)) hrow_atomic_add(dj, Qj, nk); }
                    else hrow_store(qj, Qj, nk);
                    if (c.use_bias && l32 == 0) {
                        if (at_j) { if (!(q.study & 1)) atomic_add_f32(Bj, dbj); }
                        else *Bj = bj;
                    }
                }
                processed += 1;
                since_flush += 1;
                prev_u = u; prev_neg = neg; prev_hj = at_j;
                want_flush = since_flush >= flush_n && j + 1 < n_mine && item_next == item;
            }
            // a row re-read behind the previous flush has arrived by now: it carries the other waves' steps.  The bias came back
            // in lane32 0 only: spread it (uniform control flow)
            const float bre0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bi_re), 0));
            const float bre1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bi_re), 32));
            if (act && re_pending) {
#pragma unroll
                for (int k = 0; k < 4; ++k) qi.v[k] = qi_re.v[k] + dqi.v[k];
                if (c.use_bias) bi = (c.update_i ? (half ? bre1 : bre0) : bi_re) + dbi_acc;
                re_pending = false;
            }
            if (want_flush) flush_item(true);
        }
        if (n_mine > 0) flush_item(false);
        cur_i = -1;
    }
    if (l32 == 0 && processed) atomicAdd(q.done, processed);
    if (c.compute_loss && l32 == 0 && loss != 0.0) atomicAdd(c.loss_out, loss);
}

}  // namespace bfh
