// als_pc_kernel -- the in-place iALS++ row update (als.cc:211-358, block_size 32, d = 64 / 96 / 128) as PRODUCER / CONSUMER wave
// pairs.  Included by als_kernels.hpp (needs als_ialspp_inreg, als_tile_row / als_tile_col, AlsParams, AlsWork).
//
// Why.  The split-f16 Gramian pass of als_gram_kernel<SPLIT> holds two sets of rows and pieces next to 160 accumulator registers:
// 512 registers, ONE wave per SIMD, and a lone wave pays every dependent chain in full (profiles/r03_als_split_counters.txt: 47 %
// of its cycles issue VALU, 32 % wait on memory, the matrix pipe is busy 22-32 %; the per-row block solve, ~8 k cycles of
// latency-bound reductions, idles both pipes).  Here every SIMD of a CU carries TWO waves of 256 registers with complementary jobs:
//   * the PRODUCER streams the work list: keys -> gather of the other side's rows (two groups of 16 entries in flight ahead of
//     the one being prepared, across row boundaries) -> per-entry residual q.p0 - 1 and h = sum alpha v (q.p0 - 1) q
//     (als.cc:292-296) -> x = S sqrt(alpha v) q cut into two f16 pieces (v_fma_mixlo/hi_f16: 4 instructions per pair of
//     elements incl. the scaling) -> an 8 KB slot of a 3-deep LDS ring.  No matrix instruction, no accumulator.
//   * the CONSUMER owns the accumulators: per group 8 ds_read_b128 + 3 T(T+1)/2 v_mfma_f32_32x32x16_f16, and at the end of a row the
//     block recurrence from the registers (als_ialspp_inreg, unchanged).  While it solves, its producer fills the ring with the
//     next row; while the producer waits on HBM, the matrix pipe runs.
// The two waves of a pair talk through LDS only (sequence counters polled with s_sleep; every spin is bounded and raises an error
// flag the host turns into an exception).  Roles are dealt from HW_REG_HW_ID's SIMD id so that each SIMD gets one of each; any
// other placement is still correct.
//
// Operand layout (als_gram_kernel<SPLIT>'s): lane (half, col) works on the entries 16 g + 8 half + r, r = 0..7, of group g and on
// the elements [32 b + col], b = 0..T-1 of their rows.  The rows are read from a block-interleaved copy of the other factor
// (als_interleave_stats_kernel: position T col + b holds element 32 b + col), so a lane's T elements are ONE 4T-byte load and a
// half-wave reads one contiguous row.
//
// Entries the f16 pieces cannot carry (weight alpha v negative, or above the cut) never reach this kernel: a scan of the weights
// (als_defer_scan_kernel, cached per chunk) routes their rows through the fp32 instruction + the dense-solve kernel.
#pragma once

namespace bfh {

template <int T>
struct AlsPc {
    static constexpr int NT = T * (T + 1) / 2;
    static constexpr int VD = 32 * T;
    static constexpr int NSLOT = 3;                        // ring depth per pair (groups of 16 entries)
    static constexpr int NK = 2;                           // staged 64-entry key chunks per pair
    static constexpr int SLOT_B = 2 * T * 1024;            // H[0..T-1] | L[0..T-1], each 64 lanes x 16 B
    static constexpr int FF_B = NT * 4096;                 // the FF tiles in accumulator layout, scaled by S^2
    static constexpr int KEY_B = NK * 3 * 64 * 4;          // (row id, weight, S sqrt(weight)) x 64 entries
    static constexpr int BOX_B = 16 + 2 * VD * 4;          // row header (row, n, slot, -) | h | g1
    static constexpr int VEC_B = (2 * VD + 64) * 4;        // the consumer's solve vectors: p | delta | 64 exchange floats
    static constexpr int FLAG_B = 128;                     // six counters | 16 floats of exchange space for the producer's lane reduction
    static constexpr int PAIR_B = NSLOT * SLOT_B + KEY_B + 2 * BOX_B + VEC_B + FLAG_B;
    static constexpr int ROLE_B = 64;
    static constexpr int LDS_B = FF_B + 4 * PAIR_B + ROLE_B;
    static_assert(LDS_B <= 160 * 1024, "one workgroup per CU: the layout has to fit the 160 KB of LDS");
};
// flag words of a pair (ints at the end of its LDS block)
enum { PC_PROD = 0, PC_CONS = 1, PC_ROWS_PUB = 2, PC_ROWS_DONE = 3, PC_ROWS_FREE = 4, PC_ABORT = 5 };
// A wait gives up after PC_WAIT_TICKS of the constant 100 MHz clock (s_memrealtime) -- 20 s of WALL CLOCK, not a poll count (until round 5:
// 2^21 polls, tenths of a second -- a wave slowed or preempted by a profiler / debugger / a second process on the device turned a
// legitimate wait into BFH_ERR_HIP with rows already overwritten in place).  After that error the factors of the call's rows are
// undefined: re-upload them (bfh_als_initialize_model) before the next call.
constexpr unsigned long long PC_WAIT_TICKS = 2000000000ull;

// Qi[row][T col + b] = Q[row][32 b + col]  and  part[block] = max |Q| over the block's share (the split pass's scale,
// als_split_scale_kernel): one pass over the other factor per half-epoch
template <int T>
__global__ __launch_bounds__(256) void als_interleave_stats_kernel(const float* __restrict__ Q, size_t rows, float* __restrict__ Qi, float* __restrict__ part) {
    __shared__ float smq[256];
    float qm = 0.f;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t total = rows * 32;
    for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t row = e >> 5;
        const int col = static_cast<int>(e & 31);
        const float* src = Q + row * (32 * T) + col;
        float v[T];
#pragma unroll
        for (int b = 0; b < T; ++b) {
            v[b] = src[32 * b];
            qm = fmaxf(qm, fabsf(v[b]));
        }
        float* dst = Qi + row * (32 * T) + T * col;
#pragma unroll
        for (int b = 0; b < T; ++b) dst[b] = v[b];
    }
    smq[threadIdx.x] = qm;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) smq[threadIdx.x] = fmaxf(smq[threadIdx.x], smq[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = smq[0];
}

// One wave per work item: does it hold a weight the f16 pieces cannot carry (negative: no square root; above the cut: S sqrt(w) q
// could overflow; NaN)?  Such items are flagged (als_pc_kernel skips them) and listed for the fp32-instruction pass: whole rows get a
// scratch slot behind the heavy rows' (slot = n_heavy + j) and an entry in the solve list, chunks of heavy rows keep their row's slot.
// count[0] = listed items, count[1] = listed whole rows.
__global__ __launch_bounds__(256) void als_defer_scan_kernel(const AlsWork* __restrict__ work, int n_items, const float* __restrict__ vals, float alpha,
                                                             float wcut, int n_heavy, int* __restrict__ defer, AlsWork* __restrict__ dlist,
                                                             AlsHeavy* __restrict__ dsolve, int* __restrict__ count) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= n_items) return;
    const AlsWork wk = work[item];
    bool bad = false;
    for (int k = wk.kbeg + lane; k < wk.kend; k += 64) {
        const float w = alpha * vals[k];
        bad = bad || (!(w > 0.f && w <= wcut) && w != 0.f);
    }
    const bool any = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (lane != 0) return;
    defer[item] = any ? 1 : 0;
    if (!any) return;
    const int j = atomicAdd(count, 1);
    AlsWork out = wk;
    if (wk.slot < 0) {
        const int r = atomicAdd(count + 1, 1);
        out.slot = n_heavy + r;
        dsolve[r] = AlsHeavy{wk.row, out.slot, static_cast<int64_t>(wk.kend - wk.kbeg)};
    }
    dlist[j] = out;
}

// ---- LDS hand-off between the two waves of a pair ----------------------------------------------------------------------------
// All of a wave's DS instructions execute in issue order; the explicit lgkmcnt(0) before a counter store makes the data of the
// slot / box visible before the counter is, and keeps the compiler from moving either across it.
__device__ __forceinline__ void pc_publish(int* flag, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// spin until *flag - v > 0; false = the pair gave up (timeout here or at the partner).  `seen` caches the last value read: the
// counters only grow, so a wait the cached value already satisfies costs no LDS round trip
__device__ __forceinline__ bool pc_wait_gt(int* flag, int v, int& seen, int* flg) {
    if (seen - v > 0) return true;
    unsigned long long t0 = 0;
    for (int it = 0;; ++it) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (seen - v > 0) break;
        if ((it & 1023) == 1023) {
            const int ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flg + PC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            if (ab != 0 || now - t0 > PC_WAIT_TICKS) {   // (no global memory operation in here: the caller reports the time-out after its loop)
                if (ab == 0) __hip_atomic_store(flg + PC_ABORT, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return true;
}
// after a wave's loops: a pair that gave up raises the kernel's error flag
__device__ __forceinline__ void pc_report(int* flg, int* err, int lane) {
    const int ab = __hip_atomic_load(flg + PC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ab != 0 && lane == 0) atomicOr(err, 1);
}

// a lane's T elements of an interleaved row: one load
template <int T>
__device__ __forceinline__ void pc_load_row(const char* __restrict__ ptr, float (&q)[T]) {
    if constexpr (T == 4) {
        const float4 v = *reinterpret_cast<const float4*>(ptr);
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    } else if constexpr (T == 2) {
        const float2 v = *reinterpret_cast<const float2*>(ptr);
        q[0] = v.x; q[1] = v.y;
    } else {
        struct __attribute__((packed, aligned(4))) P3 { float a, b, c; };
        const P3 v = *reinterpret_cast<const P3*>(ptr);
        q[0] = v.a; q[1] = v.b; q[2] = v.c;
    }
}

// (q0 s0, q1 s1) -> packed f16 pairs h (rounded to nearest) and l (the remainder q s - h, exact in the fused multiply-add, rounded
// the same way): h + l carries the product to 2^-24.  v_fma_mix{lo,hi}_f16 take fp32 and f16 sources in one instruction.
__device__ __forceinline__ void pc_split_pair(float q0, float s0, float q1, float s1, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(q0), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(q1), "v"(s1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(q0), "v"(s0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(q1), "v"(s1), "v"(h));
}

template <int CTRL>
__device__ __forceinline__ float pc_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// y[r], r = 0..7  ->  the sum of y[r] over the 32 lanes of this half-wave, all eight in every lane.  A butterfly costs 5 steps x 8
// values = 40 instructions (+ 8 adds); here every step HALVES the number of values a lane carries on (lane bit 3 decides which four
// of the eight it keeps and which it hands to its mirror lane, bit 2 which two of the four, bit 0 which one), so the tree costs
// 12 + 6 + 3 + 1 + 2 instructions, and one LDS round trip hands all eight sums to every lane: 31 issue slots instead of 48 -- the
// producer is bound by its instruction count.  `tmp`: 16 floats of LDS private to the wave.
__device__ __forceinline__ void pc_sum8_over_half(float (&y)[8], float* tmp, int lane) {
    const bool b3 = lane & 8, b2 = lane & 4, b0 = lane & 1;
    float z[4], w[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = (b3 ? y[i + 4] : y[i]) + pc_dpp<0x140>(b3 ? y[i] : y[i + 4]);        // row_mirror: lane ^ 15
#pragma unroll
    for (int i = 0; i < 2; ++i) w[i] = (b2 ? z[i + 2] : z[i]) + pc_dpp<0x141>(b2 ? z[i] : z[i + 2]);        // row_half_mirror: lane ^ 7
    float v = (b0 ? w[1] : w[0]) + pc_dpp<0xB1>(b0 ? w[0] : w[1]);                                            // quad_perm [1,0,3,2]: lane ^ 1
    v += pc_dpp<0x4E>(v);                                                                                     // quad_perm [2,3,0,1]: lane ^ 2
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));          // the half's other 16-lane row
    float* t8 = tmp + 8 * (lane >> 5);
    t8[(lane & 1) + ((lane & 12) >> 1)] = v;   // this lane carries the sum of y[b0 + 2 b2 + 4 b3]
    wave_lds_sync();
    const float4 a = *reinterpret_cast<const float4*>(t8), b = *reinterpret_cast<const float4*>(t8 + 4);
    y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w; y[4] = b.x; y[5] = b.y; y[6] = b.z; y[7] = b.w;
}

struct PcChunk {   // where a pipeline stage stands in the pair's stream: one 64-entry chunk of one work item (all wave-uniform)
    int valid;
    int row, kbeg, n, slot;   // the work item
    int chunk, ng;            // chunk inside the item, groups (of 16 entries) of the item
    int rseq;                 // running number of the item (row box = rseq & 1)
    int buf;                  // key staging buffer of the chunk (0 / 1)
};

// ---- producer ------------------------------------------------------------------------------------------------------------------
// The stream of a pair is walked CHUNK by chunk (64 entries = 4 groups): everything that decides where the stream goes next -- the
// work list, the row boundaries, the staging of a chunk's keys and weights -- runs once per chunk, and the four group steps inside
// are straight-line code with compile-time sub-indices.  (A lone wave issues about one instruction every four cycles whatever
// its kind, so the instruction COUNT of the loop is the producer's speed: the first version spent more issue slots on its
// per-group cursor than on the arithmetic -- profiles/r04_als_pc_steps.txt.)
// Four register sets hold the rows of four groups; step s loads group (chunk, s) into set s and prepares set (s + 1) % 4, loaded
// three steps earlier.  Every vector-memory operation of a trip is issued unconditionally and in a fixed order (rows past an item's
// end are row 0 with weight 0): the compiler joins control-flow paths by the FEWEST loads issued since the one it waits for, and
// one path without loads turns every wait into vmcnt(0).
template <int T, bool BIG, bool LOSS>
__device__ __forceinline__ void als_pc_producer(const AlsParams& p, const AlsWork* __restrict__ work, int n_items, const float* __restrict__ Qi,
                                                const int* __restrict__ defer, char* pl, int* err, int lane) {
    using C = AlsPc<T>;
    constexpr int VD = C::VD;
    const int half = lane >> 5, col = lane & 31;
    char* ring = pl;
    int* ks = reinterpret_cast<int*>(pl + C::NSLOT * C::SLOT_B);
    char* box = pl + C::NSLOT * C::SLOT_B + C::KEY_B;
    int* flg = reinterpret_cast<int*>(pl + C::PAIR_B - C::FLAG_B);
    float* redtmp = reinterpret_cast<float*>(pl + C::PAIR_B - C::FLAG_B + 64);
    const float sS = p.split[0], wcut = p.split[3], alpha = p.alpha;
    const bool lossk = LOSS && p.compute_loss && p.axis == 1;
    const int dbg = p.debug;
    double nume_k = 0.0, deno_k = 0.0;
    const char* qbase = reinterpret_cast<const char*>(Qi);
    const unsigned lane_off = static_cast<unsigned>(col) * (4u * T);
    const int32_t* __restrict__ keys = p.keys;
    const float* __restrict__ vals = p.vals;
    const float* __restrict__ Pm = p.P;

    // ---- the work list, drawn in batches (same-address atomics serialise at ~12 ns: one draw per row would cost 1.7 ms per half-epoch).
    // The ticket of the NEXT batch is always under way while the current one [b_base, b_base + b_len) is worked through; the items
    // themselves are wave-uniform reads (scalar loads: they do not count against the vector-memory counter the row loads are
    // tracked with).
    int b_base = 0, b_len = 0, b_pos = 0;
    int tk_v = 0, tk_rows = 0;
    const int batch_max = p.batch;
    auto draw = [&](int rows) {
        tk_rows = rows;
        if (lane == 0) tk_v = atomicAdd(p.ticket, rows);
    };
    bool list_end = false;
    draw(1);
    // the next item that is not deferred; false at the end of the list
    auto next_item = [&](int& row, int& kbeg, int& n, int& slot) -> bool {
        for (;;) {
            if (list_end) return false;
            if (b_pos >= b_len) {
                const int base = __builtin_amdgcn_readfirstlane(tk_v);
                int len = n_items - base;
                len = len < 0 ? 0 : (len > tk_rows ? tk_rows : len);
                b_base = base; b_len = len; b_pos = 0;
                if (len == 0) { list_end = true; return false; }
                // the list is sorted longest first: nothing later is longer than the item at hand, so about 1024 entries per draw
                const int l0 = work[base].kend - work[base].kbeg;
                int fit = 1024 / (l0 > 0 ? l0 : 1);
                fit = fit < 1 ? 1 : (fit > batch_max ? batch_max : fit);
                draw(fit);
            }
            const int idx = b_base + b_pos++;
            const AlsWork w = work[idx];
            row = w.row; kbeg = w.kbeg; n = w.kend - w.kbeg; slot = w.slot;
            if (!(defer && defer[idx])) return true;
        }
    };

    // ---- key chunks: fetched one chunk ahead into (pk_c, pk_v), weighted and staged in LDS when stage A enters the chunk ----
    int pk_c = 0, pk_rseq = -1, pk_chunk = -1;
    float pk_v = 0.f;
    auto load_keys = [&](int kbeg, int n, int chunk, int rseq) {
        int kk = chunk * 64 + lane;
        kk = kk < n ? kk : n - 1;
        kk = kk < 0 ? 0 : kk;
        pk_c = keys[kbeg + kk];
        pk_v = vals[kbeg + kk];
        pk_rseq = rseq;
        pk_chunk = chunk;
    };

    PcChunk CA{}, CB{};
    int nx_valid = 0, nx_row = 0, nx_kbeg = 0, nx_n = 1, nx_slot = -1;
    auto fetch_next = [&]() {
        nx_valid = next_item(nx_row, nx_kbeg, nx_n, nx_slot) ? 1 : 0;
        if (!nx_valid) { nx_row = 0; nx_kbeg = 0; nx_n = 1; nx_slot = -1; }
    };
    CA.n = 1;
    fetch_next();
    if (nx_valid) {
        CA.valid = 1; CA.row = nx_row; CA.kbeg = nx_kbeg; CA.n = nx_n; CA.slot = nx_slot;
        CA.chunk = 0; CA.ng = (nx_n + 15) >> 4; CA.rseq = 0; CA.buf = 0;
        fetch_next();
    }
    load_keys(CA.kbeg, CA.n, 0, 0);
    auto advance = [&]() {   // stage A moves to the next chunk of the stream
        CA.buf ^= 1;
        if (!CA.valid) return;
        if ((CA.chunk + 1) * 4 < CA.ng) { ++CA.chunk; return; }
        if (!nx_valid) { CA.valid = 0; CA.row = 0; CA.kbeg = 0; CA.n = 1; CA.chunk = 0; CA.ng = 0; return; }
        CA.row = nx_row; CA.kbeg = nx_kbeg; CA.n = nx_n; CA.slot = nx_slot;
        CA.chunk = 0; CA.ng = (nx_n + 15) >> 4; ++CA.rseq;
        fetch_next();
    };

    float raw[4][8][T];      // the rows of four groups: one being prepared, three on their way
    float p0A[T];            // the row at entry (this lane's elements [32 b + col]) of stage A's item
    float p0cur[T], gpart[T], g1part[T];
#pragma unroll
    for (int b = 0; b < T; ++b) { p0A[b] = 0.f; p0cur[b] = 0.f; gpart[b] = 0.f; g1part[b] = 0.f; }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int b = 0; b < T; ++b) raw[s4][r][b] = 0.f;
    int gseq = 0;            // groups written to the ring
    int gpub = 0;            // groups published
    int rows_opened = 0;
    int seen_cons = 0, seen_free = 0;
    bool ok = true, bad_keys = false, bad_weight = false;

    // the flag store of a group is put off until the NEXT step's LDS reads have been waited for: the wait for the slot's eight
    // 16-byte stores then costs nothing
    auto flush_pub = [&]() {
        if (gpub != gseq) {
            gpub = gseq;
            pc_publish(flg + PC_PROD, gseq);
        }
    };

    auto load_group = [&](const int* kbA, int sg, float (&rw)[8][T]) {
        const int4 c0 = *reinterpret_cast<const int4*>(kbA + 16 * sg + 8 * half);
        const int4 c1 = *reinterpret_cast<const int4*>(kbA + 16 * sg + 8 * half + 4);
        const int cid[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        // (no run-time switch around these loads: a path without them would put the compiler's wait counts back to "everything")
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (BIG) {
                const size_t voff = static_cast<size_t>(static_cast<unsigned>(cid[r])) * (4u * VD) + lane_off;
                pc_load_row<T>(qbase + voff, rw[r]);
            } else {
                const unsigned voff = static_cast<unsigned>(cid[r]) + lane_off;
                pc_load_row<T>(qbase + voff, rw[r]);
            }
        }
    };

    // group sg (compile-time) of chunk Cc: residuals, h, pieces -> ring slot
    auto prep_group = [&](const PcChunk& Cc, int sg, float (&q)[8][T]) {
        const int g = 4 * Cc.chunk + sg;
        const bool live = Cc.valid && g < Cc.ng && ok;
        if (!live) return;
        char* bx = box + (Cc.rseq & 1) * C::BOX_B;
        const int* kb = ks + Cc.buf * 192 + 16 * sg + 8 * half;
        const float4 w0 = *reinterpret_cast<const float4*>(kb + 64), w1 = *reinterpret_cast<const float4*>(kb + 64 + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(kb + 128), s1 = *reinterpret_cast<const float4*>(kb + 128 + 4);
        const float wgt[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        if (sg == 0 && Cc.chunk == 0) {   // the item opens: its row at entry, fresh sums, the header for the consumer
#pragma unroll
            for (int b = 0; b < T; ++b) { p0cur[b] = p0A[b]; gpart[b] = 0.f; g1part[b] = 0.f; }
            if (!pc_wait_gt(flg + PC_ROWS_FREE, Cc.rseq - 2, seen_free, flg)) { ok = false; return; }
            if (lane == 0) *reinterpret_cast<int4*>(bx) = make_int4(Cc.row, Cc.n, Cc.slot, 0);
            ++rows_opened;
            pc_publish(flg + PC_ROWS_PUB, Cc.rseq + 1);
        }
        u32x4 H[T], L[T];
        if (dbg & 32) {   // (timing probe: bit 32 drops the arithmetic)
#pragma unroll
            for (int b = 0; b < T; ++b) { H[b] = u32x4{0u, 0u, 0u, 0u}; L[b] = H[b]; }
        } else {
            // als.cc:292-296: residual = Yui - 1 against the row at entry; laid out in stages over the eight entries so that the
            // dependent chains run side by side
            float y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                y[r] = q[r][0] * p0cur[0];
#pragma unroll
                for (int b = 1; b < T; ++b) y[r] = __builtin_fmaf(q[r][b], p0cur[b], y[r]);
            }
            pc_sum8_over_half(y, redtmp, lane);
            // the previous group goes out HERE: its stores have drained behind the reduction's LDS round trip, and the wait costs nothing
            // (placed before the dot products it exposes the latency of the weight reads: +0.1 ms per half-epoch, profiles/r04_als_pc_steps.txt)
            flush_pub();
            const int k0 = 16 * g + 8 * half;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float cial = __builtin_fmaf(wgt[r], y[r], -wgt[r]);   // alpha v (q.p0 - 1)
                const float one = (LOSS && lossk && k0 + r < Cc.n) ? 1.0f : 0.f;
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    gpart[b] = __builtin_fmaf(cial, q[r][b], gpart[b]);
                    if (LOSS) g1part[b] = __builtin_fmaf(one, q[r][b], g1part[b]);
                }
            }
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    unsigned h_, l_;
                    pc_split_pair(q[2 * j2][b], sw[2 * j2], q[2 * j2 + 1][b], sw[2 * j2 + 1], h_, l_);
                    H[b][j2] = h_;
                    L[b][j2] = l_;
                }
        }
        if (seen_cons - (gseq - C::NSLOT) <= 0) {   // the ring looks full: whatever is still unpublished goes out before the wait
            flush_pub();
            if (!pc_wait_gt(flg + PC_CONS, gseq - C::NSLOT, seen_cons, flg)) { ok = false; return; }
        }
        char* sl = ring + (gseq % C::NSLOT) * C::SLOT_B + lane * 16;
#pragma unroll
        for (int b = 0; b < T; ++b) {
            *reinterpret_cast<u32x4*>(sl + b * 1024) = H[b];
            *reinterpret_cast<u32x4*>(sl + (T + b) * 1024) = L[b];
        }
        ++gseq;
        if (g == Cc.ng - 1) {   // the item closes: h (and g1) for the consumer; the two halves hold the k-parities of the same element
            float* hb = reinterpret_cast<float*>(bx + 16);
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const float gs = gpart[b] + __shfl_xor(gpart[b], 32, 64);
                const float g1s = g1part[b] + __shfl_xor(g1part[b], 32, 64);
                if (half == 0) {
                    hb[b * 32 + col] = gs;
                    if (LOSS) hb[VD + b * 32 + col] = g1s;
                }
            }
            flush_pub();
            pc_publish(flg + PC_ROWS_DONE, Cc.rseq + 1);
        }
    };

    // Everything fetched so far is waited for HERE, by reading it: whatever the loop header inherits as "in flight" from the code
    // before the loop, the compiler keeps waiting for on every trip.
    asm volatile("" ::"v"(pk_c), "v"(pk_v), "v"(tk_v));
    do {
        // ---- stage A enters chunk CA: weigh and stage its keys (fetched during the previous trip), fetch the next chunk's ----
        int* kbA = ks + CA.buf * 192;
        if (CA.valid) {
            if (!(pk_rseq == CA.rseq && pk_chunk == CA.chunk)) bad_keys = true;   // (cannot happen: every chunk is fetched ahead)
            const bool in = CA.chunk * 64 + lane < CA.n;
            const float ww = in ? alpha * pk_v : 0.f;     // padding lanes: row 0 of the other factor with weight 0
            const float ss = (ww > 0.f && ww <= wcut) ? sS * __builtin_amdgcn_sqrtf(ww) : 0.f;
            if (lossk && in) {   // constant and denominator of the loss (als_gram_kernel's header)
                const double w = static_cast<double>(ww);
                deno_k += w;
                nume_k += 1.0 + w;
            }
            if (ww != 0.f && ss == 0.f) bad_weight = true;   // a weight the scan should have routed elsewhere
            // (the row's BYTE offset into the interleaved factor when that fits 32 bits: one instruction less per gathered row)
            kbA[lane] = in ? (BIG ? pk_c : pk_c * (4 * VD)) : 0;
            kbA[64 + lane] = __builtin_bit_cast(int, ww);
            kbA[128 + lane] = __builtin_bit_cast(int, ss);
        } else {
            kbA[lane] = 0;
        }
        wave_lds_sync();
        if ((CA.chunk + 1) * 64 < CA.n) load_keys(CA.kbeg, CA.n, CA.chunk + 1, CA.rseq);
        else load_keys(nx_kbeg, nx_n, 0, CA.rseq + 1);
        {
            const float* Pu0 = Pm + static_cast<size_t>(CA.row) * VD;
#pragma unroll
            for (int b = 0; b < T; ++b) p0A[b] = Pu0[b * 32 + col];
        }
        // ---- four steps: load group s of CA, prepare the group loaded three steps ago ----
        load_group(kbA, 0, raw[0]); prep_group(CB, 1, raw[1]);
        load_group(kbA, 1, raw[1]); prep_group(CB, 2, raw[2]);
        load_group(kbA, 2, raw[2]); prep_group(CB, 3, raw[3]);
        load_group(kbA, 3, raw[3]); prep_group(CA, 0, raw[0]);
        CB = CA;
        advance();
    } while (ok && (CA.valid | CB.valid));
    flush_pub();
    if (ok) {   // end of the list: a header with row -1 sends the consumer home
        const int rs = rows_opened;
        if (pc_wait_gt(flg + PC_ROWS_FREE, rs - 2, seen_free, flg)) {
            if (lane == 0) *reinterpret_cast<int4*>(box + (rs & 1) * C::BOX_B) = make_int4(-1, 0, -1, 0);
            pc_publish(flg + PC_ROWS_PUB, rs + 1);
        }
    }
    pc_report(flg, err, lane);
    if (__builtin_amdgcn_ballot_w64(bad_weight) != 0 && lane == 0) atomicOr(err, 2);
    if (bad_keys && lane == 0) atomicOr(err, 4);
    if (lossk) {
        nume_k = wave_sum_f64(nume_k);
        deno_k = wave_sum_f64(deno_k);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}

// ---- consumer ------------------------------------------------------------------------------------------------------------------
template <int T, bool BIG, bool LOSS>
__device__ __forceinline__ void als_pc_consumer(const AlsParams& p, float* __restrict__ scratch, const float* ff_acc, char* pl, int* err, int lane) {
    using C = AlsPc<T>;
    constexpr int VD = C::VD, NT = C::NT;
    const int half = lane >> 5, col = lane & 31;
    const char* ring = pl;
    const char* box = pl + C::NSLOT * C::SLOT_B + C::KEY_B;
    float* pc = reinterpret_cast<float*>(pl + C::NSLOT * C::SLOT_B + C::KEY_B + 2 * C::BOX_B);
    int* flg = reinterpret_cast<int*>(pl + C::PAIR_B - C::FLAG_B);
    const float sI2 = p.split[2];
    const bool lossk = LOSS && p.compute_loss && p.axis == 1;
    double nume_k = 0.0, deno_k = 0.0;
    int gseq = 0;
    int seen_prod = 0, seen_pub = 0, seen_done = 0;
    for (int rseq = 0;; ++rseq) {
        if (!pc_wait_gt(flg + PC_ROWS_PUB, rseq, seen_pub, flg)) break;
        const char* bx = box + (rseq & 1) * C::BOX_B;
        const int4 hd = *reinterpret_cast<const int4*>(bx);
        const int row = __builtin_amdgcn_readfirstlane(hd.x), n = __builtin_amdgcn_readfirstlane(hd.y), slot = __builtin_amdgcn_readfirstlane(hd.z);
        if (row < 0) break;
        const bool solve_here = slot < 0;
        // the row at entry and FF p0 (als_rowff_kernel): wanted by the solve only -- the loads ride under the whole Gramian pass
        float p0r[T], f0r[T];
#pragma unroll
        for (int b = 0; b < T; ++b) { p0r[b] = 0.f; f0r[b] = 0.f; }
        if (solve_here) {
            const float* Pu0 = p.P + static_cast<size_t>(row) * VD;
            const float* Fu0 = p.F0 + static_cast<size_t>(row - p.start_x) * VD;
#pragma unroll
            for (int b = 0; b < T; ++b) { p0r[b] = Pu0[b * 32 + col]; f0r[b] = Fu0[b * 32 + col]; }
        }
        f32x16 acc[NT];
        if (solve_here && !(p.debug & 2)) {   // M = FF + G: start from the FF tiles (accumulator layout, scaled by S^2)
            const float* fl = ff_acc + lane * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float4 v = *reinterpret_cast<const float4*>(fl + (t * 4 + e4) * 256);
                    acc[t][4 * e4 + 0] = v.x; acc[t][4 * e4 + 1] = v.y; acc[t][4 * e4 + 2] = v.z; acc[t][4 * e4 + 3] = v.w;
                }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        }
        const int ng = (n + 15) >> 4;
        u32x4 HA[T], LA[T], HB[T], LB[T];
        auto read_slot = [&](int seq, u32x4 (&H)[T], u32x4 (&L)[T]) {
            const char* sl = ring + (seq % C::NSLOT) * C::SLOT_B + lane * 16;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                L[b] = *reinterpret_cast<const u32x4*>(sl + (T + b) * 1024);
                H[b] = *reinterpret_cast<const u32x4*>(sl + b * 1024);
            }
        };
        // part 0 = the l h and h l products (2 T(T+1)/2 instructions), part 1 = h h: the slot of the NEXT group is released between the
        // two -- its pieces have landed in registers behind part 0, and the producer gets the slot back early (a release straight after
        // the reads stalls the matrix pipe for an LDS round trip: measured +0.2 ms on the item half-epoch)
        auto mfmas = [&](const u32x4 (&H)[T], const u32x4 (&L)[T], int part) {
            if (p.debug & 16) return;
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {   // small terms first: l h, h l, h h
                if ((pr < 2) != (part == 0)) continue;
                int t = 0;
#pragma unroll
                for (int a = 0; a < T; ++a)
#pragma unroll
                    for (int b = a; b < T; ++b, ++t) {
                        const u32x4 X = pr == 0 ? L[a] : H[a];
                        const u32x4 Y = pr == 1 ? L[b] : H[b];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, X), __builtin_bit_cast(f16x8_t, Y), acc[t], 0, 0, 0);
                    }
            }
        };
        bool ok = true;
        if (ng > 0) {
            if (!pc_wait_gt(flg + PC_PROD, gseq, seen_prod, flg)) break;
            read_slot(gseq, HA, LA);
            for (int g = 0;;) {
                const bool more = g + 1 < ng;
                if (more) {   // the next group's pieces are on their way from LDS while this group's matrix instructions issue
                    if (!pc_wait_gt(flg + PC_PROD, gseq + 1, seen_prod, flg)) { ok = false; break; }
                    read_slot(gseq + 1, HB, LB);
                }
                mfmas(HA, LA, 0);
                mfmas(HA, LA, 1);
                // both slots read so far are released AFTER the group's matrix instructions: a release in their middle (or straight
                // after the reads) was measured slower (+0.2 ms on the item half-epoch: the wait for the reads stalls the matrix pipe)
                pc_publish(flg + PC_CONS, gseq + (more ? 2 : 1));
                ++gseq; ++g;
                if (!more) break;
                const bool more2 = g + 1 < ng;
                if (more2) {
                    if (!pc_wait_gt(flg + PC_PROD, gseq + 1, seen_prod, flg)) { ok = false; break; }
                    read_slot(gseq + 1, HA, LA);
                }
                mfmas(HB, LB, 0);
                mfmas(HB, LB, 1);
                pc_publish(flg + PC_CONS, gseq + (more2 ? 2 : 1));
                ++gseq; ++g;
                if (!more2) break;
            }
        }
        if (!ok) break;
        if (!pc_wait_gt(flg + PC_ROWS_DONE, rseq, seen_done, flg)) break;
        float gs[T], g1s[T];
        {
            const float* hb = reinterpret_cast<const float*>(bx + 16);
#pragma unroll
            for (int b = 0; b < T; ++b) {
                gs[b] = hb[b * 32 + col];
                g1s[b] = LOSS ? hb[VD + b * 32 + col] : 0.f;
            }
        }
        pc_publish(flg + PC_ROWS_FREE, rseq + 1);   // (waits for the reads above)
        if (solve_here) {
            float* Pu = p.P + static_cast<size_t>(row) * VD;
            double nume = 0.0, deno = 0.0;
            if (!(p.debug & 1)) {
                wave_lds_sync();
                if (half == 0) {
#pragma unroll
                    for (int b = 0; b < T; ++b) { pc[b * 32 + col] = p0r[b]; pc[VD + b * 32 + col] = 0.f; }
                }
                wave_lds_sync();
                als_ialspp_inreg<T>(acc, gs, g1s, f0r, p, pc, pc + VD, pc + 2 * VD, lane, half, col, p.adaptive_reg ? static_cast<float>(n) : 1.0f, nume, deno, sI2);
                wave_lds_sync();
                for (int e = lane; e < VD; e += 64) Pu[e] = pc[e];
            }
            if (p.compute_loss && lane == 0) {
                nume_k += nume;
                deno_k += deno;
            }
        } else {   // chunk of a heavy row: the tiles, h and g1 are summed in the row's scratch slot (zeroed by the host) for als_solve_kernel
            float* S = scratch + static_cast<size_t>(slot) * als_slot_floats(VD);
            float* Sl = S + half * 4 * VD + col;
            const float osc = p.out_scale * sI2, osg = p.out_scale;
            int t = 0;
#pragma unroll
            for (int a = 0; a < T; ++a) {
#pragma unroll
                for (int b = a; b < T; ++b, ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) atomic_add_f32(Sl + (a * 32 + (e & 3) + 8 * (e >> 2)) * VD + b * 32, acc[t][e] * osc);
                if (half == 0) {
                    float* gdst = S + VD * VD + a * 32 + col;
                    atomic_add_f32(gdst, gs[a] * osg);
                    if (lossk) atomic_add_f32(gdst + VD, g1s[a]);
                }
            }
        }
    }
    pc_report(flg, err, lane);
    if (p.compute_loss) {
        nume_k = wave_sum_f64(nume_k);
        deno_k = wave_sum_f64(deno_k);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}

// One 512-thread workgroup per CU: four producer / consumer pairs.  err[0]: bit 0 a wait timed out, bit 1 a weight outside the f16
// path reached the kernel; err[1]: workgroups whose pairs all sit on one SIMD each (placement statistic, not an error).
template <int T, bool BIG, bool LOSS>
__global__ __launch_bounds__(512, 2) void als_pc_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_items, float* __restrict__ scratch,
                                                        const float* __restrict__ Qi, const int* __restrict__ defer, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char pc_lds[];
    using C = AlsPc<T>;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (als_debug bit 1024: the shader clock this workgroup sees over the kernel -- s_memtime counts core cycles, s_memrealtime the constant 100 MHz)
    const bool clk_probe = (p.debug & 1024) && blockIdx.x == 0 && tid == 0;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_probe) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    float* ff_acc = reinterpret_cast<float*>(pc_lds);
    int* role_tab = reinterpret_cast<int*>(pc_lds + C::FF_B + 4 * C::PAIR_B);   // [0..7] pair * 2 + role, [8..15] SIMD id of wave w
    {
        const float sS2 = p.split[1];
        for (int idx = tid; idx < C::NT * 1024; idx += 512) {
            const int t = idx >> 10, rem = idx & 1023, e4 = rem >> 8, ln = (rem >> 2) & 63, e3 = rem & 3;
            const int a = als_tile_row<T>(t), b = als_tile_col<T>(t);
            const int row = a * 32 + e3 + 8 * e4 + 4 * (ln >> 5), cc = b * 32 + (ln & 31);
            ff_acc[idx] = sS2 * p.FF[row * C::VD + cc];
        }
    }
    unsigned simd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(simd));
    if (lane == 0) role_tab[8 + wv] = static_cast<int>(simd);
    if (tid < 64) reinterpret_cast<int*>(pc_lds + C::FF_B + (tid >> 4) * C::PAIR_B + C::PAIR_B - C::FLAG_B)[tid & 15] = 0;
    __syncthreads();
    if (tid == 0) {   // one consumer + one producer per SIMD where the placement allows, any pairing otherwise
        int np = 0, same = 0;
        unsigned used = 0;
        for (int s = 0; s < 4; ++s) {
            int first = -1;
            for (int w = 0; w < 8; ++w)
                if (role_tab[8 + w] == s) {
                    if (first < 0) first = w;
                    else {
                        role_tab[first] = np * 2; role_tab[w] = np * 2 + 1;
                        used |= (1u << first) | (1u << w);
                        ++np; ++same; first = -1;
                    }
                }
        }
        int first = -1;
        for (int w = 0; w < 8; ++w)
            if (!((used >> w) & 1)) {
                if (first < 0) first = w;
                else { role_tab[first] = np * 2; role_tab[w] = np * 2 + 1; ++np; first = -1; }
            }
        if (same == 4) atomicAdd(err + 1, 1);
    }
    __syncthreads();
    const int rl = __builtin_amdgcn_readfirstlane(role_tab[wv]);
    char* pl = pc_lds + C::FF_B + (rl >> 1) * C::PAIR_B;
    if (p.debug & 128) {   // (probe: static issue priority for the producers / 256: for the consumers)
        if (rl & 1) __builtin_amdgcn_s_setprio(2);
    } else if (p.debug & 256) {
        if (!(rl & 1)) __builtin_amdgcn_s_setprio(2);
    }
    if (rl & 1) als_pc_producer<T, BIG, LOSS>(p, work, n_items, Qi, defer, pl, err, lane);
    else als_pc_consumer<T, BIG, LOSS>(p, scratch, ff_acc, pl, err, lane);
    if (clk_probe) {   // wave 0 of workgroup 0 at the end of ITS stream (the kernel's tail may run a little longer on other waves)
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk_c0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
        err[2] = static_cast<int>(dc & 0xffffffffull); err[3] = static_cast<int>(dc >> 32);
        err[4] = static_cast<int>(dr & 0xffffffffull); err[5] = static_cast<int>(dr >> 32);
    }
}

}  // namespace bfh
