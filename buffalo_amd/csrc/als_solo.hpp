// als_solo_kernel -- the in-place iALS++ row update (als.cc:211-358, block_size 32, d = 128) with EVERY wave carrying a whole row:
// gather, residuals, f16 cut, its own matrix instructions and the block solve.  Two waves of 256 registers per SIMD, no hand-off.
// Included by als_kernels.hpp after als_pc.hpp (shares its helpers: pc_split_pair, pc_sum8_over_half, the interleaved factor copy, the
// deferred-weight scan).  EXPERIMENTAL: compiled only with -DBFH_WITH_ALS_SOLO (BFH_EXTRA_FLAGS), then "als_pc" = 3 selects it at d = 128;
// first contact in profiles/r04_als_solo_first_contact.txt (user half -6 %, item half +4 % against the pairs; the d = 96 instantiation, 12-byte
// DMA loads, aborted on its first launch and is not dispatched).
//
// Why (profiles/r04_micro_*.txt).  The pair kernel's user half sits 1.75 ms above the 0.95 ms its loads need: per row the
// producer's and the consumer's VALU work share one VALU, the ring couples them to a third of a row, a quarter of the producer's steps is
// padding, and the handshakes cost ~500 instructions per row.  The micro-benchmarks say (a) one group of lookahead per wave is enough for
// the memory system, (b) plain VALU work of one wave rides in the issue slots the other wave's matrix instructions leave free (~7 per
// instruction).  So: independent waves, each alternating between a VALU phase (prepare the group whose rows arrived) and a matrix phase (30
// instructions on the pieces just cut); the two waves of a SIMD drift into opposite phases by themselves, a solve stalls nobody, and a row
// costs exactly its groups.
//
// Registers decide the shape: 160 accumulators + the rows of a group (32, cut in place into the pieces) leave no room for a second set of
// rows in flight, so everything the loop fetches travels global -> LDS by the DMA path (global_load_lds_dword / _dwordx4: a wave instruction
// lands lane-linearly, no register is named): the NEXT group's 16 rows (issued as soon as this group's rows have been read out of the 8 KB
// buffer: the whole VALU + matrix phase to arrive), the next chunk's keys and values, the next row's p0 and FF p0.
// The DMA loads are inline assembly: hipcc orders every later LDS read behind a DMA load it knows of (a vmcnt(0) that would drain the
// prefetch at the first ds_read of the VALU phase).  They are counted by hand instead -- ONE s_waitcnt opens a trip, and everything a trip
// reads from a DMA buffer was issued at least one trip earlier.  hipcc's own waits (the ticket atomic) stay correct: extra operations in
// flight only make them conservative.
// Loop shape: ONE group per trip, a single copy of the step (the row end with the solve exists once); everything that decides where the
// stream goes is wave-uniform.
#pragma once

namespace bfh {

template <int T>
struct AlsSolo {
    static constexpr int NT = T * (T + 1) / 2;
    static constexpr int VD = 32 * T;
    static constexpr int FF_B = NT * 4096;                 // the FF tiles in accumulator layout, scaled by S^2 (shared by the 8 waves)
    static constexpr int KEY_B = 2 * 3 * 64 * 4;           // two staged 64-entry chunks: (row offset, weight, S sqrt(weight))
    static constexpr int RK_B = 2 * 2 * 64 * 4;            // two landing buffers for the next chunk's (key, value) as the DMA brings them
    static constexpr int NV_B = 2 * 2 * 128 * 4;           // two rows' p0 | FF p0 (128 floats each; vdim 96 uses 96), by item parity: the row at hand keeps its
                                                           // buffer until it closes while the next row's vectors land in the other
    static constexpr int VEC_B = (2 * VD + 64) * 4;        // solve vectors: p | delta | 64 exchange floats
    static constexpr int RED_B = 64;                       // 16 floats for the lane reduction
    static constexpr int LOSS_B = 2 * 64 * 8;              // per-lane loss sums (numerator | denominator) in double: kept out of the register file
    static constexpr int ROW_B = 8 * 64 * 4 * T;           // the 16 rows of a group as the DMA lands them: 8 wave instructions x 64 lanes x 4T bytes
    static constexpr int WAVE_B = KEY_B + RK_B + NV_B + VEC_B + RED_B + LOSS_B + ROW_B;
    static constexpr int LDS_B = FF_B + 8 * WAVE_B;
    static_assert(WAVE_B % 16 == 0 && (KEY_B + RK_B + NV_B + VEC_B + RED_B + LOSS_B) % 16 == 0, "16-byte aligned DMA targets");
    static_assert(LDS_B <= 160 * 1024, "one workgroup per CU");
};

// one DMA wave instruction: lane i's BYTES bytes at `src` -> LDS byte address `dst` + i * BYTES (dst wave-uniform, in an SGPR)
template <int BYTES>
__device__ __forceinline__ void solo_dma(const void* src, unsigned dst) {
    unsigned keep;
    static_assert(BYTES == 4 || BYTES == 12 || BYTES == 16, "global_load_lds moves 4, 12 or 16 bytes per lane here");
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else if constexpr (BYTES == 12)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

template <int T, bool BIG, bool LOSS>
__global__ __launch_bounds__(512, 2) void als_solo_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_items, float* __restrict__ scratch,
                                                          const float* __restrict__ Qi, const int* __restrict__ defer, int* __restrict__ err) {
    static_assert(T == 4 || T == 3, "rows of 16 or 12 bytes per lane");
    extern __shared__ __attribute__((aligned(16))) char solo_lds[];
    using C = AlsSolo<T>;
    constexpr int VD = C::VD, NT = C::NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    float* ff_acc = reinterpret_cast<float*>(solo_lds);
    {
        const float sS2 = p.split[1];
        for (int idx = tid; idx < NT * 1024; idx += 512) {
            const int t = idx >> 10, rem = idx & 1023, e4 = rem >> 8, ln = (rem >> 2) & 63, e3 = rem & 3;
            const int a = als_tile_row<T>(t), b = als_tile_col<T>(t);
            const int row = a * 32 + e3 + 8 * e4 + 4 * (ln >> 5), cc = b * 32 + (ln & 31);
            ff_acc[idx] = sS2 * p.FF[row * VD + cc];
        }
    }
    __syncthreads();
    char* wl = solo_lds + C::FF_B + wv * C::WAVE_B;
    int* ks = reinterpret_cast<int*>(wl);
    char* rk = wl + C::KEY_B;
    float* nv = reinterpret_cast<float*>(wl + C::KEY_B + C::RK_B);
    float* pc = reinterpret_cast<float*>(wl + C::KEY_B + C::RK_B + C::NV_B);
    float* redtmp = reinterpret_cast<float*>(wl + C::KEY_B + C::RK_B + C::NV_B + C::VEC_B);
    double* lsum = reinterpret_cast<double*>(wl + C::KEY_B + C::RK_B + C::NV_B + C::VEC_B + C::RED_B);
    char* rowbuf = wl + C::KEY_B + C::RK_B + C::NV_B + C::VEC_B + C::RED_B + C::LOSS_B;
    lsum[lane] = 0.0;
    lsum[64 + lane] = 0.0;
    // LDS byte addresses of the DMA targets (the low 32 bits of a generic LDS pointer are its LDS offset), wave-uniform
    const unsigned rowbuf_lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(rowbuf)));
    const unsigned rk_lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(rk)));
    const unsigned nv_lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(nv)));

    const float sS = p.split[0], sI2 = p.split[2], wcut = p.split[3], alpha = p.alpha;
    const bool lossk = LOSS && p.compute_loss && p.axis == 1;
    const char* qbase = reinterpret_cast<const char*>(Qi);
    const unsigned lane_off = static_cast<unsigned>(col) * (4u * T);
    const int32_t* __restrict__ keys = p.keys;
    const float* __restrict__ vals = p.vals;
    const float* __restrict__ Pm = p.P;

    // ---- the work list, drawn in batches with the next ticket always under way (als_pc_producer's scheme) ----
    int b_base = 0, b_len = 0, b_pos = 0;
    int tk_v = 0, tk_rows = 0;
    const int batch_max = p.batch;
    auto draw = [&](int rows) {
        tk_rows = rows;
        if (lane == 0) tk_v = atomicAdd(p.ticket, rows);
    };
    bool list_end = false;
    draw(1);
    auto next_item = [&](int& row, int& kbeg, int& n, int& slot) -> bool {
        for (;;) {
            if (list_end) return false;
            if (b_pos >= b_len) {
                const int base = __builtin_amdgcn_readfirstlane(tk_v);
                int len = n_items - base;
                len = len < 0 ? 0 : (len > tk_rows ? tk_rows : len);
                b_base = base; b_len = len; b_pos = 0;
                if (len == 0) { list_end = true; return false; }
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

    // ---- cursors (all wave-uniform).  N = the group whose rows are on their way / in the row buffer; X = the item after N's ----
    int N_valid = 0, N_row = 0, N_kbeg = 0, N_n = 1, N_slot = -1, N_ng = 0, N_g = 0, N_buf = 0, N_par = 0;
    int X_valid = 0, X_row = 0, X_kbeg = 0, X_n = 1, X_slot = -1;
    auto fetch_next = [&]() {
        X_valid = next_item(X_row, X_kbeg, X_n, X_slot) ? 1 : 0;
        if (!X_valid) { X_row = 0; X_kbeg = 0; X_n = 1; X_slot = -1; }
    };
    // (key, value) of the chunk AFTER the one N stands in: DMA into the landing buffer the last staging did not read
    int rk_rd = 0;   // the landing buffer the next staging reads
    auto dma_keys = [&](int kbeg, int n, int chunk) {
        int kk = chunk * 64 + lane;
        kk = kk < n ? kk : n - 1;
        kk = kk < 0 ? 0 : kk;
        rk_rd ^= 1;
        solo_dma<4>(keys + kbeg + kk, rk_lds + rk_rd * 512);
        solo_dma<4>(vals + kbeg + kk, rk_lds + rk_rd * 512 + 256);
    };
    bool bad_weight = false;
    // N enters chunk `chunk` of its item: weigh and stage the chunk's keys (landed at least a trip ago), start the following chunk's
    auto stage_chunk = [&](int chunk) {
        int* kb = ks + N_buf * 192;
        if (N_valid) {
            const int pk_c = reinterpret_cast<const int*>(rk + rk_rd * 512)[lane];
            const float pk_v = reinterpret_cast<const float*>(rk + rk_rd * 512 + 256)[lane];
            const bool in = chunk * 64 + lane < N_n;
            const float ww = in ? alpha * pk_v : 0.f;
            const float ss = (ww > 0.f && ww <= wcut) ? sS * __builtin_amdgcn_sqrtf(ww) : 0.f;
            if (lossk && in) {   // constant and denominator of the loss (als_gram_kernel's header)
                const double w = static_cast<double>(ww);
                lsum[lane] += 1.0 + w;
                lsum[64 + lane] += w;
            }
            if (ww != 0.f && ss == 0.f) bad_weight = true;
            kb[lane] = in ? (BIG ? pk_c : pk_c * (4 * VD)) : 0;
            kb[64 + lane] = __builtin_bit_cast(int, ww);
            kb[128 + lane] = __builtin_bit_cast(int, ss);
        } else {
            kb[lane] = 0;
            kb[64 + lane] = 0;
            kb[128 + lane] = 0;
        }
        wave_lds_sync();
        if ((chunk + 1) * 64 < N_n) dma_keys(N_kbeg, N_n, chunk + 1);
        else dma_keys(X_kbeg, X_n, 0);
    };
    // the 16 rows of group N_g of N's item -> the row buffer: always issued (past the end: row 0, whose weight is 0)
    auto dma_rows = [&]() {
        const int* kbN = ks + N_buf * 192 + 16 * (N_g & 3) + 8 * half;
        const int4 c0 = *reinterpret_cast<const int4*>(kbN);
        const int4 c1 = *reinterpret_cast<const int4*>(kbN + 4);
        const int cid[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const char* src;
            if constexpr (BIG) src = qbase + (static_cast<size_t>(static_cast<unsigned>(cid[r])) * (4u * VD) + lane_off);
            else src = qbase + (static_cast<unsigned>(cid[r]) + lane_off);
            solo_dma<4 * T>(src, rowbuf_lds + r * (64 * 4 * T));
        }
    };
    // p0 and FF p0 (als_rowff_kernel) of N's item -> nv: read when the row opens, a trip later at the earliest
    auto dma_row_vectors = [&]() {
        const float* Pu0 = Pm + static_cast<size_t>(N_row) * VD;
        const float* Fu0 = p.F0 + static_cast<size_t>(N_valid ? N_row - p.start_x : 0) * VD;
        const int e1 = lane + 64 < VD ? lane + 64 : VD - 1;
        const unsigned dst = nv_lds + N_par * 1024;
        solo_dma<4>(Pu0 + lane, dst);
        solo_dma<4>(Pu0 + e1, dst + 256);
        solo_dma<4>(Fu0 + lane, dst + 512);
        solo_dma<4>(Fu0 + e1, dst + 768);
    };

    float p0cur[T], gpart[T], g1part[T];
#pragma unroll
    for (int b = 0; b < T; ++b) { p0cur[b] = 0.f; gpart[b] = 0.f; g1part[b] = 0.f; }

    // ---- prologue: the first item, its first chunk, its first group ----
    {
        int r0, k0, n0, s0;
        if (next_item(r0, k0, n0, s0)) {
            N_valid = 1; N_row = r0; N_kbeg = k0; N_n = n0; N_slot = s0; N_ng = (n0 + 15) >> 4; N_g = 0; N_buf = 0;
        }
        fetch_next();
        dma_keys(N_kbeg, N_n, 0);
        dma_row_vectors();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_chunk(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_rows();
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bool stored = false;   // the previous trip ended with the two row stores of a solve behind its DMA loads

    while (N_valid) {
        // ---- C = the group to work on now; its rows are in the row buffer once every DMA load issued so far has landed ----
        const int C_row = N_row, C_n = N_n, C_slot = N_slot, C_ng = N_ng, C_g = N_g, C_buf = N_buf, C_par = N_par;
        const bool solve_here = C_slot < 0;
        if (stored) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // (memory operations retire in order: the two stores came last)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stored = false;
        float raw[8][T];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (T == 4) {
                const float4 v = *reinterpret_cast<const float4*>(rowbuf + r * 1024 + lane * 16);
                raw[r][0] = v.x; raw[r][1] = v.y; raw[r][2] = v.z; raw[r][3] = v.w;
            } else {
                const float* v = reinterpret_cast<const float*>(rowbuf + r * 768 + lane * 12);
                raw[r][0] = v[0]; raw[r][1] = v[1]; raw[r][2] = v[2];
            }
        }
        if (C_g == 0) {   // the row opens: its vectors landed when N entered it
#pragma unroll
            for (int b = 0; b < T; ++b) { p0cur[b] = nv[C_par * 256 + b * 32 + col]; gpart[b] = 0.f; g1part[b] = 0.f; }
        }
        // ---- N moves on one group: whatever it enters is set up, then the DMA of its rows goes out -- the buffers are free once the reads
        // above have returned ----
        bool new_item = false;
        if (N_g + 1 < N_ng) {
            ++N_g;
            if ((N_g & 3) == 0) { N_buf ^= 1; stage_chunk(N_g >> 2); }
        } else {
            N_valid = X_valid; N_row = X_row; N_kbeg = X_kbeg; N_n = X_n; N_slot = X_slot;
            N_ng = (X_n + 15) >> 4; N_g = 0; N_buf ^= 1; N_par ^= 1;
            if (!N_valid) { N_ng = 1; N_n = 1; }
            fetch_next();
            stage_chunk(0);
            new_item = true;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (new_item) dma_row_vectors();
        dma_rows();
        if (C_g == 0) {   // accumulators of the new row
            if (solve_here) {
                const float* fl = ff_acc + lane * 4;   // M = FF + G: start from the FF tiles
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
        }
        // ---- VALU phase: residuals, h, the f16 pieces (als_pc_producer's arithmetic) ----
        u32x4 H[T], L[T];
        {
            const int* kb = ks + C_buf * 192 + 16 * (C_g & 3) + 8 * half;
            float y[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                y[r] = raw[r][0] * p0cur[0];
#pragma unroll
                for (int b = 1; b < T; ++b) y[r] = __builtin_fmaf(raw[r][b], p0cur[b], y[r]);
            }
            pc_sum8_over_half(y, redtmp, lane);
            const float4 w0 = *reinterpret_cast<const float4*>(kb + 64), w1 = *reinterpret_cast<const float4*>(kb + 64 + 4);
            const float wgt[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const int k0 = 16 * C_g + 8 * half;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float cial = __builtin_fmaf(wgt[r], y[r], -wgt[r]);   // alpha v (q.p0 - 1), als.cc:292-296
                const float one = (LOSS && lossk && k0 + r < C_n) ? 1.0f : 0.f;
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    gpart[b] = __builtin_fmaf(cial, raw[r][b], gpart[b]);
                    if (LOSS) g1part[b] = __builtin_fmaf(one, raw[r][b], g1part[b]);
                }
            }
            const float4 s0 = *reinterpret_cast<const float4*>(kb + 128), s1 = *reinterpret_cast<const float4*>(kb + 128 + 4);
            const float sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    unsigned h_, l_;
                    pc_split_pair(raw[2 * j2][b], sw[2 * j2], raw[2 * j2 + 1][b], sw[2 * j2 + 1], h_, l_);
                    H[b][j2] = h_;
                    L[b][j2] = l_;
                }
        }
        // ---- matrix phase: l h, h l, h h per tile (small terms first, as everywhere) ----
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            int t = 0;
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = a; b < T; ++b, ++t) {
                    const u32x4 Xo = pr == 0 ? L[a] : H[a];
                    const u32x4 Yo = pr == 1 ? L[b] : H[b];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, Xo), __builtin_bit_cast(f16x8_t, Yo), acc[t], 0, 0, 0);
                }
        }
        // ---- the row closes ----
        if (C_g == C_ng - 1) {
            float gs[T], g1s[T];
#pragma unroll
            for (int b = 0; b < T; ++b) {
                gs[b] = gpart[b] + __shfl_xor(gpart[b], 32, 64);
                g1s[b] = LOSS ? g1part[b] + __shfl_xor(g1part[b], 32, 64) : 0.f;
            }
            if (solve_here) {
                float* Pu = p.P + static_cast<size_t>(C_row) * VD;
                double nume = 0.0, deno = 0.0;
                float f0cur[T];   // FF p0 of the row (als_rowff_kernel), still in the row's landing buffer
#pragma unroll
                for (int b = 0; b < T; ++b) f0cur[b] = nv[C_par * 256 + 128 + b * 32 + col];
                wave_lds_sync();
                if (half == 0) {
#pragma unroll
                    for (int b = 0; b < T; ++b) { pc[b * 32 + col] = p0cur[b]; pc[VD + b * 32 + col] = 0.f; }
                }
                wave_lds_sync();
                als_ialspp_inreg<T>(acc, gs, g1s, f0cur, p, pc, pc + VD, pc + 2 * VD, lane, half, col, p.adaptive_reg ? static_cast<float>(C_n) : 1.0f, nume, deno, sI2);
                wave_lds_sync();
                static_assert((VD + 63) / 64 == 2, "the wait that opens a trip counts two row stores");
                for (int e = lane; e < VD; e += 64) Pu[e] = pc[e];
                stored = true;
                if (p.compute_loss && lane == 0) {
                    lsum[0] += nume;
                    lsum[64] += deno;
                }
            } else {   // chunk of a heavy row: tiles, h and g1 into the row's scratch slot for als_solve_kernel
                float* S = scratch + static_cast<size_t>(C_slot) * als_slot_floats(VD);
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
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__builtin_amdgcn_ballot_w64(bad_weight) != 0 && lane == 0) atomicOr(err, 2);
    if (p.compute_loss) {
        double nume_k = wave_sum_f64(lsum[lane]);
        double deno_k = wave_sum_f64(lsum[64 + lane]);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}

}  // namespace bfh
