// als_ts_kernel -- the in-place iALS++ row update (als.cc:211-358, block_size 32, d = 128) with the TILES of a row split over the two
// waves of a SIMD.  Included by als_kernels.hpp after als_pc.hpp (uses its hand-off helpers, pc_split_pair, pc_sum8_over_half, pc_load_row).
//
// Why (the round-4 / round-5 reviews).  In als_pc_kernel one wave of a SIMD prepares every group of 16 entries (gather, residual, h,
// f16 cut: ~300 instructions on ONE in-order wave) and the other owns all 160 accumulators and is idle two thirds of the time.  Here both waves
// do both jobs on the SAME row:
//   * wave 0 holds the tiles (0,0) (0,1) (0,2) (0,3) (1,1), wave 1 the tiles (1,2) (1,3) (2,2) (2,3) (3,3): 80 accumulators each, 15 matrix
//     instructions per group each;
//   * the groups of the pair's stream are prepared ALTERNATELY (wave 0 the even steps of a 64-entry chunk, wave 1 the odd ones): two
//     preparation streams per SIMD, each in the shadow of the other's memory waits and matrix instructions; the pieces travel through the same
//     3-slot LDS ring as in the pair kernel, but BOTH waves read every slot (wave 1 needs the blocks 1..3 only);
//   * the block recurrence at the end of a row is PASSED ALONG: wave 0 solves the blocks 0 and 1 from its diagonal tiles and hands delta_0,
//     delta_1 and the products of its tiles (0,2) (0,3) with delta_0 to wave 1, which solves the blocks 2 and 3 while wave 0 is already on the
//     next row.  The sums are formed in the order of als_ialspp_inreg (the per-half partial products travel uncombined), so given the same
//     accumulators and h the solved row has the same bits as the pair kernel's.
// Both waves walk the SAME work list in lockstep of control flow (wave 0 draws the tickets and posts them in an LDS mailbox); what a wave
// does not own it skips.  Time t of a wave: issue the row loads of position t + 6 (own positions only), prepare position t + 2 (own), consume
// position t (always).  Every wait is bounded (error flag -> exception), like the pair kernel's.
//
// Loss (als.cc:298-303, item half-epoch): per entry  w y^2 - 2 (1 + w) y + (1 + w)  with y = q.p0; the preparer has y and w in hand, so the
// row's  sum_k [w_k y_k^2 + 2 y_k]  is accumulated there (double, per lane) instead of being rebuilt from p0^T M p0 and g_1 at the end of the
// row: with h = sum w (y - 1) q the row terms are  p0.f0 - sum w y^2 + 2 p0.h - 2 sum y  (= p FF p + [p G p - 2 p.(g_w + g_1)]), the block
// separable parts p0.f0 and p0.h are added by the wave that owns the block.  Heavy-row chunks keep the g_1 sum for the scratch path.
#pragma once

namespace bfh {

template <int T>
struct AlsTs {
    static_assert(T == 4, "the tile split is written for vdim 128 (10 tiles = 5 + 5, blocks 0-1 | 2-3)");
    static constexpr int NT = T * (T + 1) / 2;
    static constexpr int NTW = NT / 2;                     // tiles per wave
    static constexpr int VD = 32 * T;
    static constexpr int NSLOT = 3;
    static constexpr int SLOT_B = 2 * T * 1024;            // H[0..T-1] | L[0..T-1], each 64 lanes x 16 B
    static constexpr int FF_B = NT * 4096;                 // the FF tiles in accumulator layout, scaled by S^2 (shared by the four pairs)
    static constexpr int KEY_W = 2 * 3 * 32 * 4;           // per wave: 2 buffers x (byte offset | weight | S sqrt(weight)) x the 32 entries of its two groups of a chunk
    static constexpr int PBOX_W = 2 * 64 * 4;              // per wave: 2 row parities x its h partial of the PARTNER's two blocks
    static constexpr int SHARE_B = 2 * (64 + 128) * 4;     // 2 row parities x (delta_0 | delta_1 | per-half products of (0,2), (0,3) with delta_0)
    static constexpr int PRIV_W = 128 * 4;                 // per wave: 16 reduction floats | 16 pad | 32 CG direction | 32 delta_2 | 32 spare
    static constexpr int FLAG_B = 128;                     // nine counters + the 8-deep ticket mailbox
    static constexpr int PAIR_B = NSLOT * SLOT_B + 2 * KEY_W + 2 * PBOX_W + SHARE_B + 2 * PRIV_W + FLAG_B;
    static constexpr int LDS_B = FF_B + 4 * PAIR_B + 64;
    static_assert(LDS_B <= 160 * 1024, "one workgroup per CU: the layout has to fit the 160 KB of LDS");
};
enum { TS_PUB = 0 /* +wave */, TS_CONS = 2 /* +wave */, TS_HDONE = 4 /* +wave */, TS_SOLVE0 = 6, TS_MAILN = 7, TS_ABORT = 8, TS_MAIL = 16 /* .. 23 */ };

// pc_wait_gt with the abort word at TS_ABORT
__device__ __forceinline__ bool ts_wait_gt(int* flag, int v, int& seen, int* flg) {
    if (seen - v > 0) return true;
    unsigned long long t0 = 0;
    for (int it = 0;; ++it) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (seen - v > 0) break;
        if ((it & 1023) == 1023) {
            const int ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flg + TS_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            if (ab != 0 || now - t0 > PC_WAIT_TICKS) {
                if (ab == 0) __hip_atomic_store(flg + TS_ABORT, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return true;
}

// three CG steps on (ms tile + reg I) x = b (als.cc:313-345) -- the statements of als_ialspp_inreg's inner loop, in its order
__device__ __forceinline__ float ts_cg32(const f32x16& tile, float bi, float reg, float ms, float cg_tol, float* pvs, int half, int col) {
    float xr = 0.f, rr = bi, pvr = bi;
    double rsold = static_cast<double>(wave_sum(half == 0 ? rr * rr : 0.f));
    if (rsold > static_cast<double>(cg_tol)) {
        for (int step = 0; step < 3; ++step) {
            wave_lds_sync();
            if (half == 0) pvs[col] = pvr;
            wave_lds_sync();
            float ap = als_tile_colpart(tile, pvs, half);
            ap = ms * (ap + __shfl_xor(ap, 32, 64));
            ap += reg * pvr;
            const float pap = wave_sum(half == 0 ? pvr * ap : 0.f);
            const float step_size = static_cast<float>(rsold) / pap;   // (one fp32 division: see als_ialspp_inreg)
            xr += step_size * pvr;
            rr -= step_size * ap;
            const double rsnew = static_cast<double>(wave_sum(half == 0 ? rr * rr : 0.f));
            if (rsnew < static_cast<double>(cg_tol)) break;
            pvr = rr + (static_cast<float>(rsnew) / static_cast<float>(rsold)) * pvr;
            rsold = rsnew;
        }
    }
    return xr;
}

struct TsChunk {   // one 64-entry chunk of one work item in the pair's stream (wave-uniform)
    int valid;
    int row, kbeg, n, slot;
    int chunk, ng;
    int rseq;      // running number of the work item
    int buf;       // key staging buffer (0 / 1)
};

template <int T, int ME, bool BIG, bool LOSS>
__device__ __forceinline__ void als_ts_wave(const AlsParams& p, const AlsWork* __restrict__ work, int n_items, float* __restrict__ scratch,
                                            const float* __restrict__ Qi, const int* __restrict__ defer, const float* ff_acc, char* pl, int* err, int lane) {
    using C = AlsTs<T>;
    constexpr int VD = C::VD, NTW = C::NTW;
    constexpr int B0 = ME == 0 ? 0 : 1;                    // first block whose pieces this wave's tiles need (wave 1: blocks 1..3)
    constexpr int OB = 2 * ME;                             // the two blocks this wave solves: OB, OB + 1
    const int half = lane >> 5, col = lane & 31;
    char* ring = pl;
    int* ks = reinterpret_cast<int*>(pl + C::NSLOT * C::SLOT_B + ME * C::KEY_W);
    float* pbox_mine = reinterpret_cast<float*>(pl + C::NSLOT * C::SLOT_B + 2 * C::KEY_W + ME * C::PBOX_W);          // written here, read by the partner
    const float* pbox_other = reinterpret_cast<const float*>(pl + C::NSLOT * C::SLOT_B + 2 * C::KEY_W + (1 - ME) * C::PBOX_W);
    float* share = reinterpret_cast<float*>(pl + C::NSLOT * C::SLOT_B + 2 * C::KEY_W + 2 * C::PBOX_W);
    float* priv = reinterpret_cast<float*>(pl + C::NSLOT * C::SLOT_B + 2 * C::KEY_W + 2 * C::PBOX_W + C::SHARE_B + ME * C::PRIV_W);
    float* redtmp = priv;            // 16 floats
    float* pvs = priv + 32;          // 32
    float* dl2 = priv + 64;          // 32 (wave 1)
    int* flg = reinterpret_cast<int*>(pl + C::PAIR_B - C::FLAG_B);
    const float sS = p.split[0], sI2 = p.split[2], wcut = p.split[3], alpha = p.alpha;
    const bool lossk = LOSS && p.compute_loss && p.axis == 1;
    const int dbg = p.debug;
    double nume_k = 0.0, deno_k = 0.0;   // per lane; reduced once at the end
    const char* qbase = reinterpret_cast<const char*>(Qi);
    const unsigned lane_off = static_cast<unsigned>(col) * (4u * T);
    const int32_t* __restrict__ keys = p.keys;
    const float* __restrict__ vals = p.vals;
    const float* __restrict__ Pm = p.P;

    // ---- the work list: wave 0 draws (in batches, the next one always under way), wave 1 follows through the mailbox ----
    int b_base = 0, b_len = 0, b_pos = 0;
    int tk_v = 0, tk_rows = 0, n_draws = 0, seen_mail = 0;
    const int batch_max = p.batch;
    bool list_end = false, ok = true;
    auto draw = [&](int rows) {
        tk_rows = rows;
        if (ME == 0 && lane == 0) tk_v = atomicAdd(p.ticket, rows);
    };
    draw(1);
    auto next_item = [&](int& row, int& kbeg, int& n, int& slot) -> bool {
        for (;;) {
            if (list_end || !ok) return false;
            if (b_pos >= b_len) {
                int base;
                if (ME == 0) {
                    base = __builtin_amdgcn_readfirstlane(tk_v);
                    if (lane == 0) flg[TS_MAIL + (n_draws & 7)] = base;
                    pc_publish(flg + TS_MAILN, n_draws + 1);
                } else {
                    if (!ts_wait_gt(flg + TS_MAILN, n_draws, seen_mail, flg)) { ok = false; return false; }
                    base = __builtin_amdgcn_readfirstlane(flg[TS_MAIL + (n_draws & 7)]);
                }
                ++n_draws;
                int len = n_items - base;
                len = len < 0 ? 0 : (len > tk_rows ? tk_rows : len);
                b_base = base; b_len = len; b_pos = 0;
                if (len == 0) { list_end = true; return false; }
                const int l0 = work[base].kend - work[base].kbeg;   // the list is sorted longest first: about 1024 entries per draw
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

    // ---- keys: one entry per lane, fetched a chunk ahead; the wave stages the 32 entries of ITS two groups ----
    int pk_c = 0;
    float pk_v = 0.f;
    auto load_keys = [&](int kbeg, int n, int chunk) {
        int kk = chunk * 64 + lane;
        kk = kk < n ? kk : n - 1;
        kk = kk < 0 ? 0 : kk;
        pk_c = keys[kbeg + kk];
        pk_v = vals[kbeg + kk];
    };

    TsChunk CA{}, CB{}, CC{};
    int nx_valid = 0, nx_row = 0, nx_kbeg = 0, nx_n = 1, nx_slot = -1;
    auto fetch_next = [&]() {
        nx_valid = next_item(nx_row, nx_kbeg, nx_n, nx_slot) ? 1 : 0;
        if (!nx_valid) { nx_row = 0; nx_kbeg = 0; nx_n = 1; nx_slot = -1; }
    };
    CA.n = 1; CB.n = 1; CC.n = 1;
    fetch_next();
    if (nx_valid) {
        CA.valid = 1; CA.row = nx_row; CA.kbeg = nx_kbeg; CA.n = nx_n; CA.slot = nx_slot;
        CA.chunk = 0; CA.ng = (nx_n + 15) >> 4; CA.rseq = 0; CA.buf = 0;
        fetch_next();
    }
    load_keys(CA.kbeg, CA.n, 0);
    auto advance = [&]() {
        CA.buf ^= 1;
        if (!CA.valid) return;
        if ((CA.chunk + 1) * 4 < CA.ng) { ++CA.chunk; return; }
        if (!nx_valid) { CA.valid = 0; CA.row = 0; CA.kbeg = 0; CA.n = 1; CA.chunk = 0; CA.ng = 0; return; }
        CA.row = nx_row; CA.kbeg = nx_kbeg; CA.n = nx_n; CA.slot = nx_slot;
        CA.chunk = 0; CA.ng = (nx_n + 15) >> 4; ++CA.rseq;
        fetch_next();
    };

    float raw[2][8][T];       // the rows of this wave's two groups in flight / in preparation
    float p0A[T], p0B[T], p0cur[T], gpart[T], g1part[T];
    float rvB[4], rvC[4];     // (p0, FF p0) of this wave's two blocks for the rows of CB / CC: wanted at the end of a row
    float hown[2][2];         // [row parity] this wave's h partial of its own two blocks (a row can close in PREP while the row before it has not closed in CONSUME)
    f32x16 acc[NTW];
#pragma unroll
    for (int b = 0; b < T; ++b) { p0A[b] = 0.f; p0B[b] = 0.f; p0cur[b] = 0.f; gpart[b] = 0.f; g1part[b] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { rvB[i] = 0.f; rvC[i] = 0.f; }
    hown[0][0] = 0.f; hown[0][1] = 0.f; hown[1][0] = 0.f; hown[1][1] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int b = 0; b < T; ++b) raw[s2][r][b] = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    int gprep = 0;            // live positions prepared so far (by either wave): the ring slot of the next one is gprep % 3
    int nown = 0;             // ... of them this wave's
    int npub = 0;             // ... of them published
    int gcons = 0;            // live positions consumed
    int ncons_w[2] = {0, 0};  // ... per owner
    int seen_pub = 0, seen_cons = 0, seen_hdone = 0, seen_solve0 = 0;
    bool bad_weight = false;

    auto load_group = [&](const int* kbA, int j, float (&rw)[8][T]) {   // j = 0 / 1: this wave's first / second group of the chunk
        const int4 c0 = *reinterpret_cast<const int4*>(kbA + 16 * j + 8 * half);
        const int4 c1 = *reinterpret_cast<const int4*>(kbA + 16 * j + 8 * half + 4);
        const int cid[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
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

    // ---- PREP, time t: position (Cc, sg) = t + 2.  Both waves walk every position; `own` (compile time) does the arithmetic ----
    auto prep_pos = [&](const TsChunk& Cc, int sg, bool own, int j, float (&q)[8][T]) {
        const int g = 4 * Cc.chunk + sg;
        const bool live = Cc.valid && g < Cc.ng && ok;
        if (!live) return;
        if (sg == 0 && Cc.chunk == 0) {   // the item opens
#pragma unroll
            for (int b = 0; b < T; ++b) { p0cur[b] = p0B[b]; gpart[b] = 0.f; g1part[b] = 0.f; }
        }
        if (own) {
            const int* kb = ks + Cc.buf * 96 + 16 * j + 8 * half;
            const float4 w0 = *reinterpret_cast<const float4*>(kb + 32), w1 = *reinterpret_cast<const float4*>(kb + 32 + 4);
            const float4 s0 = *reinterpret_cast<const float4*>(kb + 64), s1 = *reinterpret_cast<const float4*>(kb + 64 + 4);
            const float wgt[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const float sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            // the slot is free once the partner has read position gprep - 3 (this wave's own reads precede its stores in program order)
            if (!ts_wait_gt(flg + TS_CONS + (1 - ME), gprep - 3, seen_cons, flg)) { ok = false; return; }
            char* sl = ring + (gprep % C::NSLOT) * C::SLOT_B + lane * 16;
            if (dbg & 32) {   // (timing probe: no arithmetic)
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    *reinterpret_cast<u32x4*>(sl + b * 1024) = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(sl + (T + b) * 1024) = u32x4{0u, 0u, 0u, 0u};
                }
            } else {
                float y[8];   // als.cc:292-296: residual = Yui - 1 against the row at entry
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    y[r] = q[r][0] * p0cur[0];
#pragma unroll
                    for (int b = 1; b < T; ++b) y[r] = __builtin_fmaf(q[r][b], p0cur[b], y[r]);
                }
                pc_sum8_over_half(y, redtmp, lane);
                const int k0 = 16 * g + 8 * half;
                float la = 0.f, lb = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float cial = __builtin_fmaf(wgt[r], y[r], -wgt[r]);   // alpha v (q.p0 - 1)
                    const float one = (LOSS && lossk && k0 + r < Cc.n) ? 1.0f : 0.f;
                    if (LOSS) {
                        la = __builtin_fmaf(wgt[r] * y[r], y[r], la);
                        lb = __builtin_fmaf(one, y[r], lb);
                    }
#pragma unroll
                    for (int b = 0; b < T; ++b) {
                        gpart[b] = __builtin_fmaf(cial, q[r][b], gpart[b]);
                        if (LOSS) g1part[b] = __builtin_fmaf(one, q[r][b], g1part[b]);
                    }
                }
                // in-place rows: - sum w y^2 - 2 sum y (see the header); every lane of a half holds the same eight y: one lane per half counts
                if (LOSS && lossk && Cc.slot < 0 && col == 0) nume_k -= static_cast<double>(la) + 2.0 * static_cast<double>(lb);
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    u32x4 Hb, Lb;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        unsigned h_, l_;
                        pc_split_pair(q[2 * j2][b], sw[2 * j2], q[2 * j2 + 1][b], sw[2 * j2 + 1], h_, l_);
                        Hb[j2] = h_;
                        Lb[j2] = l_;
                    }
                    *reinterpret_cast<u32x4*>(sl + b * 1024) = Hb;
                    *reinterpret_cast<u32x4*>(sl + (T + b) * 1024) = Lb;
                }
            }
            ++nown;
        }
        ++gprep;
        if (g == Cc.ng - 1) {   // the item closes: this wave's h partial -- the partner's blocks to the box, its own two kept; heavy-row chunks: to the slot
            float gs[T], g1s[T];
#pragma unroll
            for (int b = 0; b < T; ++b) {   // the two halves hold the k-parities of the same element
                gs[b] = gpart[b] + __shfl_xor(gpart[b], 32, 64);
                g1s[b] = LOSS ? g1part[b] + __shfl_xor(g1part[b], 32, 64) : 0.f;
            }
            if (Cc.slot < 0) {
                float* bx = pbox_mine + (Cc.rseq & 1) * 64;
                if (half == 0) {
                    bx[col] = gs[2 - OB];
                    bx[32 + col] = gs[3 - OB];
                }
                if (Cc.rseq & 1) { hown[1][0] = gs[OB]; hown[1][1] = gs[OB + 1]; }
                else { hown[0][0] = gs[OB]; hown[0][1] = gs[OB + 1]; }
                pc_publish(flg + TS_HDONE + ME, Cc.rseq + 1);
            } else if (half == 0) {
                float* gdst = scratch + static_cast<size_t>(Cc.slot) * als_slot_floats(VD) + VD * VD + col;
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    atomic_add_f32(gdst + b * 32, gs[b] * p.out_scale);
                    if (LOSS && lossk) atomic_add_f32(gdst + VD + b * 32, g1s[b]);
                }
            }
        }
    };

    // ---- CONSUME, time t: position (Cc, sg) ----
    auto consume_pos = [&](const TsChunk& Cc, int sg, const float (&rv)[4]) {
        const int g = 4 * Cc.chunk + sg;
        const bool live = Cc.valid && g < Cc.ng && ok;
        if (!live) return;
        const int owner = sg & 1;
        const bool solve_here = Cc.slot < 0;
        if (g == 0) {   // the row opens: M = FF + G starts from the FF tiles (accumulator layout, scaled by S^2); chunks of heavy rows from zero
            if (solve_here && !(dbg & 2)) {
                const float* fl = ff_acc + lane * 4;
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 v = *reinterpret_cast<const float4*>(fl + ((NTW * ME + t) * 4 + e4) * 256);
                        acc[t][4 * e4 + 0] = v.x; acc[t][4 * e4 + 1] = v.y; acc[t][4 * e4 + 2] = v.z; acc[t][4 * e4 + 3] = v.w;
                    }
            } else {
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            }
        }
        if (owner != ME) {
            if (!ts_wait_gt(flg + TS_PUB + owner, ncons_w[owner], seen_pub, flg)) { ok = false; return; }
        }
        u32x4 H[T], L[T];
        {
            const char* sl = ring + (gcons % C::NSLOT) * C::SLOT_B + lane * 16;
#pragma unroll
            for (int b = B0; b < T; ++b) {
                L[b] = *reinterpret_cast<const u32x4*>(sl + (T + b) * 1024);
                H[b] = *reinterpret_cast<const u32x4*>(sl + b * 1024);
            }
            if (B0 > 0) { H[0] = H[1]; L[0] = L[1]; }   // (never used: wave 1 has no tile in block row / column 0)
        }
        // the reads have landed (and with them every earlier store of this wave): the slot is released, this wave's last preparation published
        pc_publish(flg + TS_CONS + ME, gcons + 1);
        if (npub != nown) {
            npub = nown;
            __hip_atomic_store(flg + TS_PUB + ME, nown, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (!(dbg & 16)) {
            // small terms first: l h, h l, h h -- per tile the order of the pair kernel.  (Compile-time tile indices: with run-time ones the
            // operand arrays go to scratch and every matrix instruction waits for two scratch loads -- first contact, 10.2 ms per epoch.)
            als_static_for<3 * NTW>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
                constexpr int pr = i / NTW, t = i % NTW;
                constexpr int a = als_tile_row<T>(NTW * ME + t), b = als_tile_col<T>(NTW * ME + t);
                const u32x4 X = pr == 0 ? L[a] : H[a];
                const u32x4 Y = pr == 1 ? L[b] : H[b];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, X), __builtin_bit_cast(f16x8_t, Y), acc[t], 0, 0, 0);
            });
        }
        ++gcons;
        ++ncons_w[owner];
        if (g != Cc.ng - 1) return;
        // ================= the row closes =================
        if (!solve_here) {   // chunk of a heavy row: this wave's tiles are summed in the row's scratch slot (zeroed by the host) for als_solve_kernel
            float* S = scratch + static_cast<size_t>(Cc.slot) * als_slot_floats(VD);
            float* Sl = S + half * 4 * VD + col;
            const float osc = p.out_scale * sI2;
            als_static_for<NTW>([&](auto Tc) {
                constexpr int t = decltype(Tc)::value;
                constexpr int a = als_tile_row<T>(NTW * ME + t), b = als_tile_col<T>(NTW * ME + t);
#pragma unroll
                for (int e = 0; e < 16; ++e) atomic_add_f32(Sl + (a * 32 + (e & 3) + 8 * (e >> 2)) * VD + b * 32, acc[t][e] * osc);
            });
            return;
        }
        const int par = Cc.rseq & 1;
        if (!ts_wait_gt(flg + TS_HDONE + (1 - ME), Cc.rseq, seen_hdone, flg)) { ok = false; return; }
        float hh[2];
        {
            const float* bx = pbox_other + par * 64;
            hh[0] = (par ? hown[1][0] : hown[0][0]) + bx[col];
            hh[1] = (par ? hown[1][1] : hown[0][1]) + bx[32 + col];
        }
        const float ada = p.adaptive_reg ? static_cast<float>(Cc.n) : 1.0f;
        const float reg = p.reg;
        float* shp = share + par * 192;   // delta_0 | delta_1 | (0,2).delta_0 per half | (0,3).delta_0 per half
        float xr0 = 0.f, xr1 = 0.f;
        if (!(dbg & 1)) {
            if (ME == 0) {
                // block 0: gradient f0 + h + reg p (delta is still zero), CG on the tile (0,0)
                const float bi0 = rv[2] + hh[0] + sI2 * 0.f + reg * rv[0];
                xr0 = ts_cg32(acc[0], bi0, reg, sI2, p.cg_tol, pvs, half, col);
                wave_lds_sync();
                if (half == 0) shp[col] = -xr0;
                wave_lds_sync();
                // block 1: + (M delta)_1 = tile (0,1)^T delta_0
                float md = als_tile_colpart(acc[1], shp, half);
                md = sI2 * (md + __shfl_xor(md, 32, 64));
                const float bi1 = rv[3] + hh[1] + md + reg * rv[1];
                xr1 = ts_cg32(acc[4], bi1, reg, sI2, p.cg_tol, pvs, half, col);
                // what the blocks 2 and 3 need of this wave's tiles, per half and uncombined (wave 1 continues the sums in als_ialspp_inreg's order)
                const float m2 = als_tile_colpart(acc[2], shp, half), m3 = als_tile_colpart(acc[3], shp, half);
                wave_lds_sync();
                if (half == 0) shp[32 + col] = -xr1;
                shp[64 + lane] = m2;
                shp[128 + lane] = m3;
                pc_publish(flg + TS_SOLVE0, Cc.rseq + 1);
            } else {
                if (!ts_wait_gt(flg + TS_SOLVE0, Cc.rseq, seen_solve0, flg)) { ok = false; return; }
                // block 2: tiles (0,2) [wave 0's share] and (1,2)
                float md = shp[64 + lane] + als_tile_colpart(acc[0], shp + 32, half);
                md = sI2 * (md + __shfl_xor(md, 32, 64));
                const float bi2 = rv[2] + hh[0] + md + reg * rv[0];
                xr0 = ts_cg32(acc[2], bi2, reg, sI2, p.cg_tol, pvs, half, col);
                wave_lds_sync();
                if (half == 0) dl2[col] = -xr0;
                wave_lds_sync();
                // block 3: tiles (0,3) [wave 0's share], (1,3), (2,3)
                float md3 = shp[128 + lane] + als_tile_colpart(acc[1], shp + 32, half);
                md3 += als_tile_colpart(acc[3], dl2, half);
                md3 = sI2 * (md3 + __shfl_xor(md3, 32, 64));
                const float bi3 = rv[3] + hh[1] + md3 + reg * rv[1];
                xr1 = ts_cg32(acc[4], bi3, reg, sI2, p.cg_tol, pvs, half, col);
            }
            // als.cc:346: the two solved blocks of the row, one 256-byte store (half 0: block OB, half 1: block OB + 1)
            float* Pu = p.P + static_cast<size_t>(Cc.row) * VD + (OB + half) * 32 + col;
            *Pu = half ? (rv[1] - xr1) : (rv[0] - xr0);
        }
        if (p.compute_loss && half == 0) {   // row-level loss terms of this wave's blocks (als.cc:288-309 on the row at entry; see the header)
            double t = static_cast<double>(ada * reg * (rv[0] * rv[0] + rv[1] * rv[1]));
            if (p.axis == 1) {
                t += static_cast<double>(rv[0] * rv[2] + rv[1] * rv[3]) + 2.0 * static_cast<double>(rv[0] * hh[0] + rv[1] * hh[1]);
                if (ME == 0 && col == 0) deno_k += static_cast<double>(p.op_rows);
            }
            nume_k += t;
        }
    };

    // Everything fetched so far is waited for HERE (see als_pc_producer)
    asm volatile("" ::"v"(pk_c), "v"(pk_v), "v"(tk_v));
    do {
        // ---- the trip loads chunk CA, prepares the positions of CB, consumes (CC, 2) (CC, 3) (CB, 0) (CB, 1) ----
        int* kbA = ks + CA.buf * 96;
        {
            // this wave's entries of the chunk: groups ME and ME + 2, i.e. the lanes whose bit 4 equals ME; entry index among its 32: (lane >> 5) * 16 + (lane & 15)
            const bool mine = ((lane >> 4) & 1) == ME;
            const int j32 = ((lane >> 5) << 4) | (lane & 15);
            if (CA.valid) {
                const bool in = CA.chunk * 64 + lane < CA.n;
                const float ww = in ? alpha * pk_v : 0.f;     // padding lanes: row 0 of the other factor with weight 0
                const float ss = (ww > 0.f && ww <= wcut) ? sS * __builtin_amdgcn_sqrtf(ww) : 0.f;
                if (mine) {
                    if (lossk && in) {   // constant and denominator of the loss (als_gram_kernel's header)
                        const double w = static_cast<double>(ww);
                        deno_k += w;
                        nume_k += 1.0 + w;
                    }
                    if (ww != 0.f && ss == 0.f) bad_weight = true;   // a weight the scan should have routed elsewhere
                    kbA[j32] = in ? (BIG ? pk_c : pk_c * (4 * VD)) : 0;
                    kbA[32 + j32] = __builtin_bit_cast(int, ww);
                    kbA[64 + j32] = __builtin_bit_cast(int, ss);
                }
            } else if (mine) {
                kbA[j32] = 0;
            }
        }
        wave_lds_sync();
        if ((CA.chunk + 1) * 64 < CA.n) load_keys(CA.kbeg, CA.n, CA.chunk + 1);
        else load_keys(nx_kbeg, nx_n, 0);
        {
            const float* Pu0 = Pm + static_cast<size_t>(CA.row) * VD;
#pragma unroll
            for (int b = 0; b < T; ++b) p0A[b] = Pu0[b * 32 + col];
            // (p0, FF p0) of this wave's blocks for CB's row: at hand when the row closes, two to six steps from here
            const float* Pb = Pm + static_cast<size_t>(CB.row) * VD + OB * 32 + col;
            const float* Fb = p.F0 + static_cast<size_t>(CB.valid ? CB.row - p.start_x : 0) * VD + OB * 32 + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) rvC[i] = rvB[i];
            rvB[0] = Pb[0]; rvB[1] = Pb[32]; rvB[2] = Fb[0]; rvB[3] = Fb[32];
        }
        // step 0: wave 0 prepares (CB, 0) [its set 0] and reloads the set with (CA, 0); both consume (CC, 2)
        if (ME == 0) { prep_pos(CB, 0, true, 0, raw[0]); load_group(kbA, 0, raw[0]); } else { prep_pos(CB, 0, false, 0, raw[0]); }
        consume_pos(CC, 2, rvC);
        // step 1: wave 1 prepares (CB, 1) and reloads with (CA, 1); both consume (CC, 3)
        if (ME == 1) { prep_pos(CB, 1, true, 0, raw[0]); load_group(kbA, 0, raw[0]); } else { prep_pos(CB, 1, false, 0, raw[0]); }
        consume_pos(CC, 3, rvC);
        // step 2
        if (ME == 0) { prep_pos(CB, 2, true, 1, raw[1]); load_group(kbA, 1, raw[1]); } else { prep_pos(CB, 2, false, 1, raw[1]); }
        consume_pos(CB, 0, rvB);
        // step 3
        if (ME == 1) { prep_pos(CB, 3, true, 1, raw[1]); load_group(kbA, 1, raw[1]); } else { prep_pos(CB, 3, false, 1, raw[1]); }
        consume_pos(CB, 1, rvB);
#pragma unroll
        for (int b = 0; b < T; ++b) p0B[b] = p0A[b];
        CC = CB;
        CB = CA;
        advance();
    } while (ok && (CA.valid | CB.valid | CC.valid));
    if (npub != nown) pc_publish(flg + TS_PUB + ME, nown);
    {
        const int ab = __hip_atomic_load(flg + TS_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((ab != 0 || !ok) && lane == 0) atomicOr(err, 1);
    }
    if (__builtin_amdgcn_ballot_w64(bad_weight) != 0 && lane == 0) atomicOr(err, 2);
    if (p.compute_loss) {
        nume_k = wave_sum_f64(nume_k);
        deno_k = wave_sum_f64(deno_k);
        if (lane == 0) {
            if (nume_k != 0.0) atomicAdd(p.loss, nume_k);
            if (deno_k != 0.0) atomicAdd(p.loss + 1, deno_k);
        }
    }
}

// One 512-thread workgroup per CU: four pairs, the two waves of a pair on one SIMD where the placement allows (any pairing is correct).
// err[0]: bit 0 a wait timed out, bit 1 a weight outside the f16 path reached the kernel; err[1]: workgroups with one pair per SIMD.
template <int T, bool BIG, bool LOSS>
__global__ __launch_bounds__(512, 2) void als_ts_kernel(AlsParams p, const AlsWork* __restrict__ work, int n_items, float* __restrict__ scratch,
                                                        const float* __restrict__ Qi, const int* __restrict__ defer, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char ts_lds[];
    using C = AlsTs<T>;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (als_debug bit 1024: the shader clock this workgroup sees over the kernel -- s_memtime counts core cycles, s_memrealtime the constant 100 MHz)
    const bool clk_probe = (p.debug & 1024) && blockIdx.x == 0 && tid == 0;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_probe) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    float* ff_acc = reinterpret_cast<float*>(ts_lds);
    int* role_tab = reinterpret_cast<int*>(ts_lds + C::FF_B + 4 * C::PAIR_B);   // [0..7] pair * 2 + wave-in-pair, [8..15] SIMD id of wave w
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
    if (tid < 128) reinterpret_cast<int*>(ts_lds + C::FF_B + (tid >> 5) * C::PAIR_B + C::PAIR_B - C::FLAG_B)[tid & 31] = 0;
    __syncthreads();
    if (tid == 0) {
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
    char* pl = ts_lds + C::FF_B + (rl >> 1) * C::PAIR_B;
    if (rl & 1) als_ts_wave<T, 1, BIG, LOSS>(p, work, n_items, scratch, Qi, defer, ff_acc, pl, err, lane);
    else als_ts_wave<T, 0, BIG, LOSS>(p, work, n_items, scratch, Qi, defer, ff_acc, pl, err, lane);
    if (clk_probe) {   // wave 0 of workgroup 0 at the end of ITS stream (the kernel's tail may run a little longer on other waves)
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk_c0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
        err[2] = static_cast<int>(dc & 0xffffffffull); err[3] = static_cast<int>(dc >> 32);
        err[4] = static_cast<int>(dr & 0xffffffffull); err[5] = static_cast<int>(dr >> 32);
    }
}

}  // namespace bfh
