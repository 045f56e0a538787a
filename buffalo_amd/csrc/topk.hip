// Top-k selection over factor products on gfx950 -- kernels, handle, C ABI.
//
// Reference semantics: parallel::dot_topn and parallel::quickselect
// (/root/reference/buffalo/parallel/_core.hpp:37-142) -- the consumer of P, Q right after training
// (parallel/base.py:21-60, evaluate/base.py:31-42,80-82).
//
// Two kernels per batch of queries:
//   topk_scores_kernel  S[b][j] = P[q_b] . Q[j] on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32
//                       products, fp32 accumulation).  A wave owns 32 queries; its A operands (the query
//                       rows, <= 128 columns per K-chunk) stay in registers while it sweeps item tiles of
//                       32 rows whose B operands stream in as float4s; the four waves of a block sweep
//                       the same tiles for different queries, so each Q row leaves L2 once per 128
//                       queries.  Lane (i, h) supplies columns [h*W/2, (h+1)*W/2) of row i to both
//                       operands -- the MFMA sums over k in any order, so the two half-waves simply take
//                       the two halves of the chunk (contiguous float4 loads, no transposition).
//   topk_select_kernel  one block per query row: 4-pass radix select (8 bits per pass, LDS histogram)
//                       of the k-th largest admissible score, ordered collection of the boundary ties,
//                       bitonic sort of the <= k survivors in LDS by (score desc, index desc).
// Selection is exact (bit-level on the scores the first kernel produced); the scores differ from the
// reference's Eigen dot products only by fp32 summation order.
#include <cfloat>

#include "common.hpp"

namespace bfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TOPK_MAX_K = 16384;

// ------------------------------------------------------------------------------------------------
// Candidate matrix -> MFMA operand order, once per call (14 MB at ML-20M): Qp[((t*16 + v)*64 + lane)] (float4) =
// Q[32 t + (lane&31)][kc + (lane>>5)*W/2 + 4v .. +3].  A wave's B-operand load in the score kernel is then ONE
// contiguous KiB instead of 64 row-strided 16-byte pieces in 64 different cache lines -- with the strided form the
// texture-address unit of the CU was as busy as the matrix cores.  Rows beyond q_rows and float4s beyond the chunk
// are zero.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_pack_kernel(const float* __restrict__ Q, int q_rows, int ld, int kc, int W, float4* __restrict__ Qp,
                                                        int n_tiles) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;   // (t, v, lane)
    if (idx >= static_cast<int64_t>(n_tiles) * 16 * 64) return;
    const int lane = static_cast<int>(idx & 63), v = static_cast<int>((idx >> 6) & 15), t = static_cast<int>(idx >> 10);
    const int j = t * 32 + (lane & 31);
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < q_rows && v < W / 8) out = *reinterpret_cast<const float4*>(Q + static_cast<int64_t>(j) * ld + kc + (lane >> 5) * (W / 2) + 4 * v);
    Qp[idx] = out;
}

// ------------------------------------------------------------------------------------------------
// S[b][j] (+)= sum_{c in [kc, kc+W)} A[row(b)][c] * Q[j][c]
//   A row of query b: P + (qidx ? qidx[q0+b] : q0+b) * ld.  W = min(128, d_pad - kc), W % 8 == 0.
// grid.x = item-tile groups, grid.y = query blocks of 128; block = 256 threads.
// ------------------------------------------------------------------------------------------------
// The fused form (FILTER): the tile's scores never reach HBM.  Every query row carries a threshold -- the kk-th best
// admissible score of a SAMPLE of the columns (the first C0), i.e. a lower bound of the final kk-th best -- and the
// epilogue appends the (column, score) pairs at or above it to the row's candidate segment of this tile group: a few
// hundred of 27 K columns.  Slots come from a per-wave LDS counter (one wave owns a row within a tile group, so no
// global atomics); a segment that overflows is noticed by the select kernel, which sends the row to the dense path.
struct FilterArgs {
    const float* thr;       // [nq] batch-local thresholds on score (+ bias)
    const float* Qb;        // nullable: added to every score before the comparison (as topk_select_kernel does)
    const uint32_t* pool;   // nullable bitmap over columns: columns outside it are never candidates
    uint2* cand;            // [(b * gridDim.x + blockIdx.x) * cap_seg + slot] = (column, bits of the raw score)
    int* cand_cnt;          // [b * gridDim.x + blockIdx.x] candidates seen (> cap_seg: overflow)
    int cap_seg;
    int t_first;            // first tile of the sweep: the sampled columns in front of it reach the selection from their dense scores
};

// FULL: the K-chunk is a whole 128 columns (nv == 16): no per-float4 guards, straight-line MFMA stream
template <bool FULL, bool FILTER>
__global__ __launch_bounds__(256, 3) void topk_scores_kernel(const float* __restrict__ P, const int32_t* __restrict__ qidx, int q0, int nq,
                                                             const float4* __restrict__ Qp, int q_rows, int ld, int kc, int W, float* __restrict__ S,
                                                             size_t ld_s, int tiles_per_block, int accumulate, FilterArgs f) {
    __shared__ int s_cnt[FILTER ? 4 : 1][32];
    __shared__ float s_thr[FILTER ? 4 : 1][32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b0q = (blockIdx.y * 4 + wv) * 32;   // first query (batch-local) of this wave
    if (b0q >= nq) return;
    if constexpr (FILTER) {
        if (lane < 32) {
            s_cnt[wv][lane] = 0;
            s_thr[wv][lane] = b0q + lane < nq ? f.thr[b0q + lane] : __builtin_inff();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int nv = W / 8;                        // float4s per lane and row
    const int koff = kc + half * (W / 2);
    // A operands: query row b0 + col, this half's columns
    int bq = b0q + col;
    if (bq >= nq) bq = nq - 1;                   // clamped rows compute garbage that is never stored
    const int64_t prow = qidx ? qidx[q0 + bq] : (q0 + bq);
    const float4* ap = reinterpret_cast<const float4*>(P + prow * ld + koff);
    float4 a[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) a[v] = (FULL || v < nv) ? ap[v] : make_float4(0.f, 0.f, 0.f, 0.f);

    const int n_tiles = (q_rows + 31) / 32;
    const int t_begin = (FILTER ? f.t_first : 0) + blockIdx.x * tiles_per_block;
    int t_end = t_begin + tiles_per_block;
    if (t_end > n_tiles) t_end = n_tiles;
    // B operands of a tile in two halves of 8 float4s: the second half of tile t and the first half of tile t+1 are in
    // flight while the first / second half's 32 MFMAs run (no wave waits for a whole tile's loads with an idle pipe)
    auto tile_row = [&](int t) { return Qp + (static_cast<int64_t>(t) * 16 * 64 + lane); };   // float4 v of the tile at [v * 64]
    float4 b0[8], b1[8];
    if (t_begin < t_end) {
        const float4* bp = tile_row(t_begin);
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (FULL || v < nv) b0[v] = bp[v * 64];
    }
    for (int t = t_begin; t < t_end; ++t) {
        const bool jok = t * 32 + col < q_rows;
        const float4* bp = tile_row(t);
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (FULL || 8 + v < nv) b1[v] = bp[(8 + v) * 64];
        f32x16 acc;
        float* Sl = FILTER ? nullptr : S + static_cast<size_t>(b0q + 4 * half) * ld_s + t * 32 + col;   // C layout: row (e&3)+8(e>>2)+4half, col lane&31
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2);
            if constexpr (FILTER) acc[e] = 0.f;
            else acc[e] = (accumulate && jok && b0q + 4 * half + r < nq) ? Sl[static_cast<size_t>(r) * ld_s] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (FULL || v < nv) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].x, b0[v].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].y, b0[v].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].z, b0[v].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].w, b0[v].w, acc, 0, 0, 0);
            }
        if (t + 1 < t_end) {
            const float4* bn = tile_row(t + 1);
#pragma unroll
            for (int v = 0; v < 8; ++v)
                if (FULL || v < nv) b0[v] = bn[v * 64];
        }
#pragma unroll
        for (int v = 0; v < 8; ++v)
            if (FULL || 8 + v < nv) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 + v].x, b1[v].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 + v].y, b1[v].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 + v].z, b1[v].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 + v].w, b1[v].w, acc, 0, 0, 0);
            }
        if constexpr (FILTER) {
            const int j = t * 32 + col;
            bool colok = jok;
            if (f.pool && jok) colok = ((f.pool[j >> 5] >> (j & 31)) & 1u) != 0u;
            const float qb = (f.Qb && jok) ? f.Qb[j] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = 4 * half + (e & 3) + 8 * (e >> 2);
                const float sc = f.Qb ? acc[e] + qb : acc[e];   // the very sum the select kernel forms
                if (colok && sc >= s_thr[wv][rr]) {           // rows beyond nq carry +inf
                    const int slot = atomicAdd(&s_cnt[wv][rr], 1);
                    const float raw = acc[e];   // (a bit_cast applied to the vector element itself reads element 0)
                    if (slot < f.cap_seg)
                        f.cand[(static_cast<size_t>(b0q + rr) * gridDim.x + blockIdx.x) * f.cap_seg + slot] =
                            make_uint2(static_cast<uint32_t>(j), __float_as_uint(raw));
                }
            }
        } else if (jok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2);
                if (b0q + 4 * half + r < nq) Sl[static_cast<size_t>(r) * ld_s] = acc[e];
            }
        }
    }
    if constexpr (FILTER) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 32 && b0q + lane < nq) f.cand_cnt[static_cast<size_t>(b0q + lane) * gridDim.x + blockIdx.x] = s_cnt[wv][lane];
    }
}

// thr[b] = the kk-th best admissible score of the sampled columns (row q0 + b of the select output), or "everything":
// with the admission rule only scores > FLT_MIN can be listed, so FLT_MIN itself is a valid bound then
__global__ void topk_thr_kernel(const int32_t* __restrict__ keys, const float* __restrict__ scores, int q0, int nb, int k, int kk, int rule_flt_min,
                                float* __restrict__ thr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const size_t at = static_cast<size_t>(q0 + b) * k + (kk - 1);
    thr[b] = (kk > 0 && keys[at] >= 0) ? scores[at] : (rule_flt_min ? FLT_MIN : -__builtin_inff());
}

// order-preserving map: smaller key <=> larger score (-0 and +0 coincide)
__device__ __forceinline__ uint32_t desc_key(float s) {
    s += 0.0f;
    uint32_t u = __builtin_bit_cast(uint32_t, s);
    u = (u >> 31) ? ~u : (u | 0x80000000u);
    return ~u;
}
__device__ __forceinline__ float key_score(uint32_t k) {
    const uint32_t u = ~k;
    return __builtin_bit_cast(float, (u >> 31) ? (u ^ 0x80000000u) : ~u);
}

struct SelectArgs {
    const float* S;          // [rows, ld_s]
    size_t ld_s;
    int cols;
    const float* Qb;         // nullable: added to every score
    const uint32_t* pool;    // nullable bitmap over columns
    const int32_t* self_idx; // nullable: column excluded for row b (dot_topn with P == Q)
    int q0;                  // self_idx / output row offset of S row 0
    int rule_flt_min;        // dot_topn: only scores > FLT_MIN are admissible
    int k, kk;               // output width, min(k, cols[, pool_size])
    int32_t* out_keys;       // [.., k] (row q0 + b)
    float* out_scores;       // nullable (quickselect)
    int p2;                  // power of two >= kk: sort buffer entries
    int cand_cap;            // entries of the candidate buffer behind the sort buffer (0: multi-pass path only)
    const int32_t* out_row;  // nullable: output row of S row b (else q0 + b); self_idx is then indexed by b
    // list mode (the fused path): the row is not a dense score row but the candidate segments topk_scores_kernel<.., FILTER>
    // wrote -- every admissible column at or above a lower bound of the kk-th best score, in no particular order
    const uint2* cand;       // nullable: [(b * n_seg + g) * cap_seg + slot] = (column, bits of the raw score)
    const int* cand_cnt;     // [b * n_seg + g]
    int n_seg, cap_seg;
    int list_cap;            // entries of the LDS list behind the candidate buffer
    int* redo;               // [0]: rows sent to the dense path (a segment or the list overflowed), [1 + i]: their b
    const int* row_list;     // nullable: block x works on row row_list[x] (the rows topk_list_wave_kernel passed on)
    int* general;            // topk_list_wave_kernel: [0] rows passed on to topk_select_kernel (ties at the k-th place), [1 + i]: their b
    float* thr;              // topk_thr_wave_kernel: [b] the kk-th best admissible score of the row, or "everything"
    // the sampled columns' own candidates (written by topk_thr_wave_kernel from the dense sample scores; the filtered sweep
    // then starts behind the sample): one more segment per row for the list selection.  Null: the sweep covered every column.
    uint2* s0_cand;          // [b * s0_cap + slot]
    int* s0_cnt;             // [b] (> s0_cap: overflow)
    int s0_cap;
};

__global__ __launch_bounds__(256) void topk_select_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sel[];   // p2 sort entries, then cand_cap candidates
    __shared__ int hist[4096];
    __shared__ int part[256];
    __shared__ int s_misc[8];   // 0: chosen bin, 1: remaining, 2: n_gt slots, 3: run_eq, 4..7: wave eq counts / fast-path counters
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = a.row_list ? a.row_list[blockIdx.x] : blockIdx.x;
    const bool list = a.cand != nullptr;
    const float* row = list ? nullptr : a.S + static_cast<size_t>(b) * a.ld_s;
    const int orow = a.out_row ? a.out_row[b] : a.q0 + b;
    const int self = a.self_idx ? a.self_idx[a.out_row ? b : a.q0 + b] : -1;
    uint2* lst = reinterpret_cast<uint2*>(sel + a.p2 + a.cand_cap);
    int cols = a.cols;   // positions the passes run over: columns of the dense row, or entries of the list
    if (list) {
        // gather the segments into one LDS list; a row whose segments or list overflowed is redone densely by the host
        if (tid == 0) {
            int tot = 0, over = 0;
            for (int g = 0; g < a.n_seg; ++g) {
                const int c = a.cand_cnt[static_cast<size_t>(b) * a.n_seg + g];
                over |= c > a.cap_seg;
                tot += c;
            }
            if (a.s0_cand) {
                const int c = a.s0_cnt[b];
                over |= c > a.s0_cap;
                tot += c;
            }
            s_misc[6] = tot;
            s_misc[7] = (over || tot > a.list_cap) ? 1 : 0;
        }
        __syncthreads();
        cols = s_misc[6];
        if (s_misc[7]) {   // block-uniform
            if (tid == 0) a.redo[1 + atomicAdd(a.redo, 1)] = b;
            return;
        }
        int off = 0;
        for (int g = 0; g < a.n_seg; ++g) {
            const int c = a.cand_cnt[static_cast<size_t>(b) * a.n_seg + g];
            const uint2* seg = a.cand + (static_cast<size_t>(b) * a.n_seg + g) * a.cap_seg;
            for (int i = tid; i < c; i += 256) lst[off + i] = seg[i];
            off += c;
        }
        if (a.s0_cand) {
            const int c = a.s0_cnt[b];
            const uint2* seg = a.s0_cand + static_cast<size_t>(b) * a.s0_cap;
            for (int i = tid; i < c; i += 256) lst[off + i] = seg[i];
        }
        __syncthreads();
    }
    // position i -> (admissible?, key, column j)
    auto key_of = [&](int i, uint32_t& key, int& j) -> bool {
        float s;
        bool biased = false;   // sample-segment entries carry the bias already (bit 31 of the column)
        if (list) {
            const uint2 c = lst[i];
            j = static_cast<int>(c.x & 0x7FFFFFFFu);
            biased = (c.x >> 31) != 0u;
            s = __builtin_bit_cast(float, c.y);
        } else {
            j = i;
            s = row[i];
        }
        if (j == self) return false;
        if (a.pool && !((a.pool[j >> 5] >> (j & 31)) & 1u)) return false;
        if (a.Qb && !biased) s += a.Qb[j];
        if (a.rule_flt_min && !(s > FLT_MIN)) return false;
        key = desc_key(s);
        return true;
    };
    auto pack = [](uint32_t key, int j) { return (static_cast<unsigned long long>(key) << 32) | (0xFFFFFFFFu - static_cast<uint32_t>(j)); };

    // ---------------- fast path: two reads of the row ----------------
    // 12-bit histogram of the key's top bits (sign, exponent, 3 mantissa bits), then ONE more pass that sends
    // everything above the threshold bin to the output list and the bin's members (~1 % of the row) to an LDS
    // candidate buffer, where the remaining 20 bits are resolved.  Falls through to the multi-pass path when the
    // bin overflows the buffer or when ties straddle the k-th place (the reference's tie rule needs column order).
    bool done = false;
    int fast_kk_eff = 0;
    if (a.cand_cap > 0) {
        unsigned long long* cand = sel + a.p2;
        // histogram `hist[0..nbins)` is filled; finds the bin where the running count reaches `want`
        auto find_bin = [&](int nbins, int want) {   // -> s_misc[0] bin (-1: fewer than want in total), [1] remaining inside it, [2] total, [3] bin count
            const int per = nbins / 256;
            int ps = 0;
            for (int q = 0; q < per; ++q) ps += hist[tid * per + q];
            part[tid] = ps;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int t = 0; t < 256; ++t) tot += part[t];
                int bin = -1, rem = want, cnt = 0;
                if (tot >= want) {
                    int cum = 0, t = 0;
                    while (cum + part[t] < want) cum += part[t++];
                    int q = t * per;
                    while (cum + hist[q] < want) cum += hist[q++];
                    bin = q;
                    rem = want - cum;
                    cnt = hist[q];
                }
                s_misc[0] = bin; s_misc[1] = rem; s_misc[2] = tot; s_misc[3] = cnt;
            }
            __syncthreads();
        };
        for (int i = tid; i < 4096; i += 256) hist[i] = 0;
        for (int i = tid; i < a.p2; i += 256) sel[i] = ~0ull;
        __syncthreads();
        for (int i = tid; i < cols; i += 256) {
            uint32_t key;
            int j;
            if (key_of(i, key, j)) atomicAdd(&hist[key >> 20], 1);
        }
        __syncthreads();
        find_bin(4096, a.kk);
        const int bin1 = s_misc[0], rem1 = s_misc[1], total1 = s_misc[2];
        __syncthreads();
        if (tid == 0) { s_misc[4] = 0; s_misc[5] = 0; }
        __syncthreads();
        const bool all1 = bin1 < 0;
        for (int i = tid; i < cols; i += 256) {
            uint32_t key;
            int j;
            if (!key_of(i, key, j)) continue;
            const int top = static_cast<int>(key >> 20);
            if (all1 || top < bin1) sel[atomicAdd(&s_misc[4], 1)] = pack(key, j);
            else if (top == bin1) {
                const int c = atomicAdd(&s_misc[5], 1);
                if (c < a.cand_cap) cand[c] = pack(key, j);
            }
        }
        __syncthreads();
        const int n_cand = s_misc[5];
        if (all1) {
            done = true;
            fast_kk_eff = total1;
        } else if (n_cand <= a.cand_cap) {
            for (int i = tid; i < 1024; i += 256) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < n_cand; i += 256) atomicAdd(&hist[(static_cast<uint32_t>(cand[i] >> 32) >> 10) & 1023u], 1);
            __syncthreads();
            find_bin(1024, rem1);
            const int bin2 = s_misc[0], rem2 = s_misc[1];
            __syncthreads();
            for (int i = tid; i < 1024; i += 256) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < n_cand; i += 256) {
                const uint32_t k = static_cast<uint32_t>(cand[i] >> 32);
                if (static_cast<int>((k >> 10) & 1023u) == bin2) atomicAdd(&hist[k & 1023u], 1);
            }
            __syncthreads();
            find_bin(1024, rem2);
            const int bin3 = s_misc[0], need_eq = s_misc[1], eq_cnt = s_misc[3];
            __syncthreads();
            if (need_eq == eq_cnt) {   // no tie straddles the k-th place: everything up to the threshold key is in
                const uint32_t thr = (static_cast<uint32_t>(bin1) << 20) | (static_cast<uint32_t>(bin2) << 10) | static_cast<uint32_t>(bin3);
                for (int i = tid; i < n_cand; i += 256)
                    if (static_cast<uint32_t>(cand[i] >> 32) <= thr) sel[atomicAdd(&s_misc[4], 1)] = cand[i];
                done = true;
                fast_kk_eff = a.kk;
            }
        }
        __syncthreads();
    }

    int kk_eff = fast_kk_eff;
    if (!done) {   // ---------------- multi-pass path (8 bits per pass over the row) ----------------
        uint32_t prefix = 0, mask = 0;
        int remaining = a.kk, total = 0, eq_total = 0;
        bool take_all = false;
        for (int pass = 0; pass < 4 && !take_all; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < cols; i += 256) {
                uint32_t key;
                int j;
                if (key_of(i, key, j) && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, bin = 255, rem = remaining;
                int tot = 0;
                for (int i = 0; i < 256; ++i) tot += hist[i];
                if (pass == 0 && tot < remaining) {
                    bin = -1;   // fewer admissible candidates than slots: take them all
                } else {
                    for (int i = 0; i < 256; ++i) {
                        if (cum + hist[i] >= rem) { bin = i; break; }
                        cum += hist[i];
                    }
                    rem -= cum;
                }
                s_misc[0] = bin;
                s_misc[1] = rem;
                s_misc[2] = tot;
                s_misc[3] = bin >= 0 ? hist[bin] : 0;
            }
            __syncthreads();
            const int bin = s_misc[0];
            if (pass == 0) total = s_misc[2];
            if (bin < 0) { take_all = true; break; }
            remaining = s_misc[1];
            eq_total = s_misc[3];
            prefix |= static_cast<uint32_t>(bin) << shift;
            mask |= 255u << shift;
            __syncthreads();
        }
        kk_eff = take_all ? total : a.kk;
        // now: keys < prefix are in, `remaining` of the eq_total keys == prefix are in (the first ones by column)
        for (int i = tid; i < a.p2; i += 256) sel[i] = ~0ull;
        if (tid == 0) { s_misc[2] = 0; s_misc[3] = 0; }
        __syncthreads();
        const int n_gt = kk_eff - (take_all ? 0 : remaining);
        // Boundary ties (more candidates equal to the k-th score than slots left): the reference's running list
        // (_core.hpp:115-128) admits an equal-score candidate only while fewer than kk candidates >= that score
        // have been seen, and every later better candidate then evicts the OLDEST of them.  Closed form: let F be
        // the first kk candidates (by index) with score >= t and A the candidates == t inside F; the survivors are
        // the `remaining` members of A with the HIGHEST indices.
        const bool ordered = !take_all && remaining < eq_total;
        if (kk_eff > 0) {
            for (int i = tid; i < cols; i += 256) {
                uint32_t key = 0;
                int j;
                if (!key_of(i, key, j)) continue;
                if (take_all || key < prefix) sel[atomicAdd(&s_misc[2], 1)] = pack(key, j);
                else if (!ordered && key == prefix) sel[n_gt + atomicAdd(&s_misc[3], 1)] = pack(key, j);   // all eq_total == remaining of them
            }
        }
        if (ordered) {
            __shared__ int s_run[4];    // 0: candidates >= t so far, 1: candidates == t so far, 2: |A|, 3: done
            __shared__ int s_wave[8];   // per-wave counts of the current 256-column step: [0..3] >= t, [4..7] == t
            if (list) {
                // the tie rule walks the candidates in COLUMN order; the list is in arrival order: sort it by column
                // (every column at or above the k-th score is in the list, so the walk sees what the dense walk sees)
                int n2 = 2;
                while (n2 < cols) n2 <<= 1;
                __syncthreads();
                for (int i = cols + tid; i < n2; i += 256) lst[i] = make_uint2(0xFFFFFFFFu, 0u);
                __syncthreads();
                for (int size = 2; size <= n2; size <<= 1)
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        for (int t = tid; t < (n2 >> 1); t += 256) {
                            const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                            const int hi = lo | stride;
                            const bool up = (lo & size) == 0;
                            const uint2 x = lst[lo], y = lst[hi];
                            if (((x.x & 0x7FFFFFFFu) > (y.x & 0x7FFFFFFFu)) == up) { lst[lo] = y; lst[hi] = x; }   // (bit 31: bias flag)
                        }
                        __syncthreads();
                    }
            }
            if (tid < 4) s_run[tid] = 0;
            __syncthreads();
            for (int phase = 0; phase < 2; ++phase) {
                // phase 0 finds |A| (the == t count when the kk-th candidate >= t arrives); phase 1 places the survivors
                const int cnt_a = s_run[2];
                __syncthreads();
                if (tid < 2) s_run[tid] = 0;
                __syncthreads();
                for (int base = 0; base < cols; base += 256) {
                    const int i = base + tid;
                    uint32_t key = 0;
                    int j = 0;
                    const bool ok = i < cols && key_of(i, key, j);
                    const bool ge = ok && key <= prefix, eq = ok && key == prefix;
                    const unsigned long long bge = __ballot(ge), beq = __ballot(eq);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (lane == 0) { s_wave[wv] = __popcll(bge); s_wave[4 + wv] = __popcll(beq); }
                    __syncthreads();
                    int ge_rank = s_run[0] + __popcll(bge & below), eq_rank = s_run[1] + __popcll(beq & below);
                    for (int w = 0; w < wv; ++w) { ge_rank += s_wave[w]; eq_rank += s_wave[4 + w]; }
                    if (phase == 0) {
                        if (ge && ge_rank == a.kk - 1) s_run[2] = eq_rank + (eq ? 1 : 0);
                    } else if (eq && eq_rank < cnt_a && eq_rank >= cnt_a - remaining) {
                        sel[n_gt + (eq_rank - (cnt_a - remaining))] = pack(key, j);
                    }
                    __syncthreads();
                    if (tid == 0) {
                        s_run[0] += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
                        s_run[1] += s_wave[4] + s_wave[5] + s_wave[6] + s_wave[7];
                    }
                    __syncthreads();
                    if (s_run[phase == 0 ? 0 : 1] >= (phase == 0 ? a.kk : cnt_a)) break;   // block-uniform
                }
                __syncthreads();
            }
        }

    }
    __syncthreads();
    // bitonic sort, ascending composite = (score desc, column desc)
    for (int size = 2; size <= a.p2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (a.p2 >> 1); t += 256) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int hi = lo | stride;
                const bool up = (lo & size) == 0;
                const unsigned long long x = sel[lo], y = sel[hi];
                if ((x > y) == up) { sel[lo] = y; sel[hi] = x; }
            }
            __syncthreads();
        }
    int32_t* ok = a.out_keys + static_cast<size_t>(orow) * a.k;
    float* os = a.out_scores ? a.out_scores + static_cast<size_t>(orow) * a.k : nullptr;
    for (int r = tid; r < a.k; r += 256) {
        if (r < kk_eff) {
            const unsigned long long c = sel[r];
            ok[r] = static_cast<int32_t>(0xFFFFFFFFu - static_cast<uint32_t>(c & 0xFFFFFFFFull));
            if (os) os[r] = key_score(static_cast<uint32_t>(c >> 32));
        } else {
            ok[r] = -1;
            if (os) os[r] = r < a.kk ? FLT_MIN : 0.0f;   // _core.hpp:26 / :134-137
        }
    }
}


// ------------------------------------------------------------------------------------------------
// One WAVE per row, for rows that fit in registers: no block barriers, no LDS histograms.  The k-th smallest key of the
// row is found bit by bit (32 rounds of "how many live keys have a 0 here", one DPP wave sum each) over the keys the
// lanes hold; a 256-thread block per row spends most of its time in the fixed cost of its barriers when the row has a
// few hundred entries, as the candidate lists of the fused path do.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);   // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);   // row_ror:1
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

__device__ __forceinline__ void wave_lds_sync() {   // LDS traffic of ONE wave: program order is enough, keep the compiler from moving it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, v);
        if (lane >= d) v += o;
    }
    return v;
}

constexpr int kWaveHistBins = 4096;   // uint32 per wave
constexpr int kSampleCap = 512;      // entries of a row's sample segment (normally kk plus the ties at the threshold)

// key[s], s < SLOTS, live where bit s of `valid` is set.  Returns the number of live keys m; when m >= kk: kth = the kk-th
// smallest, need_eq = how many of the keys == kth belong to the kk smallest, eq_total = how many there are.
// Three histogram levels over the key's bits 31..20, 19..8, 7..0 in the wave's own LDS histogram `hist` (kWaveHistBins
// words): the live keys that match the prefix found so far are counted by their next digit (LDS atomics), the digit
// holding the kk-th key is located with two wave scans (row totals of the [rows][64] bin matrix, then inside the row).
template <int SLOTS>
__device__ __forceinline__ int wave_kth_key(const uint32_t (&key)[SLOTS], uint64_t valid, int kk, uint32_t* hist, int lane, uint32_t& kth, int& need_eq,
                                            int& eq_total) {
    const int m = wave_sum_i32(__popcll(valid));
    kth = 0u; need_eq = 0; eq_total = 0;
    if (m < kk) return m;
    uint32_t prefix = 0u, mask = 0u;
    int remaining = kk, bin_count = 0;
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const int shift = level == 0 ? 20 : (level == 1 ? 8 : 0);
        const int nb = level == 2 ? 256 : 4096;
        const int rows = nb / 64;
        uint4* h4 = reinterpret_cast<uint4*>(hist);
        for (int i = lane; i < nb / 4; i += 64) h4[i] = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const uint32_t k = key[sl];
            if (((valid >> sl) & 1ull) && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & static_cast<uint32_t>(nb - 1)], 1u);
        }
        wave_lds_sync();
        // row totals: lane r < rows sums bins [64 r, 64 r + 64), read skewed so that the lanes spread over the banks
        int rt = 0;
        if (lane < rows)
            for (int j = 0; j < 64; ++j) rt += static_cast<int>(hist[lane * 64 + ((j + lane) & 63)]);
        const int rincl = wave_incl_scan_i32(rt, lane);
        const unsigned long long rb = __ballot(lane < rows && rincl >= remaining);
        const int r = __builtin_ctzll(rb);   // rb != 0: the matching keys number at least `remaining`
        remaining -= __builtin_amdgcn_readlane(rincl - rt, r);
        const int bv = static_cast<int>(hist[r * 64 + lane]);
        const int bincl = wave_incl_scan_i32(bv, lane);
        const unsigned long long bb = __ballot(bincl >= remaining);
        const int c = __builtin_ctzll(bb);
        remaining -= __builtin_amdgcn_readlane(bincl - bv, c);
        bin_count = __builtin_amdgcn_readlane(bv, c);
        prefix |= static_cast<uint32_t>(r * 64 + c) << shift;
        mask |= static_cast<uint32_t>(nb - 1) << shift;
        wave_lds_sync();
    }
    kth = prefix; need_eq = remaining; eq_total = bin_count;
    return m;
}

// admission rules of topk_select_kernel::key_of for column j with raw score s
__device__ __forceinline__ bool topk_admit(const SelectArgs& a, int j, int self, float s, uint32_t& key) {
    if (j == self) return false;
    if (a.pool && !((a.pool[j >> 5] >> (j & 31)) & 1u)) return false;
    if (a.Qb) s += a.Qb[j];
    if (a.rule_flt_min && !(s > FLT_MIN)) return false;
    key = desc_key(s);
    return true;
}

// Both wave kernels fetch a row's entries in straight-line groups of 16 loads per lane: a load that sits behind the
// admission branches of the previous entry is not issued before that entry is done, and 64 serialised round trips per row
// made the first version of these kernels 10x slower than their arithmetic.

// thresholds of the fused path from the dense scores of the sampled columns (a.cols <= 4096: 64 keys per lane):
// thr[b] = the kk-th best admissible score, or -- with fewer than kk of them -- "everything" (with the admission rule only
// scores > FLT_MIN can be listed, so FLT_MIN is a valid bound then).  The sampled columns that reach the threshold are
// written out as the row's sample segment (s0_cand), so that the filtered sweep can start behind the sample.
// grid: ceil(rows / 4) blocks of 4 waves; dynamic LDS: 4 histograms.
__global__ __launch_bounds__(256, 2) void topk_thr_wave_kernel(SelectArgs a, int rows) {
    extern __shared__ __attribute__((aligned(16))) uint32_t whist_dyn[];   // 4 * kWaveHistBins words
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= rows) return;
    const float* row = a.S + static_cast<size_t>(b) * a.ld_s;
    const int self = a.self_idx ? a.self_idx[a.q0 + b] : -1;
    uint32_t key[64];
    uint64_t valid = 0ull;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float rawv[16], qb[16];
        uint32_t pw[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = (c * 16 + t) * 64 + lane;
            rawv[t] = row[j < a.cols ? j : a.cols - 1];
        }
        if (a.Qb) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = (c * 16 + t) * 64 + lane;
                qb[t] = a.Qb[j < a.cols ? j : a.cols - 1];
            }
        }
        if (a.pool) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = (c * 16 + t) * 64 + lane;
                pw[t] = a.pool[(j < a.cols ? j : a.cols - 1) >> 5];
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int sl = c * 16 + t;
            const int j = sl * 64 + lane;
            float sc = rawv[t];
            if (a.Qb) sc += qb[t];
            bool ok = j < a.cols && j != self;
            if (a.pool) ok = ok && ((pw[t] >> (j & 31)) & 1u);
            if (a.rule_flt_min) ok = ok && sc > FLT_MIN;
            key[sl] = ok ? desc_key(sc) : 0u;
            valid |= static_cast<uint64_t>(ok ? 1 : 0) << sl;
        }
        __builtin_amdgcn_sched_barrier(0);   // one group's loads in flight at a time: 64 keys + 48 group registers, not 256
    }
    uint32_t kth; int need_eq, eq_total;
    const int m = wave_kth_key<64>(key, valid, a.kk, whist_dyn + (threadIdx.x >> 6) * kWaveHistBins, lane, kth, need_eq, eq_total);
    if (lane == 0) a.thr[b] = m >= a.kk ? key_score(kth) : (a.rule_flt_min ? FLT_MIN : -__builtin_inff());
    if (a.s0_cand) {   // the admissible sampled columns at or above the threshold: the kk best plus the ties at the k-th place
        uint2* out = a.s0_cand + static_cast<size_t>(b) * a.s0_cap;
        int n0 = 0;
#pragma unroll
        for (int sl = 0; sl < 64; ++sl) {
            const bool win = ((valid >> sl) & 1ull) && (m < a.kk || key[sl] <= kth);
            const unsigned long long mask = __ballot(win);
            const int at = n0 + __popcll(mask & ((1ull << lane) - 1ull));
            // bit 31 of the column: the score already carries the bias (key -> score is exact, so the selection sees the same key)
            if (win && at < a.s0_cap) out[at] = make_uint2(static_cast<uint32_t>(sl * 64 + lane) | 0x80000000u, __float_as_uint(key_score(key[sl])));
            n0 += __popcll(mask);
            if ((sl & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep the 64 ballots from being formed all at once (SGPR spills)
        }
        if (lane == 0) a.s0_cnt[b] = n0;
    }
}

// selection over the candidate lists of the fused path, one wave per row (lists of <= 2048 entries: 32 per lane).
// A row whose segments or list overflowed goes to `redo` (dense path); a row with ties straddling the k-th place goes to
// `general` (topk_select_kernel's list mode, which walks the ties in column order).  Dynamic LDS: 4 histograms + 4 * p2 * 8 bytes.
__global__ __launch_bounds__(256) void topk_list_wave_kernel(SelectArgs a, int rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long wsel[];   // 4 histograms (kWaveHistBins words), then 4 * p2 sort entries
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= rows) return;
    uint32_t* whist = reinterpret_cast<uint32_t*>(wsel) + static_cast<size_t>(wv) * kWaveHistBins;
    unsigned long long* sel = wsel + (4 * kWaveHistBins) / 2 + static_cast<size_t>(wv) * a.p2;
    const int self = a.self_idx ? a.self_idx[a.q0 + b] : -1;
    // segment ends (n_seg <= 8 sweep segments, then the sample segment): seg_end[g] = entries of the segments 0..g
    int seg_end[9];
    int over = 0, tot = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        int c = 0;
        if (g < a.n_seg) {
            c = a.cand_cnt[static_cast<size_t>(b) * a.n_seg + g];
            over |= c > a.cap_seg;
        }
        tot += c;
        seg_end[g] = tot;
    }
    {
        int c = 0;
        if (a.s0_cand) {
            c = a.s0_cnt[b];
            over |= c > a.s0_cap;
        }
        tot += c;
        seg_end[8] = tot;
    }
    if (over || tot > a.list_cap) {
        if (lane == 0) a.redo[1 + atomicAdd(a.redo, 1)] = b;
        return;
    }
    uint32_t key[32], col[32];
    uint64_t valid = 0ull;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        uint2 cv[16];
        float qb[16];
        uint32_t pw[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int i = (c * 16 + t) * 64 + lane;
            if (i >= tot) i = tot > 0 ? tot - 1 : 0;
            int g = 0, beg = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i >= seg_end[q]) { g = q + 1; beg = seg_end[q]; }
            const uint2* src = g < 8 ? a.cand + (static_cast<size_t>(b) * a.n_seg + g) * a.cap_seg : a.s0_cand + static_cast<size_t>(b) * a.s0_cap;
            cv[t] = tot > 0 ? src[i - beg] : make_uint2(0u, 0u);
        }
        if (a.Qb) {
#pragma unroll
            for (int t = 0; t < 16; ++t) qb[t] = a.Qb[cv[t].x & 0x7FFFFFFFu];
        }
        if (a.pool) {
#pragma unroll
            for (int t = 0; t < 16; ++t) pw[t] = a.pool[(cv[t].x & 0x7FFFFFFFu) >> 5];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int sl = c * 16 + t;
            const int i = sl * 64 + lane;
            const int j = static_cast<int>(cv[t].x & 0x7FFFFFFFu);
            float sc = __uint_as_float(cv[t].y);
            if (a.Qb && !(cv[t].x >> 31)) sc += qb[t];   // sample-segment entries carry the bias already
            bool ok = i < tot && j != self;
            if (a.pool) ok = ok && ((pw[t] >> (j & 31)) & 1u);
            if (a.rule_flt_min) ok = ok && sc > FLT_MIN;
            key[sl] = ok ? desc_key(sc) : 0u;
            col[sl] = static_cast<uint32_t>(j);
            valid |= static_cast<uint64_t>(ok ? 1 : 0) << sl;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t kth; int need_eq, eq_total;
    const int m = wave_kth_key<32>(key, valid, a.kk, whist, lane, kth, need_eq, eq_total);
    const bool take_all = m < a.kk;
    if (!take_all && need_eq < eq_total) {   // ties straddle the k-th place: the reference's rule needs column order
        if (lane == 0) a.general[1 + atomicAdd(a.general, 1)] = b;
        return;
    }
    const int kk_eff = take_all ? m : a.kk;
    for (int i = lane; i < a.p2; i += 64) sel[i] = ~0ull;
    wave_lds_sync();
    int base = 0;
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) {
        const bool win = ((valid >> sl) & 1ull) && (take_all || key[sl] <= kth);
        const unsigned long long mask = __ballot(win);
        if (win) sel[base + __popcll(mask & ((1ull << lane) - 1ull))] = (static_cast<unsigned long long>(key[sl]) << 32) | (0xFFFFFFFFu - col[sl]);
        base += __popcll(mask);
    }
    // bitonic sort of the wave's p2 entries, ascending composite = (score desc, column desc)
    for (int size = 2; size <= a.p2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            wave_lds_sync();
            for (int t = lane; t < (a.p2 >> 1); t += 64) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int hi = lo | stride;
                const bool up = (lo & size) == 0;
                const unsigned long long x = sel[lo], y = sel[hi];
                if ((x > y) == up) { sel[lo] = y; sel[hi] = x; }
            }
        }
    wave_lds_sync();
    const int orow = a.q0 + b;
    int32_t* ok = a.out_keys + static_cast<size_t>(orow) * a.k;
    float* os = a.out_scores ? a.out_scores + static_cast<size_t>(orow) * a.k : nullptr;
    for (int r = lane; r < a.k; r += 64) {
        if (r < kk_eff) {
            const unsigned long long c = sel[r];
            ok[r] = static_cast<int32_t>(0xFFFFFFFFu - static_cast<uint32_t>(c & 0xFFFFFFFFull));
            if (os) os[r] = key_score(static_cast<uint32_t>(c >> 32));
        } else {
            ok[r] = -1;
            if (os) os[r] = r < a.kk ? FLT_MIN : 0.0f;   // _core.hpp:26 / :134-137
        }
    }
}

// ------------------------------------------------------------------------------------------------
class TopkHandle : public HandleBase {
 public:
    ~TopkHandle() override {
        if (stream) (void)hipStreamDestroy(stream);
    }
    void ensure() {
        BFH_HIP(hipSetDevice(device));
        if (!stream) {
            BFH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            hipDeviceProp_t prop;
            BFH_HIP(hipGetDeviceProperties(&prop, device));
            num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
    }

    // candidate buffer of the select kernel's fast path (entries behind the sort buffer)
    static int cand_capacity(int p2) {
        const int room = (140 * 1024 - p2 * 8) / 8;
        return room >= 1024 ? 1024 : 0;   // small on purpose: LDS per block decides how many rows a CU works on at once
    }
    static int pow2_at_least(int n) {
        int p = 2;
        while (p < n) p <<= 1;
        return p;
    }

    // ---- launch helpers ----
    // dense scores of `nb` queries (rows qidx ? qidx[q0 + b] : q0 + b of dP) against the first `cols` candidates -> S_ [nb, ld_s]
    void launch_scores(const float* dP, const int32_t* qidx, int q0, int nb, int cols, int ld, int d_pad, size_t ld_s, size_t tiles_all) {
        const int n_tiles = (cols + 31) / 32;
        const int qblocks = (nb + 127) / 128;
        int tpb = static_cast<int>((static_cast<int64_t>(n_tiles) * qblocks + num_cus_ * 8 - 1) / (num_cus_ * 8));
        if (tpb < 1) tpb = 1;
        const dim3 grid((n_tiles + tpb - 1) / tpb, qblocks);
        for (int kc = 0; kc < d_pad; kc += 128) {
            const int W = std::min(128, d_pad - kc);
            const float4* qp = Qp_.get() + static_cast<size_t>(kc / 128) * tiles_all * 16 * 64;
            if (W == 128)
                hipLaunchKernelGGL((topk_scores_kernel<true, false>), grid, dim3(256), 0, stream, dP, qidx, q0, nb, qp, cols, ld, kc, W, S_.get(), ld_s,
                                   tpb, kc > 0 ? 1 : 0, FilterArgs{});
            else
                hipLaunchKernelGGL((topk_scores_kernel<false, false>), grid, dim3(256), 0, stream, dP, qidx, q0, nb, qp, cols, ld, kc, W, S_.get(), ld_s,
                                   tpb, kc > 0 ? 1 : 0, FilterArgs{});
            BFH_HIP(hipGetLastError());
        }
    }
    void launch_select(const SelectArgs& a, int rows, size_t lds) {
        hipLaunchKernelGGL(topk_select_kernel, dim3(rows), dim3(256), lds, stream, a);
        BFH_HIP(hipGetLastError());
    }

    // The fused path's shape for a call, or `on = false`: the dense path.  C0 = sampled columns (the thresholds' source):
    // the filter is expected to pass kk * q_rows / C0 columns per query, which must sit well inside the LDS list.
    struct FusedPlan {
        bool on = false;
        int c0 = 0, c0_tiles = 0, n_seg = 1, tpb = 1, cap_seg = 0;
        bool sample_seg = false;
    };
    static constexpr int kListCap = 2048;
    FusedPlan fused_plan(int nq, int q_rows, int d_pad, int kk) const {
        FusedPlan fp;
        if (fused_ == 0 || d_pad > 128) return fp;   // two K-chunks accumulate through the score buffer
        const bool force = fused_ > 0;
        if (!force && (nq < 8192 || q_rows < 8192)) return fp;   // small sweeps: the dense path's item-tile parallelism matters more
        const int n_tiles = (q_rows + 31) / 32;
        int64_t need = (static_cast<int64_t>(kk) * q_rows + kListCap / 3 - 1) / (kListCap / 3);
        int c0 = force ? 32 : 2048;
        while (c0 < need) c0 <<= 1;
        if (force && fused_c0_ > 0) c0 = fused_c0_;   // tests: exactly this sample (too small a sample overflows the lists: the dense redo path)
        c0 = (c0 + 31) / 32 * 32;
        if (force) c0 = std::min(c0, n_tiles * 32);   // tests: any shape goes through (overflowing rows take the dense path)
        else if (c0 > q_rows / 4) return fp;
        if (c0 >= q_rows + 32) return fp;
        fp.c0 = std::min(c0, q_rows);
        fp.c0_tiles = (fp.c0 + 31) / 32;              // c0 is a multiple of 32 or the whole matrix
        fp.sample_seg = wave_select_ && fp.c0 <= 4096;   // topk_thr_wave_kernel writes the sample's own candidates ...
        const int sweep_tiles = n_tiles - (fp.sample_seg ? fp.c0_tiles : 0);   // ... and the filtered sweep starts behind the sample
        const int qblocks = (nq + 127) / 128;
        int tpb = static_cast<int>((static_cast<int64_t>(sweep_tiles) * qblocks + num_cus_ * 8 - 1) / (num_cus_ * 8));
        tpb = std::max(tpb, (sweep_tiles + 7) / 8);   // at most 8 segments per query
        fp.tpb = std::max(1, tpb);
        fp.n_seg = std::max(1, (sweep_tiles + fp.tpb - 1) / fp.tpb);
        fp.cap_seg = static_cast<int>(std::min<int64_t>(static_cast<int64_t>(fp.tpb) * 32, std::max(64, 2 * kListCap / fp.n_seg)));
        fp.on = true;
        return fp;
    }

    // core: factor matrices in HBM, [rows, ld], ld % 8 == 0, columns [d, ld) zero
    void run_device(const int32_t* indexes, int nq, const float* dP, bool gather, const float* dQ, int q_rows, int d, int ld, const float* dQb,
                    bool same, int32_t* out_keys, float* out_scores, const int32_t* pool, int pool_size, int k) {
        BFH_REQUIRE(k > 0 && k <= TOPK_MAX_K, "k must be in [1, 16384]");
        BFH_REQUIRE(ld % 8 == 0 && d <= ld && d > 0, "factor matrices need a leading dimension that is a multiple of 8 and >= d");
        BFH_REQUIRE(nq >= 0 && q_rows > 0, "empty candidate matrix");
        if (nq == 0) return;
        ensure();
        int kk = std::min(q_rows, k);
        if (pool_size) kk = std::min(pool_size, kk);
        d_idx_.resize(std::max<size_t>(d_idx_.size(), nq));
        BFH_HIP(hipMemcpyAsync(d_idx_.get(), indexes, sizeof(int32_t) * nq, hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += 4.0 * nq;
        const uint32_t* d_pool = nullptr;
        if (pool_size) {
            const size_t words = (static_cast<size_t>(q_rows) + 31) / 32;
            std::vector<uint32_t> bm(words, 0u);
            for (int i = 0; i < pool_size; ++i) {
                const int32_t j = pool[i];
                if (j >= 0 && j < q_rows) bm[j >> 5] |= 1u << (j & 31);   // ids outside the matrix can never match a candidate
            }
            d_pool_.resize(std::max(d_pool_.size(), words));
            BFH_HIP(hipMemcpyAsync(d_pool_.get(), bm.data(), words * 4, hipMemcpyHostToDevice, stream));
            BFH_HIP(hipStreamSynchronize(stream));   // bm is a local
            stats.h2d_bytes += 4.0 * words;
            d_pool = d_pool_.get();
        }
        d_keys_.resize(std::max(d_keys_.size(), static_cast<size_t>(nq) * k));
        d_scores_.resize(std::max(d_scores_.size(), static_cast<size_t>(nq) * k));
        const size_t ld_s = (static_cast<size_t>(q_rows) + 31) / 32 * 32;
        const int d_pad = (d + 7) / 8 * 8;
        const int n_tiles = (q_rows + 31) / 32;
        const int p2 = pow2_at_least(kk);
        const int cand_cap = cand_capacity(p2);
        const FusedPlan fp = fused_plan(nq, q_rows, d_pad, kk);
        // query batch, multiple of 128 rows: dense -- score buffer <= 2 GiB; fused -- sample scores + candidate segments <= 2 GiB,
        // and room in the score buffer for 128 dense rows (the rows the fused path hands back)
        const size_t per_query = fp.on ? static_cast<size_t>(fp.c0) * 4 + static_cast<size_t>(fp.n_seg) * fp.cap_seg * 8 : ld_s * 4;
        const int batch = static_cast<int>(std::min<size_t>(nq, std::max<size_t>(128, ((size_t(1) << 31) / per_query) / 128 * 128)));
        const int redo_rows = fp.on ? static_cast<int>(std::max<size_t>(128, std::min<size_t>(batch, (size_t(1) << 28) / ld_s) / 128 * 128)) : 0;
        S_.resize(std::max(S_.size(), fp.on ? std::max(static_cast<size_t>(batch) * fp.c0, static_cast<size_t>(redo_rows) * ld_s)
                                             : static_cast<size_t>(batch) * ld_s));
        const size_t lds_dense = static_cast<size_t>(p2 + cand_cap) * 8;
        const size_t lds_list = lds_dense + static_cast<size_t>(kListCap) * 8;
        BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(fp.on ? lds_list : lds_dense)));
        const size_t kWaveLds = static_cast<size_t>(4) * kWaveHistBins * 4;   // the wave kernels' four histograms
        const size_t kListLds = kWaveLds;
        if (fp.on && wave_select_) {
            BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_thr_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(kWaveLds)));
            BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_list_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(kListLds + static_cast<size_t>(4) * std::min(p2, 1024) * 8)));
        }
        {   // candidate matrix in operand order, one slab per K-chunk
            const int n_chunks = (d_pad + 127) / 128;
            const size_t per = static_cast<size_t>(n_tiles) * 16 * 64;
            Qp_.resize(std::max(Qp_.size(), per * n_chunks));
            const int slot = t_aux_.begin(stream);
            for (int c = 0; c < n_chunks; ++c) {
                const int W = std::min(128, d_pad - c * 128);
                hipLaunchKernelGGL(topk_pack_kernel, dim3(static_cast<unsigned>((per + 255) / 256)), dim3(256), 0, stream, dQ, q_rows, ld, c * 128, W,
                                   Qp_.get() + per * c, n_tiles);
                BFH_HIP(hipGetLastError());
            }
            t_aux_.end(slot, stream);
        }
        SelectArgs base{};
        base.Qb = dQb; base.pool = d_pool; base.rule_flt_min = flt_min_rule_ ? 1 : 0; base.k = k; base.kk = kk;
        base.out_keys = d_keys_.get(); base.out_scores = d_scores_.get(); base.p2 = p2; base.cand_cap = fast_select_ ? cand_cap : 0;
        const int32_t* qidx = gather ? d_idx_.get() : nullptr;
        if (fp.on) {
            thr_.resize(std::max(thr_.size(), static_cast<size_t>(batch)));
            cand_.resize(std::max(cand_.size(), static_cast<size_t>(batch) * fp.n_seg * fp.cap_seg));
            cnt_.resize(std::max(cnt_.size(), static_cast<size_t>(batch) * fp.n_seg));
            redo_.resize(std::max(redo_.size(), static_cast<size_t>(batch) + 1));
            general_.resize(std::max(general_.size(), static_cast<size_t>(batch) + 1));
            s0_cand_.resize(std::max(s0_cand_.size(), static_cast<size_t>(batch) * kSampleCap));
            s0_cnt_.resize(std::max(s0_cnt_.size(), static_cast<size_t>(batch)));
        }
        for (int q0 = 0; q0 < nq; q0 += batch) {
            const int nb = std::min(batch, nq - q0);
            if (!fp.on) {
                int slot = t_main_.begin(stream);
                launch_scores(dP, qidx, q0, nb, q_rows, ld, d_pad, ld_s, n_tiles);
                t_main_.end(slot, stream);
                SelectArgs a = base;
                a.S = S_.get(); a.ld_s = ld_s; a.cols = q_rows; a.self_idx = same ? d_idx_.get() : nullptr; a.q0 = q0;
                slot = t_aux_.begin(stream);
                launch_select(a, nb, lds_dense);
                t_aux_.end(slot, stream);
                continue;   // the scores kernel of the next batch reuses S_: the stream orders it after this select
            }
            // ---- fused: (1) thresholds from the first c0 columns, (2) filtered sweep, (3) selection over the candidate lists ----
            int slot = t_main_.begin(stream);
            launch_scores(dP, qidx, q0, nb, fp.c0, ld, d_pad, static_cast<size_t>(fp.c0), n_tiles);
            t_main_.end(slot, stream);
            slot = t_aux_.begin(stream);
            SelectArgs a = base;
            a.S = S_.get(); a.ld_s = static_cast<size_t>(fp.c0); a.cols = fp.c0; a.self_idx = same ? d_idx_.get() : nullptr; a.q0 = q0;
            // the wave kernel also writes the sample's own candidates, and the sweep then starts behind the sample; the block-level
            // route (samples beyond 4096 columns) only yields thresholds, and the sweep covers every column
            const bool sample_seg = fp.sample_seg;
            if (sample_seg) {
                a.thr = thr_.get();
                a.s0_cand = s0_cand_.get(); a.s0_cnt = s0_cnt_.get(); a.s0_cap = kSampleCap;
                hipLaunchKernelGGL(topk_thr_wave_kernel, dim3((nb + 3) / 4), dim3(256), kWaveLds, stream, a, nb);
            } else {
                launch_select(a, nb, lds_dense);
                hipLaunchKernelGGL(topk_thr_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, d_keys_.get(), d_scores_.get(), q0, nb, k, kk,
                                   flt_min_rule_ ? 1 : 0, thr_.get());
            }
            BFH_HIP(hipGetLastError());
            BFH_HIP(hipMemsetAsync(redo_.get(), 0, sizeof(int), stream));
            BFH_HIP(hipMemsetAsync(general_.get(), 0, sizeof(int), stream));
            t_aux_.end(slot, stream);
            FilterArgs f{thr_.get(), dQb, d_pool, cand_.get(), cnt_.get(), fp.cap_seg, sample_seg ? fp.c0_tiles : 0};
            slot = t_main_.begin(stream);
            if (d_pad == 128)
                hipLaunchKernelGGL((topk_scores_kernel<true, true>), dim3(fp.n_seg, (nb + 127) / 128), dim3(256), 0, stream, dP, qidx, q0, nb, Qp_.get(),
                                   q_rows, ld, 0, d_pad, static_cast<float*>(nullptr), size_t(0), fp.tpb, 0, f);
            else
                hipLaunchKernelGGL((topk_scores_kernel<false, true>), dim3(fp.n_seg, (nb + 127) / 128), dim3(256), 0, stream, dP, qidx, q0, nb, Qp_.get(),
                                   q_rows, ld, 0, d_pad, static_cast<float*>(nullptr), size_t(0), fp.tpb, 0, f);
            BFH_HIP(hipGetLastError());
            t_main_.end(slot, stream);
            slot = t_aux_.begin(stream);
            a.S = nullptr; a.cand = cand_.get(); a.cand_cnt = cnt_.get(); a.n_seg = fp.n_seg; a.cap_seg = fp.cap_seg; a.list_cap = kListCap;
            a.redo = redo_.get(); a.general = general_.get(); a.thr = nullptr;   // (a.s0_* stay: the sample segment, if there is one)
            const bool wave_list = wave_select_ && p2 <= 1024;
            if (wave_list) {
                hipLaunchKernelGGL(topk_list_wave_kernel, dim3((nb + 3) / 4), dim3(256), kListLds + static_cast<size_t>(4) * p2 * 8, stream, a, nb);
                BFH_HIP(hipGetLastError());
            } else {
                launch_select(a, nb, lds_list);
            }
            t_aux_.end(slot, stream);
            int n_redo = 0, n_general = 0;
            BFH_HIP(hipMemcpyAsync(&n_redo, redo_.get(), sizeof(int), hipMemcpyDeviceToHost, stream));
            BFH_HIP(hipMemcpyAsync(&n_general, general_.get(), sizeof(int), hipMemcpyDeviceToHost, stream));
            BFH_HIP(hipStreamSynchronize(stream));
            stats.merges += n_redo;      // top-k: rows the fused path handed back to the dense path
            stats.exchanges += n_general;   // top-k: rows with ties at the k-th place (block-level list selection)
            if (n_general > 0) {
                slot = t_aux_.begin(stream);
                a.row_list = general_.get() + 1;
                launch_select(a, n_general, lds_list);
                a.row_list = nullptr;
                t_aux_.end(slot, stream);
            }
            if (n_redo == 0) continue;
            // ---- rows whose candidates did not fit (ties at the threshold, all-inadmissible rows, tiny pools): dense path ----
            std::vector<int32_t> rows(static_cast<size_t>(n_redo));
            BFH_HIP(hipMemcpy(rows.data(), redo_.get() + 1, sizeof(int32_t) * n_redo, hipMemcpyDeviceToHost));
            std::sort(rows.begin(), rows.end());
            std::vector<int32_t> side(static_cast<size_t>(n_redo) * 3);   // row of dP | original index (self exclusion) | output row
            for (int i = 0; i < n_redo; ++i) {
                const int q = q0 + rows[i];
                side[i] = gather ? indexes[q] : q;
                side[n_redo + i] = indexes[q];
                side[2 * static_cast<size_t>(n_redo) + i] = q;
            }
            redo_side_.resize(std::max(redo_side_.size(), side.size()));
            BFH_HIP(hipMemcpy(redo_side_.get(), side.data(), side.size() * 4, hipMemcpyHostToDevice));
            for (int r0 = 0; r0 < n_redo; r0 += redo_rows) {
                const int nr = std::min(redo_rows, n_redo - r0);
                slot = t_main_.begin(stream);
                launch_scores(dP, redo_side_.get() + r0, 0, nr, q_rows, ld, d_pad, ld_s, n_tiles);
                t_main_.end(slot, stream);
                SelectArgs r = base;
                r.S = S_.get(); r.ld_s = ld_s; r.cols = q_rows; r.q0 = 0;
                r.self_idx = same ? redo_side_.get() + n_redo + r0 : nullptr;
                r.out_row = redo_side_.get() + 2 * static_cast<size_t>(n_redo) + r0;
                slot = t_aux_.begin(stream);
                launch_select(r, nr, lds_dense);
                t_aux_.end(slot, stream);
            }
        }
        BFH_HIP(hipMemcpyAsync(out_keys, d_keys_.get(), sizeof(int32_t) * nq * k, hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipMemcpyAsync(out_scores, d_scores_.get(), sizeof(float) * nq * k, hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        stats.d2h_bytes += 8.0 * nq * k;
        stats.samples += static_cast<int64_t>(nq) * q_rows;
        stats.kernel_ms += t_main_.drain();
        stats.aux_ms += t_aux_.drain();
    }

    // host matrices: upload the query rows (gathered) and the candidate matrix, zero-padded to ld = d_pad
    void run_host(const int32_t* indexes, int nq, const float* P, int p_rows, int p_cols, const float* Q, int q_rows, int q_cols, const float* Qb,
                  int qb_rows, int32_t* out_keys, float* out_scores, const int32_t* pool, int pool_size, int k) {
        BFH_REQUIRE(p_cols == q_cols, "P and Q must have the same number of columns");
        BFH_REQUIRE(qb_rows == 0 || qb_rows == q_rows, "Qb must have one row per row of Q");
        if (nq == 0) return;
        ensure();
        const int d = p_cols, ld = (d + 7) / 8 * 8;
        for (int i = 0; i < nq; ++i) BFH_REQUIRE(indexes[i] >= 0 && indexes[i] < p_rows, "query index outside P");
        // query rows: when most of P is asked for and needs no padding, P goes up as it is and the kernel gathers by
        // index; otherwise the rows are gathered (and zero-padded to ld) on the host first
        const bool whole = ld == d && static_cast<int64_t>(nq) * 2 >= p_rows;
        std::vector<float> stage;
        if (whole) {
            hP_.resize(std::max(hP_.size(), static_cast<size_t>(p_rows) * ld));
            BFH_HIP(hipMemcpyAsync(hP_.get(), P, static_cast<size_t>(p_rows) * d * 4, hipMemcpyHostToDevice, stream));
        } else {
            stage.assign(static_cast<size_t>(nq) * ld, 0.f);
            for (int i = 0; i < nq; ++i)
                std::memcpy(&stage[static_cast<size_t>(i) * ld], P + static_cast<size_t>(indexes[i]) * p_cols, sizeof(float) * d);
            hP_.resize(std::max(hP_.size(), stage.size()));
            BFH_HIP(hipMemcpyAsync(hP_.get(), stage.data(), stage.size() * 4, hipMemcpyHostToDevice, stream));
        }
        hQ_.resize(std::max(hQ_.size(), static_cast<size_t>(q_rows) * ld));
        if (ld == d) {
            BFH_HIP(hipMemcpyAsync(hQ_.get(), Q, static_cast<size_t>(q_rows) * d * 4, hipMemcpyHostToDevice, stream));
        } else {
            BFH_HIP(hipMemsetAsync(hQ_.get(), 0, static_cast<size_t>(q_rows) * ld * 4, stream));
            BFH_HIP(hipMemcpy2DAsync(hQ_.get(), static_cast<size_t>(ld) * 4, Q, static_cast<size_t>(d) * 4, static_cast<size_t>(d) * 4, q_rows,
                                     hipMemcpyHostToDevice, stream));
        }
        const float* dQb = nullptr;
        if (qb_rows) {
            hQb_.resize(std::max(hQb_.size(), static_cast<size_t>(q_rows)));
            BFH_HIP(hipMemcpyAsync(hQb_.get(), Qb, static_cast<size_t>(q_rows) * 4, hipMemcpyHostToDevice, stream));
            dQb = hQb_.get();
        }
        BFH_HIP(hipStreamSynchronize(stream));   // `stage` is a local
        stats.h2d_bytes += 4.0 * ((whole ? static_cast<double>(p_rows) * d : static_cast<double>(stage.size())) + static_cast<double>(q_rows) * d + (qb_rows ? q_rows : 0));
        // host-gathered: row b of hP_ is query b (the self-exclusion still needs the original ids, which run_device uploads)
        run_device(indexes, nq, hP_.get(), whole, hQ_.get(), q_rows, d, ld, dQb, P == Q, out_keys, out_scores, pool, pool_size, k);
    }

    void quickselect(const float* scores, int rows, int cols, int32_t* result, int k) {
        BFH_REQUIRE(rows >= 0 && cols > 0, "empty score matrix");
        BFH_REQUIRE(k > 0 && k <= cols && k <= TOPK_MAX_K, "k must be in [1, min(cols, 16384)]");
        if (rows == 0) return;
        ensure();
        const size_t n = static_cast<size_t>(rows) * cols;
        S_.resize(std::max(S_.size(), n));
        BFH_HIP(hipMemcpyAsync(S_.get(), scores, n * 4, hipMemcpyHostToDevice, stream));
        d_keys_.resize(std::max(d_keys_.size(), static_cast<size_t>(rows) * k));
        const int p2 = pow2_at_least(k);
        const int cand_cap = cand_capacity(p2);
        const size_t lds = static_cast<size_t>(p2 + cand_cap) * 8;
        BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        SelectArgs a{};
        a.S = S_.get(); a.ld_s = cols; a.cols = cols; a.q0 = 0; a.rule_flt_min = 0; a.k = k; a.kk = k;
        a.out_keys = d_keys_.get(); a.out_scores = nullptr; a.p2 = p2; a.cand_cap = fast_select_ ? cand_cap : 0;
        const int slot = t_aux_.begin(stream);
        hipLaunchKernelGGL(topk_select_kernel, dim3(rows), dim3(256), lds, stream, a);
        BFH_HIP(hipGetLastError());
        t_aux_.end(slot, stream);
        BFH_HIP(hipMemcpyAsync(result, d_keys_.get(), sizeof(int32_t) * rows * k, hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        stats.h2d_bytes += 4.0 * n;
        stats.d2h_bytes += 4.0 * rows * k;
        stats.aux_ms += t_aux_.drain();
    }

    void set_mode(const std::string& name, int64_t v) {
        if (name == "flt_min_rule") flt_min_rule_ = v != 0;
        else if (name == "fast_select") fast_select_ = v != 0;   // 0: multi-pass radix select only (debug / comparison)
        else if (name == "fused") fused_ = static_cast<int>(v);      // -1: by size (default), 0: dense path only, 1: whenever d <= 128 (tests)
        else if (name == "wave_select") wave_select_ = v != 0;       // 0: block-per-row selection everywhere (comparison)
        else if (name == "fused_c0") fused_c0_ = static_cast<int>(v);   // with fused = 1: columns sampled for the thresholds (0: by rule)
        else if (name == "timing") timing = v != 0;
        else throw Error(BFH_ERR_INVALID, "unknown mode '" + name + "'");
    }

 private:
    int num_cus_ = 256;
    bool fast_select_ = true;
    int fused_ = -1, fused_c0_ = 0;
    bool wave_select_ = true;
    bool flt_min_rule_ = true;   // _core.hpp:26,115: the running list starts at FLT_MIN, so scores <= FLT_MIN are never admitted
    DevBuf<int32_t> d_idx_, d_keys_;
    DevBuf<uint32_t> d_pool_;
    DevBuf<float> d_scores_, S_, hP_, hQ_, hQb_;
    DevBuf<float4> Qp_;   // candidate matrix in MFMA operand order (topk_pack_kernel)
    // fused path: per-query thresholds, candidate segments + counts, rows handed back to the dense path
    DevBuf<float> thr_;
    DevBuf<uint2> cand_, s0_cand_;
    DevBuf<int> cnt_, redo_, general_, s0_cnt_;
    DevBuf<int32_t> redo_side_;
    EventTimer t_main_, t_aux_;
};

}  // namespace bfh

using bfh::guarded;
using bfh::TopkHandle;

extern "C" {

void* bfh_topk_create(void) {
    try {
        TopkHandle* h = new TopkHandle();
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
void bfh_topk_destroy(void* h) { delete static_cast<TopkHandle*>(h); }
int bfh_topk_set_device(void* h, int device) {
    return guarded(h, [&] { static_cast<TopkHandle*>(h)->device = device; BFH_HIP(hipSetDevice(device)); return BFH_OK; });
}
int bfh_topk_dot_topn(void* h, const int32_t* indexes, int num_queries, const float* P, int p_rows, int p_cols, const float* Q, int q_rows,
                      int q_cols, const float* Qb, int qb_rows, int32_t* out_keys, float* out_scores, const int32_t* pool, int pool_size, int k) {
    return guarded(h, [&] {
        static_cast<TopkHandle*>(h)->run_host(indexes, num_queries, P, p_rows, p_cols, Q, q_rows, q_cols, Qb, qb_rows, out_keys, out_scores, pool,
                                              pool_size, k);
        return BFH_OK;
    });
}
int bfh_topk_dot_topn_device(void* h, const int32_t* indexes, int num_queries, const float* dP, int p_rows, const float* dQ, int q_rows, int d,
                             int ld, const float* dQb, int qb_rows, int same, int32_t* out_keys, float* out_scores, const int32_t* pool,
                             int pool_size, int k) {
    return guarded(h, [&] {
        for (int i = 0; i < num_queries; ++i)
            if (indexes[i] < 0 || indexes[i] >= p_rows) throw bfh::Error(BFH_ERR_INVALID, "query index outside P");
        if (qb_rows != 0 && qb_rows != q_rows) throw bfh::Error(BFH_ERR_INVALID, "Qb must have one row per row of Q");
        static_cast<TopkHandle*>(h)->run_device(indexes, num_queries, dP, true, dQ, q_rows, d, ld, qb_rows ? dQb : nullptr, same != 0, out_keys,
                                                out_scores, pool, pool_size, k);
        return BFH_OK;
    });
}
int bfh_topk_quickselect(void* h, const float* scores, int rows, int cols, int32_t* result, int k, int sorted) {
    (void)sorted;   // always sorted
    return guarded(h, [&] { static_cast<TopkHandle*>(h)->quickselect(scores, rows, cols, result, k); return BFH_OK; });
}
int bfh_topk_set_mode(void* h, const char* name, int64_t value) {
    return guarded(h, [&] { static_cast<TopkHandle*>(h)->set_mode(name ? name : "", value); return BFH_OK; });
}
int bfh_topk_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { *out = static_cast<TopkHandle*>(h)->stats; return BFH_OK; });
}
int bfh_topk_reset_stats(void* h) {
    return guarded(h, [&] { static_cast<TopkHandle*>(h)->stats = bfh_stats{}; return BFH_OK; });
}

}  // extern "C"
