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
// FULL: the K-chunk is a whole 128 columns (nv == 16): no per-float4 guards, straight-line MFMA stream
template <bool FULL>
__global__ __launch_bounds__(256, 3) void topk_scores_kernel(const float* __restrict__ P, const int32_t* __restrict__ qidx, int q0, int nq,
                                                             const float4* __restrict__ Qp, int q_rows, int ld, int kc, int W, float* __restrict__ S,
                                                             size_t ld_s, int tiles_per_block, int accumulate) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int b0q = (blockIdx.y * 4 + wv) * 32;   // first query (batch-local) of this wave
    if (b0q >= nq) return;
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
    const int t_begin = blockIdx.x * tiles_per_block;
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
        float* Sl = S + static_cast<size_t>(b0q + 4 * half) * ld_s + t * 32 + col;   // C layout: row (e&3)+8(e>>2)+4half, col lane&31
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2);
            acc[e] = (accumulate && jok && b0q + 4 * half + r < nq) ? Sl[static_cast<size_t>(r) * ld_s] : 0.f;
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
        if (jok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2);
                if (b0q + 4 * half + r < nq) Sl[static_cast<size_t>(r) * ld_s] = acc[e];
            }
        }
    }
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
};

__global__ __launch_bounds__(256) void topk_select_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sel[];   // p2 sort entries, then cand_cap candidates
    __shared__ int hist[4096];
    __shared__ int part[256];
    __shared__ int s_misc[8];   // 0: chosen bin, 1: remaining, 2: n_gt slots, 3: run_eq, 4..7: wave eq counts / fast-path counters
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    const float* row = a.S + static_cast<size_t>(b) * a.ld_s;
    const int self = a.self_idx ? a.self_idx[a.q0 + b] : -1;
    auto key_of = [&](int j, uint32_t& key) -> bool {
        if (j == self) return false;
        if (a.pool && !((a.pool[j >> 5] >> (j & 31)) & 1u)) return false;
        float s = row[j];
        if (a.Qb) s += a.Qb[j];
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
        for (int j = tid; j < a.cols; j += 256) {
            uint32_t key;
            if (key_of(j, key)) atomicAdd(&hist[key >> 20], 1);
        }
        __syncthreads();
        find_bin(4096, a.kk);
        const int bin1 = s_misc[0], rem1 = s_misc[1], total1 = s_misc[2];
        __syncthreads();
        if (tid == 0) { s_misc[4] = 0; s_misc[5] = 0; }
        __syncthreads();
        const bool all1 = bin1 < 0;
        for (int j = tid; j < a.cols; j += 256) {
            uint32_t key;
            if (!key_of(j, key)) continue;
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
            for (int j = tid; j < a.cols; j += 256) {
                uint32_t key;
                if (key_of(j, key) && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
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
            for (int j = tid; j < a.cols; j += 256) {
                uint32_t key = 0;
                if (!key_of(j, key)) continue;
                if (take_all || key < prefix) sel[atomicAdd(&s_misc[2], 1)] = pack(key, j);
                else if (!ordered && key == prefix) sel[n_gt + atomicAdd(&s_misc[3], 1)] = pack(key, j);   // all eq_total == remaining of them
            }
        }
        if (ordered) {
            __shared__ int s_run[4];    // 0: candidates >= t so far, 1: candidates == t so far, 2: |A|, 3: done
            __shared__ int s_wave[8];   // per-wave counts of the current 256-column step: [0..3] >= t, [4..7] == t
            if (tid < 4) s_run[tid] = 0;
            __syncthreads();
            for (int phase = 0; phase < 2; ++phase) {
                // phase 0 finds |A| (the == t count when the kk-th candidate >= t arrives); phase 1 places the survivors
                const int cnt_a = s_run[2];
                __syncthreads();
                if (tid < 2) s_run[tid] = 0;
                __syncthreads();
                for (int base = 0; base < a.cols; base += 256) {
                    const int j = base + tid;
                    uint32_t key = 0;
                    const bool ok = j < a.cols && key_of(j, key);
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
    int32_t* ok = a.out_keys + static_cast<size_t>(a.q0 + b) * a.k;
    float* os = a.out_scores ? a.out_scores + static_cast<size_t>(a.q0 + b) * a.k : nullptr;
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
        // query batch: score buffer <= 2 GiB, multiple of 128 rows
        int batch = static_cast<int>(std::min<size_t>(nq, std::max<size_t>(128, ((size_t(1) << 31) / (ld_s * 4)) / 128 * 128)));
        S_.resize(std::max(S_.size(), static_cast<size_t>(batch) * ld_s));
        const int d_pad = (d + 7) / 8 * 8;
        const int n_tiles = (q_rows + 31) / 32;
        const int p2 = pow2_at_least(kk);
        const int cand_cap = cand_capacity(p2);
        const size_t lds = static_cast<size_t>(p2 + cand_cap) * 8;
        BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
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
        for (int q0 = 0; q0 < nq; q0 += batch) {
            const int nb = std::min(batch, nq - q0);
            const int qblocks = (nb + 127) / 128;
            int tpb = static_cast<int>((static_cast<int64_t>(n_tiles) * qblocks + num_cus_ * 8 - 1) / (num_cus_ * 8));
            if (tpb < 1) tpb = 1;
            const int slot = t_main_.begin(stream);
            for (int kc = 0; kc < d_pad; kc += 128) {
                const int W = std::min(128, d_pad - kc);
                if (W == 128)
                    hipLaunchKernelGGL(topk_scores_kernel<true>, dim3((n_tiles + tpb - 1) / tpb, qblocks), dim3(256), 0, stream, dP,
                                       gather ? d_idx_.get() : nullptr, q0, nb, Qp_.get() + static_cast<size_t>(kc / 128) * n_tiles * 16 * 64, q_rows, ld,
                                       kc, W, S_.get(), ld_s, tpb, kc > 0 ? 1 : 0);
                else
                    hipLaunchKernelGGL(topk_scores_kernel<false>, dim3((n_tiles + tpb - 1) / tpb, qblocks), dim3(256), 0, stream, dP,
                                       gather ? d_idx_.get() : nullptr, q0, nb, Qp_.get() + static_cast<size_t>(kc / 128) * n_tiles * 16 * 64, q_rows, ld,
                                       kc, W, S_.get(), ld_s, tpb, kc > 0 ? 1 : 0);
                BFH_HIP(hipGetLastError());
            }
            t_main_.end(slot, stream);
            SelectArgs a{};
            a.S = S_.get(); a.ld_s = ld_s; a.cols = q_rows; a.Qb = dQb; a.pool = d_pool;
            a.self_idx = same ? d_idx_.get() : nullptr;
            a.q0 = q0; a.rule_flt_min = flt_min_rule_ ? 1 : 0; a.k = k; a.kk = kk;
            a.out_keys = d_keys_.get(); a.out_scores = d_scores_.get(); a.p2 = p2; a.cand_cap = fast_select_ ? cand_cap : 0;
            const int slot2 = t_aux_.begin(stream);
            hipLaunchKernelGGL(topk_select_kernel, dim3(nb), dim3(256), lds, stream, a);
            BFH_HIP(hipGetLastError());
            t_aux_.end(slot2, stream);
            // the scores kernel of the next batch reuses S_: the stream orders it after this select
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
        else if (name == "timing") timing = v != 0;
        else throw Error(BFH_ERR_INVALID, "unknown mode '" + name + "'");
    }

 private:
    int num_cus_ = 256;
    bool fast_select_ = true;
    bool flt_min_rule_ = true;   // _core.hpp:26,115: the running list starts at FLT_MIN, so scores <= FLT_MIN are never admitted
    DevBuf<int32_t> d_idx_, d_keys_;
    DevBuf<uint32_t> d_pool_;
    DevBuf<float> d_scores_, S_, hP_, hQ_, hQb_;
    DevBuf<float4> Qp_;   // candidate matrix in MFMA operand order (topk_pack_kernel)
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
