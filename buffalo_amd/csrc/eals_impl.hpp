// eALS (element-wise ALS, He et al. SIGIR'16) on gfx950 -- handle + kernels.
//
// Reference semantics: CEALS (/root/reference/lib/algo_impl/eals/eals.cc:33-281) behind CyEALS's surface
// (/root/reference/buffalo/algo/_eals.pyx:23-67); SURVEY.md section 8(f) rank 4.  Coordinate descent over the
// latent dimensions of every row, driven by a cache of the predictions vhat of the observed entries that
// is kept in BOTH orientations and linked by index maps (eals.cc:49-100).
//
// Device formulation: one wave per row (one 1024-thread block above 4096 entries).  The row's entries sit one per lane (up to 4 per lane
// in registers: key, value, vhat, weight; longer rows keep vhat in HBM and a transposed 16-dimension slice of the gathered rows in a
// scratch slot); for every dimension d the lanes form
// the numerator / denominator terms of eals.cc:196-213 against the gathered column element Y[key][d], three
// wave reductions finish the step (the p.S[:,d] product rides on the numerator), the new coordinate is
// broadcast through LDS and folded back into vhat.  The mirrored cache of the other orientation is written once
// per row at the end instead of twice per (entry, dimension): nobody reads it before the other half-epoch and
// both copies receive the same +-pq sequence, so the stored values are identical.
// Arrays are the reference's CPU layout ([rows, d], unpadded); the device copies are padded to vdim so the
// MFMA Gramian kernel of the ALS path serves S^p = P^T P and S^q = sum_i C_i q_i q_i^T.
#pragma once
#include <algorithm>

#include "als_kernels.hpp"

namespace bfh {

struct EalsParams {
    float* X;               // side being updated [rows, vdim]
    const float* Y;         // other side
    const float* Cw;        // item weights c_i [items]
    const float* S;         // [vdim, vdim]: axis 0 -> sum_i C_i q q^T, axis 1 -> P^T P
    const int64_t* indptr;  // END offsets of this orientation
    const int32_t* keys;
    const float* vals;
    float* own;             // vhat cache of this orientation
    float* other;           // vhat cache of the other orientation
    const int64_t* map;     // own position -> other position
    int rows, d, vdim, axis;
    float alpha, reg;
    int* ticket;
    const int32_t* row_list;   // rows this launch works on (light or heavy rows of the orientation)
    int n_list;
};

constexpr int EALS_REG = 4;    // entries per lane kept in registers (rows up to 256 entries)
constexpr int EALS_DB = 16;    // dimensions per block: 16 floats = the 64-byte sector a 4-byte gather of Y[key][d] pulls anyway
constexpr int EALS_LIGHT = 64 * EALS_REG;
constexpr int EALS_HEAVY = 4096;   // above: one 1024-thread block per row; (EALS_LIGHT, EALS_HEAVY]: one wave per row, both over HBM scratch

// Round 3: the coordinate of dimension d needs Y[key][d] of EVERY entry of the row -- a 4-byte gather that costs a 64-byte sector, 128
// times per entry at d = 128: the epoch was bound by 16x its useful traffic (131 ms on the ML-20M shape).  Both kernels now walk the
// dimensions in blocks of 16: the 64-byte segment Y[key][16 b .. 16 b + 15] of every entry is fetched ONCE per block (registers for
// rows up to 256 entries; otherwise transposed into a per-row scratch [16][n] that the 16 steps then read coalesced), and the entry
// constants w - c_i (a second sector gather per step on the user side) once per row.

// rows up to EALS_LIGHT entries: one wave per row, everything in registers
__global__ __launch_bounds__(256) void eals_update_kernel(EalsParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* xs = lds + wv * p.vdim;
    const int D = p.d, vdim = p.vdim;
    while (true) {
        int item = 0;
        if (lane == 0) item = atomicAdd(p.ticket, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= p.n_list) break;
        const int x = p.row_list[item];
        const int64_t beg = x == 0 ? 0 : p.indptr[x - 1], end = p.indptr[x];
        float* xrow = p.X + static_cast<size_t>(x) * vdim;
        wave_lds_sync();
        for (int e = lane; e < vdim; e += 64) xs[e] = xrow[e];
        wave_lds_sync();
        const float cx = p.axis == 1 ? p.Cw[x] : 0.f;
        int rk[EALS_REG];
        float rv[EALS_REG], rh[EALS_REG], rw[EALS_REG], rm[EALS_REG];   // value, vhat, w = 1 + alpha v, w - c
        bool ok[EALS_REG];
#pragma unroll
        for (int s = 0; s < EALS_REG; ++s) {
            const int64_t ind = beg + s * 64 + lane;
            ok[s] = ind < end;
            rk[s] = ok[s] ? p.keys[ind] : 0;
            rv[s] = ok[s] ? p.vals[ind] : 0.f;
            rh[s] = ok[s] ? p.own[ind] : 0.f;
            rw[s] = 1.f + p.alpha * rv[s];
            rm[s] = rw[s] - (p.axis == 0 ? (ok[s] ? p.Cw[rk[s]] : 0.f) : cx);
        }
        for (int db = 0; db < D; db += EALS_DB) {
            float yb[EALS_REG][EALS_DB];
#pragma unroll
            for (int s = 0; s < EALS_REG; ++s) {
                const float4* src = reinterpret_cast<const float4*>(p.Y + static_cast<size_t>(rk[s]) * vdim + db);
#pragma unroll
                for (int q4 = 0; q4 < EALS_DB / 4; ++q4) {
                    const float4 v = ok[s] ? src[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    yb[s][4 * q4] = v.x; yb[s][4 * q4 + 1] = v.y; yb[s][4 * q4 + 2] = v.z; yb[s][4 * q4 + 3] = v.w;
                }
            }
#pragma unroll
            for (int dd = 0; dd < EALS_DB; ++dd) {
                const int d = db + dd;
                if (d < D) {
                    const float xd = xs[d];
                    float num = 0.f, den = 0.f;
#pragma unroll
                    for (int s = 0; s < EALS_REG; ++s) {
                        const float yd = yb[s][dd];
                        const float pq = xd * yd;
                        const float vf = rh[s] - pq;
                        if (ok[s]) {
                            num += (rw[s] * rv[s] - rm[s] * vf) * yd;
                            den += rm[s] * yd * yd;
                        }
                        rh[s] = vf;
                    }
                    float dot = 0.f;   // x . S[:, d] (S symmetric)
                    for (int e = lane; e < D; e += 64) dot += xs[e] * p.S[static_cast<size_t>(d) * vdim + e];
                    num = wave_sum(num);
                    den = wave_sum(den);
                    dot = wave_sum(dot);
                    const float sdd = p.S[static_cast<size_t>(d) * vdim + d];
                    if (p.axis == 0) {   // eals.cc:214-215
                        num += -dot + xd * sdd;
                        den += sdd + p.reg;
                    } else {             // eals.cc:261-262
                        num += -cx * (dot - xd * sdd);
                        den += cx * sdd + p.reg;
                    }
                    const float xn = num / den;
                    wave_lds_sync();
                    if (lane == 0) xs[d] = xn;
                    wave_lds_sync();
#pragma unroll
                    for (int s = 0; s < EALS_REG; ++s) rh[s] += xn * yb[s][dd];
                }
            }
        }
        for (int e = lane; e < D; e += 64) xrow[e] = xs[e];
#pragma unroll
        for (int s = 0; s < EALS_REG; ++s) {
            const int64_t ind = beg + s * 64 + lane;
            if (ind < end) {
                p.own[ind] = rh[s];
                p.other[p.map[ind]] = rh[s];
            }
        }
    }
}

// Rows above EALS_LIGHT entries.  BS = 64: one wave per row (ticketed, longest first); BS = 1024: one block per row, the per-dimension
// sums combined through LDS -- three barriers per dimension instead of a 131 K-entry row crawling through one wave.
// `scratch`: per thread group (1 + EALS_DB) * cap floats: w - c of every entry | Y[key][16 b + dd] as [dd][entry].
// vhat lives in HBM (coalesced); a step's closing update vhat += x_new * y is folded into the NEXT step's pass (the same two
// roundings in the same order), so every dimension costs one pass over the row: 24 bytes per entry instead of 220.
// `row_off` (BS = 1024): the scratch of list item k starts at (1 + EALS_DB) * row_off[k] floats and is as long as the row (rounded up
// to 64 entries) -- a slot of the longest row's size per block would cost gigabytes for the few 10^5-entry rows of a real catalogue.
template <int BS>
__global__ __launch_bounds__(BS) void eals_update_long_kernel(EalsParams p, float* __restrict__ scratch, int64_t cap, const int64_t* __restrict__ row_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                 // [vdim]
    float* red = lds + p.vdim;       // [3][16]  (BS = 1024)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = p.d, vdim = p.vdim;
    float* wmc = scratch + static_cast<size_t>(blockIdx.x) * (1 + EALS_DB) * cap;
    auto group_sync = [&]() {
        if constexpr (BS == 64) wave_lds_sync();
        else __syncthreads();
    };
    while (true) {
        int item = 0;
        if constexpr (BS == 64) {
            if (lane == 0) item = atomicAdd(p.ticket, 1);
            item = __builtin_amdgcn_readfirstlane(item);
        } else {
            __syncthreads();
            if (tid == 0) reinterpret_cast<int*>(red)[0] = atomicAdd(p.ticket, 1);
            __syncthreads();
            item = reinterpret_cast<int*>(red)[0];
        }
        if (item >= p.n_list) break;
        const int x = p.row_list[item];
        const int64_t beg = x == 0 ? 0 : p.indptr[x - 1], end = p.indptr[x];
        const int64_t n = end - beg;
        if (row_off) {   // this row's own slot
            cap = ((n + 63) / 64) * 64;
            wmc = scratch + static_cast<size_t>(row_off[item]) * (1 + EALS_DB);
        }
        float* const yb_ = wmc + cap;
        float* xrow = p.X + static_cast<size_t>(x) * vdim;
        group_sync();
        for (int e = tid; e < vdim; e += BS) xs[e] = xrow[e];
        const float cx = p.axis == 1 ? p.Cw[x] : 0.f;
        for (int64_t i = tid; i < n; i += BS) {
            const float w = 1.f + p.alpha * p.vals[beg + i];
            wmc[i] = w - (p.axis == 0 ? p.Cw[p.keys[beg + i]] : cx);
        }
        group_sync();
        for (int db = 0; db < D; db += EALS_DB) {
            for (int64_t i = tid; i < n; i += BS) {   // the block's 64-byte segment of every entry, transposed into [dd][entry]
                const float4* src = reinterpret_cast<const float4*>(p.Y + static_cast<size_t>(p.keys[beg + i]) * vdim + db);
#pragma unroll
                for (int q4 = 0; q4 < EALS_DB / 4; ++q4) {
                    const float4 v = src[q4];
                    yb_[(4 * q4 + 0) * cap + i] = v.x; yb_[(4 * q4 + 1) * cap + i] = v.y;
                    yb_[(4 * q4 + 2) * cap + i] = v.z; yb_[(4 * q4 + 3) * cap + i] = v.w;
                }
            }
            float pend = 0.f;   // x_new of the previous step of this block, still to be folded into vhat
            const int nd = D - db < EALS_DB ? D - db : EALS_DB;
            for (int dd = 0; dd < nd; ++dd) {
                const int d = db + dd;
                const float xd = xs[d];
                float num = 0.f, den = 0.f;
                const float* yd_ = yb_ + static_cast<size_t>(dd) * cap;
                const float* yp_ = yb_ + static_cast<size_t>(dd > 0 ? dd - 1 : 0) * cap;
                for (int64_t i = tid; i < n; i += BS) {
                    const float yd = yd_[i];
                    float vh = p.own[beg + i];
                    if (dd > 0) vh += pend * yp_[i];
                    const float pq = xd * yd;
                    const float vf = vh - pq;
                    const float v = p.vals[beg + i];
                    const float w = 1.f + p.alpha * v;
                    const float wm = wmc[i];
                    num += (w * v - wm * vf) * yd;
                    den += wm * yd * yd;
                    p.own[beg + i] = vf;
                }
                float dot = 0.f;   // x . S[:, d] (S symmetric)
                for (int e = tid; e < D; e += BS) dot += xs[e] * p.S[static_cast<size_t>(d) * vdim + e];
                num = wave_sum(num);
                den = wave_sum(den);
                dot = wave_sum(dot);
                if constexpr (BS != 64) {
                    if (lane == 0) { red[wv] = num; red[16 + wv] = den; red[32 + wv] = dot; }
                    __syncthreads();
                    num = den = dot = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < BS / 64; ++w2) { num += red[w2]; den += red[16 + w2]; dot += red[32 + w2]; }
                }
                const float sdd = p.S[static_cast<size_t>(d) * vdim + d];
                if (p.axis == 0) {   // eals.cc:214-215
                    num += -dot + xd * sdd;
                    den += sdd + p.reg;
                } else {             // eals.cc:261-262
                    num += -cx * (dot - xd * sdd);
                    den += cx * sdd + p.reg;
                }
                const float xn = num / den;
                group_sync();
                if (tid == 0) xs[d] = xn;
                group_sync();
                pend = xn;
            }
            const float* yl_ = yb_ + static_cast<size_t>(nd - 1) * cap;   // close the block: the last step's update of vhat
            for (int64_t i = tid; i < n; i += BS) p.own[beg + i] += pend * yl_[i];
            group_sync();
        }
        for (int e = tid; e < D; e += BS) xrow[e] = xs[e];
        for (int64_t i = tid; i < n; i += BS) p.other[p.map[beg + i]] = p.own[beg + i];
    }
}

// entry `ind` of a compressed side -> sort key (other id << 32 | own id); the payload is the entry's position
__global__ __launch_bounds__(256) void eals_coord_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ keys, int rows, int64_t nnz,
                                                         uint64_t* __restrict__ key_out, int64_t* __restrict__ pos_out) {
    const int64_t ind = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (ind >= nnz) return;
    const uint64_t x = static_cast<uint64_t>(lower_bound_dev<int64_t>(indptr, rows, ind + 1));   // first row whose END offset is > ind
    key_out[ind] = (static_cast<uint64_t>(static_cast<uint32_t>(keys[ind])) << 32) | x;
    pos_out[ind] = ind;
}
__global__ __launch_bounds__(256) void eals_rank_kernel(const int64_t* __restrict__ sorted_pos, int64_t nnz, int64_t* __restrict__ map) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (r < nnz) map[sorted_pos[r]] = r;
}

// vhat[ind] = X[row(ind)] . Y[key(ind)]   (eals.cc:72-78); one wave per row
__global__ __launch_bounds__(256) void eals_cache_kernel(const float* __restrict__ X, const float* __restrict__ Y, int vdim, int rows,
                                                         const int64_t* __restrict__ indptr, const int32_t* __restrict__ keys, float* __restrict__ vhat) {
    const int lane = threadIdx.x & 63;
    const int x = static_cast<int>((static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    if (x >= rows) return;
    const int64_t beg = x == 0 ? 0 : indptr[x - 1], end = indptr[x];
    const float* xr = X + static_cast<size_t>(x) * vdim;
    for (int64_t ind = beg; ind < end; ++ind) {
        const float* yr = Y + static_cast<size_t>(keys[ind]) * vdim;
        float part = 0.f;
        for (int e = lane; e < vdim; e += 64) part += xr[e] * yr[e];
        part = wave_sum(part);
        if (lane == 0) vhat[ind] = part;
    }
}

// out[r][e] = sqrt(w[r]) * F[r][e]
__global__ __launch_bounds__(256) void eals_scale_rows_kernel(const float* __restrict__ F, const float* __restrict__ w, int rows, int vdim, float* __restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < static_cast<int64_t>(rows) * vdim) out[i] = sqrtf(w[i / vdim]) * F[i];
}

// eals.cc:134-150: per-entry loss terms; out[0] += feedbacks, out[1] += squared error
__global__ __launch_bounds__(256) void eals_loss_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ keys, const float* __restrict__ vals,
                                                        const float* __restrict__ vhat, const float* __restrict__ Cw, int rows, int axis, float alpha,
                                                        double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int x = static_cast<int>((static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    if (x >= rows) return;
    const int64_t beg = x == 0 ? 0 : indptr[x - 1], end = indptr[x];
    double fb = 0.0, se = 0.0;
    for (int64_t ind = beg + lane; ind < end; ind += 64) {
        const float v = vals[ind], vh = vhat[ind], err = v - vh;
        fb += static_cast<double>((1.f + alpha * v) * err * err) - static_cast<double>(Cw[axis == 0 ? keys[ind] : x] * vh * vh);
        se += static_cast<double>(err * err);
    }
    fb = wave_sum_f64(fb);
    se = wave_sum_f64(se);
    if (lane == 0 && end > beg) {
        atomicAdd(out, fb);
        atomicAdd(out + 1, se);
    }
}

// out += sum F^2 over [rows, vdim] (pad columns are zero)
__global__ __launch_bounds__(256) void eals_sqsum_kernel(const float* __restrict__ F, int64_t n, double* __restrict__ out) {
    double part = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256)
        part += static_cast<double>(F[i]) * static_cast<double>(F[i]);
    part = wave_sum_f64(part);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, part);
}

class EalsHandle : public AlsHandle {
 public:
    ~EalsHandle() override {
        for (int k = 0; k < 2; ++k) {
            if (side_done_[k]) (void)hipEventDestroy(side_done_[k]);
            if (side_stream_[k]) (void)hipStreamDestroy(side_stream_[k]);
        }
        if (side_go_) (void)hipEventDestroy(side_go_);
    }
    bool init_eals(const char* opt_path) {   // eals.cc:19-31
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
        FF_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        FF64_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        S2_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        loss_.resize(4, true, stream);
        ticket_.resize(1, true, stream);
        inited_ = true;
        BFH_HIP(hipStreamSynchronize(stream));
        return true;
    }
    // eals.cc:33-47
    void initialize_model_eals(float* P, float* Q, float* Cw, int P_rows, int Q_rows) {
        BFH_REQUIRE(inited_, "initialize_model called before init");
        BFH_REQUIRE(P && Q && Cw && P_rows > 0 && Q_rows > 0, "initialize_model: null arrays or empty shapes");
        hostP_ = P; hostQ_ = Q; hostC_ = Cw; P_rows_ = P_rows; Q_rows_ = Q_rows;
        P_.resize(static_cast<size_t>(P_rows) * vdim_, true, stream);
        Q_.resize(static_cast<size_t>(Q_rows) * vdim_, true, stream);
        Cw_.resize(Q_rows);
        push_factors();
        BFH_HIP(hipMemcpyAsync(Cw_.get(), Cw, sizeof(float) * Q_rows, hipMemcpyHostToDevice, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        cached_[0] = cached_[1] = false;
        model_ = true;
    }
    void push_factors() {
        BFH_HIP(hipMemcpy2DAsync(P_.get(), static_cast<size_t>(vdim_) * 4, hostP_, static_cast<size_t>(d_) * 4, static_cast<size_t>(d_) * 4, P_rows_,
                                 hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemcpy2DAsync(Q_.get(), static_cast<size_t>(vdim_) * 4, hostQ_, static_cast<size_t>(d_) * 4, static_cast<size_t>(d_) * 4, Q_rows_,
                                 hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += 4.0 * d_ * (static_cast<double>(P_rows_) + Q_rows_);
    }
    void pull_factor(int axis) {
        float* host = axis == 0 ? hostP_ : hostQ_;
        const float* dev = axis == 0 ? P_.get() : Q_.get();
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        BFH_HIP(hipMemcpy2DAsync(host, static_cast<size_t>(d_) * 4, dev, static_cast<size_t>(vdim_) * 4, static_cast<size_t>(d_) * 4, rows,
                                 hipMemcpyDeviceToHost, stream));
        stats.d2h_bytes += 4.0 * d_ * rows;
    }
    struct Side {
        DevBuf<int64_t> indptr, map;
        DevBuf<int32_t> keys;
        DevBuf<float> vals, vhat;
        DevBuf<int32_t> light, mid, heavy;   // row ids with <= EALS_LIGHT / <= EALS_HEAVY / more entries, the long ones longest first (empty rows are light: the regulariser still moves them)
        int n_light = 0, n_mid = 0, n_heavy = 0;
        int64_t mid_longest = 0;
        int64_t longest = 0, heavy_entries = 0;   // sum of the heavy rows' lengths, each rounded up to 64
        DevBuf<int64_t> heavy_off;                // [n_heavy] scratch offset (entries) of every heavy row, in list order
        int64_t nnz = 0;
    };
    // eals.cc:49-100: the orientation's structure stays resident; vhat from the current factors; the index map by a host sort
    void precompute_cache(int nnz, const int64_t* indptr, const int32_t* keys, int axis) {
        BFH_REQUIRE(model_, "precompute_cache before initialize_model");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        if (cached_[axis]) return;   // eals.cc:54-56
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        BFH_REQUIRE(indptr && keys && nnz >= 0 && (rows == 0 || indptr[rows - 1] == nnz), "precompute_cache: indptr does not end at nnz");
        Side& s = side_[axis];
        s.nnz = nnz;
        s.indptr.resize(rows);
        s.keys.resize(std::max(nnz, 1));
        s.vals.resize(std::max(nnz, 1));
        s.vhat.resize(std::max(nnz, 1));
        s.map.resize(std::max(nnz, 1));
        BFH_HIP(hipMemcpyAsync(s.indptr.get(), indptr, sizeof(int64_t) * rows, hipMemcpyHostToDevice, stream));
        if (nnz) BFH_HIP(hipMemcpyAsync(s.keys.get(), keys, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, stream));
        {
            std::vector<int32_t> li, md, hv;
            int64_t prev = 0;
            s.longest = 0;
            for (int x = 0; x < rows; ++x) {
                const int64_t n = indptr[x] - prev;
                (n > EALS_HEAVY ? hv : (n > EALS_LIGHT ? md : li)).push_back(x);
                s.longest = std::max(s.longest, n);
                prev = indptr[x];
            }
            auto longer = [&](int a, int b) { return indptr[a] - (a ? indptr[a - 1] : 0) > indptr[b] - (b ? indptr[b - 1] : 0); };
            std::stable_sort(hv.begin(), hv.end(), longer);   // longest first
            std::stable_sort(md.begin(), md.end(), longer);
            s.n_light = static_cast<int>(li.size());
            s.n_mid = static_cast<int>(md.size());
            s.mid_longest = md.empty() ? 0 : indptr[md[0]] - (md[0] ? indptr[md[0] - 1] : 0);
            s.n_heavy = static_cast<int>(hv.size());
            s.light.resize(std::max<size_t>(1, li.size()));
            s.mid.resize(std::max<size_t>(1, md.size()));
            s.heavy.resize(std::max<size_t>(1, hv.size()));
            if (!li.empty()) BFH_HIP(hipMemcpyAsync(s.light.get(), li.data(), li.size() * 4, hipMemcpyHostToDevice, stream));
            if (!md.empty()) BFH_HIP(hipMemcpyAsync(s.mid.get(), md.data(), md.size() * 4, hipMemcpyHostToDevice, stream));
            if (!hv.empty()) BFH_HIP(hipMemcpyAsync(s.heavy.get(), hv.data(), hv.size() * 4, hipMemcpyHostToDevice, stream));
            std::vector<int64_t> off(hv.size());
            s.heavy_entries = 0;
            for (size_t k = 0; k < hv.size(); ++k) {
                off[k] = s.heavy_entries;
                const int64_t n = indptr[hv[k]] - (hv[k] ? indptr[hv[k] - 1] : 0);
                s.heavy_entries += ((n + 63) / 64) * 64;
            }
            s.heavy_off.resize(std::max<size_t>(1, off.size()));
            if (!off.empty()) BFH_HIP(hipMemcpyAsync(s.heavy_off.get(), off.data(), off.size() * 8, hipMemcpyHostToDevice, stream));
            BFH_HIP(hipStreamSynchronize(stream));   // off is a local
            BFH_HIP(hipStreamSynchronize(stream));   // li / hv are locals
        }
        const int slot = t_aux_.begin(stream);
        if (nnz) {
            // rank of every entry in (other id, own id) order = its position in the other orientation (eals.cc:81-99): one
            // radix sort of (other id << 32 | own id) with the entry's position as payload, then map[payload[r]] = r
            DevBuf<uint64_t> ka, kb;
            DevBuf<int64_t> va, vb;
            DevBuf<char> tmp;
            ka.resize(nnz); kb.resize(nnz); va.resize(nnz); vb.resize(nnz);
            const unsigned blocks = static_cast<unsigned>((static_cast<int64_t>(nnz) + 255) / 256);
            hipLaunchKernelGGL(eals_coord_kernel, dim3(blocks), dim3(256), 0, stream, s.indptr.get(), s.keys.get(), rows, static_cast<int64_t>(nnz), ka.get(),
                               va.get());
            BFH_HIP(hipGetLastError());
            int bits = 33;
            while (bits < 64 && (uint64_t(1) << (bits - 32)) < static_cast<uint64_t>(axis == 0 ? Q_rows_ : P_rows_)) ++bits;
            device_sort_pairs_u64(ka.get(), kb.get(), va.get(), vb.get(), nnz, bits, tmp, stream);
            hipLaunchKernelGGL(eals_rank_kernel, dim3(blocks), dim3(256), 0, stream, vb.get(), static_cast<int64_t>(nnz), s.map.get());
            BFH_HIP(hipGetLastError());
            BFH_HIP(hipStreamSynchronize(stream));   // the sort buffers are locals
        }
        hipLaunchKernelGGL(eals_cache_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, stream, axis == 0 ? P_.get() : Q_.get(),
                           axis == 0 ? Q_.get() : P_.get(), vdim_, rows, s.indptr.get(), s.keys.get(), s.vhat.get());
        BFH_HIP(hipGetLastError());
        t_aux_.end(slot, stream);
        BFH_HIP(hipStreamSynchronize(stream));
        stats.h2d_bytes += 8.0 * rows + 4.0 * nnz;
        stats.aux_ms += t_aux_.drain();
        cached_[axis] = true;
    }
    // S for the update of `axis`: axis 0 -> sum_i C_i q q^T (eals.cc:179-191), axis 1 -> P^T P (:234-235); lands in FF_
    void gram_for(int axis) {
        if (axis == 0) {
            CQ_.resize(static_cast<size_t>(Q_rows_) * vdim_);
            const int64_t n = static_cast<int64_t>(Q_rows_) * vdim_;
            hipLaunchKernelGGL(eals_scale_rows_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, Q_.get(), Cw_.get(), Q_rows_, vdim_,
                               CQ_.get());
            BFH_HIP(hipGetLastError());
            gramian_of(CQ_.get(), Q_rows_);
        } else {
            gramian_of(P_.get(), P_rows_);
        }
    }
    // eals.cc:102-115: false until both caches exist
    bool update(const int64_t* indptr, const int32_t* keys, const float* vals, int axis) {
        BFH_REQUIRE(model_, "update before initialize_model");
        BFH_REQUIRE(axis == 0 || axis == 1, "axis must be 0 or 1");
        if (!(cached_[0] && cached_[1])) return false;
        (void)indptr; (void)keys;   // the structure was bound by precompute_cache
        Side& s = side_[axis];
        if (s.nnz) BFH_HIP(hipMemcpyAsync(s.vals.get(), vals, sizeof(float) * s.nnz, hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += 4.0 * s.nnz;
        gram_for(axis);
        EalsParams p{};
        p.X = axis == 0 ? P_.get() : Q_.get();
        p.Y = axis == 0 ? Q_.get() : P_.get();
        p.Cw = Cw_.get(); p.S = FF_.get();
        p.indptr = s.indptr.get(); p.keys = s.keys.get(); p.vals = s.vals.get();
        p.own = s.vhat.get(); p.other = side_[1 - axis].vhat.get(); p.map = s.map.get();
        p.rows = axis == 0 ? P_rows_ : Q_rows_;
        p.d = d_; p.vdim = vdim_; p.axis = axis; p.alpha = alpha_; p.reg = axis == 0 ? reg_u_ : reg_i_;
        // three launches over disjoint rows (heavy / mid / light), each with its own ticket, side by side on three streams: the few
        // blocks of the longest rows would otherwise leave most of the chip idle while they crawl through their 128 steps
        if (tickets3_.size() < 3) tickets3_.resize(3);
        if (!side_stream_[0]) {
            for (int k = 0; k < 2; ++k) {
                BFH_HIP(hipStreamCreateWithFlags(&side_stream_[k], hipStreamNonBlocking));
                BFH_HIP(hipEventCreateWithFlags(&side_done_[k], hipEventDisableTiming));
            }
            BFH_HIP(hipEventCreateWithFlags(&side_go_, hipEventDisableTiming));
        }
        BFH_HIP(hipMemsetAsync(tickets3_.get(), 0, 3 * sizeof(int), stream));
        const int slot = t_main_.begin(stream);
        BFH_HIP(hipEventRecord(side_go_, stream));
        if (s.n_heavy) {   // the longest rows, one 1024-thread block each (main stream)
            EalsParams ph = p;
            ph.row_list = s.heavy.get();
            ph.n_list = s.n_heavy;
            ph.ticket = tickets3_.get();
            const int groups = std::min(s.n_heavy, num_cus_ * 2);
            const size_t need = static_cast<size_t>(s.heavy_entries) * (1 + EALS_DB);   // every heavy row has its own slot
            if (scratch_h_.size() < need) scratch_h_.resize(need);
            hipLaunchKernelGGL(eals_update_long_kernel<1024>, dim3(groups), dim3(1024), (static_cast<size_t>(vdim_) + 48) * sizeof(float), stream, ph,
                               scratch_h_.get(), int64_t(0), static_cast<const int64_t*>(s.heavy_off.get()));
            BFH_HIP(hipGetLastError());
        }
        if (s.n_mid) {     // (EALS_LIGHT, EALS_HEAVY] entries: one wave per row over a scratch slot
            EalsParams pm = p;
            pm.row_list = s.mid.get();
            pm.n_list = s.n_mid;
            pm.ticket = tickets3_.get() + 1;
            const int groups = std::min(s.n_mid, num_cus_ * 16);
            // slot = the longest mid row (the list is sorted longest first) rounded up to a wave, not the class bound EALS_HEAVY: mid rows
            // may be as short as EALS_LIGHT + 1 entries, and 4,096 groups x 17 x EALS_HEAVY floats were 1.1 GB whatever the rows' lengths
            const int64_t cap = std::min<int64_t>(EALS_HEAVY, ((s.mid_longest + 63) / 64) * 64);
            const size_t need = static_cast<size_t>(groups) * (1 + EALS_DB) * cap;
            if (scratch_m_.size() < need) scratch_m_.resize(need);
            BFH_HIP(hipStreamWaitEvent(side_stream_[0], side_go_, 0));
            hipLaunchKernelGGL(eals_update_long_kernel<64>, dim3(groups), dim3(64), (static_cast<size_t>(vdim_) + 48) * sizeof(float), side_stream_[0], pm,
                               scratch_m_.get(), cap, static_cast<const int64_t*>(nullptr));
            BFH_HIP(hipGetLastError());
            BFH_HIP(hipEventRecord(side_done_[0], side_stream_[0]));
            BFH_HIP(hipStreamWaitEvent(stream, side_done_[0], 0));
        }
        if (s.n_light) {
            p.row_list = s.light.get();
            p.n_list = s.n_light;
            p.ticket = tickets3_.get() + 2;
            int blocks = (s.n_light + 3) / 4;
            if (blocks > num_cus_ * 4) blocks = num_cus_ * 4;
            BFH_HIP(hipStreamWaitEvent(side_stream_[1], side_go_, 0));
            hipLaunchKernelGGL(eals_update_kernel, dim3(blocks), dim3(256), static_cast<size_t>(4) * vdim_ * sizeof(float), side_stream_[1], p);
            BFH_HIP(hipGetLastError());
            BFH_HIP(hipEventRecord(side_done_[1], side_stream_[1]));
            BFH_HIP(hipStreamWaitEvent(stream, side_done_[1], 0));
        }
        t_main_.end(slot, stream);
        pull_factor(axis);
        BFH_HIP(hipStreamSynchronize(stream));
        stats.kernel_ms += t_main_.drain();
        stats.samples += s.nnz;
        stats.launches += 1;
        return true;
    }
    // eals.cc:117-174 -> (rmse, loss); accumulated in double (the reference sums tens of millions of floats in a float)
    void estimate_loss(int nnz, const int64_t* indptr, const int32_t* keys, const float* vals, int axis, float* rmse, float* loss) {
        (void)indptr; (void)keys;
        *rmse = 0.f;
        *loss = 0.f;
        if (!(cached_[0] && cached_[1])) return;
        Side& s = side_[axis];
        BFH_REQUIRE(nnz == s.nnz, "estimate_loss: nnz differs from the cached structure");
        if (s.nnz) BFH_HIP(hipMemcpyAsync(s.vals.get(), vals, sizeof(float) * s.nnz, hipMemcpyHostToDevice, stream));
        BFH_HIP(hipMemsetAsync(loss_.get(), 0, 4 * sizeof(double), stream));
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        hipLaunchKernelGGL(eals_loss_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, stream, s.indptr.get(), s.keys.get(), s.vals.get(),
                           s.vhat.get(), Cw_.get(), rows, axis, alpha_, loss_.get());
        hipLaunchKernelGGL(eals_sqsum_kernel, dim3(num_cus_ * 4), dim3(256), 0, stream, P_.get(), static_cast<int64_t>(P_rows_) * vdim_, loss_.get() + 2);
        hipLaunchKernelGGL(eals_sqsum_kernel, dim3(num_cus_ * 4), dim3(256), 0, stream, Q_.get(), static_cast<int64_t>(Q_rows_) * vdim_, loss_.get() + 3);
        BFH_HIP(hipGetLastError());
        // <S^p, S^q> over the full symmetric matrices (misc/blas.hpp:49-63 mirrors the computed triangle)
        const size_t nn = static_cast<size_t>(vdim_) * vdim_;
        gram_for(1);
        BFH_HIP(hipMemcpyAsync(S2_.get(), FF_.get(), nn * sizeof(float), hipMemcpyDeviceToDevice, stream));
        gram_for(0);
        std::vector<float> sp(nn), sq(nn);
        double acc[4];
        BFH_HIP(hipMemcpyAsync(sp.data(), S2_.get(), nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipMemcpyAsync(sq.data(), FF_.get(), nn * sizeof(float), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipMemcpyAsync(acc, loss_.get(), 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        double ip = 0.0;
        for (size_t k = 0; k < nn; ++k) ip += static_cast<double>(sp[k]) * static_cast<double>(sq[k]);
        const double reg = static_cast<double>(reg_u_) * acc[2] + static_cast<double>(reg_i_) * acc[3];
        *rmse = static_cast<float>(std::sqrt(acc[1] / std::max(nnz, 1)));
        *loss = static_cast<float>(acc[0] + ip + reg);
    }

    float* hostC_ = nullptr;
    bool cached_[2] = {false, false};
    Side side_[2];
    DevBuf<float> Cw_, CQ_, S2_;
    DevBuf<float> scratch_m_, scratch_h_;   // per thread group (1 + EALS_DB) * cap floats (eals_update_long_kernel)
    DevBuf<int> tickets3_;
    hipStream_t side_stream_[2] = {nullptr, nullptr};
    hipEvent_t side_done_[2] = {nullptr, nullptr}, side_go_ = nullptr;
};

}  // namespace bfh
