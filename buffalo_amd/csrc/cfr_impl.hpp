// CFR (CoFactor) on gfx950 -- handle + the small kernels that are not shared with ALS.
//
// Reference semantics: CCFR (/root/reference/lib/algo_impl/cfr/cfr.cc:29-313) behind CyCFR's surface
// (/root/reference/buffalo/algo/_cfr.pyx:25-71); SURVEY.md section 8(f) rank 4: "CFR calls the same
// _leastsquare".  The three row updates are instances of the ALS dense path (als_gram_kernel ->
// scratch slot -> als_solve_kernel / als_dense_solve):
//   user     A = l (FF_I + sum alpha v i i^T) + reg_u I,                         y = l sum (1 + alpha v) i
//   item     A = l (FF_U + sum alpha v u u^T) + sum c c^T + reg_i I,             y = l sum (1 + alpha v) u + sum (v - Ib_x - Cb_c) c
//   context  A = sum i i^T + reg_c I,                                            y = sum (v - Cb_x - Ib_i) i
// -- the item system is built by two Gramian passes adding into one slot (p.accumulate, p.out_scale,
// p.ctx) -- followed by the bias refresh from the UPDATED row (cfr.cc:244-250, 303-309).
// Host arrays are [rows, d] unpadded (the reference's CPU layout); the device copies are padded to vdim.
#pragma once
#include "als_kernels.hpp"

namespace bfh {

// b[x] = sum_k (v_k - F[x].G[key_k] - bias_other[key_k]) / (n + 1e-10) over the row's entries; one wave per row.
// `indptr_also`: rows that were solved because of entries in a second matrix get 0 / 1e-10 = 0 when they have none here
// (cfr.cc:244-250 runs for every item that was not skipped at :177-180).
__global__ __launch_bounds__(256) void cfr_bias_kernel(const float* __restrict__ F, const float* __restrict__ G, int vdim, const int64_t* __restrict__ indptr,
                                                       int64_t shift, int start_x, int next_x, const int32_t* __restrict__ keys,
                                                       const float* __restrict__ vals, const float* __restrict__ bias_other, float* __restrict__ bias_self,
                                                       const int64_t* __restrict__ indptr_also) {
    const int lane = threadIdx.x & 63;
    const int x = start_x + static_cast<int>((static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    if (x >= next_x) return;
    const int64_t beg = (x == 0 ? 0 : indptr[x - 1]) - shift, end = indptr[x] - shift;
    if (end == beg) {
        if (indptr_also && lane == 0 && indptr_also[x] - (x == 0 ? 0 : indptr_also[x - 1]) > 0) bias_self[x] = 0.f;
        return;
    }
    const float* fx = F + static_cast<size_t>(x) * vdim;
    float b = 0.f;
    for (int64_t k = beg; k < end; ++k) {
        const int c = keys[k];
        const float* gc = G + static_cast<size_t>(c) * vdim;
        float part = 0.f;
        for (int e = lane; e < vdim; e += 64) part += fx[e] * gc[e];
        b += vals[k] - wave_sum(part) - bias_other[c];
    }
    if (lane == 0) bias_self[x] = b / (static_cast<float>(end - beg) + 1e-10f);
}

// out += scale * sum_x |F[x]|^2 over rows of [start_x, next_x) that have entries in `indptr` (or in `indptr2`)
__global__ __launch_bounds__(256) void cfr_sqnorm_kernel(const float* __restrict__ F, int vdim, const int64_t* __restrict__ indptr,
                                                         const int64_t* __restrict__ indptr2, int start_x, int next_x, double scale, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int x = start_x + static_cast<int>((static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    if (x >= next_x) return;
    const bool has = (indptr[x] - (x == 0 ? 0 : indptr[x - 1])) > 0 || (indptr2 && (indptr2[x] - (x == 0 ? 0 : indptr2[x - 1])) > 0);
    if (!has) return;
    const float* fx = F + static_cast<size_t>(x) * vdim;
    float part = 0.f;
    for (int e = lane; e < vdim; e += 64) part += fx[e] * fx[e];
    part = wave_sum(part);
    if (lane == 0) atomicAdd(out, scale * static_cast<double>(part));
}

// item loss terms of cfr.cc:176-190, 212-228 on the row BEFORE its update; one wave per row
__global__ __launch_bounds__(256) void cfr_item_loss_kernel(const float* __restrict__ I, const float* __restrict__ U, const float* __restrict__ C,
                                                            const float* __restrict__ FF, int d, int vdim, const int64_t* __restrict__ ip_u,
                                                            int64_t shift_u, const int32_t* __restrict__ keys_u, const float* __restrict__ vals_u,
                                                            const int64_t* __restrict__ ip_c, int64_t shift_c, const int32_t* __restrict__ keys_c,
                                                            const float* __restrict__ vals_c, const float* __restrict__ Ib, const float* __restrict__ Cb,
                                                            int start_x, int next_x, float alpha, float l, float reg_i, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int x = start_x + static_cast<int>((static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    if (x >= next_x) return;
    const int64_t bu = (x == 0 ? 0 : ip_u[x - 1]) - shift_u, eu = ip_u[x] - shift_u;
    const int64_t bc = (x == 0 ? 0 : ip_c[x - 1]) - shift_c, ec = ip_c[x] - shift_c;
    if (eu == bu && ec == bc) return;
    const float* ix = I + static_cast<size_t>(x) * vdim;
    auto dot = [&](const float* a, const float* b) {
        float part = 0.f;
        for (int e = lane; e < vdim; e += 64) part += a[e] * b[e];
        return wave_sum(part);
    };
    float loss = 0.f;
    {   // (I_x FF) . I_x
        float part = 0.f;
        for (int i = lane; i < d; i += 64) {
            float t = 0.f;
            for (int j = 0; j < d; ++j) t += ix[j] * FF[static_cast<size_t>(j) * vdim + i];
            part += t * ix[i];
        }
        loss += wave_sum(part);
    }
    for (int64_t k = bu; k < eu; ++k) {
        const float w = vals_u[k] * alpha;
        const float dt = dot(ix, U + static_cast<size_t>(keys_u[k]) * vdim);
        loss += (-dt * dt + (1 + w) * (dt - 1) * (dt - 1));
    }
    double total = static_cast<double>(loss * l);
    loss = 0.f;
    for (int64_t k = bc; k < ec; ++k) {
        const int c = keys_c[k];
        const float err = vals_c[k] - dot(ix, C + static_cast<size_t>(c) * vdim) - Ib[x] - Cb[c];
        loss += err * err;
    }
    total += static_cast<double>(loss) + static_cast<double>(reg_i * dot(ix, ix));
    if (lane == 0) atomicAdd(out, total);
}

class CfrHandle : public AlsHandle {
 public:
    bool init_cfr(const char* opt_path) {
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
        BFH_REQUIRE(vdim_ <= 128, "CFR: d > 128 is not supported by the gfx950 kernels yet");
        alpha_ = static_cast<float>(opt_.num("alpha"));
        l_ = static_cast<float>(opt_.num("l"));
        reg_u_ = static_cast<float>(opt_.num("reg_u"));
        reg_i_ = static_cast<float>(opt_.num("reg_i"));
        reg_c_ = static_cast<float>(opt_.num("reg_c"));
        eps_ = static_cast<float>(opt_.num_or("eps", 1e-10));
        cg_tol_ = static_cast<float>(opt_.num_or("cg_tolerance_", 0.0));   // sic (cfr.cc:38): the key nobody sets
        num_cg_max_iters_ = static_cast<int>(opt_.num_or("num_cg_max_iters", 3));
        compute_loss_ = false;                                             // ALS loss plumbing stays off
        cfr_loss_ = opt_.boolean_or("compute_loss", false);                // cfr.cc:44
        const std::string optimizer = opt_.str("optimizer");
        if (optimizer == "llt") code_ = 0;
        else if (optimizer == "ldlt") code_ = 1;
        else if (optimizer == "manual_cg") code_ = 2;
        else throw Error(BFH_ERR_UNSUPPORTED, "optimizer '" + optimizer + "' is not implemented on gfx950 (supported: llt, ldlt, manual_cg)");
        FF_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        FF64_.resize(static_cast<size_t>(vdim_) * vdim_, true, stream);
        loss_.resize(2, true, stream);
        ticket_.resize(1, true, stream);
        inited_ = true;
        BFH_HIP(hipStreamSynchronize(stream));
        return true;
    }

    struct Emb {
        float* host = nullptr;
        int rows = 0, cols = 0;   // cols: d for factor matrices, 1 for biases
        DevBuf<float> dev;
    };
    Emb& emb(const std::string& t) {
        if (t == "user") return U_;
        if (t == "item") return I_;
        if (t == "context") return C_;
        if (t == "item_bias") return Ib_;
        if (t == "context_bias") return Cb_;
        throw Error(BFH_ERR_INVALID, "unknown embedding '" + t + "' (user, item, context, item_bias, context_bias)");
    }
    // cfr.cc:70-82: binds the caller's array; the device copy is padded to vdim
    void set_embedding(float* data, int size, const std::string& t) {
        BFH_REQUIRE(inited_, "set_embedding called before init");
        BFH_REQUIRE(data && size > 0, "set_embedding: null array or empty shape");
        Emb& e = emb(t);
        const bool bias = t == "item_bias" || t == "context_bias";
        e.host = data; e.rows = size; e.cols = bias ? 1 : d_;
        const int ld = bias ? 1 : vdim_;
        e.dev.resize(static_cast<size_t>(size) * ld, true, stream);
        BFH_HIP(hipMemcpy2DAsync(e.dev.get(), static_cast<size_t>(ld) * 4, data, static_cast<size_t>(e.cols) * 4, static_cast<size_t>(e.cols) * 4, size,
                                 hipMemcpyHostToDevice, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        stats.h2d_bytes += 4.0 * size * e.cols;
    }
    void write_back(Emb& e, int start_x, int next_x) {   // rows [start_x, next_x) -> the caller's array
        const int ld = e.cols == 1 ? 1 : vdim_;
        BFH_HIP(hipMemcpy2DAsync(e.host + static_cast<size_t>(start_x) * e.cols, static_cast<size_t>(e.cols) * 4,
                                 e.dev.get() + static_cast<size_t>(start_x) * ld, static_cast<size_t>(ld) * 4, static_cast<size_t>(e.cols) * 4,
                                 next_x - start_x, hipMemcpyDeviceToHost, stream));
        stats.d2h_bytes += 4.0 * (next_x - start_x) * e.cols;
    }
    // cfr.cc:85-90
    void precompute_cfr(const std::string& t) {
        BFH_REQUIRE(t == "user" || t == "item", "precompute: obj_type must be user or item");
        Emb& e = emb(t);
        BFH_REQUIRE(e.host, "precompute before set_embedding");
        // AlsHandle::precompute(0) takes the Gramian of Q_: alias it for the call
        model_ = true;
        Q_rows_ = e.rows;
        gramian_of(e.dev.get(), e.rows);
    }

    struct Csr {   // one chunk on the device
        DevBuf<int64_t> indptr;
        DevBuf<int32_t> keys;
        DevBuf<float> vals;
        int64_t shift = 0, nnz = 0;
        const int64_t* host_ip = nullptr;
    };
    void upload(Csr& c, int rows_total_hint, int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals) {
        (void)rows_total_hint;
        // the reference hands the FULL indptr (end offsets) + the chunk's keys/vals (buffered_data.py:99-118)
        const int64_t beg = start_x == 0 ? 0 : indptr[start_x - 1];
        const int64_t end = next_x == 0 ? 0 : indptr[next_x - 1];
        c.shift = beg; c.nnz = end - beg; c.host_ip = indptr;
        c.indptr.resize(std::max<size_t>(c.indptr.size(), next_x));
        BFH_HIP(hipMemcpyAsync(c.indptr.get(), indptr, sizeof(int64_t) * next_x, hipMemcpyHostToDevice, stream));
        c.keys.resize(std::max<size_t>(c.keys.size(), std::max<int64_t>(c.nnz, 1)));
        c.vals.resize(std::max<size_t>(c.vals.size(), std::max<int64_t>(c.nnz, 1)));
        if (c.nnz) {
            BFH_HIP(hipMemcpyAsync(c.keys.get(), keys, sizeof(int32_t) * c.nnz, hipMemcpyHostToDevice, stream));
            BFH_HIP(hipMemcpyAsync(c.vals.get(), vals, sizeof(float) * c.nnz, hipMemcpyHostToDevice, stream));
        }
        stats.h2d_bytes += 8.0 * next_x + 8.0 * c.nnz;
    }

    AlsParams base_params(Emb& self, Emb& other, Csr& c, int start_x, int next_x, float reg) {
        AlsParams p{};
        p.P = self.dev.get(); p.Q = other.dev.get(); p.FF = FF_.get();
        p.indptr = c.indptr.get(); p.keys = c.keys.get(); p.vals = c.vals.get(); p.shift = c.shift;
        p.start_x = start_x; p.next_x = next_x; p.d = d_; p.vdim = vdim_; p.op_rows = other.rows; p.block_size = 32;
        p.alpha = alpha_; p.reg = reg; p.eps = eps_; p.cg_tol = cg_tol_; p.num_cg_max_iters = num_cg_max_iters_;
        p.loss = loss_.get(); p.ticket = ticket_.get(); p.solver = static_cast<int>(code_);
        p.out_scale = 1.0f; p.ff_scale = 1.0f;
        return p;
    }
    // one Gramian pass of the chunk's rows into the slots (row - start_x)
    void gram_pass(const AlsParams& p0, int cache_axis, const Csr& c, int start_x, int next_x) {
        const WorkList& wl = work_list(cache_axis, start_x, next_x, c.host_ip, c.shift);
        if (wl.n_work == 0) return;
        AlsParams p = p0;
        BFH_HIP(hipMemsetAsync(ticket_.get(), 0, sizeof(int), stream));
        const int T = vdim_ / 32;
        int blocks = (wl.n_work + 3) / 4;
        if (blocks > num_cus_ * 4) blocks = num_cus_ * 4;
        const bool big = static_cast<uint64_t>(p.op_rows) * vdim_ * 4 >= (1ull << 32);
        const int nrows = next_x - start_x;
#define BFH_GK(TT)                                                                                                                          \
    do {                                                                                                                                    \
        if (big) hipLaunchKernelGGL((als_gram_kernel<TT, false, false, true>), dim3(blocks), dim3(256), 0, stream, p, wl.work.get(), wl.n_work, \
                                    gscratch_.get(), nrows);                                                                                \
        else hipLaunchKernelGGL((als_gram_kernel<TT, false, false, false>), dim3(blocks), dim3(256), 0, stream, p, wl.work.get(), wl.n_work,    \
                                gscratch_.get(), nrows);                                                                                    \
    } while (0)
        if (T <= 1) BFH_GK(1);
        else if (T <= 2) BFH_GK(2);
        else if (T <= 3) BFH_GK(3);
        else BFH_GK(4);
#undef BFH_GK
        BFH_HIP(hipGetLastError());
    }
    void zero_slots(int nrows) {
        const size_t need = static_cast<size_t>(nrows) * als_slot_floats(vdim_);
        if (gscratch_.size() < need) gscratch_.resize(need);
        BFH_HIP(hipMemsetAsync(gscratch_.get(), 0, need * sizeof(float), stream));
    }
    // solves every row of [start_x, next_x) that has entries in ip_a (or ip_b) from its slot
    void solve_rows(const AlsParams& p0, int start_x, int next_x, const int64_t* ip_a, const int64_t* ip_b) {
        std::vector<AlsHeavy> sv;
        for (int x = start_x; x < next_x; ++x) {
            const int64_t na = ip_a[x] - (x == 0 ? 0 : ip_a[x - 1]);
            const int64_t nb = ip_b ? ip_b[x] - (x == 0 ? 0 : ip_b[x - 1]) : 0;
            if (na + nb > 0) sv.push_back({x, x - start_x, na + nb});
        }
        if (sv.empty()) return;
        solve_list_.resize(std::max(solve_list_.size(), sv.size()));
        BFH_HIP(hipMemcpyAsync(solve_list_.get(), sv.data(), sv.size() * sizeof(AlsHeavy), hipMemcpyHostToDevice, stream));
        BFH_HIP(hipStreamSynchronize(stream));   // sv is a local
        const size_t lds_h = als_gs_lds_bytes(vdim_);
        BFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(als_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_h)));
        int sblocks = num_cus_ * static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_h)));
        if (sblocks > static_cast<int>(sv.size())) sblocks = static_cast<int>(sv.size());
        hipLaunchKernelGGL(als_solve_kernel, dim3(sblocks), dim3(256), lds_h, stream, p0, solve_list_.get(), static_cast<int>(sv.size()), gscratch_.get(),
                           static_cast<int>(code_));
        BFH_HIP(hipGetLastError());
    }
    double read_loss() {
        double v = 0.0;
        BFH_HIP(hipMemcpyAsync(&v, loss_.get(), sizeof(double), hipMemcpyDeviceToHost, stream));
        BFH_HIP(hipStreamSynchronize(stream));
        return v;
    }
    static unsigned wave_blocks(int nrows) { return static_cast<unsigned>((nrows + 3) / 4); }

    // cfr.cc:92-146
    double partial_update_user(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals) {
        BFH_REQUIRE(U_.host && I_.host, "partial_update_user before set_embedding(user / item)");
        BFH_REQUIRE(0 <= start_x && start_x <= next_x && next_x <= U_.rows, "partial_update_user: bad row range");
        if (next_x == start_x) return 0.0;
        upload(cu_, U_.rows, start_x, next_x, indptr, keys, vals);
        const int slot = t_main_.begin(stream);
        AlsParams p = base_params(U_, I_, cu_, start_x, next_x, reg_u_);
        p.out_scale = l_; p.ff_scale = l_;
        zero_slots(next_x - start_x);
        p.accumulate = 1;
        gram_pass(p, 20, cu_, start_x, next_x);
        solve_rows(p, start_x, next_x, indptr, nullptr);
        BFH_HIP(hipMemsetAsync(loss_.get(), 0, 2 * sizeof(double), stream));
        if (cfr_loss_)   // reg_u * |U_x|^2 of the UPDATED rows (cfr.cc:139-140)
            hipLaunchKernelGGL(cfr_sqnorm_kernel, dim3(wave_blocks(next_x - start_x)), dim3(256), 0, stream, U_.dev.get(), vdim_, cu_.indptr.get(), nullptr,
                               start_x, next_x, static_cast<double>(reg_u_), loss_.get());
        t_main_.end(slot, stream);
        write_back(U_, start_x, next_x);
        const double loss = read_loss();
        stats.kernel_ms += t_main_.drain();
        stats.samples += cu_.nnz;
        return loss;
    }
    // cfr.cc:148-255
    double partial_update_item(int start_x, int next_x, const int64_t* ip_u, const int32_t* keys_u, const float* vals_u, const int64_t* ip_c,
                               const int32_t* keys_c, const float* vals_c) {
        BFH_REQUIRE(U_.host && I_.host && C_.host && Ib_.host && Cb_.host, "partial_update_item before set_embedding of all five arrays");
        BFH_REQUIRE(0 <= start_x && start_x <= next_x && next_x <= I_.rows, "partial_update_item: bad row range");
        if (next_x == start_x) return 0.0;
        upload(ciu_, I_.rows, start_x, next_x, ip_u, keys_u, vals_u);
        upload(cic_, I_.rows, start_x, next_x, ip_c, keys_c, vals_c);
        const int slot = t_main_.begin(stream);
        BFH_HIP(hipMemsetAsync(loss_.get(), 0, 2 * sizeof(double), stream));
        if (cfr_loss_)
            hipLaunchKernelGGL(cfr_item_loss_kernel, dim3(wave_blocks(next_x - start_x)), dim3(256), 0, stream, I_.dev.get(), U_.dev.get(), C_.dev.get(),
                               FF_.get(), d_, vdim_, ciu_.indptr.get(), ciu_.shift, ciu_.keys.get(), ciu_.vals.get(), cic_.indptr.get(), cic_.shift,
                               cic_.keys.get(), cic_.vals.get(), Ib_.dev.get(), Cb_.dev.get(), start_x, next_x, alpha_, l_, reg_i_, loss_.get());
        zero_slots(next_x - start_x);
        AlsParams pu = base_params(I_, U_, ciu_, start_x, next_x, reg_i_);   // user-item part, weighted by l
        pu.out_scale = l_; pu.ff_scale = l_; pu.accumulate = 1;
        gram_pass(pu, 21, ciu_, start_x, next_x);
        AlsParams pc = base_params(I_, C_, cic_, start_x, next_x, reg_i_);   // item-context part
        pc.ctx = 1; pc.bias_self = Ib_.dev.get(); pc.bias_other = Cb_.dev.get(); pc.accumulate = 1; pc.ff_scale = l_;
        gram_pass(pc, 22, cic_, start_x, next_x);
        solve_rows(pc, start_x, next_x, ip_u, ip_c);
        hipLaunchKernelGGL(cfr_bias_kernel, dim3(wave_blocks(next_x - start_x)), dim3(256), 0, stream, I_.dev.get(), C_.dev.get(), vdim_, cic_.indptr.get(),
                           cic_.shift, start_x, next_x, cic_.keys.get(), cic_.vals.get(), Cb_.dev.get(), Ib_.dev.get(), ciu_.indptr.get());
        BFH_HIP(hipGetLastError());
        t_main_.end(slot, stream);
        write_back(I_, start_x, next_x);
        write_back(Ib_, start_x, next_x);
        const double loss = read_loss();
        stats.kernel_ms += t_main_.drain();
        stats.samples += ciu_.nnz + cic_.nnz;
        return loss;
    }
    // cfr.cc:257-313
    double partial_update_context(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals) {
        BFH_REQUIRE(I_.host && C_.host && Ib_.host && Cb_.host, "partial_update_context before set_embedding(item, context, biases)");
        BFH_REQUIRE(0 <= start_x && start_x <= next_x && next_x <= C_.rows, "partial_update_context: bad row range");
        if (next_x == start_x) return 0.0;
        upload(cc_, C_.rows, start_x, next_x, indptr, keys, vals);
        const int slot = t_main_.begin(stream);
        BFH_HIP(hipMemsetAsync(loss_.get(), 0, 2 * sizeof(double), stream));
        if (cfr_loss_)   // reg_c * |C_x|^2 of the rows BEFORE the update (cfr.cc:297-298)
            hipLaunchKernelGGL(cfr_sqnorm_kernel, dim3(wave_blocks(next_x - start_x)), dim3(256), 0, stream, C_.dev.get(), vdim_, cc_.indptr.get(), nullptr,
                               start_x, next_x, static_cast<double>(reg_c_), loss_.get());
        zero_slots(next_x - start_x);
        AlsParams p = base_params(C_, I_, cc_, start_x, next_x, reg_c_);
        p.ctx = 1; p.bias_self = Cb_.dev.get(); p.bias_other = Ib_.dev.get(); p.accumulate = 1; p.ff_scale = 0.0f;
        gram_pass(p, 23, cc_, start_x, next_x);
        solve_rows(p, start_x, next_x, indptr, nullptr);
        hipLaunchKernelGGL(cfr_bias_kernel, dim3(wave_blocks(next_x - start_x)), dim3(256), 0, stream, C_.dev.get(), I_.dev.get(), vdim_, cc_.indptr.get(),
                           cc_.shift, start_x, next_x, cc_.keys.get(), cc_.vals.get(), Ib_.dev.get(), Cb_.dev.get(), nullptr);
        BFH_HIP(hipGetLastError());
        t_main_.end(slot, stream);
        write_back(C_, start_x, next_x);
        write_back(Cb_, start_x, next_x);
        const double loss = read_loss();
        stats.kernel_ms += t_main_.drain();
        stats.samples += cc_.nnz;
        return loss;
    }

    float l_ = 1.f, reg_c_ = 0.f;
    bool cfr_loss_ = false;
    Emb U_, I_, C_, Ib_, Cb_;
    Csr cu_, ciu_, cic_, cc_;
    DevBuf<AlsHeavy> solve_list_;
};

}  // namespace bfh
