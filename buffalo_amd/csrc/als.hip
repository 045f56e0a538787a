// ALS on gfx950 -- C ABI + handle.  Kernels live in als_kernels.hpp.
//
// Reference semantics: CALS (/root/reference/lib/algo_impl/als/als.cc:30-358) + Algorithm::_leastsquare
// (/root/reference/lib/algo.cc:39-82) behind CuALS's object surface
// (/root/reference/include/buffalo/cuda/als/als.hpp:20-35).
#include "als_kernels.hpp"
#include "cfr_impl.hpp"
#include "eals_impl.hpp"

using bfh::AlsHandle;
using bfh::CfrHandle;
using bfh::EalsHandle;
using bfh::guarded;

extern "C" {

void* bfh_als_create(void) {
    try {
        AlsHandle* h = new AlsHandle();
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
void bfh_als_destroy(void* h) { delete static_cast<AlsHandle*>(h); }
int bfh_als_set_device(void* h, int device) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->device = device; BFH_HIP(hipSetDevice(device)); return BFH_OK; });
}
int bfh_als_init(void* h, const char* opt_json_path) {
    int ok = 0;
    int rc = guarded(h, [&] { ok = static_cast<AlsHandle*>(h)->init(opt_json_path) ? 1 : 0; return BFH_OK; });
    return rc == BFH_OK ? ok : rc;
}
int bfh_als_get_vdim(void* h) { return h ? static_cast<AlsHandle*>(h)->vdim_ : BFH_ERR_INVALID; }
int bfh_als_initialize_model(void* h, float* P, int P_rows, float* Q, int Q_rows) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->initialize_model(P, P_rows, Q, Q_rows); return BFH_OK; });
}
int bfh_als_set_placeholder(void* h, const int64_t* lindptr, const int64_t* rindptr, size_t batch_size) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->set_placeholder(lindptr, rindptr, batch_size); return BFH_OK; });
}
int bfh_als_precompute(void* h, int axis) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->precompute(axis); return BFH_OK; });
}
int bfh_als_partial_update(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals,
                           int axis, double* loss_nume, double* loss_deno) {
    return guarded(h, [&] {
        double a = 0, b = 0;
        static_cast<AlsHandle*>(h)->partial_update(start_x, next_x, indptr, keys, vals, axis, &a, &b);
        if (loss_nume) *loss_nume = a;
        if (loss_deno) *loss_deno = b;
        return BFH_OK;
    });
}
int bfh_als_set_resident_csr(void* h, int axis, const int64_t* indptr, const int32_t* keys, const float* vals, int64_t nnz) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->set_resident_csr(axis, indptr, keys, vals, nnz); return BFH_OK; });
}
int bfh_als_synchronize(void* h, int device_to_host) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->synchronize(device_to_host != 0); return BFH_OK; });
}
int bfh_als_set_mode(void* h, const char* name, int64_t value) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->set_mode(name ? name : "", value); return BFH_OK; });
}
int bfh_als_device_buffer(void* h, const char* name, void** dptr, size_t* bytes) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->device_buffer(name ? name : "", dptr, bytes); return BFH_OK; });
}
void* bfh_als_stream(void* h) { return h ? static_cast<void*>(static_cast<AlsHandle*>(h)->stream) : nullptr; }
int bfh_als_set_comm(void* h, void* comm) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->set_comm(static_cast<bfh::Comm*>(comm)); return BFH_OK; });
}
int bfh_als_publish_rows(void* h, int axis, const int* bounds, int n_bounds) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->publish_rows(axis, bounds, n_bounds); return BFH_OK; });
}
int bfh_als_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->flush_timers(); *out = static_cast<AlsHandle*>(h)->stats; return BFH_OK; });
}
int bfh_als_reset_stats(void* h) {
    return guarded(h, [&] { static_cast<AlsHandle*>(h)->stats = bfh_stats{}; return BFH_OK; });
}

// ------------------------------------------------------------------------------------------------
// CFR -- CCFR (/root/reference/lib/algo_impl/cfr/cfr.cc) behind CyCFR's surface (buffalo/algo/_cfr.pyx:25-71)
// ------------------------------------------------------------------------------------------------
void* bfh_cfr_create(void) {
    try {
        CfrHandle* h = new CfrHandle();
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
void bfh_cfr_destroy(void* h) { delete static_cast<CfrHandle*>(h); }
int bfh_cfr_set_device(void* h, int device) {
    return guarded(h, [&] { static_cast<CfrHandle*>(h)->device = device; BFH_HIP(hipSetDevice(device)); return BFH_OK; });
}
int bfh_cfr_init(void* h, const char* opt_json_path) {
    int ok = 0;
    int rc = guarded(h, [&] { ok = static_cast<CfrHandle*>(h)->init_cfr(opt_json_path) ? 1 : 0; return BFH_OK; });
    return rc == BFH_OK ? ok : rc;
}
int bfh_cfr_set_embedding(void* h, float* data, int size, const char* obj_type) {
    return guarded(h, [&] { static_cast<CfrHandle*>(h)->set_embedding(data, size, obj_type ? obj_type : ""); return BFH_OK; });
}
int bfh_cfr_precompute(void* h, const char* obj_type) {
    return guarded(h, [&] { static_cast<CfrHandle*>(h)->precompute_cfr(obj_type ? obj_type : ""); return BFH_OK; });
}
int bfh_cfr_partial_update_user(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals, double* loss) {
    return guarded(h, [&] {
        const double v = static_cast<CfrHandle*>(h)->partial_update_user(start_x, next_x, indptr, keys, vals);
        if (loss) *loss = v;
        return BFH_OK;
    });
}
int bfh_cfr_partial_update_item(void* h, int start_x, int next_x, const int64_t* indptr_u, const int32_t* keys_u, const float* vals_u,
                                const int64_t* indptr_c, const int32_t* keys_c, const float* vals_c, double* loss) {
    return guarded(h, [&] {
        const double v = static_cast<CfrHandle*>(h)->partial_update_item(start_x, next_x, indptr_u, keys_u, vals_u, indptr_c, keys_c, vals_c);
        if (loss) *loss = v;
        return BFH_OK;
    });
}
int bfh_cfr_partial_update_context(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals, double* loss) {
    return guarded(h, [&] {
        const double v = static_cast<CfrHandle*>(h)->partial_update_context(start_x, next_x, indptr, keys, vals);
        if (loss) *loss = v;
        return BFH_OK;
    });
}
int bfh_cfr_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { static_cast<CfrHandle*>(h)->flush_timers(); *out = static_cast<CfrHandle*>(h)->stats; return BFH_OK; });
}
int bfh_cfr_reset_stats(void* h) {
    return guarded(h, [&] { static_cast<CfrHandle*>(h)->stats = bfh_stats{}; return BFH_OK; });
}

// ------------------------------------------------------------------------------------------------
// eALS -- CEALS (/root/reference/lib/algo_impl/eals/eals.cc) behind CyEALS's surface (buffalo/algo/_eals.pyx:23-67)
// ------------------------------------------------------------------------------------------------
void* bfh_eals_create(void) {
    try {
        EalsHandle* h = new EalsHandle();
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
void bfh_eals_destroy(void* h) { delete static_cast<EalsHandle*>(h); }
int bfh_eals_set_device(void* h, int device) {
    return guarded(h, [&] { static_cast<EalsHandle*>(h)->device = device; BFH_HIP(hipSetDevice(device)); return BFH_OK; });
}
int bfh_eals_init(void* h, const char* opt_json_path) {
    int ok = 0;
    int rc = guarded(h, [&] { ok = static_cast<EalsHandle*>(h)->init_eals(opt_json_path) ? 1 : 0; return BFH_OK; });
    return rc == BFH_OK ? ok : rc;
}
int bfh_eals_initialize_model(void* h, float* P, float* Q, float* C, int P_rows, int Q_rows) {
    return guarded(h, [&] { static_cast<EalsHandle*>(h)->initialize_model_eals(P, Q, C, P_rows, Q_rows); return BFH_OK; });
}
int bfh_eals_precompute_cache(void* h, int nnz, const int64_t* indptr, const int32_t* keys, int axis) {
    return guarded(h, [&] { static_cast<EalsHandle*>(h)->precompute_cache(nnz, indptr, keys, axis); return BFH_OK; });
}
int bfh_eals_update(void* h, const int64_t* indptr, const int32_t* keys, const float* vals, int axis) {
    int ok = 0;
    int rc = guarded(h, [&] { ok = static_cast<EalsHandle*>(h)->update(indptr, keys, vals, axis) ? 1 : 0; return BFH_OK; });
    return rc == BFH_OK ? ok : rc;
}
int bfh_eals_estimate_loss(void* h, int nnz, const int64_t* indptr, const int32_t* keys, const float* vals, int axis, float* rmse, float* loss) {
    return guarded(h, [&] {
        float a = 0.f, b = 0.f;
        static_cast<EalsHandle*>(h)->estimate_loss(nnz, indptr, keys, vals, axis, &a, &b);
        if (rmse) *rmse = a;
        if (loss) *loss = b;
        return BFH_OK;
    });
}
int bfh_eals_get_stats(void* h, bfh_stats* out) {
    return guarded(h, [&] { static_cast<EalsHandle*>(h)->flush_timers(); *out = static_cast<EalsHandle*>(h)->stats; return BFH_OK; });
}
int bfh_eals_reset_stats(void* h) {
    return guarded(h, [&] { static_cast<EalsHandle*>(h)->stats = bfh_stats{}; return BFH_OK; });
}

}  // extern "C"
