// TEST INFRASTRUCTURE -- not product code.  A C-callable door to the REFERENCE's own ALS class, compiled from the sources where they
// lie (/root/reference/lib/algo.cc, lib/algo_impl/als/als.cc, lib/misc/log.cc) against the stand-in headers of oracle/stand_in_3rd
// (see its README: not the reference binary; the dense products, Cholesky solves and their summation orders are the stand-in's).
// Built by `make -C oracle _ref_sgd` into oracle/_ref/libbuffalo_als_on_stand_ins.so.
#include "buffalo/algo_impl/als/als.hpp"

extern "C" {

void* refals_create() { return new als::CALS(); }
void refals_destroy(void* h) { delete static_cast<als::CALS*>(h); }
int refals_init(void* h, const char* opt_path) { return static_cast<als::CALS*>(h)->init(opt_path) ? 1 : 0; }
void refals_initialize_model(void* h, float* P, int P_rows, float* Q, int Q_rows) { static_cast<als::CALS*>(h)->initialize_model(P, P_rows, Q, Q_rows); }
void refals_precompute(void* h, int axis) { static_cast<als::CALS*>(h)->precompute(axis); }
void refals_partial_update(void* h, int start_x, int next_x, int64_t* indptr, int32_t* keys, float* vals, int axis, double* out2) {
    auto r = static_cast<als::CALS*>(h)->partial_update(start_x, next_x, indptr, keys, vals, axis);
    out2[0] = r.first;
    out2[1] = r.second;
}
void refals_get_ff(void* h, float* out, int d) {
    auto& FF = static_cast<als::CALS*>(h)->FF_;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) out[i * d + j] = FF(i, j);
}

}  // extern "C"
