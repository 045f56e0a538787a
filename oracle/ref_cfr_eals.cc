// TEST INFRASTRUCTURE -- not product code.  C-callable doors to the REFERENCE's own CFR and eALS classes, compiled from the sources where
// they lie (/root/reference/lib/algo.cc, lib/algo_impl/cfr/cfr.cc, lib/algo_impl/eals/eals.cc, lib/misc/log.cc) against the stand-in
// headers of oracle/stand_in_3rd (see its README: not the reference binary).  Built by `make -C oracle _ref_sgd` into
// oracle/_ref/libbuffalo_cfr_eals_on_stand_ins.so.
#include "buffalo/algo_impl/cfr/cfr.hpp"
#include "buffalo/algo_impl/eals/eals.hpp"

// eals.hpp forms its Gramians through the Fortran BLAS routine ssyrk (buffalo/misc/blas.hpp:6-13) and no CPU BLAS exists in this image:
// the reference semantics of the routine, written out (C := alpha * op(A) * op(A)^T + beta * C on one triangle, column-major).
template <class T> static void syrk_stand_in(char uplo, char trans, int n, int k, T alpha, const T* A, int lda, T beta, T* C, int ldc) {
    const bool upper = uplo == 'u' || uplo == 'U', plain = trans == 'n' || trans == 'N';
    for (int j = 0; j < n; ++j)
        for (int i = upper ? 0 : j; i <= (upper ? j : n - 1); ++i) {
            T s = 0;
            for (int l = 0; l < k; ++l) s += plain ? A[i + l * lda] * A[j + l * lda] : A[l + i * lda] * A[l + j * lda];
            C[i + j * ldc] = alpha * s + (beta == T(0) ? T(0) : beta * C[i + j * ldc]);
        }
}
extern "C" {
void ssyrk_(const char* uplo, const char* trans, const int* n, const int* k, const float* alpha, const float* A, const int* lda, const float* beta,
            float* C, const int* ldc) {
    syrk_stand_in<float>(*uplo, *trans, *n, *k, *alpha, A, *lda, *beta, C, *ldc);
}
void dsyrk_(const char* uplo, const char* trans, const int* n, const int* k, const double* alpha, const double* A, const int* lda, const double* beta,
            double* C, const int* ldc) {
    syrk_stand_in<double>(*uplo, *trans, *n, *k, *alpha, A, *lda, *beta, C, *ldc);
}
}

extern "C" {

void* refcfr_create() { return new cfr::CCFR(); }
void refcfr_destroy(void* h) { delete static_cast<cfr::CCFR*>(h); }
int refcfr_init(void* h, const char* opt_path) { return static_cast<cfr::CCFR*>(h)->init(opt_path) ? 1 : 0; }
void refcfr_set_embedding(void* h, float* data, int size, const char* obj_type) { static_cast<cfr::CCFR*>(h)->set_embedding(data, size, obj_type); }
void refcfr_precompute(void* h, const char* obj_type) { static_cast<cfr::CCFR*>(h)->precompute(obj_type); }
double refcfr_partial_update_user(void* h, int start_x, int next_x, int64_t* indptrs, int32_t* keys, float* vals) {
    return static_cast<cfr::CCFR*>(h)->partial_update_user(start_x, next_x, indptrs, keys, vals);
}
double refcfr_partial_update_item(void* h, int start_x, int next_x, int64_t* indptrs_u, int32_t* keys_u, float* vals_u, int64_t* indptrs_c,
                                  int32_t* keys_c, float* vals_c) {
    return static_cast<cfr::CCFR*>(h)->partial_update_item(start_x, next_x, indptrs_u, keys_u, vals_u, indptrs_c, keys_c, vals_c);
}
double refcfr_partial_update_context(void* h, int start_x, int next_x, int64_t* indptrs, int32_t* keys, float* vals) {
    return static_cast<cfr::CCFR*>(h)->partial_update_context(start_x, next_x, indptrs, keys, vals);
}

void* refeals_create() { return new eals::CEALS(); }
void refeals_destroy(void* h) { delete static_cast<eals::CEALS*>(h); }
int refeals_init(void* h, const char* opt_path) { return static_cast<eals::CEALS*>(h)->init(opt_path) ? 1 : 0; }
void refeals_initialize_model(void* h, float* P, float* Q, float* Cw, int P_rows, int Q_rows) {
    static_cast<eals::CEALS*>(h)->initialize_model(P, Q, Cw, P_rows, Q_rows);
}
void refeals_precompute_cache(void* h, int nnz, const int64_t* indptr, const int32_t* keys, int axis) {
    static_cast<eals::CEALS*>(h)->precompute_cache(nnz, indptr, keys, axis);
}
int refeals_update(void* h, const int64_t* indptr, const int32_t* keys, const float* vals, int axis) {
    return static_cast<eals::CEALS*>(h)->update(indptr, keys, vals, axis) ? 1 : 0;
}
void refeals_estimate_loss(void* h, int nnz, const int64_t* indptr, const int32_t* keys, const float* vals, int axis, float* out2) {
    auto r = static_cast<eals::CEALS*>(h)->estimate_loss(nnz, indptr, keys, vals, axis);
    out2[0] = r.first;
    out2[1] = r.second;
}

}  // extern "C"
