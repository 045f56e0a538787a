// TEST INFRASTRUCTURE -- not product code.  A C-callable door to the REFERENCE's own top-k code, /root/reference/buffalo/parallel/_core.hpp
// compiled from where it lies against the stand-in headers of oracle/stand_in_3rd (the score of a candidate is one row-row product of the
// stand-in; the running list, its admission rule, the pool, the self-exclusion and quickselect's nth_element + sort are the reference's).
// Built by `make -C oracle _ref_sgd` into oracle/_ref/libbuffalo_core_on_stand_ins.so.
#include "buffalo/parallel/_core.hpp"

extern "C" {

void refcore_quickselect(float* scores, int rows, int cols, int32_t* result, int k, int sorted, int num_threads) {
    parallel::quickselect(scores, rows, cols, result, k, sorted != 0, num_threads);
}
void refcore_dot_topn(int32_t* indexes, int num_queries, float* P, int p_rows, int p_cols, float* Q, int q_rows, int q_cols, float* Qb, int qb_rows,
                      int32_t* out_keys, float* out_scores, int32_t* pool, int pool_size, int k, int num_threads) {
    parallel::dot_topn(indexes, num_queries, P, p_rows, p_cols, Q, q_rows, q_cols, Qb, qb_rows, out_keys, out_scores, pool, pool_size, k, num_threads);
}

}  // extern "C"
