// TEST INFRASTRUCTURE -- not product code.  A C-callable door to the REFERENCE's own SGD classes, compiled from the sources where
// they lie (/root/reference/lib/algo.cc, lib/algo_impl/{bpr/bpr.cc | warp/warp.cc}, lib/misc/log.cc) against the stand-in headers
// of oracle/stand_in_3rd (see its README: this is not the reference binary; Eigen's part of the arithmetic is the stand-in's).
// Built twice (-DREF_BPR / -DREF_WARP: the two class headers define the same global constant) into
// oracle/_ref/libbuffalo_{bpr,warp}_on_stand_ins.so by `make -C oracle _ref_sgd`.
#if defined(REF_BPR)
#include "buffalo/algo_impl/bpr/bpr.hpp"
typedef bpr::CBPRMF Cls;
#define FN(name) refsgd_bpr_##name
#elif defined(REF_WARP)
#include "buffalo/algo_impl/warp/warp.hpp"
typedef warp::CWARP Cls;
#define FN(name) refsgd_warp_##name
#else
#error "compile with -DREF_BPR or -DREF_WARP"
#endif

extern "C" {

void* FN(create)() { return new Cls(); }
void FN(destroy)(void* h) { delete static_cast<Cls*>(h); }
int FN(init)(void* h, const char* opt_path) { return static_cast<Cls*>(h)->init(opt_path) ? 1 : 0; }
void FN(initialize_model)(void* h, float* P, int P_rows, float* Q, int Q_rows, float* Qb, long long num_total_samples) {
    static_cast<Cls*>(h)->initialize_model(P, P_rows, Q, Q_rows, Qb, num_total_samples);
}
void FN(set_cumulative_table)(void* h, int64_t* table, int size) { static_cast<Cls*>(h)->set_cumulative_table(table, size); }
void FN(launch_workers)(void* h) { static_cast<Cls*>(h)->launch_workers(); }
void FN(add_jobs)(void* h, int start_x, int next_x, int64_t* indptr, int32_t* positives) {
    static_cast<Cls*>(h)->add_jobs(start_x, next_x, indptr, positives);
}
void FN(wait_until_done)(void* h) { static_cast<Cls*>(h)->wait_until_done(); }
void FN(update_parameters)(void* h) { static_cast<Cls*>(h)->update_parameters(); }
double FN(join)(void* h) { return static_cast<Cls*>(h)->join(); }
double FN(compute_loss)(void* h, int n, int32_t* users, int32_t* positives, int32_t* negatives) {
    return static_cast<Cls*>(h)->compute_loss(n, users, positives, negatives);
}
int FN(queue_size)(void* h) { return static_cast<int>(static_cast<Cls*>(h)->job_queue_.get_size()); }

}  // extern "C"
