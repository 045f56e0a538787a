// =============================================================================
// TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT PATH.
//
// CPU restatement ("oracle") of kakao/buffalo's ALS / BPRMF / WARP training hot
// path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load this library, and only as the checker / the timed CPU baseline.
//
// WHAT PINS THIS FILE (DESIGN.md section 7 has the table):
//  * The reference BINARY cannot be built here: its training C++ needs Eigen
//    (json11, spdlog), empty submodules absent from this image, and it ships no
//    golden vectors or known-answer tests for training results.
//    WHAT EIGEN DOES INSIDE AN EXPRESSION (evaluation order, vectorised
//    reductions, GEMM / Cholesky summation) IS THEREFORE UNPINNED BY THE
//    REFERENCE; builder-made pins only (numpy transliterations, analytic
//    micro-cases, float64 recurrences: tests/test_oracle_pins.py).
//  * Everything else in the training sources IS held to the reference: its own
//    lib/algo.cc, bpr.cc, warp.cc, als.cc, cfr.cc, eals.cc and parallel/_core.hpp
//    compile unmodified against stand-ins for Eigen / json11 / spdlog written
//    here (oracle/stand_in_3rd) and run beside this file on the same inputs
//    (tests/test_oracle_vs_reference_sources.py): BPRMF and WARP bit-identical
//    when both are built without FP contraction (incl. the learning-rate
//    thread), top-k identical, ALS / CFR / eALS within their solvers'
//    conditioning.  The stand-ins and this file share one reading of Eigen.
//  * Its own tests: tests/parallel/test_base.py unmodified over dot_topn; 62 of
//    its algorithm tests (tests/algo/test_{als,bpr,warp,eals,cfr}.py: NDCG / MAP
//    thresholds, top-k by item name, serialization) with these classes bound
//    where its fronts import CyALS / CyBPRMF / CyWARP / CyEALS / CyCFR, on
//    ML-100K-shaped synthetic files (tests/test_reference_suite_on_oracle.py).
//  * The data-ingestion restatements (orc_coo_to_csr, orc_build_sppmi): the
//    reference's own compiled buffalo/data/fileio.hpp (oracle/_ref), bit for bit
//    (tests/test_oracle_ref_fileio.py).
//
// All citations are relative to /root/reference/.
// Quirk numbers (Q-n) refer to SURVEY.md section 7.4.
//
// Eigen is replaced by plain loops; float/double promotion follows Eigen's
// rule for `double_scalar * float_matrix` (the scalar is cast to float first).
// libstdc++'s std::mt19937 / uniform_int_distribution / unordered_set are used
// directly, so sample order (Q-3, Q-4) is reproduced exactly on this toolchain.
// =============================================================================
#include <omp.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <thread>
#include <unordered_set>
#include <utility>
#include <vector>

namespace {

static const float FEPS = 1e-10f;  // lib/algo.cc:11, lib/algo_impl/warp/warp.cc:15
static const int MAX_EXP = 6;      // lib/algo_impl/bpr/bpr.cc:16
static const int EXP_TABLE_SIZE = 1000;  // include/buffalo/algo_impl/bpr/bpr.hpp:17

// ----------------------------------------------------------------------------
// Options.  The reference parses a JSON file with json11 (lib/algo.cc:19-37);
// here the Python wrapper parses the same file and pushes typed key/values.
// Accessors mimic json11: a missing key or a key of another type reads as
// 0 / "" / false (Q-7, Q-22).
// ----------------------------------------------------------------------------
struct Opt {
    std::map<std::string, double> num;
    std::map<std::string, std::string> str;
    std::map<std::string, bool> boo;
    int i(const char* k) const {
        auto it = num.find(k);
        return it == num.end() ? 0 : static_cast<int>(it->second);
    }
    double d(const char* k) const {
        auto it = num.find(k);
        return it == num.end() ? 0.0 : it->second;
    }
    bool b(const char* k) const {
        auto it = boo.find(k);
        return it == boo.end() ? false : it->second;
    }
    std::string s(const char* k) const {
        auto it = str.find(k);
        return it == str.end() ? std::string() : it->second;
    }
};

// ----------------------------------------------------------------------------
// Counter-based sampler used ONLY when sampler==1 ("counter" mode): replaces
// the reference's mt19937 stream with Philox4x32-10 keyed on
// (seed, stream) and counted on (nnz position, attempt, epoch) so that the HIP
// kernels -- which cannot consume a sequential mt19937 stream -- can be checked
// draw-for-draw.  This is an independent implementation of the published
// Philox4x32-10 (Salmon et al., SC'11); the product has its own copy in
// buffalo_amd/csrc.  Known-answer vectors: tests/test_oracle_pins.py.
// ----------------------------------------------------------------------------
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Draw `attempt` for nnz position `pos_idx` (global position in the rowwise key
// array), negative slot `slot`, epoch `epoch`.  Stream 0 = BPR, 1 = WARP.
static inline void counter_draw(uint32_t seed, uint32_t stream, uint64_t pos_idx, uint32_t slot,
                                uint32_t epoch, uint32_t attempt, uint32_t out[4]) {
    philox4x32_10((uint32_t)pos_idx, (uint32_t)(pos_idx >> 32), attempt, (epoch << 8) | (slot & 0xffu),
                  seed, 0x5bf03635u ^ stream, out);
}
// uniform in [0, n) from 32 random bits (multiply-shift).
static inline int64_t mulhi_u32(uint32_t r, uint32_t n) { return (int64_t)(((uint64_t)r * n) >> 32); }
// uniform in [0, n) for 64-bit n from 64 random bits.
static inline int64_t mulhi_u64(uint64_t r, uint64_t n) {
    return (int64_t)(((unsigned __int128)r * n) >> 64);
}

// ----------------------------------------------------------------------------
// include/buffalo/concurrent_queue.hpp:9-79 -- mutex + condvar MPMC queue.
// ----------------------------------------------------------------------------
template <typename T>
class Queue {
 public:
    T pop() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        T v = std::move(q_.front());
        q_.pop_front();
        return v;
    }
    void push(const T& v) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(v);
        }
        cv_.notify_one();
    }
    size_t get_size() {
        std::lock_guard<std::mutex> lk(m_);
        return q_.size();
    }

 private:
    std::deque<T> q_;
    std::mutex m_;
    std::condition_variable cv_;
};

// include/buffalo/algo.hpp:28-48
struct job_t {
    int size = 0;
    double alpha = 0.0;
    std::vector<std::vector<int>> samples;
    std::vector<int64_t> base;  // oracle-only: global nnz position of samples[i][1] (counter sampler)
    void add(const std::vector<int>& s, int64_t b) {
        samples.push_back(s);
        base.push_back(b);
        size += (int)s.size();
    }
};
// include/buffalo/algo.hpp:51-69
struct progress_t {
    int num_sents, num_processed_samples, num_total_samples;
    double loss;
};

// Eigen evaluates row dot products with packet (8 x float on AVX2) partial sums followed by a
// horizontal add; `omp simd reduction` makes GCC emit the same shape instead of a serial chain.
static inline float dotf(const float* a, const float* b, int n) {
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

// ============================================================================
// SGDAlgorithm  (include/buffalo/algo.hpp:93-148, lib/algo.cc:133-492)
// ============================================================================
class SGD {
 public:
    virtual ~SGD() {}
    Opt opt_;
    float *P_ = nullptr, *Q_ = nullptr, *Qb_ = nullptr;
    int P_rows_ = 0, Q_rows_ = 0, D_ = 0;
    int iters_ = 0;
    double lr_ = 0.0;  // written by the progress thread, read by add_jobs (racy by design, Q-8)
    std::string optimizer_;
    double total_processed_ = 0.0;
    std::vector<int> Pcnt_, Qcnt_;
    std::vector<float> gradP_, gradQ_, gradQb_, momP_, momQ_, momQb_, velP_, velQ_, velQb_;
    std::vector<std::thread> workers_;
    std::thread* progress_manager_ = nullptr;
    Queue<job_t> job_queue_;
    Queue<progress_t> progress_queue_;
    int64_t* cum_table_ = nullptr;
    int cum_table_size_ = 0;

    // ---- oracle-only switches (all default to reference behaviour) ----------
    int sampler_ = 0;    // 0: std::mt19937 (reference) | 1: Philox counter draws
    int pos_order_ = 0;  // 0: std::unordered_set order (reference, Q-3) | 1: CSR order
    int inline_ = 0;     // 0: worker threads + queue (reference) | 1: process jobs inside add_jobs,
                         //    lr stamped once per add_jobs call from completed progress (deterministic Q-8)
    uint32_t epoch_ = 0;  // counts update_parameters calls (counter sampler)
    long long inl_total_processed_ = 0;
    int64_t nnz_offset_ = 0;  // multi-GPU tests: global position of this shard's first nnz
    int num_shards_ = 1;      //                  shards that advance the lr schedule together
    std::vector<std::mt19937> inl_rng_;
    // statistics exported to the tests / bench
    std::atomic<long long> stat_samples_{0}, stat_scored_negs_{0}, stat_updates_{0};
    std::vector<int32_t>* trace_ = nullptr;  // (u,pos,neg) triples in processing order when tracing
    std::mutex trace_m_;

    virtual void worker(int worker_id) = 0;
    virtual void process_job(job_t& job, int worker_id, std::mt19937& RNG) = 0;

    // lib/algo.cc:197-208 (+ bpr.cc:38-46 / warp.cc:68-85)
    virtual bool init() {
        int num_workers = opt_.i("num_workers");
        omp_set_num_threads(std::max(1, num_workers));
        optimizer_ = opt_.s("optimizer");
        return true;
    }

    // lib/algo.cc:148-176
    virtual void initialize_model(float* P, int P_rows, float* Q, int Q_rows, float* Qb,
                                  int64_t num_total_samples) {
        D_ = opt_.i("d");
        P_ = P; P_rows_ = P_rows;
        Q_ = Q; Q_rows_ = Q_rows;
        Qb_ = Qb;
        if (optimizer_ != "sgd") initialize_adam_optimizer();
        else initialize_sgd_optimizer();
        iters_ = 0;
        int num_iters = opt_.i("num_iters");
        total_processed_ = (double)num_total_samples * num_iters;
        inl_total_processed_ = 0;
        epoch_ = 0;
    }

    // lib/algo.cc:221-255.  gradQb_ is allocated unconditionally here because
    // update_parameters touches it even when use_bias is false (Q-9).
    void initialize_adam_optimizer() {
        size_t np = (size_t)P_rows_ * D_, nq = (size_t)Q_rows_ * D_;
        gradP_.assign(np, 0.f); gradQ_.assign(nq, 0.f);
        momP_.assign(np, 0.f);  momQ_.assign(nq, 0.f);
        velP_.assign(np, 0.f);  velQ_.assign(nq, 0.f);
        gradQb_.assign(Q_rows_, 0.f); momQb_.assign(Q_rows_, 0.f); velQb_.assign(Q_rows_, 0.f);
        if (opt_.b("per_coordinate_normalize")) {
            Pcnt_.assign(P_rows_, 0);
            Qcnt_.assign(Q_rows_, 0);
        }
    }
    // lib/algo.cc:257-260
    void initialize_sgd_optimizer() { lr_ = opt_.d("lr"); }

    // lib/algo.cc:211-219
    void launch_workers() {
        int num_workers = opt_.i("num_workers");
        if (inline_) {
            inl_rng_.clear();
            for (int i = 0; i < std::max(1, num_workers); ++i)
                inl_rng_.emplace_back(opt_.i("random_seed") + i);
            lr_ = opt_.d("lr");
            return;
        }
        workers_.clear();
        for (int i = 0; i < num_workers; ++i) workers_.emplace_back(&SGD::worker, this, i);
        progress_manager_ = new std::thread(&SGD::progress_manager, this);
    }

    // lib/algo.cc:261-306 (logging dropped)
    void progress_manager() {
        double alpha = opt_.d("lr");
        double min_alpha = opt_.d("min_lr");
        lr_ = alpha;
        long long total_processed_samples = 0;
        while (true) {
            progress_t p = progress_queue_.pop();
            if (p.num_sents == -1 && p.num_processed_samples == -1) break;
            total_processed_samples += p.num_total_samples;
            double progress = total_processed_samples / total_processed_;
            double new_alpha = alpha - (alpha - min_alpha) * progress;
            new_alpha = std::max(new_alpha, min_alpha);
            lr_ = new_alpha;
        }
    }

    // lib/algo.cc:308-362.  `positives` is the chunk's key array shifted by
    // indptr[start_x-1]; indptr holds END offsets (no leading zero).
    void add_jobs(int start_x, int next_x, const int64_t* indptr, const int32_t* positives) {
        if ((next_x - start_x) == 0) return;
        int batch_size = opt_.i("batch_size");  // key absent from BPRMF/WARP options => 0 (Q-7)
        if (batch_size < 0) batch_size = 10000;

        if (inline_) {
            // deterministic idealisation of Q-8: every job of this call carries the lr that
            // results from all previously *completed* work.
            double alpha = opt_.d("lr"), min_alpha = opt_.d("min_lr");
            double progress = inl_total_processed_ / total_processed_;
            lr_ = std::max(alpha - (alpha - min_alpha) * progress, min_alpha);
        }

        job_t job;
        int job_size = 0;
        int end_loop = next_x - start_x;
        const int64_t shifted = start_x == 0 ? 0 : indptr[start_x - 1];
        std::vector<int> S;
        auto flush = [&](job_t& j) {
            j.alpha = lr_;
            if (inline_) {
                process_job(j, 0, inl_rng_[0]);
                inl_total_processed_ += (long long)j.size * num_shards_;
            } else {
                job_queue_.push(j);
            }
        };
        for (int i = 0; i < end_loop; ++i) {
            int x = start_x + i;
            const int u = x;
            int64_t beg = x == 0 ? 0 : indptr[x - 1];
            int64_t end = indptr[x];
            int64_t data_size = end - beg;
            if (data_size == 0) continue;
            S.push_back(u);
            for (int64_t it = beg; it < end; ++it) S.push_back(positives[it - shifted]);
            if (data_size + job_size <= batch_size) {
                job.add(S, beg + nnz_offset_);
                job_size += (int)S.size();
            } else {
                flush(job);  // with batch_size==0 the very first flush is an empty job (Q-7)
                job = job_t();
                job.add(S, beg + nnz_offset_);
                job_size = (int)S.size();
            }
            S.clear();
        }
        if (job.size) flush(job);
    }

    // lib/algo.cc:365-375 (Q-5: bias-correction exponent iters_+1, FEPS outside the sqrt)
    void update_adam(float* grad, float* mom, float* vel, int n, double beta1, double beta2) {
        const float b1 = (float)beta1, omb1 = (float)(1.0 - beta1);
        const float b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
        const float c1 = (float)(1.0 - std::pow(beta1, iters_ + 1));
        const float c2 = (float)(1.0 - std::pow(beta2, iters_ + 1));
        for (int k = 0; k < n; ++k) {
            mom[k] = b1 * mom[k] + omb1 * grad[k];
            vel[k] = b2 * vel[k] + omb2 * (grad[k] * grad[k]);
            float m_hat = mom[k] / c1;
            float v_hat = vel[k] / c2;
            grad[k] = m_hat / (std::sqrt(v_hat) + FEPS);
        }
    }
    // lib/algo.cc:377-380
    void update_adagrad(float* grad, float* vel, int n) {
        for (int k = 0; k < n; ++k) {
            vel[k] = vel[k] + grad[k] * grad[k];
            grad[k] = grad[k] / (std::sqrt(vel[k]) + FEPS);
        }
    }

    // lib/algo.cc:382-465.  NOTE Q-5 (beta2 read from "beta1"), Q-6 (grad buffers
    // are left holding the transformed step and are never re-zeroed), Q-9
    // (gradQb divided by the count even when use_bias is false).
    virtual void update_parameters() {
        int num_workers = std::max(1, opt_.i("num_workers"));
        omp_set_num_threads(num_workers);
        bool use_bias = opt_.b("use_bias");
        double reg_u = opt_.d("reg_u"), reg_i = opt_.d("reg_i"), reg_b = opt_.d("reg_b");
        bool pcn = opt_.b("per_coordinate_normalize");
        const int D = D_;
        if (optimizer_ == "adam" || optimizer_ == "adagrad") {
            const bool adam = optimizer_ == "adam";
            double lr = opt_.d("lr");
            double beta1 = opt_.d("beta1");
            double beta2 = opt_.d("beta1");  // sic (lib/algo.cc:396)
            const float lrf = (float)lr;
            const float ru2 = (float)(2 * reg_u), ri2 = (float)(2 * reg_i);
#pragma omp parallel for schedule(static)
            for (int u = 0; u < P_rows_; ++u) {
                float* g = &gradP_[(size_t)u * D];
                float* p = &P_[(size_t)u * D];
                if (pcn && Pcnt_[u]) {
                    const float c = (float)Pcnt_[u];
                    for (int k = 0; k < D; ++k) g[k] /= c;
                }
                for (int k = 0; k < D; ++k) g[k] -= p[k] * ru2;
                if (adam) update_adam(g, &momP_[(size_t)u * D], &velP_[(size_t)u * D], D, beta1, beta2);
                else update_adagrad(g, &velP_[(size_t)u * D], D);
                for (int k = 0; k < D; ++k) p[k] += lrf * g[k];
            }
#pragma omp parallel for schedule(static)
            for (int i = 0; i < Q_rows_; ++i) {
                float* g = &gradQ_[(size_t)i * D];
                float* q = &Q_[(size_t)i * D];
                if (pcn && Qcnt_[i]) {
                    const float c = (float)Qcnt_[i];
                    for (int k = 0; k < D; ++k) g[k] /= c;
                    gradQb_[i] /= c;
                }
                for (int k = 0; k < D; ++k) g[k] -= q[k] * ri2;
                if (adam) update_adam(g, &momQ_[(size_t)i * D], &velQ_[(size_t)i * D], D, beta1, beta2);
                else update_adagrad(g, &velQ_[(size_t)i * D], D);
                for (int k = 0; k < D; ++k) q[k] += lrf * g[k];
                if (use_bias) {
                    // lib/algo.cc:418-420 / :447-449 are SCALAR C++ statements on matrix elements, not Eigen expressions: a float
                    // element combined with the double options is computed in double and rounded once when it is stored
                    // (found by running the reference's own sources next to this file: oracle/ref_sgd.cc)
                    gradQb_[i] = (float)((double)gradQb_[i] - (double)Qb_[i] * (2 * reg_b));
                    if (adam) update_adam(&gradQb_[i], &momQb_[i], &velQb_[i], 1, beta1, beta2);
                    else update_adagrad(&gradQb_[i], &velQb_[i], 1);
                    Qb_[i] = (float)((double)Qb_[i] + lr * (double)gradQb_[i]);
                }
            }
            if (pcn) {
                Pcnt_.assign(P_rows_, 0);
                Qcnt_.assign(Q_rows_, 0);
            }
        }
        iters_ += 1;
        epoch_ += 1;
    }

    // lib/algo.cc:467-472 -- "queue empty" is not "work finished"; the oracle
    // additionally offers drain() for tests that need completed work.
    void wait_until_done() {
        if (inline_) return;
        while (job_queue_.get_size() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }

    // lib/algo.cc:474-492
    double join() {
        if (inline_) return 0.0;
        int num_workers = opt_.i("num_workers");
        for (int i = 0; i < num_workers; ++i) {
            job_t job;
            job.size = -1;
            job_queue_.push(job);
        }
        for (auto& t : workers_) t.join();
        progress_queue_.push(progress_t{-1, -1, -1, 0.0});
        progress_manager_->join();
        delete progress_manager_;
        progress_manager_ = nullptr;
        workers_.clear();
        return 0.0;
    }

    void trace_push(int u, int pos, int neg) {
        if (!trace_) return;
        std::lock_guard<std::mutex> lk(trace_m_);
        trace_->push_back(u); trace_->push_back(pos); trace_->push_back(neg);
    }
};

// ============================================================================
// CBPRMF  (lib/algo_impl/bpr/bpr.cc)
// ============================================================================
class BPR : public SGD {
 public:
    float exp_table_[EXP_TABLE_SIZE];

    // bpr.cc:47-55
    void initialize_model(float* P, int P_rows, float* Q, int Q_rows, float* Qb,
                          int64_t num_total_samples) override {
        SGD::initialize_model(P, P_rows, Q, Q_rows, Qb, num_total_samples);
        build_exp_table();
    }
    // bpr.cc:57-63 (Q-2)
    void build_exp_table() {
        for (int i = 0; i < EXP_TABLE_SIZE; ++i) {
            exp_table_[i] = (float)std::exp((i / (float)EXP_TABLE_SIZE * 2 - 1) * MAX_EXP);
            exp_table_[i] = 1.0 / (exp_table_[i] + 1);
        }
    }

    // bpr.cc:72-188 outer loop
    void worker(int worker_id) override {
        std::mt19937 RNG(opt_.i("random_seed") + worker_id);
        while (true) {
            job_t job = job_queue_.pop();
            if (job.size == -1) break;
            process_job(job, worker_id, RNG);
        }
    }

    // bpr.cc:92-186 (body of the while loop)
    void process_job(job_t& job, int worker_id, std::mt19937& RNG) override {
        (void)worker_id;
        const bool use_bias = opt_.b("use_bias");
        const bool update_i = opt_.b("update_i");
        const bool update_j = opt_.b("update_j");
        const float reg_u = (float)opt_.d("reg_u"), reg_i = (float)opt_.d("reg_i");
        const float reg_j = (float)opt_.d("reg_j");
        const int num_negative_samples = opt_.i("num_negative_samples");
        const double sample_power = opt_.d("sampling_power");
        const bool verify_neg = opt_.b("verify_neg");
        const int uniform_sampling = sample_power == 0.0 ? 1 : 0;
        const bool pcn = opt_.b("per_coordinate_normalize");
        const bool sgd = optimizer_ == "sgd";
        const int D = D_;
        const uint32_t seed = (uint32_t)opt_.i("random_seed");
        // Q-20: rng1 is only constructed when it can be drawn from.
        const int64_t cum_total = (!uniform_sampling && cum_table_size_ > 0) ? cum_table_[cum_table_size_ - 1] : 1;
        std::uniform_int_distribution<int64_t> rng1(0, std::max<int64_t>(cum_total - 1, 0));
        std::uniform_int_distribution<int64_t> rng2(0, Q_rows_ - 1);

        int processed_samples = 0, total_samples = job.size;
        const float alpha = (float)job.alpha;
        // the bias updates (bpr.cc:161, 167) are scalar C++ statements: `double alpha`, `double reg_b` and float operands give a
        // double expression that is rounded once when it is stored
        const double alpha_d = job.alpha, reg_b_d = opt_.d("reg_b");
        std::vector<float> item_deriv(D);
        for (size_t si = 0; si < job.samples.size(); ++si) {
            const auto& _seen = job.samples[si];
            const int u = _seen[0];
            std::unordered_set<int> seen(_seen.begin() + 1, _seen.end());
            float* Pu = &P_[(size_t)u * D];
            // iteration order over the user's positives
            std::vector<std::pair<int, int64_t>> order;  // (pos, global nnz index)
            order.reserve(_seen.size());
            if (pos_order_ == 0) {
                for (const auto pos : seen) order.emplace_back(pos, -1);
                if (sampler_ == 1) {  // need nnz positions: look them up
                    for (auto& pr : order)
                        for (size_t k = 1; k < _seen.size(); ++k)
                            if (_seen[k] == pr.first) { pr.second = job.base[si] + (int64_t)k - 1; break; }
                }
            } else {
                for (size_t k = 1; k < _seen.size(); ++k) order.emplace_back(_seen[k], job.base[si] + (int64_t)k - 1);
            }
            for (const auto& pr : order) {
                const int pos = pr.first;
                for (int i = 0; i < num_negative_samples; ++i) {
                    int neg = 0;
                    uint32_t attempt = 0;
                    while (true) {
                        if (sampler_ == 0) {
                            if (uniform_sampling) {
                                neg = (int)rng2(RNG);
                            } else {
                                int64_t r = rng1(RNG);
                                neg = (int)(std::lower_bound(cum_table_, cum_table_ + cum_table_size_, r) - cum_table_);
                            }
                        } else {
                            uint32_t o[4];
                            counter_draw(seed, 0u, (uint64_t)pr.second, (uint32_t)i, epoch_, attempt++, o);
                            if (uniform_sampling) {
                                neg = (int)mulhi_u32(o[0], (uint32_t)Q_rows_);
                            } else {
                                int64_t r = mulhi_u64(((uint64_t)o[1] << 32) | o[0], (uint64_t)cum_total);
                                neg = (int)(std::lower_bound(cum_table_, cum_table_ + cum_table_size_, r) - cum_table_);
                            }
                        }
                        if (!verify_neg || (seen.find(neg) == seen.end())) break;
                    }
                    trace_push(u, pos, neg);
                    float* Qp = &Q_[(size_t)pos * D];
                    float* Qn = &Q_[(size_t)neg * D];
                    float x_uij = 0.f;
#pragma omp simd reduction(+ : x_uij)
                    for (int k = 0; k < D; ++k) x_uij += Pu[k] * (Qp[k] - Qn[k]);
                    if (use_bias) x_uij += (Qb_[pos] - Qb_[neg]);

                    float logit = 0.0;
                    if (MAX_EXP < x_uij) {
                        logit = 0.0;
                    } else if (x_uij < -MAX_EXP) {
                        logit = 1.0;
                    } else {
                        // integer arithmetic: EXP_TABLE_SIZE / MAX_EXP / 2 == 83 (Q-2)
                        logit = exp_table_[(int)((x_uij + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))];
                    }
                    if (update_i || update_j)
                        for (int k = 0; k < D; ++k) item_deriv[k] = logit * Pu[k];

                    if (!sgd) {
                        if (pcn) {
#pragma omp atomic
                            Qcnt_[neg] += 1;
                        }
                        float* gP = &gradP_[(size_t)u * D];
                        for (int k = 0; k < D; ++k) gP[k] += logit * (Qp[k] - Qn[k]);
                        if (update_i) {
                            float* g = &gradQ_[(size_t)pos * D];
                            for (int k = 0; k < D; ++k) g[k] += item_deriv[k];
                            if (use_bias) gradQb_[pos] += logit;
                        }
                        if (update_j) {
                            float* g = &gradQ_[(size_t)neg * D];
                            for (int k = 0; k < D; ++k) g[k] -= item_deriv[k];
                            if (use_bias) gradQb_[neg] -= logit;
                        }
                    } else {
                        // Q-1: `g` is a lazy Eigen expression in the reference: it is evaluated at
                        // the final `P_.row(u) += alpha * g`, i.e. AFTER Q rows were updated, while
                        // item_deriv (a concrete matrix) holds logit * OLD P_u.
                        if (update_i) {
                            for (int k = 0; k < D; ++k) Qp[k] += alpha * (item_deriv[k] - reg_i * Qp[k]);
                            if (use_bias) Qb_[pos] = (float)((double)Qb_[pos] + alpha_d * ((double)logit - reg_b_d * (double)Qb_[pos]));
                        }
                        if (update_j) {
                            for (int k = 0; k < D; ++k) Qn[k] += alpha * (-item_deriv[k] - reg_j * Qn[k]);
                            if (use_bias) Qb_[neg] = (float)((double)Qb_[neg] + alpha_d * (-(double)logit - reg_b_d * (double)Qb_[neg]));
                        }
                        for (int k = 0; k < D; ++k)
                            Pu[k] += alpha * (logit * (Qp[k] - Qn[k]) - reg_u * Pu[k]);
                    }
                    stat_updates_ += 1;
                }
                if (!sgd && pcn) {
                    Pcnt_[u] += 1;
#pragma omp atomic
                    Qcnt_[pos] += 1;
                }
            }
            processed_samples += (int)_seen.size() - 1;
        }
        stat_samples_ += processed_samples;
        if (!inline_)
            progress_queue_.push(progress_t{(int)job.samples.size(), processed_samples, total_samples, 0.0});
    }

    // The SGD branch of the loop body above (bpr.cc:119-131, 157-171) applied to a GIVEN list of triples, one after
    // the other: lets a test replay any schedule of an epoch (e.g. the item-major walk of the HIP backend) through
    // the same arithmetic.  Sequential, single thread; `alpha` is the learning rate of every step.
    void apply_triples(int64_t n, const int32_t* users, const int32_t* positives, const int32_t* negatives, double alpha_d) {
        const float alpha = (float)alpha_d;
        const double reg_b_d = opt_.d("reg_b");     // the bias statements are scalar C++ in double, as in process_job
        const bool use_bias = opt_.b("use_bias");
        const bool update_i = opt_.b("update_i");
        const bool update_j = opt_.b("update_j");
        const float reg_u = (float)opt_.d("reg_u"), reg_i = (float)opt_.d("reg_i");
        const float reg_j = (float)opt_.d("reg_j");
        const int D = D_;
        std::vector<float> item_deriv(D);
        for (int64_t t = 0; t < n; ++t) {
            const int u = users[t], pos = positives[t], neg = negatives[t];
            float* Pu = &P_[(size_t)u * D];
            float* Qp = &Q_[(size_t)pos * D];
            float* Qn = &Q_[(size_t)neg * D];
            float x_uij = 0.f;
#pragma omp simd reduction(+ : x_uij)
            for (int k = 0; k < D; ++k) x_uij += Pu[k] * (Qp[k] - Qn[k]);
            if (use_bias) x_uij += (Qb_[pos] - Qb_[neg]);
            float logit = 0.0;
            if (MAX_EXP < x_uij) {
                logit = 0.0;
            } else if (x_uij < -MAX_EXP) {
                logit = 1.0;
            } else {
                logit = exp_table_[(int)((x_uij + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))];
            }
            if (update_i || update_j)
                for (int k = 0; k < D; ++k) item_deriv[k] = logit * Pu[k];
            if (update_i) {
                for (int k = 0; k < D; ++k) Qp[k] += alpha * (item_deriv[k] - reg_i * Qp[k]);
                if (use_bias) Qb_[pos] = (float)((double)Qb_[pos] + alpha_d * ((double)logit - reg_b_d * (double)Qb_[pos]));
            }
            if (update_j) {
                for (int k = 0; k < D; ++k) Qn[k] += alpha * (-item_deriv[k] - reg_j * Qn[k]);
                if (use_bias) Qb_[neg] = (float)((double)Qb_[neg] + alpha_d * (-(double)logit - reg_b_d * (double)Qb_[neg]));
            }
            for (int k = 0; k < D; ++k) Pu[k] += alpha * (logit * (Qp[k] - Qn[k]) - reg_u * Pu[k]);
        }
    }

    // bpr.cc:217-225
    double distance(size_t p, size_t q) {
        bool use_bias = opt_.b("use_bias");
        float ret = dotf(&P_[p * D_], &Q_[q * D_], D_);
        if (use_bias) ret += Qb_[q];
        return ret;
    }
    // bpr.cc:227-244
    double compute_loss(int32_t n, const int32_t* users, const int32_t* positives, const int32_t* negatives) {
        int num_workers = std::max(1, opt_.i("num_workers"));
        omp_set_num_threads(num_workers);
        std::vector<double> loss(num_workers, 0.0);
#pragma omp parallel for schedule(static)
        for (int idx = 0; idx < n; ++idx) {
            int u = users[idx], i = positives[idx], j = negatives[idx];
            double x_uij = distance(u, i) - distance(u, j);
            loss[omp_get_thread_num()] += std::log(1.0 + std::exp(-x_uij));
        }
        double l = std::accumulate(loss.begin(), loss.end(), 0.0);
        return l / (double)n;
    }
};

// ============================================================================
// CWARP  (lib/algo_impl/warp/warp.cc)
// ============================================================================
class WARP : public SGD {
 public:
    bool l2_ = false;

    // warp.cc:68-85: only the exact string "l2" selects the L2 score (Q-23)
    bool init() override {
        SGD::init();
        l2_ = opt_.s("score_func") == "l2";
        return true;
    }
    // warp.cc:21-28
    float score(const float* u, const float* i) const {
        if (!l2_) return dotf(u, i, D_);
        float s = 0.f;
#pragma omp simd reduction(+ : s)
        for (int k = 0; k < D_; ++k) {
            float df = u[k] - i[k];
            s += df * df;
        }
        return -s;
    }

    void worker(int worker_id) override {
        std::mt19937 RNG(opt_.i("random_seed") + worker_id);
        while (true) {
            job_t job = job_queue_.pop();
            if (job.size == -1) break;
            process_job(job, worker_id, RNG);
        }
    }

    // warp.cc:103-173 (body).  Q-10, Q-11.
    void process_job(job_t& job, int worker_id, std::mt19937& RNG) override {
        (void)worker_id;
        const int max_trial = opt_.i("max_trials");
        const double threshold = opt_.d("threshold");
        const float reg_u = (float)opt_.d("reg_u"), reg_i = (float)opt_.d("reg_i"), reg_j = (float)opt_.d("reg_j");
        const int Q_rows = Q_rows_;
        const int D = D_;
        const bool pcn = opt_.b("per_coordinate_normalize");
        const uint32_t seed = (uint32_t)opt_.i("random_seed");
        std::uniform_int_distribution<int64_t> rng(0, Q_rows - 1);
        int processed_samples = 0, total_samples = job.size;
        double partial_loss = 0.0;
        std::vector<float> ud(D), id(D), jd(D);
        for (size_t si = 0; si < job.samples.size(); ++si) {
            const auto& _seen = job.samples[si];
            const int u = _seen[0];
            std::unordered_set<int> seen(_seen.begin() + 1, _seen.end());
            const float* Pu = &P_[(size_t)u * D];
            std::vector<std::pair<int, int64_t>> order;
            if (pos_order_ == 0) {
                for (const auto pos : seen) order.emplace_back(pos, -1);
                if (sampler_ == 1)
                    for (auto& pr : order)
                        for (size_t k = 1; k < _seen.size(); ++k)
                            if (_seen[k] == pr.first) { pr.second = job.base[si] + (int64_t)k - 1; break; }
            } else {
                for (size_t k = 1; k < _seen.size(); ++k) order.emplace_back(_seen[k], job.base[si] + (int64_t)k - 1);
            }
            for (const auto& pr : order) {
                const int pos = pr.first;
                const float* Qp = &Q_[(size_t)pos * D];
                float ui = score(Pu, Qp);
                float uj = 0;
                int neg = 0;
                int trial = 1;
                uint32_t attempt = 0;
                while (trial <= max_trial) {
                    if (sampler_ == 0) {
                        neg = (int)rng(RNG);
                    } else {
                        uint32_t o[4];
                        counter_draw(seed, 1u, (uint64_t)pr.second, 0u, epoch_, attempt++, o);
                        neg = (int)mulhi_u32(o[0], (uint32_t)Q_rows);
                    }
                    if (seen.find(neg) != seen.end()) continue;  // false negative: not counted
                    trial += 1;
                    uj = score(Pu, &Q_[(size_t)neg * D]);
                    stat_scored_negs_ += 1;
                    if ((ui - uj) < threshold) break;  // violating pair
                    trial += 1;
                }
                if (trial >= max_trial) continue;
                const float* Qn = &Q_[(size_t)neg * D];
                float Phi = std::log(std::max(1, int((Q_rows - seen.size() - 1) / trial)));
                trace_push(u, pos, neg);
                if (!l2_) {  // warp.cc:30-40
                    for (int k = 0; k < D; ++k) {
                        ud[k] = Phi * (Qp[k] - Qn[k]);
                        id[k] = Phi * Pu[k];
                        jd[k] = -id[k];
                    }
                } else {  // warp.cc:42-52
                    for (int k = 0; k < D; ++k) {
                        ud[k] = Phi * 2 * (Qp[k] - Qn[k]);
                        id[k] = Phi * (Pu[k] - Qp[k]);
                        jd[k] = -Phi * (Pu[k] - Qn[k]);
                    }
                }
                float* gP = &gradP_[(size_t)u * D];
                float* gI = &gradQ_[(size_t)pos * D];
                float* gJ = &gradQ_[(size_t)neg * D];
                for (int k = 0; k < D; ++k) gP[k] += ud[k] - reg_u * Pu[k];
                for (int k = 0; k < D; ++k) gI[k] += id[k] - reg_i * Qp[k];
                for (int k = 0; k < D; ++k) gJ[k] += jd[k] - reg_j * Qn[k];
                if (pcn) {
#pragma omp atomic
                    Pcnt_[u] += 1;
                    Qcnt_[pos] += 1;  // the pragma covers only the first increment (warp.cc:161-164)
                    Qcnt_[neg] += 1;
                }
                partial_loss += (uj - ui + threshold);
                stat_updates_ += 1;
            }
            processed_samples += (int)_seen.size() - 1;
        }
        stat_samples_ += processed_samples;
        if (!inline_)
            progress_queue_.push(progress_t{(int)job.samples.size(), processed_samples, total_samples, partial_loss});
    }

    // warp.cc:192-201 (Q-12)
    void update_parameters() override {
        SGD::update_parameters();
        const int D = D_;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < Q_rows_; ++i) {
            float* q = &Q_[(size_t)i * D];
            float n = std::max(1.0f, std::sqrt(dotf(q, q, D)));
            for (int k = 0; k < D; ++k) q[k] /= n;
        }
#pragma omp parallel for schedule(static)
        for (int u = 0; u < P_rows_; ++u) {
            float* p = &P_[(size_t)u * D];
            float n = std::max(1.0f, std::sqrt(dotf(p, p, D)));
            for (int k = 0; k < D; ++k) p[k] /= n;
        }
    }

    // warp.cc:205-226
    double compute_loss(int32_t n, const int32_t* users, const int32_t* positives, const int32_t* negatives) {
        int num_workers = std::max(1, opt_.i("num_workers"));
        omp_set_num_threads(num_workers);
        double threshold = opt_.d("threshold");
        std::vector<int> loss(num_workers, 0);
#pragma omp parallel for schedule(static)
        for (int idx = 0; idx < n; ++idx) {
            int u = users[idx], i = positives[idx], j = negatives[idx];
            double x_ui = score(&P_[(size_t)u * D_], &Q_[(size_t)i * D_]);
            double x_uj = score(&P_[(size_t)u * D_], &Q_[(size_t)j * D_]);
            loss[omp_get_thread_num()] += int((x_ui - x_uj) < threshold);
        }
        double l = double(std::accumulate(loss.begin(), loss.end(), 0));
        return l / (double)n;
    }
};

// ============================================================================
// CALS  (lib/algo_impl/als/als.cc) + Algorithm::_leastsquare (lib/algo.cc:39-131)
// ============================================================================
class ALS {
 public:
    Opt opt_;
    float *P_ = nullptr, *Q_ = nullptr;
    int P_rows_ = 0, Q_rows_ = 0, D_ = 0;
    std::vector<float> FF_;  // D x D; symmetric so the reference's col-major storage is immaterial
    bool use_ialspp_ = false;
    char optimizer_code_ = 0;
    int num_cg_max_iters_ = 3;      // include/buffalo/algo.hpp:88
    float cg_tolerance_ = 1e-10f;   // :89
    float eps_ = 1e-10f;            // :90

    // als.cc:30-69
    bool init() {
        int num_workers = std::max(1, opt_.i("num_workers"));
        omp_set_num_threads(num_workers);
        int d = opt_.i("d");
        D_ = d;
        FF_.assign((size_t)d * d, 0.f);
        eps_ = (float)opt_.d("eps");
        cg_tolerance_ = (float)opt_.d("cg_tolerance");
        num_cg_max_iters_ = opt_.i("num_cg_max_iters");
        std::string optimizer = opt_.s("optimizer");
        if (d >= 128) optimizer = "ialspp";  // Q-13
        use_ialspp_ = false;
        if (optimizer == "llt") optimizer_code_ = 0;
        else if (optimizer == "ldlt") optimizer_code_ = 1;
        else if (optimizer == "manual_cg") optimizer_code_ = 2;
        else if (optimizer == "ialspp") { use_ialspp_ = true; optimizer_code_ = 8; }
        else return false;  // eigen_* Krylov solvers are out of scope
        return true;
    }
    // als.cc:77-84
    void initialize_model(float* P, int P_rows, float* Q, int Q_rows) {
        P_ = P; P_rows_ = P_rows; Q_ = Q; Q_rows_ = Q_rows;
    }
    // als.cc:86-93
    void precompute(int axis) {
        const float* F = axis == 0 ? Q_ : P_;
        const int rows = axis == 0 ? Q_rows_ : P_rows_;
        const int D = D_;
        // float products accumulated blockwise; Eigen's GEMM order is not reproducible anyway.
        const int nth = std::max(1, opt_.i("num_workers"));
        std::vector<std::vector<float>> part(nth, std::vector<float>((size_t)D * D, 0.f));
#pragma omp parallel num_threads(nth)
        {
            std::vector<float>& a = part[omp_get_thread_num()];
#pragma omp for schedule(static)
            for (int r = 0; r < rows; ++r) {
                const float* f = F + (size_t)r * D;
                for (int i = 0; i < D; ++i) {
                    const float fi = f[i];
                    float* ai = &a[(size_t)i * D];
                    for (int j = 0; j < D; ++j) ai[j] += fi * f[j];
                }
            }
        }
        for (size_t k = 0; k < (size_t)D * D; ++k) {
            float s = 0.f;
            for (int t = 0; t < nth; ++t) s += part[t][k];
            FF_[k] = s;
        }
    }

    // lib/algo.cc:39-82.  A is D x D row-major symmetric, y length D, x = row to update.
    void leastsquare(float* x, std::vector<float>& A, const std::vector<float>& y) {
        const int D = D_;
        if (optimizer_code_ == 0 || optimizer_code_ == 1) {
            // case 0: A.llt().solve(y); case 1: A.ldlt().solve(y).  Both are exact solves of an SPD
            // system; restated as an unpivoted Cholesky (llt) / LDL^T (ldlt) in float.
            std::vector<float> L(A);
            std::vector<float> z(y);
            if (optimizer_code_ == 0) {
                for (int j = 0; j < D; ++j) {
                    float s = L[(size_t)j * D + j];
                    for (int k = 0; k < j; ++k) s -= L[(size_t)j * D + k] * L[(size_t)j * D + k];
                    float ljj = std::sqrt(s);
                    L[(size_t)j * D + j] = ljj;
                    for (int i = j + 1; i < D; ++i) {
                        float t = L[(size_t)i * D + j];
                        for (int k = 0; k < j; ++k) t -= L[(size_t)i * D + k] * L[(size_t)j * D + k];
                        L[(size_t)i * D + j] = t / ljj;
                    }
                }
                for (int i = 0; i < D; ++i) {
                    float t = z[i];
                    for (int k = 0; k < i; ++k) t -= L[(size_t)i * D + k] * z[k];
                    z[i] = t / L[(size_t)i * D + i];
                }
                for (int i = D - 1; i >= 0; --i) {
                    float t = z[i];
                    for (int k = i + 1; k < D; ++k) t -= L[(size_t)k * D + i] * z[k];
                    z[i] = t / L[(size_t)i * D + i];
                }
            } else {
                std::vector<float> dg(D);
                for (int j = 0; j < D; ++j) {
                    float s = L[(size_t)j * D + j];
                    for (int k = 0; k < j; ++k) s -= L[(size_t)j * D + k] * L[(size_t)j * D + k] * dg[k];
                    dg[j] = s;
                    for (int i = j + 1; i < D; ++i) {
                        float t = L[(size_t)i * D + j];
                        for (int k = 0; k < j; ++k) t -= L[(size_t)i * D + k] * L[(size_t)j * D + k] * dg[k];
                        L[(size_t)i * D + j] = t / s;
                    }
                }
                for (int i = 0; i < D; ++i) {
                    float t = z[i];
                    for (int k = 0; k < i; ++k) t -= L[(size_t)i * D + k] * z[k];
                    z[i] = t;
                }
                for (int i = 0; i < D; ++i) z[i] /= dg[i];
                for (int i = D - 1; i >= 0; --i) {
                    float t = z[i];
                    for (int k = i + 1; k < D; ++k) t -= L[(size_t)k * D + i] * z[k];
                    z[i] = t;
                }
            }
            for (int i = 0; i < D; ++i) x[i] = z[i];
            return;
        }
        // case 2: manual conjugate gradient (algo.cc:58-82, Q-17)
        std::vector<float> r(D), p(D), Ap(D);
        auto rowmat = [&](const float* v, std::vector<float>& out) {  // out = v * A
            for (int j = 0; j < D; ++j) out[j] = 0.f;
            for (int i = 0; i < D; ++i) {
                const float vi = v[i];
                const float* Ai = &A[(size_t)i * D];
                for (int j = 0; j < D; ++j) out[j] += vi * Ai[j];
            }
        };
        rowmat(x, Ap);
        for (int i = 0; i < D; ++i) r[i] = y[i] - Ap[i];
        if (dotf(y.data(), y.data(), D) < dotf(r.data(), r.data(), D)) {
            for (int i = 0; i < D; ++i) x[i] = 0.f;
            r = y;
        }
        p = r;
        float rs_old = dotf(r.data(), r.data(), D);
        for (int it = 0; it < num_cg_max_iters_; ++it) {
            rowmat(p.data(), Ap);
            float alpha = rs_old / (dotf(Ap.data(), p.data(), D) + eps_);
            for (int i = 0; i < D; ++i) x[i] += alpha * p[i];
            for (int i = 0; i < D; ++i) r[i] -= alpha * Ap[i];
            float rs_new = dotf(r.data(), r.data(), D);
            if (rs_new < cg_tolerance_) break;
            float beta = rs_new / (rs_old + eps_);
            for (int i = 0; i < D; ++i) p[i] = r[i] + beta * p[i];
            rs_old = rs_new;
        }
    }

    // als.cc:95-105
    std::pair<double, double> partial_update(int start_x, int next_x, const int64_t* indptr,
                                             const int32_t* keys, const float* vals, int axis) {
        if (use_ialspp_) return partial_update_ialspp(start_x, next_x, indptr, keys, vals, axis);
        return partial_update_dense(start_x, next_x, indptr, keys, vals, axis);
    }

    // als.cc:107-209
    std::pair<double, double> partial_update_dense(int start_x, int next_x, const int64_t* indptr,
                                                   const int32_t* keys, const float* vals, int axis) {
        if ((next_x - start_x) == 0) return std::make_pair(0.0, 0.0);
        float reg = axis == 0 ? (float)opt_.d("reg_u") : (float)opt_.d("reg_i");
        float* P = axis == 0 ? P_ : Q_;
        const float* Q = axis == 0 ? Q_ : P_;
        const int Q_rows = axis == 0 ? Q_rows_ : P_rows_;
        const int D = D_;
        const int num_workers = std::max(1, opt_.i("num_workers"));
        const bool adaptive_reg = opt_.b("adaptive_reg");
        const bool closs = opt_.b("compute_loss_on_training");
        const float alpha = (float)opt_.d("alpha");
        omp_set_num_threads(num_workers);
        std::vector<double> loss_nume(num_workers, 0.0), loss_deno(num_workers, 0.0);
        const int end_loop = next_x - start_x;
        const int64_t shifted = start_x == 0 ? 0 : indptr[start_x - 1];
#pragma omp parallel num_threads(num_workers)
        {
            int worker_id = omp_get_thread_num();
            std::vector<float> m((size_t)D * D), Fxy(D), tmp(D);
#pragma omp for schedule(dynamic, 4)
            for (int i = 0; i < end_loop; ++i) {
                int x = start_x + i;
                const int u = x;
                int64_t beg = x == 0 ? 0 : indptr[x - 1];
                int64_t end = indptr[x];
                int64_t data_size = end - beg;
                if (data_size == 0) continue;  // Q-16: empty rows are left unchanged
                float* Pu = &P[(size_t)u * D];
                std::fill(m.begin(), m.end(), 0.f);
                std::fill(Fxy.begin(), Fxy.end(), 0.f);
                if (closs && axis == 1) {
                    // P.row(u).dot(P.row(u) * FF_)
                    for (int j = 0; j < D; ++j) {
                        float s = 0.f;
                        for (int k = 0; k < D; ++k) s += Pu[k] * FF_[(size_t)k * D + j];
                        tmp[j] = s;
                    }
                    loss_nume[worker_id] += dotf(Pu, tmp.data(), D);
                    loss_deno[worker_id] += Q_rows;
                }
                for (int64_t it = beg; it < end; ++it) {
                    const int c = keys[it - shifted];
                    const float v = vals[it - shifted];
                    const float* q = &Q[(size_t)c * D];
                    const float coef = (float)(1.0 + v * alpha);
                    for (int k = 0; k < D; ++k) Fxy[k] += q[k] * coef;
                    // FiF += (v q)^T q   (Fs^T * Fs2, scaled by alpha below)
                    for (int a = 0; a < D; ++a) {
                        const float va = v * q[a];
                        float* ma = &m[(size_t)a * D];
                        for (int b = 0; b < D; ++b) ma[b] += va * q[b];
                    }
                    if (closs && axis == 1) {
                        float dot = dotf(Pu, q, D);
                        loss_nume[worker_id] -= dot * dot;
                        loss_nume[worker_id] += (dot - 1) * (dot - 1) * (1.0 + v * alpha);
                        loss_deno[worker_id] += v * alpha;
                    }
                }
                for (size_t k = 0; k < (size_t)D * D; ++k) m[k] = FF_[k] + m[k] * alpha;
                float ada_reg = adaptive_reg ? (float)data_size : 1.0f;
                if (closs) loss_nume[worker_id] += ada_reg * reg * dotf(Pu, Pu, D);
                for (int d = 0; d < D; ++d) m[(size_t)d * D + d] += (reg * ada_reg);
                leastsquare(Pu, m, Fxy);
            }
        }
        return std::make_pair(std::accumulate(loss_nume.begin(), loss_nume.end(), 0.0),
                              std::accumulate(loss_deno.begin(), loss_deno.end(), 0.0));
    }

    // als.cc:211-358 (Q-14).  Rows are independent (Yui is indexed per nnz), so the reference's
    // "for block: for row" nest is restated as "for row: for block".
    std::pair<double, double> partial_update_ialspp(int start_x, int next_x, const int64_t* indptr,
                                                    const int32_t* keys, const float* vals, int axis) {
        if ((next_x - start_x) == 0) return std::make_pair(0.0, 0.0);
        float reg = axis == 0 ? (float)opt_.d("reg_u") : (float)opt_.d("reg_i");
        float* P = axis == 0 ? P_ : Q_;
        const float* Q = axis == 0 ? Q_ : P_;
        const int Q_rows = axis == 0 ? Q_rows_ : P_rows_;
        const int D = D_;
        const int num_workers = std::max(1, opt_.i("num_workers"));
        const bool adaptive_reg = opt_.b("adaptive_reg");
        const bool closs = opt_.b("compute_loss_on_training");
        const float alpha = (float)opt_.d("alpha");
        const int block_size_ = std::min(D, opt_.i("block_size"));
        omp_set_num_threads(num_workers);
        std::vector<double> loss_nume(num_workers, 0.0), loss_deno(num_workers, 0.0);
        const int end_loop = next_x - start_x;
        const int64_t shifted = start_x == 0 ? 0 : indptr[start_x - 1];
        // Q-15: the reference sizes Yui as indptr[end_loop-1], which is the chunk nnz only when the
        // chunk starts at row 0; sized correctly here (identical in single-chunk mode).
        std::vector<float> Yui((size_t)(indptr[next_x - 1] - shifted));
#pragma omp parallel for schedule(dynamic, 16)
        for (int i = 0; i < end_loop; ++i) {
            int x = start_x + i;
            int64_t beg = x == 0 ? 0 : indptr[x - 1];
            int64_t end = indptr[x];
            for (int64_t it = beg; it < end; ++it)
                Yui[it - shifted] = dotf(&P[(size_t)x * D], &Q[(size_t)keys[it - shifted] * D], D);
        }
#pragma omp parallel num_threads(num_workers)
        {
            int worker_id = omp_get_thread_num();
            std::vector<float> pc(D), b, xs, r, pv, Ap, tmp(D);
#pragma omp for schedule(dynamic, 4)
            for (int i = 0; i < end_loop; ++i) {
                int x = start_x + i;
                const int u = x;
                int64_t beg = x == 0 ? 0 : indptr[x - 1];
                int64_t end = indptr[x];
                int64_t data_size = end - beg;
                if (data_size == 0) continue;
                float* Pu = &P[(size_t)u * D];
                for (int block_beg = 0; block_beg < D; block_beg += block_size_) {
                    int bs = block_size_;
                    if (block_beg + bs >= D) bs = D - block_beg;
                    // p: copy of the whole row at block start (const FactorType& p = P.row(u))
                    for (int k = 0; k < D; ++k) pc[k] = Pu[k];
                    b.assign(bs, 0.f);
                    // b = p * gramian + reg * block_p,  gramian = FF_.block(0, block_beg, D, bs)
                    for (int j = 0; j < bs; ++j) {
                        float s = 0.f;
#pragma omp simd reduction(+ : s)
                        for (int k = 0; k < D; ++k) s += pc[k] * FF_[(size_t)k * D + block_beg + j];
                        b[j] = s + reg * pc[block_beg + j];
                    }
                    if (block_beg == 0 && closs && axis == 1) {
                        for (int j = 0; j < D; ++j) {
                            float s = 0.f;
                            for (int k = 0; k < D; ++k) s += Pu[k] * FF_[(size_t)k * D + j];
                            tmp[j] = s;
                        }
                        loss_nume[worker_id] += dotf(Pu, tmp.data(), D);
                        loss_deno[worker_id] += Q_rows;
                    }
                    for (int64_t it = beg; it < end; ++it) {
                        const int col = keys[it - shifted];
                        const float val = vals[it - shifted];
                        float residual = Yui[it - shifted] - 1.0;
                        const float* v = &Q[(size_t)col * D + block_beg];
                        const float coef = residual * val * alpha;
                        for (int j = 0; j < bs; ++j) b[j] += coef * v[j];
                        if ((block_beg == 0) && closs && axis == 1) {
                            float dot = dotf(Pu, &Q[(size_t)col * D], D);
                            loss_nume[worker_id] -= dot * dot;
                            loss_nume[worker_id] += (dot - 1) * (dot - 1) * (1.0 + val * alpha);
                            loss_deno[worker_id] += val * alpha;
                        }
                    }
                    float ada_reg = adaptive_reg ? (float)data_size : 1.0f;
                    if ((block_beg == 0) && closs) loss_nume[worker_id] += ada_reg * reg * dotf(Pu, Pu, D);
                    // CG (hard-coded 3 steps, plain reg, no eps; rs in double)
                    xs.assign(bs, 0.f);
                    r = b;
                    pv = r;
                    Ap.assign(bs, 0.f);
                    double rsold = dotf(r.data(), r.data(), bs);
                    if (rsold > cg_tolerance_) {
                        for (int cg_step = 0; cg_step < 3; ++cg_step) {
                            // Ap = A * p,  A = FF[blk,blk] + I*reg
                            for (int a = 0; a < bs; ++a) {
                                float s = 0.f;
#pragma omp simd reduction(+ : s)
                                for (int c = 0; c < bs; ++c)
                                    s += (FF_[(size_t)(block_beg + a) * D + block_beg + c] + (a == c ? reg : 0.f)) * pv[c];
                                Ap[a] = s;
                            }
                            for (int64_t it = beg; it < end; ++it) {
                                const int col = keys[it - shifted];
                                const float val = vals[it - shifted];
                                const float* v = &Q[(size_t)col * D + block_beg];
                                const float coef = val * alpha * dotf(v, pv.data(), bs);
                                for (int j = 0; j < bs; ++j) Ap[j] += coef * v[j];
                            }
                            float step_size = rsold / dotf(pv.data(), Ap.data(), bs);
                            for (int j = 0; j < bs; ++j) xs[j] += step_size * pv[j];
                            for (int j = 0; j < bs; ++j) r[j] -= step_size * Ap[j];
                            double rsnew = dotf(r.data(), r.data(), bs);
                            if (rsnew < cg_tolerance_) break;
                            const float ratio = (float)(rsnew / rsold);
                            for (int j = 0; j < bs; ++j) pv[j] = r[j] + ratio * pv[j];
                            rsold = rsnew;
                        }
                    }
                    for (int j = 0; j < bs; ++j) Pu[block_beg + j] -= xs[j];
                    for (int64_t it = beg; it < end; ++it) {
                        const int col = keys[it - shifted];
                        Yui[it - shifted] -= dotf(&Q[(size_t)col * D + block_beg], xs.data(), bs);
                    }
                }
            }
        }
        return std::make_pair(std::accumulate(loss_nume.begin(), loss_nume.end(), 0.0),
                              std::accumulate(loss_deno.begin(), loss_deno.end(), 0.0));
    }
};

// ============================================================================
// CCFR  (lib/algo_impl/cfr/cfr.cc) -- CoFactor: user-item ALS regularised by an item-context (SPPMI)
// factorisation with biases.  SURVEY.md section 8(f) rank 4: "CFR calls the same _leastsquare".
// Quirks kept: `cg_tolerance_` is read from the key "cg_tolerance_" (cfr.cc:38), which no option
// object defines, so the manual CG never stops on its tolerance; `compute_loss` (not
// compute_loss_on_training) switches the loss terms (:44); d >= 128 does NOT force iALS++ here.
// ============================================================================
class CFR : public ALS {
 public:
    float *U_ = nullptr, *I_ = nullptr, *C_ = nullptr, *Ib_ = nullptr, *Cb_ = nullptr;
    int U_rows_ = 0, I_rows_ = 0, C_rows_ = 0;
    float alpha_ = 0.f, l_ = 1.f, reg_u_ = 0.f, reg_i_ = 0.f, reg_c_ = 0.f;
    bool compute_loss_ = false;

    // cfr.cc:29-63
    bool init_cfr() {
        omp_set_num_threads(std::max(1, opt_.i("num_workers")));
        D_ = opt_.i("d");
        num_cg_max_iters_ = opt_.i("num_cg_max_iters");
        alpha_ = (float)opt_.d("alpha");
        l_ = (float)opt_.d("l");
        cg_tolerance_ = (float)opt_.d("cg_tolerance_");   // sic: 0 unless someone sets that key
        eps_ = (float)opt_.d("eps");
        reg_u_ = (float)opt_.d("reg_u");
        reg_i_ = (float)opt_.d("reg_i");
        reg_c_ = (float)opt_.d("reg_c");
        compute_loss_ = opt_.b("compute_loss");
        const std::string optimizer = opt_.s("optimizer");
        if (optimizer == "llt") optimizer_code_ = 0;
        else if (optimizer == "ldlt") optimizer_code_ = 1;
        else if (optimizer == "manual_cg") optimizer_code_ = 2;
        else return false;   // eigen_* Krylov solvers are out of scope
        use_ialspp_ = false;
        FF_.assign((size_t)D_ * D_, 0.f);
        return true;
    }
    // cfr.cc:70-82
    void set_embedding(float* data, int size, const std::string& t) {
        if (t == "user") { U_ = data; U_rows_ = size; }
        else if (t == "item") { I_ = data; I_rows_ = size; }
        else if (t == "context") { C_ = data; C_rows_ = size; }
        else if (t == "item_bias") Ib_ = data;
        else if (t == "context_bias") Cb_ = data;
    }
    // cfr.cc:85-90
    void precompute_cfr(const std::string& t) {
        if (t == "user") { Q_ = U_; Q_rows_ = U_rows_; precompute(0); }
        else if (t == "item") { Q_ = I_; Q_rows_ = I_rows_; precompute(0); }
    }
    // A += sum_k w_k f_k f_k^T (row-major, symmetric), y += sum_k c_k f_k
    void accumulate(std::vector<float>& A, std::vector<float>& y, const float* F, const int32_t* keys, const std::vector<float>& w,
                    const std::vector<float>& c, size_t n) {
        const int D = D_;
        for (size_t k = 0; k < n; ++k) {
            const float* f = F + (size_t)keys[k] * D;
            for (int i = 0; i < D; ++i) {
                const float wi = w[k] * f[i];
                float* Ai = &A[(size_t)i * D];
                for (int j = 0; j < D; ++j) Ai[j] += wi * f[j];
                y[i] += c[k] * f[i];
            }
        }
    }
    // cfr.cc:92-146
    double partial_update_user(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals) {
        if (next_x == start_x) return 0.0;
        const int D = D_;
        const int64_t shifted = start_x == 0 ? 0 : indptr[start_x - 1];
        double loss = 0.0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : loss)
        for (int x = start_x; x < next_x; ++x) {
            const size_t beg = x == 0 ? 0 : indptr[x - 1] - shifted, end = indptr[x] - shifted, n = end - beg;
            if (n == 0) continue;
            std::vector<float> A(FF_), y(D, 0.f), w(n), c(n);
            for (size_t k = 0; k < n; ++k) { w[k] = vals[beg + k] * alpha_; c[k] = w[k] + 1.f; }
            accumulate(A, y, I_, keys + beg, w, c, n);
            for (auto& a : A) a *= l_;
            for (auto& v : y) v *= l_;
            for (int d = 0; d < D; ++d) A[(size_t)d * D + d] += reg_u_;
            float* ux = U_ + (size_t)x * D;
            leastsquare(ux, A, y);
            if (compute_loss_) loss += dotf(ux, ux, D);
        }
        return reg_u_ * loss;
    }
    // cfr.cc:148-255
    double partial_update_item(int start_x, int next_x, const int64_t* indptr_u, const int32_t* keys_u, const float* vals_u,
                               const int64_t* indptr_c, const int32_t* keys_c, const float* vals_c) {
        if (next_x == start_x) return 0.0;
        const int D = D_;
        const int64_t shifted_u = start_x == 0 ? 0 : indptr_u[start_x - 1];
        const int64_t shifted_c = start_x == 0 ? 0 : indptr_c[start_x - 1];
        double total = 0.0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
        for (int x = start_x; x < next_x; ++x) {
            const size_t beg_u = x == 0 ? 0 : indptr_u[x - 1] - shifted_u, end_u = indptr_u[x] - shifted_u, nu = end_u - beg_u;
            const size_t beg_c = x == 0 ? 0 : indptr_c[x - 1] - shifted_c, end_c = indptr_c[x] - shifted_c, nc = end_c - beg_c;
            if (nu == 0 && nc == 0) continue;
            float* ix = I_ + (size_t)x * D;
            std::vector<float> A(FF_), y(D, 0.f), w(nu), c(nu);
            float loss = 0.f;
            if (compute_loss_) {   // (I_x FF) . I_x
                for (int i = 0; i < D; ++i) {
                    float t = 0.f;
                    for (int j = 0; j < D; ++j) t += ix[j] * FF_[(size_t)j * D + i];
                    loss += t * ix[i];
                }
            }
            for (size_t k = 0; k < nu; ++k) {
                w[k] = vals_u[beg_u + k] * alpha_;
                if (compute_loss_) {
                    const float dot = dotf(ix, U_ + (size_t)keys_u[beg_u + k] * D, D);
                    loss += (-dot * dot + (1 + w[k]) * (dot - 1) * (dot - 1));
                }
                c[k] = w[k] + 1.f;
            }
            if (compute_loss_) total += loss * l_;
            accumulate(A, y, U_, keys_u + beg_u, w, c, nu);
            for (auto& a : A) a *= l_;
            for (auto& v : y) v *= l_;
            std::vector<float> w1(nc, 1.f), cc(nc);
            loss = 0.f;
            for (size_t k = 0; k < nc; ++k) {
                const int cidx = keys_c[beg_c + k];
                const float v = vals_c[beg_c + k];
                cc[k] = v - Ib_[x] - Cb_[cidx];
                if (compute_loss_) {
                    const float err = v - dotf(ix, C_ + (size_t)cidx * D, D) - Ib_[x] - Cb_[cidx];
                    loss += err * err;
                }
            }
            if (compute_loss_) total += loss + reg_i_ * dotf(ix, ix, D);
            accumulate(A, y, C_, keys_c + beg_c, w1, cc, nc);
            for (int d = 0; d < D; ++d) A[(size_t)d * D + d] += reg_i_;
            leastsquare(ix, A, y);
            float b = 0.f;   // bias from the UPDATED row (cfr.cc:244-250)
            for (size_t k = 0; k < nc; ++k) {
                const int cidx = keys_c[beg_c + k];
                b += (vals_c[beg_c + k] - dotf(ix, C_ + (size_t)cidx * D, D) - Cb_[cidx]);
            }
            Ib_[x] = b / ((float)nc + 1e-10f);
        }
        return total;
    }
    // cfr.cc:257-313
    double partial_update_context(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, const float* vals) {
        if (next_x == start_x) return 0.0;
        const int D = D_;
        const int64_t shifted = start_x == 0 ? 0 : indptr[start_x - 1];
        double loss = 0.0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : loss)
        for (int x = start_x; x < next_x; ++x) {
            const size_t beg = x == 0 ? 0 : indptr[x - 1] - shifted, end = indptr[x] - shifted, n = end - beg;
            if (n == 0) continue;
            float* cx = C_ + (size_t)x * D;
            std::vector<float> A((size_t)D * D, 0.f), y(D, 0.f), w(n, 1.f), c(n);
            for (size_t k = 0; k < n; ++k) c[k] = vals[beg + k] - Cb_[x] - Ib_[keys[beg + k]];
            accumulate(A, y, I_, keys + beg, w, c, n);
            for (int d = 0; d < D; ++d) A[(size_t)d * D + d] += reg_c_;
            if (compute_loss_) loss += dotf(cx, cx, D);   // the row BEFORE the update (cfr.cc:297-298)
            leastsquare(cx, A, y);
            float b = 0.f;
            for (size_t k = 0; k < n; ++k) b += (vals[beg + k] - dotf(cx, I_ + (size_t)keys[beg + k] * D, D) - Ib_[keys[beg + k]]);
            Cb_[x] = b / ((float)n + 1e-10f);
        }
        return reg_c_ * loss;
    }
};

// ============================================================================
// CEALS  (lib/algo_impl/eals/eals.cc) -- element-wise ALS (He et al., SIGIR'16): coordinate descent over the
// latent dimensions of every row with a cache of the predictions vhat for the observed entries, kept in both
// orientations (vhat_u in user-major, vhat_i in item-major order) and linked by index maps.
// Whole-matrix calls (no chunking): indptr are END offsets of the full matrix.
// ============================================================================
class EALS {
 public:
    Opt opt_;
    float *P_ = nullptr, *Q_ = nullptr, *C_ = nullptr;
    int P_rows_ = 0, Q_rows_ = 0;
    bool cached_[2] = {false, false};
    std::vector<float> vhat_[2];       // [0] user-major, [1] item-major
    std::vector<int64_t> map_[2];      // [0] u2i: position in the item-major order of user-major entry ind; [1] i2u
    bool init() { omp_set_num_threads(std::max(1, opt_.i("num_workers"))); return true; }   // eals.cc:19-26
    void initialize_model(float* P, float* Q, float* C, int P_rows, int Q_rows) {           // eals.cc:33-47
        P_ = P; Q_ = Q; C_ = C; P_rows_ = P_rows; Q_rows_ = Q_rows;
        cached_[0] = cached_[1] = false;
    }
    // eals.cc:49-100
    void precompute_cache(int nnz, const int64_t* indptr, const int32_t* keys, int axis) {
        if (cached_[axis]) return;
        const float* X = axis == 0 ? P_ : Q_;
        const float* Y = axis == 0 ? Q_ : P_;
        const int rows = axis == 0 ? P_rows_ : Q_rows_, D = opt_.i("d");
        vhat_[axis].assign(nnz, 0.f);
        std::vector<std::pair<std::pair<int32_t, int32_t>, int64_t>> coord(nnz);   // ((other id, own id), position)
#pragma omp parallel for schedule(dynamic, 8)
        for (int x = 0; x < rows; ++x) {
            const int64_t beg = x == 0 ? 0 : indptr[x - 1], end = indptr[x];
            for (int64_t ind = beg; ind < end; ++ind) {
                const int32_t y = keys[ind];
                float acc = 0.f;   // std::inner_product: sequential
                for (int c = 0; c < D; ++c) acc += X[(size_t)x * D + c] * Y[(size_t)y * D + c];
                vhat_[axis][ind] = acc;
                coord[ind] = {{y, x}, ind};
            }
        }
        std::sort(coord.begin(), coord.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        map_[axis].assign(nnz, 0);
        for (int64_t r = 0; r < nnz; ++r) map_[axis][coord[r].second] = r;
        cached_[axis] = true;
    }
    // S[d][e] = sum_rows w_r F[r][d] F[r][e]  (blas::syrk + fill_left_elems: the full symmetric matrix)
    std::vector<float> gram(const float* F, int rows, const float* w) const {
        const int D = opt_.i("d");
        std::vector<float> S((size_t)D * D, 0.f);
        for (int r = 0; r < rows; ++r) {
            const float wr = w ? w[r] : 1.f;   // (sqrt(C) q)(sqrt(C) q)^T = C q q^T
            const float* f = F + (size_t)r * D;
            for (int a = 0; a < D; ++a) {
                const float fa = wr * f[a];
                for (int b = 0; b < D; ++b) S[(size_t)a * D + b] += fa * f[b];
            }
        }
        return S;
    }
    // eals.cc:102-115, 176-281
    bool update(const int64_t* indptr, const int32_t* keys, const float* vals, int axis) {
        if (!(cached_[0] && cached_[1])) return false;
        const int D = opt_.i("d");
        const float alpha = (float)opt_.d("alpha");
        const float reg = (float)opt_.d(axis == 0 ? "reg_u" : "reg_i");
        float* X = axis == 0 ? P_ : Q_;
        const float* Y = axis == 0 ? Q_ : P_;
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        const std::vector<float> S = axis == 0 ? gram(Q_, Q_rows_, C_) : gram(P_, P_rows_, nullptr);
        std::vector<float>& own = vhat_[axis];
        std::vector<float>& other = vhat_[1 - axis];
        const std::vector<int64_t>& map = map_[axis];
#pragma omp parallel for schedule(dynamic, 8)
        for (int x = 0; x < rows; ++x) {
            float* xp = X + (size_t)x * D;
            const int64_t beg = x == 0 ? 0 : indptr[x - 1], end = indptr[x];
            for (int d = 0; d < D; ++d) {
                float numerator = 0.f, denominator = 0.f;
                for (int64_t ind = beg; ind < end; ++ind) {
                    const int32_t y = keys[ind];
                    const float v = vals[ind];
                    const float yd = Y[(size_t)y * D + d];
                    const float pq = xp[d] * yd;
                    const float vf = own[ind] - pq;
                    const float w = 1.f + alpha * v;
                    const float wmc = w - C_[axis == 0 ? y : x];
                    numerator += (w * v - wmc * vf) * yd;
                    denominator += wmc * yd * yd;
                    own[ind] -= pq;
                    other[map[ind]] -= pq;
                }
                float dot = 0.f;
                for (int c = 0; c < D; ++c) dot += xp[c] * S[(size_t)d * D + c];
                if (axis == 0) {
                    numerator += -dot + xp[d] * S[(size_t)d * D + d];
                    denominator += S[(size_t)d * D + d] + reg;
                } else {
                    numerator += -C_[x] * (dot - xp[d] * S[(size_t)d * D + d]);
                    denominator += C_[x] * S[(size_t)d * D + d] + reg;
                }
                xp[d] = numerator / denominator;
                for (int64_t ind = beg; ind < end; ++ind) {
                    const float pq = xp[d] * Y[(size_t)keys[ind] * D + d];
                    own[ind] += pq;
                    other[map[ind]] += pq;
                }
            }
        }
        return true;
    }
    // eals.cc:117-174: float accumulators, as the reference
    std::pair<float, float> estimate_loss(int nnz, const int64_t* indptr, const int32_t* keys, const float* vals, int axis) {
        if (!(cached_[0] && cached_[1])) return {0.f, 0.f};
        const int D = opt_.i("d");
        const float alpha = (float)opt_.d("alpha");
        float feedbacks = 0.f, squared_error = 0.f;
        const int rows = axis == 0 ? P_rows_ : Q_rows_;
        for (int x = 0; x < rows; ++x) {
            const int64_t beg = x == 0 ? 0 : indptr[x - 1], end = indptr[x];
            for (int64_t ind = beg; ind < end; ++ind) {
                const float v = vals[ind], vh = vhat_[axis][ind], err = v - vh;
                feedbacks += (1.f + alpha * v) * err * err;
                feedbacks -= C_[axis == 0 ? keys[ind] : x] * vh * vh;
                squared_error += err * err;
            }
        }
        float rp = 0.f, rq = 0.f;
        for (size_t k = 0; k < (size_t)P_rows_ * D; ++k) rp += P_[k] * P_[k];
        for (size_t k = 0; k < (size_t)Q_rows_ * D; ++k) rq += Q_[k] * Q_[k];
        const float reg = (float)opt_.d("reg_u") * rp + (float)opt_.d("reg_i") * rq;
        const std::vector<float> Sp = gram(P_, P_rows_, nullptr), Sq = gram(Q_, Q_rows_, C_);
        // <Sp, Sq> over the full symmetric D x D matrices (blas::syrk mirrors the triangle it computed, misc/blas.hpp:49-63,80)
        float ip = 0.f;
        for (size_t k = 0; k < (size_t)D * D; ++k) ip += Sp[k] * Sq[k];
        feedbacks += ip;
        return {std::sqrt(squared_error / (float)nnz), feedbacks + reg};
    }
};

struct Handle {
    int kind;  // 0 bpr, 1 warp, 2 als, 3 cfr
    SGD* sgd = nullptr;
    ALS* als = nullptr;   // kind 3: a CFR
    EALS* eals = nullptr; // kind 4
    std::vector<int32_t> trace;
    Opt* opt() { return kind == 4 ? &eals->opt_ : (kind >= 2 ? &als->opt_ : &sgd->opt_); }
};

}  // namespace

// ============================================================================
// C ABI consumed by oracle/oracle.py (ctypes)
// ============================================================================
// ------------------------------------------------------------------------------------------------
// Inference-side selection, restating buffalo/parallel/_core.hpp.
//
// orc_dot_topn  ~ parallel::dot_topn (_core.hpp:89-142): per query q = indexes[i], candidates j are
//   visited in ascending order; skipped when the two factor matrices are the same array and j == q
//   (:117-118) or when a pool is given and j is not in it (:119-120); score = P[q].Q[j] (+ Qb[j]).
//   The running list (topn_t, :37-67) is kept sorted by descending score in `correct_k` slots that
//   start as (key -1, val FLT_MIN).  A candidate is admitted only if score > last slot's val
//   (strict, :124) -- since FLT_MIN is the smallest POSITIVE normal float, non-positive scores are
//   never admitted -- and is placed in front of the first slot whose val is not greater than it
//   (lower_bound with `val > that`, :53), i.e. BEFORE earlier candidates of equal score.
//   Net effect: listed by (score desc, j desc); at the boundary a tie is admitted only while fewer than
//   k candidates >= that score have been seen, and later better candidates evict the OLDEST tie
//   (closed form: tests/topk_cases.py:spec_dot_topn).
//   Slots correct_k..k-1 are written as (-1, 0.0) (:134-137).
// orc_quickselect ~ parallel::quickselect (_core.hpp:69-87): std::nth_element over column indices
//   with `scores[l] > scores[r]`, then std::sort of the first k when `sorted`.
// ------------------------------------------------------------------------------------------------
static void dot_topn_ref(const int32_t* indexes, int num_queries, const float* P, int p_rows, int p_cols, const float* Q, int q_rows,
                         int q_cols, const float* Qb, int qb_rows, int32_t* out_keys, float* out_scores, const int32_t* pool,
                         int pool_size, int k, int same) {
    (void)p_rows;
    std::unordered_set<int32_t> allowed(pool, pool + pool_size);
    int kk = std::min(q_rows, k);
    if (pool_size) kk = std::min(pool_size, kk);
    const int d = std::min(p_cols, q_cols);
#pragma omp parallel for schedule(guided)
    for (int i = 0; i < num_queries; ++i) {
        std::vector<int32_t> keys(std::max(kk, 1), -1);
        std::vector<float> vals(std::max(kk, 1), std::numeric_limits<float>::min());
        float last = std::numeric_limits<float>::min();
        const int q = indexes[i];
        const float* pq = P + (size_t)q * p_cols;
        for (int j = 0; j < q_rows && kk > 0; ++j) {
            if (same && q == j) continue;
            if (pool_size && !allowed.count(j)) continue;
            const float* qj = Q + (size_t)j * q_cols;
            float score = 0.f;
#pragma omp simd reduction(+ : score)
            for (int c = 0; c < d; ++c) score += pq[c] * qj[c];
            if (qb_rows) score += Qb[j];
            if (!(score > last)) continue;
            int pos = 0;   // first slot whose value is not greater than the candidate
            while (pos < kk && vals[pos] > score) ++pos;
            if (pos >= kk) continue;
            for (int t = kk - 1; t > pos; --t) {
                keys[t] = keys[t - 1];
                vals[t] = vals[t - 1];
            }
            keys[pos] = j;
            vals[pos] = score;
            last = vals[kk - 1];
        }
        for (int t = 0; t < kk; ++t) {
            out_keys[(size_t)i * k + t] = keys[t];
            out_scores[(size_t)i * k + t] = vals[t];
        }
        for (int t = kk; t < k; ++t) {
            out_keys[(size_t)i * k + t] = -1;
            out_scores[(size_t)i * k + t] = 0.0f;
        }
    }
}

static void quickselect_ref(const float* scores, int rows, int cols, int32_t* result, int k, int sorted) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < rows; ++i) {
        const float* srow = scores + (size_t)i * cols;
        std::vector<int> order(cols);
        std::iota(order.begin(), order.end(), 0);
        auto higher = [&](int l, int r) { return srow[l] > srow[r]; };
        std::nth_element(order.begin(), order.begin() + k - 1, order.end(), higher);
        if (sorted) std::sort(order.begin(), order.begin() + k, higher);
        std::copy(order.begin(), order.begin() + k, result + (size_t)i * k);
    }
}

extern "C" {

void* orc_create(int kind) {
    Handle* h = new Handle();
    h->kind = kind;
    if (kind == 0) h->sgd = new BPR();
    else if (kind == 1) h->sgd = new WARP();
    else if (kind == 2) h->als = new ALS();
    else if (kind == 3) h->als = new CFR();
    else h->eals = new EALS();
    return h;
}
void orc_destroy(void* hp) {
    Handle* h = (Handle*)hp;
    delete h->sgd;
    delete h->als;
    delete h;
}
void orc_opt_num(void* hp, const char* k, double v) { ((Handle*)hp)->opt()->num[k] = v; }
void orc_opt_str(void* hp, const char* k, const char* v) { ((Handle*)hp)->opt()->str[k] = v; }
void orc_opt_bool(void* hp, const char* k, int v) { ((Handle*)hp)->opt()->boo[k] = v != 0; }
int orc_init(void* hp) {
    Handle* h = (Handle*)hp;
    if (h->kind == 4) return (int)h->eals->init();
    if (h->kind == 3) return (int)static_cast<CFR*>(h->als)->init_cfr();
    return h->kind == 2 ? (int)h->als->init() : (int)h->sgd->init();
}
void orc_set_modes(void* hp, int sampler, int pos_order, int inline_mode) {
    Handle* h = (Handle*)hp;
    h->sgd->sampler_ = sampler;
    h->sgd->pos_order_ = pos_order;
    h->sgd->inline_ = inline_mode;
}
void orc_set_shard(void* hp, int64_t nnz_offset, int num_shards) {
    Handle* h = (Handle*)hp;
    h->sgd->nnz_offset_ = nnz_offset;
    h->sgd->num_shards_ = num_shards;
}
void orc_trace(void* hp, int on) {
    Handle* h = (Handle*)hp;
    h->trace.clear();
    h->sgd->trace_ = on ? &h->trace : nullptr;
}
int64_t orc_trace_size(void* hp) { return (int64_t)((Handle*)hp)->trace.size() / 3; }
void orc_trace_copy(void* hp, int32_t* out) {
    Handle* h = (Handle*)hp;
    std::memcpy(out, h->trace.data(), h->trace.size() * sizeof(int32_t));
}
void orc_sgd_initialize_model(void* hp, float* P, int P_rows, float* Q, int Q_rows, float* Qb, int64_t n) {
    ((Handle*)hp)->sgd->initialize_model(P, P_rows, Q, Q_rows, Qb, n);
}
void orc_sgd_set_cumulative_table(void* hp, int64_t* t, int size) {
    Handle* h = (Handle*)hp;
    h->sgd->cum_table_ = t;
    h->sgd->cum_table_size_ = size;
}
void orc_sgd_launch_workers(void* hp) { ((Handle*)hp)->sgd->launch_workers(); }
void orc_sgd_add_jobs(void* hp, int s, int n, const int64_t* indptr, const int32_t* keys) {
    ((Handle*)hp)->sgd->add_jobs(s, n, indptr, keys);
}
void orc_sgd_update_parameters(void* hp) { ((Handle*)hp)->sgd->update_parameters(); }
void orc_sgd_wait_until_done(void* hp) { ((Handle*)hp)->sgd->wait_until_done(); }
double orc_sgd_join(void* hp) { return ((Handle*)hp)->sgd->join(); }
// test hook: BPR's SGD step applied to an explicit triple list (kind 0 only)
int orc_bpr_apply_triples(void* hp, int64_t n, const int32_t* u, const int32_t* p, const int32_t* q, double alpha) {
    Handle* h = (Handle*)hp;
    BPR* b = h->kind == 0 ? dynamic_cast<BPR*>(h->sgd) : nullptr;
    if (!b) return 0;
    b->apply_triples(n, u, p, q, alpha);
    return 1;
}
double orc_sgd_compute_loss(void* hp, int n, const int32_t* u, const int32_t* p, const int32_t* q) {
    Handle* h = (Handle*)hp;
    if (h->kind == 0) return static_cast<BPR*>(h->sgd)->compute_loss(n, u, p, q);
    return static_cast<WARP*>(h->sgd)->compute_loss(n, u, p, q);
}
void orc_sgd_stats(void* hp, long long* out3) {
    SGD* s = ((Handle*)hp)->sgd;
    out3[0] = s->stat_samples_;
    out3[1] = s->stat_scored_negs_;
    out3[2] = s->stat_updates_;
}
// optimizer state access for tests: which = 0 gradP,1 gradQ,2 gradQb,3 momP,4 momQ,5 momQb,6 velP,7 velQ,8 velQb
float* orc_sgd_state(void* hp, int which, int64_t* n) {
    SGD* s = ((Handle*)hp)->sgd;
    std::vector<float>* v[] = {&s->gradP_, &s->gradQ_, &s->gradQb_, &s->momP_, &s->momQ_, &s->momQb_,
                               &s->velP_, &s->velQ_, &s->velQb_};
    *n = (int64_t)v[which]->size();
    return v[which]->data();
}
void orc_bpr_exp_table(void* hp, float* out) {
    BPR* b = static_cast<BPR*>(((Handle*)hp)->sgd);
    b->build_exp_table();
    std::memcpy(out, b->exp_table_, sizeof(float) * EXP_TABLE_SIZE);
}
void orc_als_initialize_model(void* hp, float* P, int P_rows, float* Q, int Q_rows) {
    ((Handle*)hp)->als->initialize_model(P, P_rows, Q, Q_rows);
}
void orc_als_precompute(void* hp, int axis) { ((Handle*)hp)->als->precompute(axis); }
void orc_als_get_ff(void* hp, float* out) {
    ALS* a = ((Handle*)hp)->als;
    std::memcpy(out, a->FF_.data(), a->FF_.size() * sizeof(float));
}
void orc_als_partial_update(void* hp, int s, int n, const int64_t* indptr, const int32_t* keys,
                            const float* vals, int axis, double* out2) {
    auto r = ((Handle*)hp)->als->partial_update(s, n, indptr, keys, vals, axis);
    out2[0] = r.first;
    out2[1] = r.second;
}
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}
void orc_counter_draw(uint32_t seed, uint32_t stream, uint64_t pos_idx, uint32_t slot, uint32_t epoch,
                      uint32_t attempt, uint32_t* out) {
    counter_draw(seed, stream, pos_idx, slot, epoch, attempt, out);
}

void orc_dot_topn(const int32_t* indexes, int num_queries, const float* P, int p_rows, int p_cols, const float* Q, int q_rows, int q_cols,
                  const float* Qb, int qb_rows, int32_t* out_keys, float* out_scores, const int32_t* pool, int pool_size, int k, int same) {
    dot_topn_ref(indexes, num_queries, P, p_rows, p_cols, Q, q_rows, q_cols, Qb, qb_rows, out_keys, out_scores, pool, pool_size, k, same);
}
void orc_quickselect(const float* scores, int rows, int cols, int32_t* result, int k, int sorted) {
    quickselect_ref(scores, rows, cols, result, k, sorted);
}
// _sort_and_compressed_binarization (buffalo/data/fileio.hpp:263-420) on 0-based in-memory records: stable sort by
// (major, minor) (:328-339), END-offset indptr (:359-379), minor ids and values in sorted order (:389-410).
void orc_eals_initialize_model(void* hp, float* P, float* Q, float* Cw, int P_rows, int Q_rows) { ((Handle*)hp)->eals->initialize_model(P, Q, Cw, P_rows, Q_rows); }
void orc_eals_precompute_cache(void* hp, int nnz, const int64_t* indptr, const int32_t* keys, int axis) { ((Handle*)hp)->eals->precompute_cache(nnz, indptr, keys, axis); }
int orc_eals_update(void* hp, const int64_t* indptr, const int32_t* keys, const float* vals, int axis) { return ((Handle*)hp)->eals->update(indptr, keys, vals, axis) ? 1 : 0; }
void orc_eals_estimate_loss(void* hp, int nnz, const int64_t* indptr, const int32_t* keys, const float* vals, int axis, float* out2) {
    auto r = ((Handle*)hp)->eals->estimate_loss(nnz, indptr, keys, vals, axis);
    out2[0] = r.first;
    out2[1] = r.second;
}
void orc_eals_caches(void* hp, int axis, float* vhat, int64_t* map) {
    EALS* e = ((Handle*)hp)->eals;
    std::copy(e->vhat_[axis].begin(), e->vhat_[axis].end(), vhat);
    std::copy(e->map_[axis].begin(), e->map_[axis].end(), map);
}
void orc_cfr_set_embedding(void* hp, float* data, int size, const char* type) { static_cast<CFR*>(((Handle*)hp)->als)->set_embedding(data, size, type); }
void orc_cfr_precompute(void* hp, const char* type) { static_cast<CFR*>(((Handle*)hp)->als)->precompute_cfr(type); }
double orc_cfr_partial_update_user(void* hp, int s, int n, const int64_t* indptr, const int32_t* keys, const float* vals) {
    return static_cast<CFR*>(((Handle*)hp)->als)->partial_update_user(s, n, indptr, keys, vals);
}
double orc_cfr_partial_update_item(void* hp, int s, int n, const int64_t* ipu, const int32_t* ku, const float* vu, const int64_t* ipc, const int32_t* kc,
                                   const float* vc) {
    return static_cast<CFR*>(((Handle*)hp)->als)->partial_update_item(s, n, ipu, ku, vu, ipc, kc, vc);
}
double orc_cfr_partial_update_context(void* hp, int s, int n, const int64_t* indptr, const int32_t* keys, const float* vals) {
    return static_cast<CFR*>(((Handle*)hp)->als)->partial_update_context(s, n, indptr, keys, vals);
}
// fileio.hpp:280-310 restated for a buffer in memory: the file's lines in order (std::getline: split at '\n', an unterminated last line
// counts), each through the reference's own sscanf(line, "%d %d %f"), the first `total_lines` of them.  Returns the number of lines found.
int64_t orc_parse_triples(const char* text, int64_t bytes, int64_t total_lines, int32_t* rows, int32_t* cols, float* vals) {
    int64_t k = 0, beg = 0;
    std::string line;
    while (beg < bytes) {
        const char* nlp = static_cast<const char*>(memchr(text + beg, '\n', static_cast<size_t>(bytes - beg)));
        const int64_t end = nlp ? nlp - text : bytes;
        if (k < total_lines) {
            line.assign(text + beg, text + end);
            int r = 0, c = 0;
            float v = 0.f;
            sscanf(line.c_str(), "%d %d %f", &r, &c, &v);
            rows[k] = r; cols[k] = c; vals[k] = v;
        }
        ++k;
        beg = end + 1;
    }
    return k;
}
void orc_coo_to_csr(const int32_t* major, const int32_t* minor, const float* vals, int64_t nnz, int num_major, int64_t* indptr,
                    int32_t* out_minor, float* out_vals) {
    std::vector<int64_t> order(nnz);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        if (major[a] == major[b]) return minor[a] < minor[b];
        return major[a] < major[b];
    });
    int64_t pos = 0;
    for (int m = 0; m < num_major; ++m) {
        while (pos < nnz && major[order[pos]] <= m) ++pos;
        indptr[m] = pos;
    }
    for (int64_t i = 0; i < nnz; ++i) {
        out_minor[i] = minor[order[i]];
        out_vals[i] = vals[order[i]];
    }
}

// SPPMI matrix of a stream (the context input of CoFactor / CFR).  Three reference steps, restated in memory:
//   * pair lines, buffalo/data/stream.py:257-267: for every user's (training) sequence, each item and the `windows` items
//     after it are written as the two lines "w c" and "c w" (1-based ids); sppmi_total_lines counts the lines;
//   * the lines are sorted by their first id (stream.py:173, aux.psort key=1) and _parallel_build_sppmi
//     (buffalo/data/fileio.hpp:109-254) walks the groups: appearances[id] = lines starting with id (:137-160); for a group
//     `probe` and every distinct second id c <= probe (:207-208), cnt = lines "probe c" of the group,
//     pmi = log(cnt) + log(D) - log(app[probe]) - log(app[c]) (double, in this order, :210-212), sppmi = pmi - log(k);
//     when sppmi > 0 the TEXT lines "probe c sppmi" and "c probe sppmi" are written (:215-221; c == probe gives the same
//     line twice) -- `fout << double` prints six significant digits, and that is what the next step parses.  A group is
//     written when the NEXT id's first line arrives (:204-225) and nothing flushes at end of file, so the group of the
//     LARGEST id that has lines never acts as probe: every pair with that id is absent from the reference's output
//     (measured on the compiled reference, oracle/_ref; with any number of splits and workers it is dropped exactly once);
//   * the output is sorted by its row id only (stream.py:181, aux.psort: numeric + stable on the first field), split into
//     record files (fileio.hpp:25-107: "%d %d %f", ids made 0-based) and compressed (data/base.py:354-398, END-offset indptr).
//     Inside a row the reference therefore leaves the order its writer produced -- std::unordered_set iteration order and,
//     with more than one worker, thread timing.  The oracle (like the device builder) gives every row in (col) order: the
//     same multiset per row, in the one order that is reproducible.  tests/test_oracle_ref_fileio.py compares the two row
//     by row as multisets, bit for bit.
// Returns nnz; fills the outputs only when cap >= nnz.
int64_t orc_build_sppmi(const int64_t* indptr, const int32_t* items, int num_users, int num_items, int windows, int k, int64_t cap,
                        int64_t* out_indptr, int32_t* out_key, float* out_val, int64_t* total_lines_out) {
    std::vector<std::pair<int, int>> lines;
    for (int u = 0; u < num_users; ++u) {
        const int64_t beg = u ? indptr[u - 1] : 0, sz = indptr[u] - beg;
        for (int64_t i = 0; i < sz; ++i)
            for (int64_t j = i + 1; j < i + windows + 1 && j < sz; ++j) {
                const int w = items[beg + i] + 1, c = items[beg + j] + 1;
                lines.emplace_back(w, c);
                lines.emplace_back(c, w);
            }
    }
    const int64_t total_lines = (int64_t)lines.size();
    if (total_lines_out) *total_lines_out = total_lines;
    std::stable_sort(lines.begin(), lines.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    const double log_d = std::log((double)total_lines), log_k = std::log((double)k);
    std::vector<int64_t> appearances(num_items, 0);
    for (const auto& l : lines) appearances[l.first - 1] += 1;
    std::vector<int32_t> rr, cc;
    std::vector<float> vv;
    size_t g0 = 0;
    while (g0 < lines.size()) {
        size_t g1 = g0;
        const int probe_id = lines[g0].first;
        std::vector<int> chunk;
        while (g1 < lines.size() && lines[g1].first == probe_id) chunk.push_back(lines[g1++].second);
        if (g1 == lines.size()) break;   // fileio.hpp:182-250: no flush at end of file -- the last group is never written
        std::unordered_set<int> chunk_set(chunk.begin(), chunk.end());
        for (const int _c : chunk_set) {
            if (probe_id < _c) continue;
            const int64_t cnt = std::count(chunk.begin(), chunk.end(), _c);
            const double pmi = std::log((double)cnt) + log_d - std::log((double)appearances[probe_id - 1]) - std::log((double)appearances[_c - 1]);
            const double sppmi = pmi - log_k;
            if (sppmi > 0) {
                char buf[64];
                std::snprintf(buf, sizeof(buf), "%g", sppmi);   // operator<<(double): precision 6, general format
                float v = 0.f;
                std::sscanf(buf, "%f", &v);
                rr.push_back(probe_id - 1); cc.push_back(_c - 1); vv.push_back(v);
                rr.push_back(_c - 1); cc.push_back(probe_id - 1); vv.push_back(v);
            }
        }
        g0 = g1;
    }
    const int64_t nnz = (int64_t)rr.size();
    if (cap >= nnz && nnz > 0) orc_coo_to_csr(rr.data(), cc.data(), vv.data(), nnz, num_items, out_indptr, out_key, out_val);
    else if (cap >= nnz) std::fill(out_indptr, out_indptr + num_items, (int64_t)0);
    return nnz;
}

}  // extern "C"
