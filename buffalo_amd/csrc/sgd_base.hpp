// Device-resident state shared by the BPRMF and WARP backends: the GPU counterpart of
// SGDAlgorithm (/root/reference/include/buffalo/algo.hpp:93-148, lib/algo.cc:133-492) behind the
// accelerator surface of CuBPR (/root/reference/include/buffalo/cuda/bpr/bpr.hpp:29-45).
#pragma once
#include "common.hpp"

namespace bfh {

class Comm;

struct SgdParams {  // kernel-visible constants
    float* P;
    float* Q;
    float* Qb;
    float* gradP;
    float* gradQ;
    float* gradQb;
    int* cntP;
    int* cntQ;
    const int64_t* indptr;    // [P_rows] end offsets (full matrix)
    const int32_t* keys;      // chunk keys (chunk-local index)
    const int32_t* rows;      // chunk row ids (chunk-local index)
    const int64_t* cum_table; // [Q_rows] or null
    int64_t chunk_nnz;        // nnz in this chunk
    int64_t shift;            // global position of the chunk's first nnz (within this shard)
    int64_t nnz_offset;       // global position of the shard's first nnz
    int P_rows, Q_rows, d, vdim;
    uint32_t seed, epoch;
};

// ------------------------------------------------------------------------------------------------
// Gradient accumulation without per-triple atomics (adam / adagrad BPRMF: bpr.cc:138-156; WARP:
// warp.cc:151-165).  P and Q are frozen inside an epoch, so the three gradient rows of a triple are
// plain sums and any grouping of the terms is the reference's result up to fp32 summation order:
//   pass 1 (user-major, the update kernels): score, coefficient c_t (sigmoid-table logit / WARP's Phi) and
//          the negative of every triple -> two 4-byte arrays; gradP[u] is summed in registers over the
//          user's run (owner computes);
//   pass 2 (`grad_gather_kernel`): the triples seen from the item side.  The incidences (item, triple)
//          are sorted by item -- positives once per resident matrix (that is the CSC order), negatives
//          with one radix sort per call -- and a wave sums  sign * c_t * P[u_t]  over a run of one item's
//          incidences in registers: gradQ receives ONE atomic row add per run instead of one per triple.
// The item-side row term of WARP ( -reg * q, and -/+ Phi * q for the L2 score, warp.cc:42-52) factors
// out of the sum:  (a * sum c + b * n) * q_item, added at the flush.
// ------------------------------------------------------------------------------------------------
struct GatherParams {
    const uint32_t* inc_key;   // [n] item of every incidence, ascending; entries >= Q_rows (rejected positives) are ignored
    const int32_t* inc_idx;    // [n] pos_list: chunk-local nnz position; else chunk-local triple index (position * num_neg + slot)
    int64_t n;
    // uc[t] = (user as int bits, coefficient) of triple t, written by pass 1 -- ONE 8-byte gather per incidence carries everything
    // the gather needs (rounds 1-2 read a row id, a coefficient and WARP's accept flag: three 4-byte gathers, a sector each, a
    // quarter of the gathers' counter traffic at configs[4] size).  A negative coefficient = the positive was rejected (WARP).
    const float2* uc;          // [chunk nnz * num_neg]
    const float* P;
    const float* Q;            // read only when a != 0 or b != 0
    float* gradQ;
    float* gradQb;             // or null
    int* cntQ;                 // or null (per_coordinate_normalize)
    int num_neg, vdim, Q_rows;
    int pos_list;              // 1: one incidence per nnz position, coefficient = sum over its slots
    float sign, a, b;          // gradQ[item] += sign * sum(c p_u) + (a * sum(c) + b * n) * q_item
};
void launch_grad_gather(const GatherParams& g, int64_t max_waves, hipStream_t s);
// key[t] = keys[t], idx[t] = t   (the unsorted positive incidence list of a chunk)
void launch_incidence_iota(const int32_t* keys, int64_t n, uint32_t* key_out, int32_t* idx_out, hipStream_t s);

class SgdHandle : public HandleBase {
 public:
    explicit SgdHandle(int kind) : kind_(kind) {}
    ~SgdHandle() override;

    // ---- reference surface -------------------------------------------------------------------
    bool init(const char* opt_path);
    int get_vdim() const { return vdim_; }
    void initialize_model(float* P, int P_rows, float* Q, float* Qb, int Q_rows, int64_t num_nnz, bool set_gpu);
    void set_placeholder(const int64_t* indptr, size_t batch_size);
    void set_cumulative_table(const int64_t* table);
    void synchronize(bool device_to_host, bool force = false);
    void update_parameters();
    // ---- extensions --------------------------------------------------------------------------
    void set_resident_csr(const int64_t* indptr, const int32_t* keys, int64_t nnz);
    void set_mode(const std::string& name, int64_t v);
    void device_buffer(const std::string& name, void** p, size_t* bytes);
    void sync_stream() { BFH_HIP(hipStreamSynchronize(stream)); }
    void harvest_timers();
    // ---- multi-GPU (one process per GPU, users sharded, item factors replicated) -------------
    // Hogwild sgd: local SGD on the rank's replica of Q / Qb; what the rank changed since the state Z all ranks agree on is
    // summed over the ranks with ONE all-reduce per exchange point.  The exchange is pipelined one deep on the communicator's
    // stream: `exchange_begin` publishes S = Q - Z and returns, the next walk runs on the local replica, and
    // `exchange_finish` (at the next exchange point, or before anything reads Q) advances Z <- Z + R (R = summed deltas: the
    // same arithmetic on every rank, Z stays bit-identical) and folds in what the OTHER ranks sent, Q += R - S -- or, when the
    // rank has not worked since `begin` (a flush), Q <- Z exactly, so flushed replicas are bit-identical.  Every delta is
    // applied exactly once everywhere.
    // adam / adagrad / WARP: the gradient deltas (gradQ, gradQb, counts) are summed once per update_parameters, before
    // the (then identical) optimizer step -- exactly the single-GPU result up to summation order (algo.cc:382).
    void set_comm(Comm* c);
    void exchange_arm();         // first exchange point of a model: capture the state every rank starts from
    void exchange_begin();
    void exchange_finish(bool progressed = false);   // progressed: this rank changed Q since exchange_begin
    void exchange_gradients();   // synchronous; called by update_parameters
    void exchange_histogram(const int32_t* keys, int64_t n);
    void exchange_weights(double interval_triples, double lr, int num_neg, bool uniform);

 protected:
    // stage the chunk [start_x,next_x) (keys + row ids) and fill `pr`; returns chunk nnz
    int64_t stage_chunk(int start_x, int next_x, const int64_t* indptr, const int32_t* keys, SgdParams* pr);
    double current_lr() const;  // deterministic per-call schedule (see DESIGN.md "learning rate")
    void advance_progress(int start_x, int next_x, const int64_t* indptr_host);
    virtual void parse_specific() = 0;
    virtual bool project_unit_ball() const { return false; }
    // two-pass accumulation: buffers of pass 1 for a chunk of `triples` triples; the item-sorted positive incidence
    // list of the staged chunk (cached for a resident matrix); the item-side gather over both lists
    void acc_prepare(int64_t triples);
    void acc_build_positive_list(const SgdParams& p, int start_x, int next_x);
    void acc_gather(const SgdParams& p, int num_neg, bool do_pos, bool do_neg, const float sab_pos[3], const float sab_neg[3], bool with_bias);

 public:
    int kind_;  // 0 bpr, 1 warp
    Options opt_;
    bool inited_ = false, model_on_gpu_ = false, placeholder_set_ = false;
    int d_ = 0, vdim_ = 0, P_rows_ = 0, Q_rows_ = 0;
    int64_t num_nnz_ = 0;
    int num_iters_ = 0;
    std::string optimizer_;
    bool use_bias_ = false, update_i_ = true, update_j_ = true, pcn_ = false, compute_loss_ = false;
    float reg_u_ = 0, reg_i_ = 0, reg_j_ = 0, reg_b_ = 0;
    double reg_b_d_ = 0;   // reg_b as the reference holds it (a double, bpr.cc:81)
    double lr_ = 0, min_lr_ = 0, beta1_ = 0;
    uint32_t seed_ = 0, epoch_ = 0;
    int iters_ = 0;
    double processed_ = 0;  // sum of job sizes so far (Q-8)
    int64_t nnz_offset_ = 0;
    int num_shards_ = 1;

    // knobs
    int sequential_ = 0, hogwild_atomic_ = 1, prefetch_ = -1, waves_per_cu_ = 0, chunk_ = 256;  // prefetch -1: the kernel's default
    // policy 2 (BPRMF sgd): updates between two merges of the per-XCD item-factor replicas, and
    // whether the merge sums (0) or averages (1) the replicas' deltas
    int64_t xcd_sync_updates_ = -1;   // default: 2^21 (policy 2), 2^23 (policy 3)
    int xcd_merge_mean_ = 0;
    // policy 3: curvature (permille) the merge's per-row saturation weights assume for Q / Qb / the replicated P rows (xcd_item_weight_kernel); 0 = plain sum.
    // Biases: 250 = the logistic loss's own curvature bound (1/4), the constant the multi-GPU exchange uses (comm_stiffness_milli_).  Measured at
    // BASELINE scale against the threaded oracle pair (profiles/r04_bpr_merge_weights_study.txt): at the reference's lr |Qb| 93.04 -> 91.93 (oracles
    // 90.21 / 92.06), sampled loss 0.2075 -> 0.2081 (oracles 0.2075 / 0.2084), kernel time unchanged; the factor rows sit far below saturation there
    // (a weight on Q moves |Q| AWAY from the oracles, one on the replicated P rows changes nothing), and at lr 0.05 the drift rule has the item
    // rows chip-wide, so no weight reaches them.
    int xcd_stiff_q_milli_ = 0, xcd_stiff_b_milli_ = 250, xcd_stiff_p_milli_ = 0;
    // the learning rate (in units of 1e-6) the stiffness constants were calibrated at: above it they shrink like lr_ref / lr, i.e. the saturation
    // argument x = lr k m stops growing with the learning rate (0: the constants apply at every lr -- the form up to round 5, which damped the
    // negatives' bias steps to 0.36 of their sum at lr 0.035 and left |Qb| 19 % low on the reference benchmark's schedule: profiles/r06_bpr_lr005_*)
    // Measured (profiles/r06_bpr_lr005_stiffness_rule.txt): refbench |Qb| 147.7 (off) / 159.6 (5000) / 171.6 (2000) / 176.1 (1000) / 180.8 (no weight at all) against 183.0;
    // the bench case (lr 0.002) 91.93 / 91.93 / 91.93 / 92.34 / 93.04 against oracle pairs 90.2 .. 92.8 on four boxes -- 1000 sits inside every pair's band.
    int xcd_stiff_lr_ref_micro_ = 1000;
    // learning rate (permille) up to which users get per-XCD replicas.  Above it (study at lr 0.05): plain sums end at |Qb| 109 against the oracles'
    // 91.6; with xcd_stiff_p = 50 |P| 424 / |Qb| 95.5 against 428.7 / 94.8 for the owner form (oracle 430.6 / 91.6) for 6 % of the walk -- not taken.
    int im_user_lr_max_milli_ = 10;
    DevBuf<float> xcd_wq_, xcd_wb_, xcd_wp_;
    int im_single_wave_ = 0, im_force_queues_ = 0;   // test hooks: one wave drains all queues in order; number of queues for that run
    int im_drain_only_ = 0;        // test hook: skip the owner-XCD launch, the atomic drain launch does everything
    DevBuf<int32_t> im_trace_;     // test hook ("im_trace" = capacity): table index of every triple of a single-wave call
    int im_drift_budget_milli_ = 1000;  // policy 3: lr-weighted positive steps of a row per merge interval above which its negatives go chip-wide
    int im_presample_ = 1;         // policy 3: draw the call's negatives in CSR order before the walk
    int im_presample_ahead_ = 1;   // ... and the next epoch's on a side stream while this epoch's walk runs
    int im_blocks_ = 0;            // policy 3: runs an item's entries are cut into inside a queue (0 = from the learning rate)
    bool im_dual_call_ = false;    // decided per call (im_choose_dual)
    int im_dual_ = -1;             // policy 3: two triples per wave at vdim <= 128 (bpr_item_major_dual_kernel); -1: from 1024 users per queue up (6144 until round 6), 1: always, 0: never
    int im_neg_limit_ = 0;         // policy 3 study knob: fold the uniform negatives into the first rows of Q
    bool im_dual_generic_ = false; // "im_dual_generic": the two-triples walk with per-lane guards at every vdim (A/B against the whole-group instantiations)
    int im_study_ = 0;             // policy 3 measurement knob (never set by a front): bit 0 drops the chip-wide atomics of the negatives' rows (profiles/r06_bpr_lr005_*)
    int im_p_nt_ = 0;              // policy 3 study knob: non-temporal hint on the per-triple P rows
    int im_user_replicas_ = -1;    // policy 3: per-XCD replicas of P instead of one owner XCD per user (-1: for small shards, 0 / 1)
    int im_user_hybrid_ = 2;       // policy 3, whole matrices: per-XCD replicas of P for the HEAVY users only (the ones the collision rule would put on
                                   // atomics): 1 = their entries over all queues, 2 = over as few neighbouring queues as bring the share under the threshold
    int im_built_spread_mode_ = 0; // what the cached item-major keys were built with: 0 owner queues, 1 every entry spread, 2 heavy users' entries spread
    int64_t im_built_heavy_deg_ = 0;
    int im_max_stale_ = 16;        // policy 3: updates of one item row that may be in flight unseen by the other waves (at lr 0.05; x 0.05 / lr)
                                   // 16: |P| within 0.1 % of the threaded oracle's after 24 epochs at lr 0.05 (64: -1.1 %), no cost at lr 0.002
    int xcd_fresh_ = -1, xcd_v4_ = 0;  // re-read before store; float4-per-lane rows (hot-row atomics then cost 4x the line operations)
    int xcd_hot_tau_ = 100;        // permille: tolerated collision probability of a replica row (0 = no hot rows)
    // the reference's call pattern hands the chunk's keys over on EVERY call (cuda/_bpr.pyx:60-74) and copies the model back
    // after every epoch.  auto_resident: a chunk seen before -- same row range, same length, same 64-bit hash over the whole host
    // buffer -- is served from its copy in HBM (and keeps its item-major regrouping); lazy_sync: synchronize(device_to_host)
    // only marks the host arrays stale, the copy happens on synchronize(2) / destroy (default off = the reference behaviour);
    // pin_host: the caller's factor arrays are page-locked for the duration of the model (D2H at PCIe rate)
    int auto_resident_ = 1, lazy_sync_ = 0, pin_host_ = 0;   // pin_host: opt-in since round 5 (the default copies back through HostStager)
    HostStager stager_;
    bool host_stale_ = false;
    struct ChunkSig { int64_t n; uint64_t sig; };
    std::map<std::pair<int, int>, ChunkSig> chunks_;
    std::vector<std::pair<void*, size_t>> pinned_;
    void unpin_host();
    int gather_waves_per_cu_ = 0;  // grad_gather_kernel grid (0: 32 waves per CU)
    int accum_two_pass_ = 1;       // adam / adagrad / WARP: item-side gradients by the sorted gather (0: one atomic row add per triple)
    int64_t csr_generation_ = 0;   // bumped by set_resident_csr
    bool chunk_set_ = false;

    float *hostP_ = nullptr, *hostQ_ = nullptr, *hostQb_ = nullptr;
    DevBuf<float> P_, Q_, Qb_, gradP_, gradQ_, gradQb_, momP_, momQ_, momQb_, velP_, velQ_, velQb_;
    DevBuf<int> cntP_, cntQ_;
    DevBuf<int64_t> indptr_, cum_;
    std::vector<int64_t> indptr_host_;
    DevBuf<int32_t> keys_, rows_;
    DevBuf<double> scratch_;  // loss / counters returned by kernels
    bool resident_ = false;
    int64_t resident_nnz_ = 0;
    bool have_cum_ = false;
    int64_t cum_total_ = 0;
    int num_cus_ = 256;

    // two-pass accumulation
    DevBuf<float2> acc_uc_;                  // [triples] (user, coefficient) of pass 1
    DevBuf<uint32_t> acc_neg_;               // [triples] pass-1 negative (Q_rows: none)
    DevBuf<uint32_t> acc_key_a_, acc_key_b_; // sort input / output keys (negatives)
    DevBuf<int32_t> acc_iota_, acc_idx_b_;   // 0..n-1, sorted triple indices
    DevBuf<uint32_t> acc_pkey_;              // positives sorted by item
    DevBuf<int32_t> acc_pidx_;
    DevBuf<char> acc_tmp_;
    int64_t acc_iota_n_ = 0, acc_pos_gen_ = -1, acc_pos_n_ = -1;
    int acc_pos_start_ = -1, acc_pos_next_ = -1;

    Comm* comm_ = nullptr;          // not owned
    int comm_overlap_ = 1;          // leave the last exchange of a call in flight (finished by the next exchange point / reader)
    int comm_segments_ = 0;         // exchange segments per partial_update call (0 = 1: one blocking exchange; k > 1: pipelined)
    bool comm_blocking_call_ = false;   // this call is one segment: its exchange is finished before it returns
    int64_t comm_forced_segments_ = 1;  // exchange segments of the current call (identical on every rank)
    bool x_inited_ = false, x_pending_ = false;
    DevBuf<float> xZ_, xS_, xR_;    // [x_count()]: state at the last exchange, own delta, summed deltas
    DevBuf<float> xW_, xWb_;        // [Q_rows] combination weight of every factor row / bias for the exchange in flight (sum .. mean)
    DevBuf<int> x_gcnt_;            // [Q_rows] positives per item over all ranks
    double x_gcnt_total_ = 0;
    bool x_gcnt_ready_ = false;
    int comm_stiffness_milli_ = 250;   // curvature the saturation model assumes for the biases (permille); 0: plain sum
    int comm_stiffness_q_milli_ = 25;  // ... and for the factor rows (the reference's default regulariser)
    hipEvent_t x_ready_ = nullptr, x_done_ = nullptr;
    // Q | Qb (padded to 4) | 4 scalars that travel with the delta: [0] this rank's triples inside the interval, [1] its lr --
    // their sums come back with R, so the combination weights are computed from the SAME numbers on every rank
    size_t x_scalars() const { return static_cast<size_t>(Q_rows_) * vdim_ + ((static_cast<size_t>(Q_rows_) + 3) / 4) * 4; }
    size_t x_count() const { return x_scalars() + 4; }
    double x_w_interval_ = 0, x_w_lr_ = 0;   // what exchange_weights announced for the exchange that begins next
    int x_w_num_neg_ = 1;
    bool x_w_uniform_ = true;
    int x_p_num_neg_ = 1;           // ... of the exchange in flight (snapshot taken by exchange_begin)
    bool x_p_uniform_ = true;

    EventTimer t_main_, t_opt_, t_aux_, t_xk_, t_ar_;   // dominant kernel, optimizer, everything else, exchange kernels, all-reduces
};

// kernels implemented in sgd_base.hip
void launch_fill_rows(const int64_t* indptr, int start_x, int next_x, int64_t shift, int64_t n, int32_t* rows, hipStream_t s);

}  // namespace bfh
