/*
 * buffalo_hip.h -- C ABI of libbuffalo_hip.so, the MI355X (gfx950) backend for buffalo's
 * ALS / BPRMF / WARP training inner loops.
 *
 * Drop-in boundary: every object below replaces one C++ class that buffalo's Cython bindings wrap
 * (paths relative to the kakao/buffalo tree):
 *
 *   bfh_bpr_*   <->  cuda_bpr::CuBPR     include/buffalo/cuda/bpr/bpr.hpp:29-45
 *                    bound by CyBPR      buffalo/algo/cuda/_bpr.pyx:13-80
 *                    (CPU twin bpr::CBPRMF include/buffalo/algo_impl/bpr/bpr.hpp:21-58,
 *                     whose numerics this backend follows: lib/algo_impl/bpr/bpr.cc:72-188)
 *   bfh_warp_*  <->  warp::CWARP         include/buffalo/algo_impl/warp/warp.hpp:20-67
 *                    bound by CyWARP     buffalo/algo/_warp.pyx; accelerator scaffold
 *                    buffalo/algo/warp.py:212-234 (the reference has no GPU WARP)
 *   bfh_als_*   <->  cuda_als::CuALS     include/buffalo/cuda/als/als.hpp:20-35
 *                    bound by CyALS      buffalo/algo/cuda/_als.pyx:13-67
 *                    (numerics: als::CALS lib/algo_impl/als/als.cc:86-358)
 *
 * Conventions
 *   - Plain C types only. All array arguments are HOST pointers owned by the caller (numpy memory
 *     in buffalo); the backend keeps the factor pointers and writes results back into them exactly
 *     where the reference's CUDA backend does (bpr.cu:334-336, als.cu:403).
 *   - Factor matrices are C-contiguous float32 [rows, vdim], vdim = ceil(d/32)*32, pad columns zero
 *     (buffalo/algo/bpr.py:196-204, als.py:146-152).
 *   - `indptr` is int64[rows] of row END offsets without a leading zero; `keys` / `vals` hold only the
 *     current chunk, shifted by indptr[start_x-1] (buffalo/data/buffered_data.py:99-118).
 *   - Return value: BFH_OK (0) on success, a negative bfh_status on failure with the message
 *     available from bfh_last_error(handle).  `*_init` follows the reference's bool contract:
 *     1 = options parsed, 0 = file missing / invalid JSON (bpr.cu:245, _bpr.pyx:35).
 *   - Calls are synchronous (device idle on return), like the reference (bpr.cu:412,427).
 *   - One handle drives one GPU (the current HIP device at *_create, or bfh_*_set_device).
 */
#ifndef BUFFALO_HIP_H_
#define BUFFALO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    BFH_OK = 0,
    BFH_ERR_INVALID = -1,   /* bad argument / call order */
    BFH_ERR_HIP = -2,       /* a HIP runtime call or kernel failed */
    BFH_ERR_UNSUPPORTED = -3,
    BFH_ERR_NOMEM = -4
} bfh_status;

/* Kernel-level statistics, cumulative since the last bfh_*_reset_stats. */
typedef struct {
    int64_t samples;          /* (u,pos[,neg]) updates processed                                */
    int64_t scored_negatives; /* WARP: negatives actually scored (sum of T)           */
    int64_t accepted;         /* WARP: positives for which a violator was found                  */
    int64_t launches;         /* launches of the dominant kernel                                  */
    double kernel_ms;         /* HIP-event time of the dominant kernel, summed over launches      */
    double optimizer_ms;      /* HIP-event time of the epoch-end optimizer kernels                */
    double aux_ms;            /* everything else the backend launched (precompute, loss, ...)     */
    double h2d_bytes, d2h_bytes;
    int64_t merges;           /* BPRMF policy 2: reconciliations of the per-XCD item-factor replicas */
    int64_t exchanges;        /* multi-GPU: all-reduce exchange points (bfh_*_set_comm) */
    int64_t loaded_rows;      /* WARP: candidate rows fetched by the trial loop (scored + speculated), for the byte model */
    double exchange_kernel_ms; /* multi-GPU: HIP-event time of the delta / weight / apply kernels around the all-reduces */
    double allreduce_ms;       /* multi-GPU: time of the all-reduces on the stream they ran on (blocking exchanges; an exchange
                                  still in flight when the call returns is not counted) */
} bfh_stats;

const char* bfh_version(void);
/* sizeof(bfh_stats) of THIS library: the struct grows at its end between versions, a caller built against an older header can check that
 * its buffer is large enough before calling bfh_*_get_stats (which writes the whole struct). */
size_t bfh_stats_size(void);
/* Message of the last failure on `handle` (any bfh object), or of the last failed *_create when
 * handle is NULL.  Never returns NULL. */
const char* bfh_last_error(const void* handle);
int bfh_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * BPRMF   (CuBPR: include/buffalo/cuda/bpr/bpr.hpp:29-45)
 * ---------------------------------------------------------------------------------------------- */
void* bfh_bpr_create(void);                                        /* CuBPR::CuBPR      bpr.cu:181 */
void bfh_bpr_destroy(void* h);                                     /* CuBPR::~CuBPR     bpr.cu:222 */
int bfh_bpr_init(void* h, const char* opt_json_path);              /* CuBPR::init       bpr.cu:245 */
int bfh_bpr_get_vdim(void* h);                                     /* CuBPR::get_vdim   bpr.cu:346 */
/* CuBPR::initialize_model bpr.cu:286-308: set_gpu==0 only records the host pointers. */
int bfh_bpr_initialize_model(void* h, float* P, int P_rows, float* Q, float* Qb, int Q_rows,
                             int64_t num_nnz, int set_gpu);
/* CuBPR::set_placeholder bpr.cu:310-320: full rowwise indptr + max chunk nnz. */
int bfh_bpr_set_placeholder(void* h, const int64_t* indptr, size_t batch_size);
/* CuBPR::set_cumulative_table bpr.cu:322-327: int64 cumulative item counts [Q_rows]. */
int bfh_bpr_set_cumulative_table(void* h, const int64_t* table);
/* CuBPR::partial_update bpr.cu:350-430 (CyBPR.add_jobs).  keys == NULL means "use the CSR made
 * resident by bfh_bpr_set_resident_csr". Outputs: sum of log(1+e^-x) when
 * compute_loss_on_training, and the number of (u,pos,neg) samples. */
int bfh_bpr_partial_update(void* h, int start_x, int next_x, const int64_t* indptr,
                           const int32_t* keys, double* loss_sum, double* n_samples);
/* SGDAlgorithm::update_parameters lib/algo.cc:382-465 on the device (no-op for optimizer "sgd"). */
int bfh_bpr_update_parameters(void* h);
/* CuBPR::synchronize bpr.cu:330-344: device_to_host != 0 copies P,Q,Qb into the host arrays. */
int bfh_bpr_synchronize(void* h, int device_to_host);
/* CBPRMF::compute_loss bpr.cc:227-244 (same contract as CuBPR::compute_loss bpr.cu:432). */
int bfh_bpr_compute_loss(void* h, int n, const int32_t* users, const int32_t* positives,
                         const int32_t* negatives, double* loss);

/* ------------------------------------------------------------------------------------------------
 * WARP   (CWARP: include/buffalo/algo_impl/warp/warp.hpp:20-67; same surface as BPR, as
 *         buffalo/algo/warp.py:212-234 expects from an accelerator object)
 * ---------------------------------------------------------------------------------------------- */
void* bfh_warp_create(void);
void bfh_warp_destroy(void* h);
int bfh_warp_init(void* h, const char* opt_json_path);             /* CWARP::init       warp.cc:68 */
int bfh_warp_get_vdim(void* h);
int bfh_warp_initialize_model(void* h, float* P, int P_rows, float* Q, float* Qb, int Q_rows,
                              int64_t num_nnz, int set_gpu);       /* warp.cc:87-94               */
int bfh_warp_set_placeholder(void* h, const int64_t* indptr, size_t batch_size);
int bfh_warp_set_cumulative_table(void* h, const int64_t* table);  /* warp.cc:96-100 (unused)     */
/* CWARP::worker warp.cc:103-173 over one CSR chunk (CyWARP.add_jobs). loss_sum = sum(uj-ui+thr). */
int bfh_warp_partial_update(void* h, int start_x, int next_x, const int64_t* indptr,
                            const int32_t* keys, double* loss_sum, double* n_samples);
int bfh_warp_update_parameters(void* h);                           /* warp.cc:192-201             */
int bfh_warp_synchronize(void* h, int device_to_host);
int bfh_warp_compute_loss(void* h, int n, const int32_t* users, const int32_t* positives,
                          const int32_t* negatives, double* loss); /* warp.cc:205-226             */

/* ------------------------------------------------------------------------------------------------
 * ALS    (CuALS: include/buffalo/cuda/als/als.hpp:20-35)
 * ---------------------------------------------------------------------------------------------- */
void* bfh_als_create(void);                                        /* CuALS::CuALS      als.cu:124 */
void bfh_als_destroy(void* h);
int bfh_als_init(void* h, const char* opt_json_path);              /* CuALS::init       als.cu:230 */
int bfh_als_get_vdim(void* h);                                     /* als.cu:342                   */
int bfh_als_initialize_model(void* h, float* P, int P_rows, float* Q, int Q_rows); /* als.cu:271 */
int bfh_als_set_placeholder(void* h, const int64_t* lindptr, const int64_t* rindptr,
                            size_t batch_size);                    /* als.cu:292                   */
int bfh_als_precompute(void* h, int axis);                         /* als.cu:310 / als.cc:86       */
/* CuALS::partial_update als.cu:346-406 with CALS numerics (als.cc:95-358).  Updated rows
 * [start_x,next_x) are copied back into the host factor array before returning (als.cu:403).
 * keys/vals == NULL: use the CSR made resident by bfh_als_set_resident_csr(axis). */
int bfh_als_partial_update(void* h, int start_x, int next_x, const int64_t* indptr,
                           const int32_t* keys, const float* vals, int axis, double* loss_nume,
                           double* loss_deno);

/* ------------------------------------------------------------------------------------------------
 * Extensions (not in the reference): residency, determinism hooks, measurement, multi-GPU plumbing
 * ---------------------------------------------------------------------------------------------- */
/* Select the GPU for a handle; call before *_init.  Default: current HIP device at *_create. */
int bfh_bpr_set_device(void* h, int device);
int bfh_warp_set_device(void* h, int device);
int bfh_als_set_device(void* h, int device);

/* Keep the whole CSR in HBM (288 GB) instead of re-sending keys every epoch (bpr.cu:362,
 * als.cu:361-364).  indptr: int64[rows] end offsets; keys/vals: [nnz]. */
int bfh_bpr_set_resident_csr(void* h, const int64_t* indptr, const int32_t* keys, int64_t nnz);
int bfh_warp_set_resident_csr(void* h, const int64_t* indptr, const int32_t* keys, int64_t nnz);
int bfh_als_set_resident_csr(void* h, int axis, const int64_t* indptr, const int32_t* keys,
                             const float* vals, int64_t nnz);
/* With resident CSR the per-call write-back of updated rows (als.cu:403) can be deferred. */
int bfh_als_synchronize(void* h, int device_to_host);

/* Named integer knobs.  Unknown names fail with BFH_ERR_INVALID.
 *   "sequential"      1 = one wave walks the chunk in CSR order: the deterministic parity mode.
 *   "hogwild_atomic"  how the SGD (Hogwild) kernels keep the shared factor rows coherent across the 8 XCDs:
 *                       3  BPRMF sgd default: item-major walk (csrc/bpr_item_major.hpp) -- users owned by XCDs (plain
 *                          stores through the owner's L2), the positive item row in registers with bounded-staleness
 *                          atomic flushes, negatives in per-XCD replicas merged by the delta rule;
 *                       1  fp32 atomic adds on the shared item rows, user-major walk (WARP; BPRMF adam/adagrad
 *                          accumulation always uses this);
 *                       2  BPRMF: user-major walk on per-XCD replicas of the item factors, popular rows on atomics;
 *                       0  racy device-coherent write-through stores (CPU Hogwild literally; loses colliding updates).
 *   "xcd_sync_updates" updates between two merges of the per-XCD replicas (default 2^23 for 3, 2^21 for 2);
 *   "xcd_merge_mean"   1 = average instead of sum the replicas' deltas;  "xcd_hot_tau" (permille) tolerated collision
 *                      probability of a plainly stored row, above it the row is updated with atomics;
 *   "xcd_stiff_b", "xcd_stiff_q", "xcd_stiff_p"  (3, permille) curvature assumed by the merge's per-row saturation weights for the item
 *                      biases (default 250 = the logistic loss's 1/4, the multi-GPU exchange's constant), the item factor rows and the
 *                      replicated user rows (default 0 = plain sum): a row whose replicas each took m steps between two merges is merged
 *                      with w = (1 - exp(-n x)) / (n (1 - exp(-x))), x = lr k (m - 1/n) (sum for cold rows -- exactly, up to one step per row -- mean for
 *                      saturated ones);  "xcd_stiff_lr_ref" (3, units of 1e-6, default 1000 = lr 0.001): the learning rate the curvatures were
 *                      calibrated at -- above it they shrink like lr_ref / lr (at a high lr a row sits in the flat part of the sigmoid for most of
 *                      its steps; the constant at every lr, 0, left the biases 19 % low on the reference benchmark's lr 0.05 -> 0.0001 schedule);
 *   "im_user_lr_max"   (3, permille, default 10) learning rate up to which "im_user_replicas" / "im_user_hybrid" apply;
 *   "im_max_stale"     (3) updates of one item row in flight unseen by the other waves, stated at lr 0.05 (scales 1/lr; default 16);
 *   "im_p_nt", "im_neg_limit"  (3) study knobs (non-temporal hint on the P rows; uniform negatives folded into the first rows
 *                      of Q): DESIGN.md 4.1 "what bounds it" -- not for training;
 *   "im_dual"          (3) two triples per wave at vdim <= 128 (the half-waves walk two slices side by side): -1 (default) = for calls
 *                      with 1024 users per queue or more (6144 until round 6), 1 = always, 0 = never;
 *   "im_user_replicas" (3) 1 = per-XCD replicas of P (entries spread over the queues by position, delta rule at the merges)
 *                      instead of one owner XCD per user; -1 (default) = when a call has fewer than 3072 users per queue
 *                      (the shards of an 8-GPU ML-20M run) and lr <= 0.01, 0 = never;
 *   "im_user_hybrid"   (3) otherwise (whole matrices, lr <= 0.01): replicas for the HEAVY users only -- the ones "xcd_hot_tau" would put
 *                      on atomics; 2 (default) = their entries over as few neighbouring queues as needed, 1 = over all, 0 = off;
 *   "im_drift_budget"  (3, permille) lr-weighted positive steps of a row per merge interval above which its negative
 *                      updates also go to the chip-wide copy;  "im_blocks" runs an item's entries are cut into per queue (0 = ceil(160 lr));
 *   "im_presample"     (3) 1 = draw the call's negatives in CSR order before the walk;  "xcd_fresh" re-read a row right
 *                      before storing it (prefetching variants);  "im_drain_only", "im_single_wave", "im_force_queues" test hooks;
 *   "prefetch"         software prefetch of the per-triple rows (user-major: 0/1, default 1; item-major: default 0 = rows
 *                      are read where they are used, 1 = two triples ahead);
 *   "waves_per_cu", "chunk" (nnz positions per wave work item), "als_writeback" (0 = defer), "timing" (1 = record HIP
 *   events around every launch), "epoch";
 *   "als_split_f16"    (ALS, in-place iALS++ rows, d >= 64; default 1) the row Gramian through v_mfma_f32_32x32x16_f16 with every
 *                      operand cut into two round-to-nearest f16 pieces (fp32 accuracy, 5x fewer matrix-core cycles);
 *                      0 = v_mfma_f32_32x32x2_f32;  "als_split_wcut": rows holding a weight alpha*v above this (default 32768) or a
 *                      negative one go through the fp32 instruction + the dense-solve kernel (a scan of the weights, cached per
 *                      chunk, finds them);  "als_pc" (default 1) producer / consumer wave pairs for those rows where they win (d = 96, 128;
 *                      csrc/als_pc.hpp), 2 = also at d = 64, 0 = round 3's wave-per-row kernel everywhere;  "als_inreg" 0 = every row through the scratch slot + als_solve_kernel;
 *                      "als_wide_split" (default 1) / "als_wide_split_max_t" (default 7): 128 < vdim <= 32 * max_t on the split-f16 form of als_wide_kernel
 *                      (from vdim 192 up with the fourth product l l), 0 = the fp32 instruction;  "als_gram_waves" / "als_gram_upg": waves per CU and row pairs
 *                      per trip of als_gramian_kernel (0 = 4 waves at vdim 128 and 8 elsewhere / 8; the slice boundaries decide FF's last bits);  "als_debug": timing probes, results are wrong
 *                      with any bit but 1024 (the shader clock of the last als_pc_kernel launch, read back as device buffer "als_pc_clock_mhz"). */
int bfh_bpr_set_mode(void* h, const char* name, int64_t value);
int bfh_warp_set_mode(void* h, const char* name, int64_t value);
int bfh_als_set_mode(void* h, const char* name, int64_t value);

/* Multi-GPU sharding (one process per GPU, users sharded above this ABI): global position of the
 * shard's first nnz (keeps the counter-based sampler identical to the 1-GPU run) and the number of
 * shards that advance the lr schedule together. */
/* Host-only (no GPU needed): the slice schedule of the item-major sgd walk for `num_queues` (<= 8) queues holding
 * queue_entries[x] interactions each.  slice_len = triples per wave work item (the largest multiple of
 * num_negative_samples <= 64), segments = merge intervals of the call; ticket t of queue x works on slice
 * (t * queue_stride[x]) mod queue_slices[x], segment s takes the tickets [slices*s/segments, slices*(s+1)/segments).
 * What CBPRMF's job queue (algo.hpp:28-69) is to the CPU path; exported so that the schedule can be checked on its own. */
int bfh_bpr_item_major_plan(int num_queues, const int64_t* queue_entries, int num_negative_samples, int64_t sync_updates, int* slice_len,
                            int64_t* segments, int64_t* queue_slices, int64_t* queue_stride);
int bfh_bpr_set_shard(void* h, int64_t nnz_offset, int num_shards);
int bfh_warp_set_shard(void* h, int64_t nnz_offset, int num_shards);

/* Test hook: apply explicit (u,pos,neg) triples with learning rate lr (sgd) or accumulate their
 * gradients (adam/adagrad), bypassing the sampler. */
int bfh_bpr_update_triples(void* h, int64_t n, const int32_t* users, const int32_t* positives,
                           const int32_t* negatives, double lr);

/* Device pointers for collectives (RCCL through torch.distributed) and zero-copy inspection.
 * BPR/WARP names: "P" "Q" "Qb" "gradP" "gradQ" "gradQb" "countP" "countQ" "velP" "velQ" "momP" "momQ";
 * ALS names: "P" "Q" "FF".  *dptr is valid until the next initialize_model / destroy. */
int bfh_bpr_device_buffer(void* h, const char* name, void** dptr, size_t* bytes);
int bfh_warp_device_buffer(void* h, const char* name, void** dptr, size_t* bytes);
int bfh_als_device_buffer(void* h, const char* name, void** dptr, size_t* bytes);
/* hipStream_t the handle launches on (for event timing / stream ordering by the caller). */
void* bfh_bpr_stream(void* h);
void* bfh_warp_stream(void* h);
void* bfh_als_stream(void* h);

/* ---- Multi-GPU inside the library (SURVEY.md section 8(b) "_set_devices" / 8(e)): one process per GPU, one RCCL rank per
 * process.  The reference has no multi-device code; a binding drives it like this:
 *   rank 0:      bfh_comm_unique_id(id, 128)  -> hand the 128 bytes to every rank (MPI / a file / torch's store)
 *   every rank:  comm = bfh_comm_create(world, rank, id, device);  bfh_{bpr,warp,als}_set_comm(h, comm)
 * and from then on the handle exchanges by itself over RCCL / xGMI:
 *   BPRMF sgd  -- users (P rows + their CSR rows) sharded, Q / Qb replicated: at every exchange point ONE ncclAllReduce of
 *                 Q | Qb sums what the ranks changed since the state Z they agree on, and every rank advances
 *                 Z <- Z + w R with a per-row weight w between 1 (cold rows: the deltas are independent steps, SUM) and
 *                 1/world (rows every rank drove to its local equilibrium: MEAN) -- see exchange_weight_kernel.  Default: one
 *                 exchange per partial_update, finished before it returns.  "comm_segments" = k > 1 cuts a call into k
 *                 segments whose exchanges travel on the communicator's stream behind the next segment's walk, the last one
 *                 staying in flight until the next exchange point ("comm_overlap" = 0: until the call returns);
 *                 bfh_*_comm_flush / synchronize / compute_loss finish what is in flight;
 *   adam / adagrad / WARP -- update_parameters sums gradQ | gradQb (| counts) over the ranks before the (then identical)
 *                 optimizer step: exactly the single-GPU result up to summation order (lib/algo.cc:382);
 *   ALS        -- rows of the side being solved sharded, both factor matrices replicated: after partial_update on its
 *                 rows a rank calls bfh_als_publish_rows(h, axis, bounds, world + 1), the uneven all-gather of the solved
 *                 row blocks (a group of ncclBroadcast); FF is recomputed per rank from the identical replica.
 * The communicator is not owned by the handle: destroy the handles first.  bfh_comm_all_reduce_f64 sums host doubles
 * (loss sums) over the ranks. */
int bfh_comm_unique_id(char* out, size_t bytes);
void* bfh_comm_create(int n_ranks, int rank, const char* unique_id, int device);
void bfh_comm_destroy(void* comm);
int bfh_comm_rank(void* comm);
int bfh_comm_size(void* comm);        /* ranks of the LIVE communicator: ncclCommCount, not the number bfh_comm_create was asked for */
int bfh_comm_transport(void* comm, char* out, size_t bytes);   /* "rccl <major.minor.patch>" (libbuffalo_hip_test.so over shared memory: "shm-test") */
int bfh_comm_self_test(void* comm);
int bfh_comm_all_reduce_f64(void* comm, double* values, int n);
int bfh_bpr_set_comm(void* h, void* comm);
int bfh_warp_set_comm(void* h, void* comm);
int bfh_als_set_comm(void* h, void* comm);
int bfh_bpr_comm_flush(void* h);
int bfh_warp_comm_flush(void* h);
int bfh_als_publish_rows(void* h, int axis, const int* bounds, int n_bounds);

int bfh_bpr_get_stats(void* h, bfh_stats* out);
int bfh_warp_get_stats(void* h, bfh_stats* out);
int bfh_als_get_stats(void* h, bfh_stats* out);
int bfh_bpr_reset_stats(void* h);
int bfh_warp_reset_stats(void* h);
int bfh_als_reset_stats(void* h);

/* ------------------------------------------------------------------------------------------------
 * Top-k selection over factor products   (buffalo/parallel/_core.hpp)
 * The consumer of P, Q right after training: ParALS/ParBPRMF.topk_recommendation, most_similar
 * (parallel/base.py:21-28, 46-60) and the validation ranking loop (evaluate/base.py:31-42, 80-82).
 * ---------------------------------------------------------------------------------------------- */
void* bfh_topk_create(void);
void bfh_topk_destroy(void* h);
int bfh_topk_set_device(void* h, int device);
/* parallel::dot_topn _core.hpp:89-142 (Cython dot_topn _core.pyx:39-56; the num_threads argument has no
 * meaning here).  Host arrays in, host arrays out: out_keys/out_scores are [num_queries, k], C order.
 * Candidates j == indexes[i] are skipped when P == Q (same pointer), a non-empty pool restricts the
 * candidates, qb_rows == 0 means "no bias".  Admission rule, tie order and padding follow the
 * reference: only scores > FLT_MIN are ever admitted, kept set = first k by (score desc, index asc),
 * listed by (score desc, index desc); unfilled slots below min(k, q_rows[, pool_size]) read
 * (-1, FLT_MIN), the slots above it (-1, 0.0).  k <= 16384. */
int bfh_topk_dot_topn(void* h, const int32_t* indexes, int num_queries, const float* P, int p_rows, int p_cols,
                      const float* Q, int q_rows, int q_cols, const float* Qb, int qb_rows, int32_t* out_keys,
                      float* out_scores, const int32_t* pool, int pool_size, int k);
/* Same selection with the factor matrices already in HBM (e.g. the "P"/"Q"/"Qb" buffers a training
 * handle returns from bfh_*_device_buffer): no PCIe traffic except indexes/pool in and results out.
 * dP/dQ are device pointers to row-major [rows, ld] floats with ld % 8 == 0 and columns >= d zero
 * (the training layout: ld = vdim); `same` != 0 applies the P == Q self-exclusion. */
int bfh_topk_dot_topn_device(void* h, const int32_t* indexes, int num_queries, const float* dP, int p_rows, const float* dQ,
                             int q_rows, int d, int ld, const float* dQb, int qb_rows, int same, int32_t* out_keys,
                             float* out_scores, const int32_t* pool, int pool_size, int k);
/* parallel::quickselect _core.hpp:69-87 (evaluate/base.py:31-42): column indices of the k largest
 * scores of every row of the host matrix scores[rows, cols]; always returned in descending score
 * order (the reference leaves the order unspecified when sorted == 0).  Ties are broken by the
 * higher column index first (std::nth_element leaves that unspecified). */
int bfh_topk_quickselect(void* h, const float* scores, int rows, int cols, int32_t* result, int k, int sorted);
/* "flt_min_rule" (default 1): 0 admits every score in bfh_topk_dot_topn[_device] -- the selection the
 * validation loop gets from numpy scores + quickselect (algo/base.py:40-55 of the reference).
 * "fused" (default -1): -1 = sweeps of 8192 queries x 8192 candidates or more at d <= 128 take the fused path (thresholds
 * from a column sample, filtered score sweep that writes only candidates, wave-per-row selection; results identical to the
 * dense path), 0 = always the dense score buffer + select, 1 = fused whenever d <= 128; "fused_c0" (with "fused" = 1): the
 * exact number of sampled columns (tests); "wave_select" (default 1): 0 = block-per-row selection inside the fused path;
 * "fast_select" (default 1): 0 = the dense select's multi-pass radix path only.
 * bfh_topk_get_stats: kernel_ms = score kernels, aux_ms = everything else; merges = rows the fused path handed back to the
 * dense path, exchanges = rows whose ties at the k-th place went to the block-level list selection. */
int bfh_topk_set_mode(void* h, const char* name, int64_t value);
int bfh_topk_get_stats(void* h, bfh_stats* out);
int bfh_topk_reset_stats(void* h);

/* ------------------------------------------------------------------------------------------------
 * COO -> compressed rows on the device   (buffalo/data/fileio.hpp:263-420)
 * The step right before the training path: _sort_and_compressed_binarization, run once per orientation.
 * Records are stable-sorted by (major, minor) -- duplicates are kept, in input order -- and come back as
 * indptr[num_major] (END offsets: records with major id <= k, no leading zero), out_minor[nnz],
 * out_vals[nnz].  Ids are 0-based here (the reference parses 1-based text and subtracts 1, :396,:401).
 * Stateless; failures are reported through bfh_last_error(NULL).  `stats` may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int bfh_coo_to_csr(const int32_t* major, const int32_t* minor, const float* vals, int64_t nnz, int num_major,
                   int num_minor, int64_t* indptr, int32_t* out_minor, float* out_vals, bfh_stats* stats);

/* The working TEXT file of buffalo's data creation parsed on the device (buffalo/data/fileio.hpp:263-330; the file is what
 * buffalo/data/mm.py:175-234 writes: the MatrixMarket body, one "row col val" line per entry, 1-based ids).
 * bfh_parse_triples: the first `total_lines` lines as the reference's sscanf(line, "%d %d %f") reads them (ids stay 1-based) --
 * bit-identical: what the device cannot guarantee to round like strtof is re-parsed on the host with that very call
 * (stats->merges = number of such lines, 0 on ordinary rating files).
 * bfh_text_to_csr: the same followed by _sort_and_compressed_binarization (:328-420) on the device; sort_key 1 = rowwise
 * (major = row), 2 = colwise (major = col); outputs as bfh_coo_to_csr (0-based minors, END offsets). */
int bfh_parse_triples(const char* text, int64_t bytes, int64_t total_lines, int32_t* rows, int32_t* cols, float* vals, bfh_stats* stats);
int bfh_text_to_csr(const char* text, int64_t bytes, int64_t total_lines, int num_major, int num_minor, int sort_key,
                    int64_t* indptr, int32_t* out_minor, float* out_vals, bfh_stats* stats);

/* ------------------------------------------------------------------------------------------------
 * CFR / CoFactor   (CCFR: include/buffalo/algo_impl/cfr/cfr.hpp:19-45, lib/algo_impl/cfr/cfr.cc;
 * SURVEY.md section 8(f) rank 4) -- the three row updates run on the ALS Gramian / dense-solve kernels.
 * Arrays are the reference's CPU layout: C-contiguous float32 [rows, d] (NOT padded), biases [rows, 1];
 * updated rows are written back into them before a call returns.  d <= 128; optimizers llt, ldlt, manual_cg.
 * ---------------------------------------------------------------------------------------------- */
void* bfh_cfr_create(void);                                                   /* CCFR::CCFR           cfr.cc:14  */
void bfh_cfr_destroy(void* h);                                                /* CCFR::~CCFR          cfr.cc:18  */
int bfh_cfr_set_device(void* h, int device);
int bfh_cfr_init(void* h, const char* opt_json_path);                         /* CCFR::init           cfr.cc:29  */
/* CCFR::set_embedding cfr.cc:70-82: obj_type in user | item | context | item_bias | context_bias */
int bfh_cfr_set_embedding(void* h, float* data, int size, const char* obj_type);
int bfh_cfr_precompute(void* h, const char* obj_type);                        /* CCFR::precompute     cfr.cc:85  */
/* CCFR::partial_update_{user,item,context} cfr.cc:92-313: full END-offset indptr + the chunk's keys/vals; *loss
 * receives the value the reference returns (0 unless the option "compute_loss" is set). */
int bfh_cfr_partial_update_user(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys,
                                const float* vals, double* loss);
int bfh_cfr_partial_update_item(void* h, int start_x, int next_x, const int64_t* indptr_u, const int32_t* keys_u,
                                const float* vals_u, const int64_t* indptr_c, const int32_t* keys_c, const float* vals_c,
                                double* loss);
int bfh_cfr_partial_update_context(void* h, int start_x, int next_x, const int64_t* indptr, const int32_t* keys,
                                   const float* vals, double* loss);
int bfh_cfr_get_stats(void* h, bfh_stats* out);
int bfh_cfr_reset_stats(void* h);

/* ------------------------------------------------------------------------------------------------
 * eALS   (CEALS: include/buffalo/algo_impl/eals/eals.hpp:20-84, lib/algo_impl/eals/eals.cc)
 * Element-wise ALS with a prediction cache in both orientations.  Whole-matrix calls: indptr are END offsets
 * of the full matrix, P [P_rows, d] / Q [Q_rows, d] unpadded, C [Q_rows] the item weights c_i.  The structure
 * handed to precompute_cache stays bound; update / estimate_loss take the same arrays again (only vals are read).
 * ---------------------------------------------------------------------------------------------- */
void* bfh_eals_create(void);                                                       /* CEALS::CEALS            eals.cc:6   */
void bfh_eals_destroy(void* h);
int bfh_eals_set_device(void* h, int device);
int bfh_eals_init(void* h, const char* opt_json_path);                             /* CEALS::init             eals.cc:19  */
int bfh_eals_initialize_model(void* h, float* P, float* Q, float* C, int P_rows, int Q_rows);   /* eals.cc:33 */
/* CEALS::precompute_cache eals.cc:49-100: vhat of every observed entry + the index map to the other orientation */
int bfh_eals_precompute_cache(void* h, int nnz, const int64_t* indptr, const int32_t* keys, int axis);
/* CEALS::update eals.cc:102-115 -> 1, or 0 while a cache is missing; the updated side is copied back into P / Q */
int bfh_eals_update(void* h, const int64_t* indptr, const int32_t* keys, const float* vals, int axis);
/* CEALS::estimate_loss eals.cc:117-174 -> (rmse, loss); both 0 while a cache is missing */
int bfh_eals_estimate_loss(void* h, int nnz, const int64_t* indptr, const int32_t* keys, const float* vals, int axis,
                           float* rmse, float* loss);
int bfh_eals_get_stats(void* h, bfh_stats* out);
int bfh_eals_reset_stats(void* h);

/* ------------------------------------------------------------------------------------------------
 * SPPMI matrix of a stream   (CoFactor's context input)
 * Replaces, in HBM and without text files: the pair lines of buffalo/data/stream.py:257-267 (every event with the
 * `windows` events after it in its user's sequence, both orientations), _parallel_build_sppmi
 * (buffalo/data/fileio.hpp:109-254: appearances, pmi = log(cnt) + log(D) - log(app[probe]) - log(app[c]), shift
 * log(k), entries with sppmi > 0, values carried with the six significant digits of the reference's text output;
 * the group of the largest id that has lines is never written by the reference -- it flushes a group when the next id
 * begins and not at end of file, :182-250 -- and is left out here as well) and the sort + compression of its output
 * (stream.py:169-195).
 * bfh_sppmi_build: `indptr` are END offsets [num_users] over the 0-based `items` of the stream; reports the number of
 * entries and D (sppmi_total_lines).  bfh_sppmi_fetch copies the group out: indptr_out[num_items] (END offsets),
 * keys_out[nnz], vals_out[nnz], rows ascending, columns ascending inside a row, a pair (p, p) listed twice as the
 * reference lists it.  (The reference sorts its output by row only, so inside a row it keeps std::unordered_set
 * iteration order; per row the entries here are the same multiset, bit for bit -- tests/test_oracle_ref_fileio.py.)
 * ---------------------------------------------------------------------------------------------- */
void* bfh_sppmi_create(void);
void bfh_sppmi_destroy(void* h);
int bfh_sppmi_build(void* h, const int64_t* indptr, const int32_t* items, int num_users, int num_items, int windows, int k,
                    int64_t* nnz, int64_t* total_lines);
int bfh_sppmi_fetch(void* h, int64_t* indptr_out, int32_t* keys_out, float* vals_out);
int bfh_sppmi_get_stats(void* h, bfh_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* BUFFALO_HIP_H_ */
