// One RCCL rank behind the C ABI (bfh_comm_*): the multi-GPU plumbing of SURVEY.md section 8(e) inside the library.
// One process per GPU; collectives run over xGMI on the stream the caller names.  librccl is opened at run time
// (dlopen by SONAME, so a process that already loaded RCCL -- e.g. through torch -- shares that copy); a single-GPU
// user never loads it.
//
// TEST TRANSPORT (csrc/comm_test_transport.hpp, compiled ONLY into libbuffalo_hip_test.so with -DBFH_TEST_TRANSPORT since round 5; there the
// environment knob BFH_COMM_TRANSPORT=shm is read when the id is made and when the communicator is built; the product library refuses it): RCCL
// refuses two ranks on one device, so on a one-GPU box the exchange code of the handles could only ever see a world of one.
// With the knob the same Comm interface runs over a POSIX shared-memory segment: every collective waits for its stream,
// stages the operands through host slots (one per rank), sums them IN RANK ORDER -- the same arithmetic on every rank --
// and copies the result back.  Host-blocking and slow by construction: it exists so that N processes sharing one GPU
// execute exchange_begin / exchange_finish / exchange_gradients / publish_rows with N > 1 under `pytest -m gpu`.
#pragma once
#include "common.hpp"

namespace bfh {

class Comm : public HandleBase {
 public:
    static void unique_id(char* out128);                      // ncclGetUniqueId (rank 0; hand the 128 bytes to every rank)
    Comm(int n_ranks, int rank, const char* id128, int device);
    ~Comm() override;
    int rank() const { return rank_; }
    int size() const { return size_; }
    int live_size() const;                                    // ranks the LIVE communicator reports (ncclCommCount; the test transport: as attached)
    std::string transport() const;                            // "rccl <version>" | "shm-test"
    // sum all-reduce; `send == recv` is in place
    void all_reduce_f32(const float* send, float* recv, size_t count, hipStream_t s);
    void all_reduce_i32(const int* send, int* recv, size_t count, hipStream_t s);
    void all_reduce_f64(const double* send, double* recv, size_t count, hipStream_t s);
    void broadcast_bytes(void* buf, size_t bytes, int root, hipStream_t s);
    void group_start();
    void group_end();
    // the communicator's own stream: collectives that overlap the owner's compute stream run here
    hipStream_t comm_stream() const { return stream; }

 private:
    void* comm_ = nullptr;
    int rank_ = 0, size_ = 1;
    // shm test transport
    struct Shm;
    Shm* shm_ = nullptr;
    template <typename T> void shm_all_reduce(const T* send, T* recv, size_t count, hipStream_t s);
    bool shm_attach(int n_ranks, const char* id128);          // true: `id128` names a test-transport segment and this rank is attached to it
    void shm_broadcast(void* buf, size_t bytes, int root, hipStream_t s);
};

}  // namespace bfh
