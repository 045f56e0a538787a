// One RCCL rank behind the C ABI (bfh_comm_*): the multi-GPU plumbing of SURVEY.md section 8(e) inside the library.
// One process per GPU; collectives run over xGMI on the stream the caller names.  librccl is opened at run time
// (dlopen by SONAME, so a process that already loaded RCCL -- e.g. through torch -- shares that copy); a single-GPU
// user never loads it.
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
};

}  // namespace bfh
