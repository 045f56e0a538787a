// TEST INFRASTRUCTURE -- not product code.  A C-callable door to three functions of the REFERENCE's own data-ingestion
// code, compiled from the source where it lies (/root/reference/buffalo/data/fileio.hpp: standard library + OpenMP + GNU
// parallel mode, no Eigen / json11 / spdlog), into oracle/_ref/ (git-ignored).  Nothing of the reference is copied into
// this repository: the header is only #included at build time (oracle/Makefile, target _ref; built by
// __graft_entry__.build() and by the tests when /root/reference is present).  The tests use it to pin the oracle's
// restatements of COO -> CSR (orc_coo_to_csr) and of the SPPMI builder (orc_build_sppmi) against the real thing.
#include <cassert>   // fileio.hpp:325 uses assert and :103 unlink without including their headers (Cython pulls them in)
#include <unistd.h>
#include "buffalo/data/fileio.hpp"

extern "C" {

// fileio.hpp:109-254.  `from`: text lines "a b" (1-based ids) grouped by their first id; `to`: text lines "r c sppmi".
long long ref_parallel_build_sppmi(const char* from, const char* to, long long total_lines, int num_items, int k, int workers) {
    return static_cast<long long>(fileio::_parallel_build_sppmi(from, to, total_lines, num_items, k, workers));
}

// fileio.hpp:263-420.  `path`: text lines "r c v" (1-based); writes <to_dir>/indptr.bin (int64 END offsets) and
// <to_dir>/chunk<i>.bin (interleaved int32 minor id (0-based), float32 value); returns the number of files written.
int ref_sort_and_compressed_binarization(const char* path, const char* to_dir, long long total_lines, int max_key, int sort_key, int workers) {
    return static_cast<int>(fileio::_sort_and_compressed_binarization(path, to_dir, total_lines, max_key, sort_key, workers).size());
}

// fileio.hpp:25-107.  Splits the (sorted) text file into `num_chunks` binary record files <to_dir>/chunk<i>.bin (int32 r, int32 c,
// float32 v, ids made 0-based); returns the number of files.
int ref_chunking_into_bins(const char* path, const char* to_dir, long long total_lines, int num_chunks, int sep_idx, int workers) {
    return static_cast<int>(fileio::_chunking_into_bins(path, to_dir, total_lines, num_chunks, sep_idx, workers).size());
}

}  // extern "C"
