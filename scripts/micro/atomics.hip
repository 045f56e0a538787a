// Micro-benchmark: throughput of 512-B row updates (d=128 fp32) on MI355X by update form and
// address distribution.  Feeds the roofline discussion of the BPR kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: atomic add (dword per lane, 2 instr/row)  1: plain load+store  2: atomic on ONE row
// 3: atomics, Zipf-like (row = n * u^3)  4: plain load + atomic (like the BPR kernel)
__global__ __launch_bounds__(256) void upd(float* Q, int n_rows, int per_wave, int mode) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int i = 0; i < per_wave; ++i) {
        unsigned h = hash32(wave * 7919u + i * 104729u + 12345u);
        unsigned row;
        if (mode == 2) row = 0;
        else if (mode == 3) { float u = (h >> 8) * (1.0f / 16777216.0f); row = (unsigned)(n_rows * u * u * u); }
        else row = h % (unsigned)n_rows;
        row = __builtin_amdgcn_readfirstlane(row);
        float* base = Q + (size_t)row * 128;
        if (mode == 1) {
            float a = base[lane], b = base[64 + lane];
            base[lane] = a + 1e-6f; base[64 + lane] = b + 1e-6f;
        } else if (mode == 4) {
            acc += base[lane] + base[64 + lane];
            unsafeAtomicAdd(base + lane, 1e-6f); unsafeAtomicAdd(base + 64 + lane, 1e-6f);
        } else {
            unsafeAtomicAdd(base + lane, 1e-6f); unsafeAtomicAdd(base + 64 + lane, 1e-6f);
        }
    }
    if (acc == 123.456f) Q[0] = acc;
}

int main() {
    const int n_rows = 27278, per_wave = 2048, waves = 256 * 32;
    float* Q; CK(hipMalloc(&Q, (size_t)n_rows * 128 * 4)); CK(hipMemset(Q, 0, (size_t)n_rows * 128 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[] = {"atomic uniform", "plain RMW uniform", "atomic ONE row", "atomic zipf-ish", "load+atomic uniform"};
    for (int mode = 0; mode < 5; ++mode) {
        int pw = mode == 2 ? 64 : per_wave;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(upd, dim3(waves / 4), dim3(256), 0, 0, Q, n_rows, pw, mode);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        double rows = (double)waves * pw;
        printf("%-22s %8.3f ms  %7.2f ns/row-update (aggregate)  %7.1f M rows/s  %6.1f GB/s payload\n", names[mode], ms, ms * 1e6 / rows,
               rows / ms / 1e3, rows * 512 / ms / 1e6);
    }
    return 0;
}
