// Micro-benchmark behind the BPR lr cliff: what does the READ of a chip-wide 512-B row cost next to the fp32 row atomic that follows it?
// Uniform random rows of a 27,278 x 128 fp32 matrix, 8,192 waves, every wave 2,048 rows (as scripts/micro/atomics.hip, whose "load+atomic" line is
// mode 1 here).  Forms of the read: plain (L2-cached), sc1 (agent-scope relaxed atomic load: what hrow_load issues), nt (non-temporal),
// and the returning atomic as a read-and-add in one instruction.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/row_read_forms.hip -o scripts/micro/row_read_forms.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_nt(const float* p) { return __builtin_nontemporal_load(p); }

// read: 0 none, 1 plain, 2 sc1, 3 nt;  write: 0 none, 1 atomic (no return), 2 atomic with return (its result is the read), 3 plain store
// other: the atomic goes to ANOTHER random row than the one read
template <int READ, int WRITE, bool OTHER>
__global__ __launch_bounds__(256) void upd(float* Q, int n_rows, int per_wave) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int i = 0; i < per_wave; ++i) {
        const unsigned h = hash32(wave * 7919u + i * 104729u + 12345u);
        const unsigned row = __builtin_amdgcn_readfirstlane(h % (unsigned)n_rows);
        const unsigned row2 = OTHER ? __builtin_amdgcn_readfirstlane(hash32(h + 77u) % (unsigned)n_rows) : row;
        float* base = Q + (size_t)row * 128;
        float* wb = Q + (size_t)row2 * 128;
        float a = 0.f, b = 0.f;
        if (READ == 1) { a = base[lane]; b = base[64 + lane]; }
        if (READ == 2) { a = ld_sc1(base + lane); b = ld_sc1(base + 64 + lane); }
        if (READ == 3) { a = ld_nt(base + lane); b = ld_nt(base + 64 + lane); }
        if (WRITE == 1) { unsafeAtomicAdd(wb + lane, 1e-6f); unsafeAtomicAdd(wb + 64 + lane, 1e-6f); }
        if (WRITE == 2) { a = unsafeAtomicAdd(wb + lane, 1e-6f); b = unsafeAtomicAdd(wb + 64 + lane, 1e-6f); }
        if (WRITE == 3) { wb[lane] = a + 1e-6f; wb[64 + lane] = b + 1e-6f; }
        acc += a + b;
    }
    if (acc == 123.456f) Q[0] = acc;
}

template <int READ, int WRITE, bool OTHER>
static void run(const char* name, float* Q, int n_rows) {
    const int per_wave = 2048, waves = 256 * 32;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((upd<READ, WRITE, OTHER>), dim3(waves / 4), dim3(256), 0, 0, Q, n_rows, per_wave);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    const double rows = (double)waves * per_wave;
    printf("%-44s %8.3f ms  %6.3f ns/row\n", name, ms, ms * 1e6 / rows);
}

int main() {
    const int n_rows = 27278;
    float* Q; CK(hipMalloc(&Q, (size_t)n_rows * 128 * 4)); CK(hipMemset(Q, 0, (size_t)n_rows * 128 * 4));
    run<0, 1, false>("atomic only", Q, n_rows);
    run<1, 0, false>("plain load only", Q, n_rows);
    run<2, 0, false>("sc1 load only", Q, n_rows);
    run<3, 0, false>("nt load only", Q, n_rows);
    run<1, 1, false>("plain load + atomic, same row", Q, n_rows);
    run<2, 1, false>("sc1 load + atomic, same row (the walk)", Q, n_rows);
    run<3, 1, false>("nt load + atomic, same row", Q, n_rows);
    run<1, 1, true>("plain load + atomic, another row", Q, n_rows);
    run<2, 1, true>("sc1 load + atomic, another row", Q, n_rows);
    run<0, 2, false>("returning atomic (read and add in one)", Q, n_rows);
    run<1, 3, false>("plain load + plain store (one L2's view)", Q, n_rows);
    return 0;
}
