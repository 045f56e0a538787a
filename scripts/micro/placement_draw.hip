// Round-6 study: is the per-handle draw of the item-major walk's time (3.87 .. 4.33 ms, profiles/r06_walk_variance.txt) a property of WHERE the
// driver puts the buffers?  The walk's memory skeleton alone, on its own allocations: every half-wave reads + writes one 512-byte row of a
// user table (71 MB; a row is only touched from the XCD that owns it) and one row of its XCD's replica of the item table (8 x 14 MB), plus
// 12 streamed bytes per step -- the same instructions (agent-scope relaxed loads past the L1, plain stores), the same residency (5 x 256
// threads per CU).  Trials: (a) everything allocated anew, (b) only the user table anew, (c) only the replicas anew, (d) nothing anew.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/placement_draw.hip -o gpurun_out/placement_draw && gpurun_out/placement_draw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return (int)(x & 7u); }

constexpr int D = 128;
__global__ __launch_bounds__(256, 5) void skeleton(float* P, int p_rows, float* rep, int q_rows, const int* stream, long long stream_n, int steps, unsigned salt) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int x = xcc_id();
    float* const myrep = rep + (size_t)x * q_rows * D;
    const unsigned hw = ((blockIdx.x * blockDim.x + threadIdx.x) >> 5);   // half-wave id
    float acc = 0.f;
    for (int i = 0; i < steps; ++i) {
        const unsigned h = hash32(hw * 7919u + (unsigned)i * 104729u + salt);
        const unsigned h2 = hash32(h ^ 0x9e3779b9u);
        unsigned u = h % (unsigned)p_rows; u = u - (u & 7u) + (unsigned)x; if (u >= (unsigned)p_rows) u -= 8;   // a user of this XCD
        const unsigned j = h2 % (unsigned)q_rows;
        const long long s = ((long long)hw * steps + i) * 3 % (stream_n - 3);
        const int m = l32 < 3 ? stream[s + l32] : 0;       // 12 streamed bytes per step
        float* pu = P + (size_t)u * D + l32;
        float* qj = myrep + (size_t)j * D + l32;
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = __hip_atomic_load(pu + k * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b[k] = __hip_atomic_load(qj + k * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) d += a[k] * b[k];
        d += (float)m * 1e-30f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { pu[k * 32] = a[k] + d * 1e-9f; qj[k * 32] = b[k] - d * 1e-9f; }
        acc += d;
    }
    if (acc == 12345.678f) P[0] = acc;
    (void)half;
}

struct Tables { float* P = nullptr; float* rep = nullptr; int* stream = nullptr; };
int main(int argc, char** argv) {
    const int p_rows = 138493, q_rows = 27278;
    const long long stream_n = 30000000;
    const int trials = argc > 1 ? atoi(argv[1]) : 6;
    const size_t pb = (size_t)p_rows * D * 4, rb = (size_t)8 * q_rows * D * 4, sb = stream_n * 4;
    const int grid = 256 * 5, steps = 10000131 / (grid * 8);   // 8 half-waves per workgroup
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto alloc = [&](void** p, size_t n) { CK(hipMalloc(p, n)); CK(hipMemset(*p, 0, n)); };
    auto run = [&](Tables& t) {
        float best = 1e9f, worst = 0.f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0));
            skeleton<<<grid, 256>>>(t.P, p_rows, t.rep, q_rows, t.stream, stream_n, steps, 77u + r);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
        }
        printf("  %.3f .. %.3f ms   P %p rep %p stream %p\n", best, worst, (void*)t.P, (void*)t.rep, (void*)t.stream);
    };
    const char* names[4] = {"everything allocated anew", "only the user table anew", "only the replicas anew", "nothing anew"};
    for (int mode = 0; mode < 4; ++mode) {
        printf("%s\n", names[mode]);
        Tables t; alloc((void**)&t.P, pb); alloc((void**)&t.rep, rb); alloc((void**)&t.stream, sb);
        std::vector<void*> held;
        for (int k = 0; k < trials; ++k) {
            if (k) {
                // the old block is kept until the new one exists, so that the new one lands elsewhere
                if (mode == 0 || mode == 1) { void* o = t.P; alloc((void**)&t.P, pb); CK(hipFree(o)); }
                if (mode == 0 || mode == 2) { void* o = t.rep; alloc((void**)&t.rep, rb); CK(hipFree(o)); }
                if (mode == 0) { void* o = t.stream; alloc((void**)&t.stream, sb); CK(hipFree(o)); }
            }
            run(t);
        }
        CK(hipFree(t.P)); CK(hipFree(t.rep)); CK(hipFree(t.stream));
    }
    return 0;
}
