// One wave: does a load issued after an fp32 atomic add of the same wave see the add?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/atomic_then_load.hip -o /tmp/atl && /tmp/atl
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// mode 0: atomic(no return), wait vmcnt(0), sc1 load      mode 1: atomic, NO wait, sc1 load
// mode 2: atomic, wait, plain load                       mode 3: sc1 load first (line into L2), then as mode 0
// mode 4: as 3 but without the wait
__global__ void k(float* x, float* out, int mode, int iters) {
    const int lane = threadIdx.x;
    float seen_stale = 0.f;
    for (int it = 0; it < iters; ++it) {
        float* p = x + ((it * 97) % 4096) * 64 + lane;     // a different 256-B row every iteration
        float before = 0.f;
        if (mode >= 3) before = ld_sc1(p);
        else before = *reinterpret_cast<volatile float*>(p);
        unsafeAtomicAdd(p, 1.0f);
        if (mode == 0 || mode == 2 || mode == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float after = (mode == 2) ? *reinterpret_cast<volatile float*>(p) : ld_sc1(p);
        if (after != before + 1.0f) seen_stale += 1.f;
    }
    out[lane] = seen_stale;
}
int main() {
    float *x, *out;
    hipMalloc(&x, 4096 * 64 * 4);
    hipMalloc(&out, 64 * 4);
    for (int mode = 0; mode < 5; ++mode) {
        hipMemset(x, 0, 4096 * 64 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, out, mode, 20000);
        float h[64];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        float s = 0;
        for (float v : h) s += v;
        printf("mode %d: stale reads %.0f of %d lane-iterations\n", mode, s, 64 * 20000);
    }
    return 0;
}
