// Do a VALU-only wave and a matrix-only wave on ONE SIMD run side by side?  Decision input for als_pc.hpp (DESIGN "ALS"): a 512-thread
// workgroup per CU puts two waves on every SIMD (waves w and w + 4 share one: the dispatcher deals waves to SIMDs cyclically); waves
// 0-3 issue v_mfma_f32_32x32x16_f16 back to back over 10 accumulators (the consumer's stream), waves 4-7 run independent fp32 FMA
// chains (a producer-like stream: 8 chains, no memory).  Three timings: matrix waves alone, VALU waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/micro/simd_overlap.hip -o /tmp/simd_overlap && /tmp/simd_overlap
// (without -fno-slp-vectorize the FMA chains below become v_pk_fma_f32, four dependent chains: a latency-bound stream whose time ADDS to the
// matrix waves' -- the first reading of this program, corrected in profiles/r04_micro_simd_overlap.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// mode bits: 1 matrix waves run, 2 VALU waves run, 4 roles swapped (VALU = waves 0-3, the OLDER ones), 8 s_setprio 3 on the VALU waves,
// 16 s_setprio 3 on the matrix waves, 32 the VALU waves also sleep between chains (s_nop: leaves issue cycles free)
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, int valu_per_iter, float* out) {
    int wv = threadIdx.x >> 6;
    if (mode & 4) wv = (wv + 4) & 7;
    unsigned simd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(simd));
    if (threadIdx.x == 0 && blockIdx.x == 0) out[8] = 0.f;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[16 + wv] = static_cast<float>(simd);
    if (wv < 4) {
        if (!(mode & 1)) return;
        if (mode & 16) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[10];
        for (int t = 0; t < 10; ++t)
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = static_cast<_Float16>(threadIdx.x * 0.001f + e); b[e] = static_cast<_Float16>(e * 0.5f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int t = 0; t < 10; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 10; ++t) s += acc[t][0];
        if (s == 123.456f) out[0] = s;
    } else {
        if (!(mode & 2)) return;
        if (mode & 8) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 0.01f + c;
        const float m = 1.0000001f, ad = 1e-9f;
        for (int it = 0; it < iters; ++it) {
            for (int j = 0; j < valu_per_iter / 8; ++j) {
#pragma unroll
                for (int c = 0; c < 8; ++c) x[c] = __builtin_fmaf(x[c], m, ad);
            }
        }
        float s = 0.f;
        for (int c = 0; c < 8; ++c) s += x[c];
        if (s == 123.456f) out[1] = s;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int extra : {4, 8, 16, 4 + 8, 4 + 16}) {
        float t = 0.f;
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, 3 | extra, iters, 200, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, 3 | extra, iters, 200, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t, e0, e1);
        printf("both, 200 VALU per 30 matrix instructions, variant %2d (4 VALU waves are the older ones, 8 priority to VALU waves, 16 priority to matrix waves): %.3f ms (%.0f ns / group)\n",
               extra, t, t * 1e6 / iters);
    }
    for (int vpi : {120, 200, 280}) {
        float ms[4] = {0, 0, 0, 0};
        for (int mode = 1; mode <= 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, vpi, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, vpi, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
        // per "group": 30 matrix instructions and vpi FMAs
        printf("%3d VALU per 30 matrix instructions: matrix alone %.3f ms (%.0f ns / group)  VALU alone %.3f ms (%.0f ns / group)  both %.3f ms (%.0f ns / group)  sum %.3f  max %.3f\n",
               vpi, ms[1], ms[1] * 1e6 / iters, ms[2], ms[2] * 1e6 / iters, ms[3], ms[3] * 1e6 / iters, ms[1] + ms[2], ms[1] > ms[2] ? ms[1] : ms[2]);
    }
    float h[32];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("SIMD of waves 0-7 of block 0:");
    for (int w = 0; w < 8; ++w) printf(" %d", (int)h[16 + w]);
    printf("\n");
    return 0;
}
