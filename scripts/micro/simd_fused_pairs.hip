// Follow-up to simd_overlap.hip (whose first, packed-FMA build showed a VALU-only wave and a matrix-only wave taking the SUM of their times;
// with plain VALU they overlap -- profiles/r04_micro_simd_overlap.txt).
// Question for the next ALS design: when every wave carries its OWN mix -- a matrix instruction followed by a few
// independent fp32 FMAs, the way round 3's fused row kernel issues them -- how much of the VALU work hides, with one such wave per
// SIMD and with two?  A 512-thread workgroup per CU puts waves w and w + 4 on one SIMD.
//   A  one wave per SIMD (waves 4-7 leave at once), 30 matrix instructions per group, VPM FMAs behind each
//   B  two waves per SIMD, 15 matrix instructions per group EACH (half the tiles each), VPM FMAs behind each
//   C  two waves per SIMD, 30 matrix instructions per group each (two whole rows side by side)
// A "group" is 16 entries of a row: 30 x v_mfma_f32_32x32x16_f16 (10 tiles x 3 products).  Printed per case: ns per group of ONE row.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/micro/simd_fused_pairs.hip -o /tmp/simd_fused_pairs && /tmp/simd_fused_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NT, int VPM>
__global__ __launch_bounds__(512, 2) void k(int waves, int iters, float* out) {
    const int wv = threadIdx.x >> 6;
    if (wv >= waves) return;
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = static_cast<_Float16>(threadIdx.x * 0.001f + e); b[e] = static_cast<_Float16>(e * 0.5f); }
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 0.01f + c;
    const float m = 1.0000001f, ad = 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VPM; ++v) x[(t * VPM + v) & 7] = __builtin_fmaf(x[(t * VPM + v) & 7], m, ad);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) s += acc[t][0];
    for (int c = 0; c < 8; ++c) s += x[c];
    if (s == 123.456f) out[0] = s;
}

template <int NT, int VPM>
static float run(int waves, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float t = 0.f;
    hipLaunchKernelGGL((k<NT, VPM>), dim3(256), dim3(512), 0, 0, waves, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, VPM>), dim3(256), dim3(512), 0, 0, waves, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t, e0, e1);
    return t;
}

template <int VPM>
static void row(float* out) {
    const int iters = 4000;
    const float a = run<10, VPM>(4, iters, out);   // one wave per SIMD, whole groups
    const float b = run<5, VPM>(8, iters, out);    // two per SIMD, half a group each: a row's group = one iteration of both
    const float c = run<10, VPM>(8, iters, out);   // two per SIMD, whole groups: two rows advance one group per iteration
    printf("%d FMAs behind each matrix instruction (%3d per group):  A one wave / SIMD %4.0f ns per group   B two waves / SIMD, half the tiles each %4.0f ns per group   "
           "C two waves / SIMD, a row each %4.0f ns per group and row (%4.0f per iteration)\n",
           VPM, 30 * VPM, a * 1e6 / iters, b * 1e6 / iters, c * 1e6 / iters / 2, c * 1e6 / iters);
}

int main() {
    float* out;
    hipMalloc(&out, 256);
    row<0>(out);
    row<2>(out);
    row<4>(out);
    row<5>(out);
    row<7>(out);
    row<9>(out);
    return 0;
}
