// Third micro-benchmark of the series (simd_overlap.hip: a matrix-only wave beside a VALU-only wave; simd_fused_pairs.hip: waves that each
// carry matrix + VALU work; results and their reading: profiles/r04_micro_simd_overlap.txt).  This one maps the ways a row's group of 16 entries
// -- 30 x v_mfma_f32_32x32x16_f16 over 10 tiles and V independent fp32 FMAs standing in for gather + residuals + f16 cut -- can be dealt to
// the two waves of a SIMD (waves w and w + 4 of a 512-thread workgroup).  Role 0 = waves 0-3: NT0 tiles, V0 FMAs spread behind its matrix
// instructions; role 1 = waves 4-7: NT1 tiles, V1 FMAs (NT1 = 0: a plain FMA stream).  Printed: ns per group.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/micro/simd_split_map.hip -o /tmp/simd_split_map && /tmp/simd_split_map
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// matrix instruction I of M, then its share of the V FMAs; the scheduler is told to keep them in that order
template <int I, int M, int NT, int V>
__device__ __forceinline__ void steps(f32x16 (&acc)[NT], const f16x8& a, const f16x8& b, float (&x)[8], float m, float ad) {
    if constexpr (I < M) {
        acc[I % NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[I % NT], 0, 0, 0);
        constexpr int nv = (I + 1) * V / M - I * V / M;
#pragma unroll
        for (int v = 0; v < nv; ++v) x[(I + v) & 7] = __builtin_fmaf(x[(I + v) & 7], m, ad);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
        steps<I + 1, M, NT, V>(acc, a, b, x, m, ad);
    }
}

template <int NT, int V>
__device__ __forceinline__ void stream(int iters, float* out) {
    constexpr int NA = NT > 0 ? NT : 1;
    f32x16 acc[NA];
    for (int t = 0; t < NA; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = static_cast<_Float16>(threadIdx.x * 0.001f + e); b[e] = static_cast<_Float16>(e * 0.5f); }
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = threadIdx.x * 0.01f + c;
    const float m = 1.0000001f, ad = 1e-9f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (NT == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], m, ad);
        } else {
            steps<0, 3 * NT, NA, V>(acc, a, b, x, m, ad);
        }
    }
    float s = 0.f;
    for (int t = 0; t < NA; ++t) s += acc[t][0];
    for (int c = 0; c < 8; ++c) s += x[c];
    if (s == 123.456f) out[0] = s;
}

template <int NT0, int V0, int NT1, int V1>
__global__ __launch_bounds__(512, 2) void k(int iters, float* out) {
    if ((threadIdx.x >> 6) < 4) stream<NT0, V0>(iters, out);
    else stream<NT1, V1>(iters, out);
}

template <int NT0, int V0, int NT1, int V1>
static void run(const char* what, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float t = 0.f;
    hipLaunchKernelGGL((k<NT0, V0, NT1, V1>), dim3(256), dim3(512), 0, 0, iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT0, V0, NT1, V1>), dim3(256), dim3(512), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    printf("role 0: %2d tiles + %3d FMAs | role 1: %2d tiles + %3d FMAs : %5.0f ns per group   (%s)\n", NT0, V0, NT1, V1, t * 1e6 / iters, what);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256);
    // 270 FMAs per group in all
    run<10, 0, 0, 270>("today's pairs: matrix wave | VALU wave", out);
    run<10, 30, 0, 240>("the matrix wave takes 30 of the FMAs", out);
    run<10, 60, 0, 210>("... 60", out);
    run<10, 120, 0, 150>("... 120 (4 behind each matrix instruction)", out);
    run<8, 0, 2, 270>("the VALU wave takes 2 of the tiles", out);
    run<7, 0, 3, 270>("... 3", out);
    run<6, 0, 4, 270>("... 4", out);
    run<5, 0, 5, 270>("... 5", out);
    run<7, 60, 3, 210>("3 tiles and 210 FMAs | 7 tiles and 60", out);
    run<6, 90, 4, 180>("4 tiles and 180 FMAs | 6 tiles and 90", out);
    run<5, 135, 5, 135>("even split", out);
    // 200 FMAs per group in all (a leaner preparation)
    run<10, 0, 0, 200>("200 FMAs: today's pairs", out);
    run<7, 0, 3, 200>("200 FMAs: the VALU wave takes 3 tiles", out);
    run<5, 100, 5, 100>("200 FMAs: even split", out);
    // the solve's share: a plain VALU stream next to a wave with its own mix (what a row end looks like when the partner keeps working)
    run<10, 120, 0, 400>("mixed wave | 400 FMAs alone", out);
    run<0, 400, 0, 400>("two plain VALU streams of 400", out);
    run<0, 400, 0, 0>("one plain VALU stream of 400", out);
    return 0;
}
