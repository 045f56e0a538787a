// What does the memory system give for the ALS row kernel's access pattern -- 20 M random 512-byte rows out of a 14 MB table (the item factors,
// user half-epoch) or a 71 MB one (the user factors, item half-epoch) -- as a function of what is kept in flight?  The pair kernel's skeleton
// (keys + row loads + ring handshakes, no arithmetic) runs 1.16 / 1.31 ms per half-epoch with 4 producer waves per CU and three groups of 16
// rows in flight each (profiles/r04_als_pc_steps.txt).  Here: nothing but the loads -- every half-wave reads one row with one dwordx4 per
// lane, G groups of 16 rows in flight per wave, W waves per CU -- so the numbers are the ceiling for any formulation that gathers fp32 rows.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int G>
__global__ void gather(const float4* __restrict__ tab, const int* __restrict__ keys, long n_groups, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const long wave = (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const long n_waves = (static_cast<long>(gridDim.x) * blockDim.x) >> 6;
    const long per = (n_groups + n_waves - 1) / n_waves;
    long g0 = wave * per, g1 = g0 + per < n_groups ? g0 + per : n_groups;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long g = g0; g < g1; g += G) {
        float4 v[G][8];
#pragma unroll
        for (int s = 0; s < G; ++s) {
            const long gg = g + s < g1 ? g + s : g1 - 1;
            const int4 k0 = *reinterpret_cast<const int4*>(keys + gg * 16 + 8 * half);
            const int4 k1 = *reinterpret_cast<const int4*>(keys + gg * 16 + 8 * half + 4);
            const int kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int r = 0; r < 8; ++r) v[s][r] = tab[static_cast<long>(kk[r]) * 32 + col];
        }
#pragma unroll
        for (int s = 0; s < G; ++s)
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc.x += v[s][r].x; acc.y += v[s][r].y; acc.z += v[s][r].z; acc.w += v[s][r].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int G>
static float run(const float4* tab, const int* keys, long n_groups, int waves_per_cu, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int block = 64 * (waves_per_cu >= 4 ? 4 : waves_per_cu), grid = 256 * waves_per_cu * 64 / block;
    float t = 0.f, best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gather<G>, dim3(grid), dim3(block), 0, 0, tab, keys, n_groups, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && t < best) best = t;
    }
    return best;
}

int main() {
    const long nnz = 20000256;   // 16 | nnz
    const long n_groups = nnz / 16;
    float* out;
    (void)hipMalloc(&out, 256);
    for (int rows : {27278, 138493}) {
        float4* tab;
        (void)hipMalloc(&tab, static_cast<size_t>(rows) * 512);
        (void)hipMemset(tab, 0, static_cast<size_t>(rows) * 512);
        for (int sorted = 0; sorted < 2; ++sorted) {
            std::vector<int> h(nnz);
            unsigned long long s = 88172645463325252ull;
            for (long i = 0; i < nnz; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = static_cast<int>(s % static_cast<unsigned long long>(rows)); }
            if (sorted) {   // keys ascending inside runs of 144 (a CSR row's keys are sorted)
                for (long i = 0; i + 144 <= nnz; i += 144) std::qsort(&h[i], 144, sizeof(int), [](const void* a, const void* b) { return *(const int*)a - *(const int*)b; });
            }
            int* keys;
            (void)hipMalloc(&keys, nnz * sizeof(int));
            (void)hipMemcpy(keys, h.data(), nnz * sizeof(int), hipMemcpyHostToDevice);
            printf("table of %6d rows x 512 B (%.1f MB), keys %s: ms per 20 M rows (TB/s)\n", rows, rows * 512e-6, sorted ? "ascending inside runs of 144" : "uniform random");
            for (int w : {4, 8, 16}) {
                const float a = run<1>(tab, keys, n_groups, w, out), b = run<2>(tab, keys, n_groups, w, out), c = run<3>(tab, keys, n_groups, w, out),
                            d = run<4>(tab, keys, n_groups, w, out);
                auto tb = [&](float ms) { return nnz * 512.0 / (ms * 1e-3) / 1e12; };
                printf("  %2d waves per CU: 1 group in flight %.3f (%.2f)   2: %.3f (%.2f)   3: %.3f (%.2f)   4: %.3f (%.2f)\n", w, a, tb(a), b, tb(b), c, tb(c), d, tb(d));
            }
            (void)hipFree(keys);
        }
        (void)hipFree(tab);
    }
    return 0;
}
