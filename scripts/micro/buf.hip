#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* q, int vdim, int coh) {
    int lane = threadIdx.x & 63;
    const float* base = p + blockIdx.x * vdim;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, vdim * 4, 0x00020000);
    u32x4 v = coh ? __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, 0);
    float f[4];
    for (int c = 0; c < 4; ++c) f[c] = __builtin_bit_cast(float, v[c]);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(q + blockIdx.x * vdim, 0, vdim * 4, 0x00020000);
    u32x4 o;
    for (int c = 0; c < 4; ++c) o[c] = __builtin_bit_cast(unsigned int, f[c] + 1.0f);
    if (coh) __builtin_amdgcn_raw_buffer_store_b128(o, w, lane * 16, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b128(o, w, lane * 16, 0, 0);
}
int main() {
    const int rows = 3, vdim = 128;
    std::vector<float> h(rows * vdim), g(rows * vdim, -1.f);
    for (int i = 0; i < rows * vdim; ++i) h[i] = i;
    float *p, *q;
    hipMalloc(&p, h.size() * 4); hipMalloc(&q, h.size() * 4);
    for (int coh = 0; coh < 2; ++coh) {
        hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(q, 0, h.size() * 4);
        hipLaunchKernelGGL(k, dim3(rows), dim3(64), 0, 0, p, q, vdim, coh);
        hipMemcpy(g.data(), q, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < rows * vdim; ++i) if (g[i] != h[i] + 1.f) ++bad;
        printf("coh=%d bad=%d first: %g %g %g %g %g | row1: %g %g\n", coh, bad, g[0], g[1], g[2], g[3], g[4], g[128], g[129]);
    }
    return 0;
}
