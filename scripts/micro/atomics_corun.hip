// Round-6 study (b) of the BPR lr cliff: are the fp32 row atomics of the negatives and the rest of the item-major walk
// SEPARATE resources?  This is the "atomics only" half as a small shared library: row atomics on uniformly drawn rows of its OWN
// 27,278 x 128 fp32 matrix, on its OWN stream, so that scripts/r6_lr005_corun.py can run it beside the walk (whose negatives' atomics are
// masked by the `im_study` knob) and compare: each alone, both together.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -shared -fPIC scripts/micro/atomics_corun.hip -o scripts/micro/libatomics_corun.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// every wave: `per_wave` rows, one atomic instruction per 128-byte line pair (dword per lane, two instructions per 512-byte row) -- what
// hrow_atomic_add issues per half-wave, here for the whole wave
__global__ __launch_bounds__(256) void corun_atomics(float* Q, int n_rows, int per_wave, unsigned salt) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < per_wave; ++i) {
        const unsigned h = hash32(wave * 7919u + i * 104729u + salt);
        const unsigned row = __builtin_amdgcn_readfirstlane(h % (unsigned)n_rows);
        float* wb = Q + (size_t)row * 128;
        unsafeAtomicAdd(wb + lane, 1e-6f);
        unsafeAtomicAdd(wb + 64 + lane, 1e-6f);
    }
}

static hipStream_t g_stream = nullptr;
static float* g_Q = nullptr;
static int g_rows = 0;
static std::vector<hipEvent_t> g_ev;

extern "C" int corun_init(int n_rows) {
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
    g_rows = n_rows;
    if (hipMalloc(&g_Q, (size_t)n_rows * 128 * 4) != hipSuccess) return 2;
    if (hipMemsetAsync(g_Q, 0, (size_t)n_rows * 128 * 4, g_stream) != hipSuccess) return 3;
    return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 4;
}
// queue `launches` launches of `rows_per_launch` row atomics on `waves_per_cu` x 256 waves; returns at once
extern "C" int corun_start(long long rows_per_launch, int launches, int waves_per_cu) {
    const int waves = 256 * waves_per_cu;
    const int per_wave = (int)((rows_per_launch + waves - 1) / waves);
    for (auto e : g_ev) hipEventDestroy(e);
    g_ev.assign(launches + 1, nullptr);
    for (auto& e : g_ev)
        if (hipEventCreate(&e) != hipSuccess) return 1;
    hipEventRecord(g_ev[0], g_stream);
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL(corun_atomics, dim3(waves / 4), dim3(256), 0, g_stream, g_Q, g_rows, per_wave, 12345u + 977u * l);
        hipEventRecord(g_ev[l + 1], g_stream);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
// waits; ms[l] = duration of launch l
extern "C" int corun_wait(float* ms, int launches) {
    if (hipStreamSynchronize(g_stream) != hipSuccess) return 1;
    for (int l = 0; l < launches && l + 1 < (int)g_ev.size(); ++l) hipEventElapsedTime(ms + l, g_ev[l], g_ev[l + 1]);
    return 0;
}
