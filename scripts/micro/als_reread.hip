// What does re-touching the gathered rows cost on this memory system?  Decision input for the ALS row kernel (DESIGN "ALS"):
// the full-triangle Gramian pass reads every gathered row ONCE (512 B per entry at d = 128); a block-diagonal formulation with a
// tracked Yui (the reference's own, als.cc:269-352) needs only 4 of the 10 MFMA tiles but re-reads each row's 128-B block segment
// once (segment kept in LDS between its two uses) or twice per block.  One wave per row, rows and keys shaped like the ML-20M
// half-epochs (20 M entries; 138,493 rows gathering from a 27,278-row table, and 27,278 rows gathering from a 138,493-row table).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/als_reread.hip -o /tmp/als_reread && /tmp/als_reread
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>

// mode 0: one pass over full rows.  mode 1: + one pass per 128-B block segment.  mode 2: + two passes per segment.
__global__ __launch_bounds__(256) void k(const float* __restrict__ F, const int64_t* __restrict__ indptr, const int32_t* __restrict__ keys,
                                         int rows, int mode, float* __restrict__ out, int* ticket) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    float acc = 0.f;
    for (;;) {
        int r = 0;
        if (lane == 0) r = atomicAdd(ticket, 1);
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= rows) break;
        const int64_t b = r ? indptr[r - 1] : 0, e = indptr[r];
        for (int64_t t = b + half; t < e; t += 2) {       // a half-wave per entry: 4 dwords per lane = the 512-B row
            const float* q = F + static_cast<size_t>(keys[t]) * 128 + l32;
            acc += q[0] + q[32] + q[64] + q[96];
        }
        if (mode >= 1) {
            for (int blk = 0; blk < 4; ++blk)
                for (int rep = 0; rep < mode; ++rep) {
                    for (int64_t t = b + half; t < e; t += 2) acc += F[static_cast<size_t>(keys[t]) * 128 + blk * 32 + l32] * (rep + 1.5f);
                    asm volatile("" ::: "memory");
                }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const int64_t nnz = 20000263;
    struct Side { const char* name; int rows, table; } sides[2] = {{"user half-epoch (138,493 rows <- 27,278-row table, 14 MB)", 138493, 27278},
                                                                    {"item half-epoch (27,278 rows <- 138,493-row table, 71 MB)", 27278, 138493}};
    for (auto& sd : sides) {
        std::vector<double> w(sd.rows);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); };
        double tot = 0;
        for (auto& x : w) { const double u1 = rnd() + 1e-12, u2 = rnd(); x = std::exp(std::sqrt(-2 * std::log(u1)) * std::cos(6.283185307 * u2)); tot += x; }
        std::vector<int64_t> indptr(sd.rows);
        int64_t c = 0;
        for (int i = 0; i < sd.rows; ++i) { c += std::max<int64_t>(1, static_cast<int64_t>(w[i] / tot * nnz)); indptr[i] = c; }
        const int64_t n = c;
        std::vector<int32_t> keys(n);
        for (auto& kx : keys) kx = static_cast<int32_t>(rnd() * sd.table);
        float *F, *out; int64_t* ip; int32_t* ky; int* tk;
        hipMalloc(&F, static_cast<size_t>(sd.table) * 512); hipMemset(F, 0, static_cast<size_t>(sd.table) * 512);
        hipMalloc(&out, 64); hipMalloc(&ip, sd.rows * 8); hipMalloc(&ky, n * 4); hipMalloc(&tk, 4);
        hipMemcpy(ip, indptr.data(), sd.rows * 8, hipMemcpyHostToDevice);
        hipMemcpy(ky, keys.data(), n * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
        printf("%s, %lld entries\n", sd.name, (long long)n);
        for (int wpc : {8, 16}) for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipMemset(tk, 0, 4);
                hipEventRecord(a);
                hipLaunchKernelGGL(k, dim3(256 * wpc / 4), dim3(256), 0, 0, F, ip, ky, sd.rows, mode, out, tk);
                hipEventRecord(b2); hipEventSynchronize(b2);
                float ms; hipEventElapsedTime(&ms, a, b2); best = std::min(best, ms);
            }
            const double bytes = n * 512.0 * (1.0 + 0.25 * 4 * mode);
            printf("  %2d waves/CU  mode %d (%4.0f B per entry): %.3f ms  -> %.2f TB/s through the fabric\n", wpc, mode, bytes / n, best, bytes / best / 1e9);
        }
        hipFree(F); hipFree(out); hipFree(ip); hipFree(ky); hipFree(tk);
    }
    return 0;
}
