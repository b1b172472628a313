// Micro-benchmark: what does a divergent 8-byte gather cost on gfx950 -- per active lane, or per
// instruction?  65536 waves, each issuing K independent random loads from a 160 MB table with
// 64 / 32 / 16 / 4 active lanes.  Build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <int K>
__global__ void __launch_bounds__(256) gather(const double* __restrict__ t, uint64_t m, int active, double* out, int dep) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = 0.0;
    if (lane < active) {
        uint64_t idx[K];
#pragma unroll
        for (int k = 0; k < K; k++) idx[k] = mix(wave * 64 * K + lane * K + k) % m;
        double v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = t[idx[k]];
#pragma unroll
        for (int k = 0; k < K; k++) acc += v[k];
        if (dep) {  // a second, dependent round
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = t[(idx[k] + (uint64_t)(acc != 12345.0 ? 977 : 0)) % m];
#pragma unroll
            for (int k = 0; k < K; k++) acc += v[k];
        }
    }
    if (acc == 42.0) out[wave] = acc;
}

int main() {
    const uint64_t m = 20000000;
    double *t, *out;
    hipMalloc(&t, m * 8); hipMemset(t, 0, m * 8);
    hipMalloc(&out, 65536 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int waves = 65536;
    for (int dep = 0; dep < 2; dep++)
        for (int active : {64, 32, 16, 4}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(gather<16>, dim3(waves / 4), dim3(256), 0, 0, t, m, active, out, dep);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double loads = (double)waves * active * 16 * (1 + dep);
            printf("dep=%d active=%2d  %.3f ms  %.1f G lane-loads/s  %.1f G instr-slots/s (x64)\n", dep, active, best,
                   loads / best / 1e6, (double)waves * 64 * 16 * (1 + dep) / best / 1e6);
        }
    return 0;
}
