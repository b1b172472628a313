// Micro-benchmark: rate of divergent 4-byte gathers on gfx950 by where the table lives -- L1/L2-resident
// tables of 8 KB .. 160 MB in global memory, and a 32 KB table in LDS.  65536 waves x 64 lanes x 16 loads.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather2 gather2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int K>
__global__ void __launch_bounds__(256) gather_g(const uint32_t* __restrict__ t, uint32_t mask, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx[K], v[K], acc = 0;
#pragma unroll
    for (int k = 0; k < K; k++) idx[k] = mix32(tid * K + k) & mask;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = t[idx[k]];
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k];
    if (acc == 42u) out[tid] = acc;
}

template <int K>
__global__ void __launch_bounds__(256) gather_lds(const uint32_t* __restrict__ t, uint32_t* out, int rounds) {
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = t[i];
    __syncthreads();
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint32_t idx[K], v[K];
#pragma unroll
        for (int k = 0; k < K; k++) idx[k] = mix32((tid * K + k) ^ (r * 0x9e3779b9u) ^ acc) & 8191;
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = lds[idx[k]];
#pragma unroll
        for (int k = 0; k < K; k++) acc += v[k];
    }
    if (acc == 42u) out[tid] = acc;
}

int main() {
    uint32_t *t, *out;
    const size_t maxw = 40u << 20;  // 160 MB of 4-byte words
    hipMalloc(&t, maxw * 4); hipMemset(t, 0, maxw * 4);
    hipMalloc(&out, 65536 * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int waves = 65536;
    for (uint32_t words : {2048u, 1u << 16, 1u << 17 /* 512 KB */, 1u << 19, 1u << 22, 1u << 25}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(gather_g<16>, dim3(waves / 4), dim3(256), 0, 0, t, words - 1, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("global table %8.0f KB: %.3f ms  %.1f G lane-loads/s\n", words * 4 / 1024.0, best, (double)waves * 64 * 16 / best / 1e6);
    }
    for (int rounds : {1, 8}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(gather_lds<16>, dim3(waves / 4), dim3(256), 0, 0, t, out, rounds);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("LDS table 32 KB, %d round(s) of 16 gathers (incl. 32 KB staging per block): %.3f ms  %.1f G lane-loads/s\n", rounds, best,
               (double)waves * 64 * 16 * rounds / best / 1e6);
    }
    return 0;
}
