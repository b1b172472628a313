// Micro-benchmark for env_step_kernel's second load round: every lane (book) needs two adjacent 224-byte records (448 B) and
// two adjacent 128-byte track entries (256 B) of its own stream.
//   (a) per lane: 28 + 16 sixteen-byte loads, each instruction touching 64 different 64-byte sectors a quarter at a time;
//   (b) cooperative: 4 lanes fetch the 4 quads of one 64-byte chunk, an instruction covers 16 chunks; the data goes through LDS
//       (a 16 KB buffer, reused) and every lane reads its own back.
// One round, all 65 536 lanes at once (1 024 one-wave blocks), cold (a different row range per repetition).
// Build: hipcc --offload-arch=gfx950 -O3 -o round2 round2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ROW_Q 14
#define TRK_Q 8

__device__ inline const uint4* shfl_ptr(const uint4* p, int src) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return (const uint4*)(((unsigned long long)hi << 32) | lo);
}
// NQ quads per book from per-lane pointer p into out[NQ] through lds[64 * NQ]
template <int NQ>
__device__ inline void coop(const uint4* p, uint4* out, uint4* lds, int lane) {
    constexpr int NC = (NQ + 3) / 4;
    uint4 t[NC * 4];
    const int sub = lane >> 2, quad = lane & 3;
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint4* bp = shfl_ptr(p, 16 * r + sub);
            const int q = 4 * c + quad;
            t[c * 4 + r] = q < NQ ? bp[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = 4 * c + quad;
            if (q < NQ) lds[(16 * r + sub) * NQ + q] = t[c * 4 + r];
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NQ; q++) out[q] = lds[lane * NQ + q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}

template <int MODE>
__global__ void __launch_bounds__(64) round2(const uint4* __restrict__ rows, size_t row_stride_q, const uint4* __restrict__ trk, size_t trk_stride_q, int at, uint32_t* out) {
    __shared__ uint4 lds[64 * 16];
    const int lane = threadIdx.x;
    const size_t book = (size_t)blockIdx.x * 64 + lane;
    const uint4* rp = rows + book * row_stride_q + (size_t)(at + (int)(book % 7)) * ROW_Q;   // (books are not in step with each other)
    const uint4* tp = trk + book * trk_stride_q + (size_t)(at + (int)(book % 5)) * TRK_Q;
    uint4 r0[ROW_Q], r1[ROW_Q], t01[2 * TRK_Q];
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 2 * TRK_Q; q++) t01[q] = tp[q];
#pragma unroll
        for (int q = 0; q < ROW_Q; q++) r0[q] = rp[q];
#pragma unroll
        for (int q = 0; q < ROW_Q; q++) r1[q] = rp[ROW_Q + q];
    } else {
        coop<2 * TRK_Q>(tp, t01, lds, lane);
        coop<ROW_Q>(rp, r0, lds, lane);
        coop<ROW_Q>(rp + ROW_Q, r1, lds, lane);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < ROW_Q; q++) acc += r0[q].x ^ r0[q].y ^ r0[q].z ^ r0[q].w ^ r1[q].x ^ r1[q].y ^ r1[q].z ^ r1[q].w;
#pragma unroll
    for (int q = 0; q < 2 * TRK_Q; q++) acc += t01[q].x ^ t01[q].y ^ t01[q].z ^ t01[q].w;
    out[book] = acc;
}

int main() {
    const size_t books = 65536, n = 512;
    uint4 *rows, *trk; uint32_t* out;
    hipMalloc(&rows, books * n * ROW_Q * 16); hipMemset(rows, 1, books * n * ROW_Q * 16);
    hipMalloc(&trk, books * n * TRK_Q * 16); hipMemset(trk, 2, books * n * TRK_Q * 16);
    hipMalloc(&out, books * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int pass = 0; pass < 2; pass++)
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9, sum = 0;
            for (int rep = 0; rep < 8; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(round2<0>, dim3(books / 64), dim3(64), 0, 0, rows, n * ROW_Q, trk, n * TRK_Q, 16 * rep + 40 * mode + 200 * pass, out);
                else hipLaunchKernelGGL(round2<1>, dim3(books / 64), dim3(64), 0, 0, rows, n * ROW_Q, trk, n * TRK_Q, 16 * rep + 40 * mode + 200 * pass, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
                sum += ms;
            }
            printf("%-44s best %.1f us  mean %.1f us\n", mode == 0 ? "per lane: 44 x 16 B" : "cooperative 4 lanes / 64 B chunk, through LDS", best * 1000.0f, sum / 8 * 1000.0f);
        }
    return 0;
}
