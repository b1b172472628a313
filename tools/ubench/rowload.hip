// Micro-benchmark: how should a lane-per-book kernel fetch one 224-byte record per lane from 65 536 different streams?
//   (a) every lane reads its own record with twelve 16-byte loads (what env_kernel does): each load instruction touches 64
//       different cache lines, and the twelve loads of a lane go to the same 2-3 lines again and again;
//   (b) cooperative: 14 lanes read one record (16 bytes each, contiguous), so one load instruction covers 4 records and
//       touches ~10 lines; the records go through LDS and every lane then reads its own from there.
// R dependent rounds (a pass of the event loop cannot start before the previous one has finished), 1 024 one-wave blocks.
// Build: hipcc --offload-arch=gfx950 -O3 -o rowload rowload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ROW_BYTES 224
#define ROW_Q (ROW_BYTES / 16)

template <int MODE, int R>
__global__ void __launch_bounds__(64) rowload(const uint4* __restrict__ buf, size_t stride_q, int start_row, uint32_t* out) {
    __shared__ uint4 lds[64 * ROW_Q];
    const int lane = threadIdx.x;
    const size_t book = (size_t)blockIdx.x * 64 + lane;
    const uint4* base = buf + book * stride_q + (size_t)start_row * ROW_Q;
    uint32_t acc = 0;
    int skip = 0;
    for (int r = 0; r < R; r++) {
        const uint4* row = base + (size_t)(r + skip) * ROW_Q;
        uint4 v[12];
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 12; q++) v[q] = row[1 + q];
        } else {
            // 16 instructions x 4 records: lane l of an instruction reads quad (l % 14) of record 4 * i + l / 14
            const int sub = lane / 14, quad = lane % 14;
            uint4 t[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int src = 4 * i + sub;  // whose record
                const unsigned long long p = (unsigned long long)row;
                const unsigned lo = __shfl((unsigned)p, src & 63), hi = __shfl((unsigned)(p >> 32), src & 63);
                const uint4* rp = (const uint4*)(((unsigned long long)hi << 32) | lo);
                t[i] = lane < 56 ? rp[quad] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (lane < 56) lds[(4 * i + sub) * ROW_Q + quad] = t[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 12; q++) v[q] = lds[lane * ROW_Q + 1 + q];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int q = 0; q < 12; q++) acc += v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
        skip += (acc & 1);  // the next round's address depends on this round's data
    }
    out[book] = acc;
}

int main() {
    const size_t books = 65536, rows_per_book = 512;
    const size_t stride_q = rows_per_book * ROW_Q;
    uint4* buf; uint32_t* out;
    hipMalloc(&buf, books * stride_q * 16);
    hipMemset(buf, 1, books * stride_q * 16);
    hipMalloc(&out, books * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char* tag) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(books / 64), dim3(64), 0, 0, buf, stride_q, 8 * rep + 1, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-40s %.1f us\n", tag, best * 1000.0f);
    };
    run(rowload<0, 1>, "per-lane 12 x 16 B, 1 round");
    run(rowload<1, 1>, "cooperative 14 lanes / record, 1 round");
    run(rowload<0, 4>, "per-lane 12 x 16 B, 4 dependent rounds");
    run(rowload<1, 4>, "cooperative, 4 dependent rounds");
    run(rowload<0, 8>, "per-lane, 8 dependent rounds");
    run(rowload<1, 8>, "cooperative, 8 dependent rounds");
    return 0;
}
