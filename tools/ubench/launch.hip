// Micro-benchmark: what does a DEPENDENT kernel launch that finds nothing to do cost on gfx950, as a function of its
// grid, its static LDS and its register allocation?  (The step's three work-list kernels exit at once in the steady
// state; profiles/r02_kernel_stats.csv has them at 4.8-7.0 us each.)  A chain of `n` launches on one stream, each reading
// one word the previous one could have written; time per launch = chain time / n.
// Build: hipcc --offload-arch=gfx950 -O3 -o launch launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_WORDS, int REGS>
__global__ void __launch_bounds__(256) probe(const int* __restrict__ n, int* out) {
    __shared__ int lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
    if ((int)(blockIdx.x * blockDim.x) >= *n) return;
    // never executed (n = 0): keeps the LDS allocation and REGS live registers in the kernel descriptor
    int r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; i++) r[i] = out[threadIdx.x + i * 256];
    lds[threadIdx.x % (LDS_WORDS > 0 ? LDS_WORDS : 1)] = r[0];
    __syncthreads();
    int s = lds[(threadIdx.x * 7) % (LDS_WORDS > 0 ? LDS_WORDS : 1)];
#pragma unroll
    for (int i = 0; i < REGS; i++) s += r[i] * (i + 1);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDS_WORDS, int REGS>
static void run(const char* tag, const int* n, int* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {1, 16, 64, 256, 512, 1024, 4096}) {
        const int reps = 200;
        float best = 1e9;
        for (int t = 0; t < 5; t++) {
            hipEventRecord(a);
            for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe<LDS_WORDS, REGS>), dim3(grid), dim3(256), 0, 0, n, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-28s grid %5d : %.2f us per launch\n", tag, grid, best * 1000.0f / reps);
    }
}

// ... and of the size of its kernel-argument segment (the engine's kernels take DevParams / DevState by value: 1-2.5 KB)
template <int N> struct Big { int w[N / 4]; };
template <int N>
__global__ void __launch_bounds__(256) probe_arg(Big<N> a, const int* __restrict__ n, int* out) {
    if ((int)(blockIdx.x * blockDim.x) >= *n) return;
    out[blockIdx.x * 256 + threadIdx.x] = a.w[threadIdx.x % (N / 4)];
}
template <int N>
static void run_arg(const int* n, int* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    Big<N> big{};
    for (int grid : {1, 512}) {
        const int reps = 200;
        float best = 1e9;
        for (int t = 0; t < 5; t++) {
            hipEventRecord(a);
            for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe_arg<N>), dim3(grid), dim3(256), 0, 0, big, n, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("kernarg %5d B               grid %5d : %.2f us per launch\n", N, grid, best * 1000.0f / reps);
    }
}

int main() {
    int *n, *out;
    hipMalloc(&n, 4); hipMemset(n, 0, 4);
    hipMalloc(&out, 64 << 20);
    run<0, 4>("lds 0 KB, few regs", n, out);
    run<4096, 4>("lds 16 KB, few regs", n, out);
    run<16000, 4>("lds 62.5 KB, few regs", n, out);
    run<0, 200>("lds 0 KB, ~200 vgprs", n, out);
    run<4096, 200>("lds 16 KB, ~200 vgprs", n, out);
    run_arg<64>(n, out);
    run_arg<1024>(n, out);
    run_arg<2560>(n, out);
    run_arg<4000>(n, out);
    return 0;
}
