// TEST INFRASTRUCTURE ONLY (tests/host_env): a stand-in for <hip/hip_runtime.h> that lets g++ compile the
// engine's *device* headers (rl_markets_amd/csrc/lob_env.h) as plain host code, so that two implementations of the
// same device routine can be compared with each other on the CPU, bit for bit, before they go to the GPU.
// Nothing under rl_markets_amd/ includes this; the product has no host execution path.
#ifndef LOB_TEST_HIP_SHIM_H
#define LOB_TEST_HIP_SHIM_H
#include <math.h>
#include <stdint.h>
#include <string.h>
#define __HIPCC__ 1
#define __device__
#define __host__
#define __global__
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)
struct uint4 { uint32_t x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
static inline int atomicOr(int* p, int v) { int o = *p; *p |= v; return o; }
static inline long long clock64() { return 0; }
// (serial stand-ins: the host tests run one "lane" at a time)
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long c, unsigned long long v) { const unsigned long long o = *p; if (o == c) *p = v; return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p |= v; return o; }
static inline int atomicAdd(int* p, int v) { const int o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
// (wave-level intrinsics of code the host tests never call -- prepass_run's cooperative row fetch -- so that it still parses)
#define __forceinline__ inline
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_s_barrier() ((void)0)
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __any(int p) { return p; }
static const struct { unsigned x, y, z; } threadIdx = {0, 0, 0};
#endif
