// Exhaustive check of the engine's expf restatement (lob_env.h expf_glibc) against this libm's expf: every non-NaN float.
// gcc -O2 -fopenmp -mfma -ffp-contract=off -o check_expf tools/check_expf.c -lm && ./check_expf   (expected: mismatches 0)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static const uint64_t T[32] = {
0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
static inline double asd(uint64_t u){ double d; memcpy(&d,&u,8); return d; }
static inline uint64_t asu(double d){ uint64_t u; memcpy(&u,&d,8); return u; }
static inline uint32_t asu32(float f){ uint32_t u; memcpy(&u,&f,4); return u; }
static float my_expf(float x) {
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    uint32_t ax = asu32(x) & 0x7fffffff;
    if (ax >= asu32(88.0f)) {
        if (asu32(x) == asu32(-INFINITY)) return 0.0f;
        if (ax >= 0x7f800000) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double xd = (double)x;
    double kd = fma(InvLn2N, xd, SHIFT);
    uint64_t ki = asu(kd);
    kd -= SHIFT;
    double r = fma(InvLn2N, xd, -kd);
    uint64_t t = T[ki % 32];
    t += ki << (52 - 5);
    double s = asd(t);
    double zz = fma(C0, r, C1);
    volatile double r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    y = y * s;
    return (float)y;
}
int main() {
    long mism = 0, total = 0;
    #pragma omp parallel for reduction(+:mism,total) schedule(dynamic, 1<<20)
    for (long long i = 0; i < (1LL << 32); i++) {
        uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
        if (x != x) continue;
        float a = my_expf(x), b = expf(x);
        total++;
        if (asu32(a) != asu32(b)) { mism++; if (mism < 10) { printf("x=%a mine=%a libm=%a\n", x, a, b); } }
    }
    printf("checked %ld mismatches %ld\n", total, mism);
    return 0;
}
