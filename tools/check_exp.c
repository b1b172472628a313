// Check of the engine's exp(double) restatement (lob_learn.h exp_glibc, table lob_exp_table.h) against this libm's exp():
// random doubles over the ranges the policy can reach (Q / tau) and over the whole line, plus the special cases.
// gcc -O2 -fopenmp -mfma -ffp-contract=off -Irl_markets_amd/csrc -o check_exp tools/check_exp.c -lm && ./check_exp   (expected: mismatches 0)
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
#include "lob_exp_table.h"
static const uint64_t T[256] = { LOB_EXP_TABLE_VALUES };
static inline double asd(uint64_t u){ double d; memcpy(&d,&u,8); return d; }
static inline uint64_t asu(double d){ uint64_t u; memcpy(&u,&d,8); return u; }
static inline uint32_t top12(double x) { return asu(x) >> 52; }
// glibc e_exp.c specialcase(): the result may over- or underflow
static double specialcase(double tmp, uint64_t sbits, uint64_t ki) {
    double scale, y;
    if ((ki & 0x80000000) == 0) {  // k > 0: the exponent of scale might have overflowed by <= 460
        sbits -= 1009ull << 52;
        scale = asd(sbits);
        y = 0x1p1009 * fma(scale, tmp, scale);
        return y;
    }
    sbits += 1022ull << 52;  // k < 0: careful in the subnormal range
    scale = asd(sbits);
    y = scale + scale * tmp;  // (NOT fused in the library's build, unlike the two other places: found by this very comparison)
    if (y < 1.0) {
        double hi, lo;
        lo = scale - y + scale * tmp;
        hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;  // (no -0.0 with downward rounding: irrelevant in round-to-nearest)
    }
    return 0x1p-1022 * y;
}
static double my_exp(double x) {
    uint32_t abstop = top12(x) & 0x7ff;
    if (abstop - top12(0x1p-54) >= top12(512.0) - top12(0x1p-54)) {
        if (abstop - top12(0x1p-54) >= 0x80000000) return 1.0 + x;  // |x| < 2^-54 (or 0)
        if (abstop >= top12(1024.0)) {
            if (asu(x) == asu(-INFINITY)) return 0.0;
            if (abstop >= top12(INFINITY)) return 1.0 + x;
            if (asu(x) >> 63) return 0.0;       // __math_uflow
            return INFINITY;                    // __math_oflow
        }
        abstop = 0;  // large |x|: special-cased below
    }
    const double z = LOB_EXP_INVLN2N * x;
    double kd = z + LOB_EXP_SHIFT;
    const uint64_t ki = asu(kd);
    kd -= LOB_EXP_SHIFT;
    const double r = fma(kd, LOB_EXP_NEGLN2LON, fma(kd, LOB_EXP_NEGLN2HIN, x));
    const uint64_t idx = 2 * (ki % 128), top = ki << (52 - 7);
    const double tail = asd(T[idx]);
    const uint64_t sbits = T[idx + 1] + top;
    const double r2 = r * r;
    // tail + r + r2 * (C2 + r * C3) + r2 * r2 * (C4 + r * C5), as GCC contracts it for the FMA build
    const double p23 = fma(r, LOB_EXP_C3, LOB_EXP_C2), p45 = fma(r, LOB_EXP_C5, LOB_EXP_C4);
    const double tmp = fma(r2 * r2, p45, fma(r2, p23, tail + r));
    if (abstop == 0) return specialcase(tmp, sbits, ki);
    const double scale = asd(sbits);
    return fma(scale, tmp, scale);
}
static inline uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 200000000LL;
    long mism = 0, total = 0;
    #pragma omp parallel for reduction(+:mism,total) schedule(dynamic, 1<<18)
    for (long long i = 0; i < n; i++) {
        const uint64_t u = mix((uint64_t)i * 0x9E3779B97F4A7C15ull + 12345);
        double x;
        switch (i & 3) {
            case 0: x = ((double)(u >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 40.0; break;        // Q / tau of ordinary size
            case 1: x = ((double)(u >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1500.0; break;      // into over- and underflow
            case 2: x = asd(u); break;                                                                 // any bit pattern
            default: x = ((double)(u >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3; break;       // near 0
        }
        if (x != x) continue;
        const double a = my_exp(x), b = exp(x);
        total++;
        if (asu(a) != asu(b)) { mism++; if (mism < 10) printf("x=%a mine=%a libm=%a\n", x, a, b); }
    }
    printf("checked %ld mismatches %ld\n", total, mism);
    return mism != 0;
}
