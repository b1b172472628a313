// Device-side learner pieces: CMAC tile coding, linear-Q evaluation, policy
// sampling and eligibility-trace maintenance, one WAVE (64 lanes) per book.
//   tiles()/hash_UNH        src/rl/tiles.cpp:31-75,130-169
//   State::populateFeatures src/rl/state.cpp:53-65
//   Agent::getQ/argmaxQ     src/rl/agent.cpp:117-169
//   Greedy/EpsilonGreedy    src/rl/policy.cpp:37-75
//   Traces                  src/rl/traces.cpp:30-102
#ifndef LOB_LEARN_H
#define LOB_LEARN_H

#include <hip/hip_runtime.h>

#include "lob_state.h"
#include "lob_stream.h"

#define LOB_QSTRIDE 33  // 32 tiles of one group + 1 pad double: lanes a=0..8 read column i without bank conflicts
#define LOB_HSLOTS 1024 // per-wave LDS hash set for the 288 "current" tiles

// S % M for S < 2^37, M < 2^31 through a double reciprocal (+-1 fix-up).
__device__ inline i32 mod_m(u64 s, i64 M, f64 inv_M) {
    i64 q = (i64)((f64)s * inv_M);
    i64 r = (i64)s - q * M;
    if (r < 0) r += M;
    if (r >= M) r -= M;
    return (i32)r;
}

// Sum of the table terms of tiling j of group g that do not depend on the
// action: the nf float coordinates and the tiling index (tiles.cpp:50-70).
// `v` = the group's float sub-array (State::populateFeatures passes
// &state_vars[0] or &state_vars[3]).
__device__ inline u64 tile_base(const f32* v, int nf, int j, const uint32_t* rnd) {
    u64 sum = 0;
    for (int i = 0; i < nf; i++) {
        int q = (int)floorf(v[i] * 32.0f);  // (int) floor(floats[i] * num_tilings)
        int base = j * (1 + 2 * i);
        int c;
        if (q >= base) c = q - ((q - base) % 32);
        else c = q + 1 + ((base - q - 1) % 32) - 32;
        sum += (u64)rnd[(c + 449 * i) & 2047];
    }
    sum += (u64)rnd[(j + 449 * nf) & 2047];
    return sum;
}
// table term of the trailing integer coordinate (the action code)
__device__ inline u64 tile_action_term(int nf, int code, const uint32_t* rnd) {
    return (u64)rnd[(code + 449 * (nf + 1)) & 2047];
}

// ---- policy RNG (counter-based; DESIGN.md "RNG") ----------------------------
struct Rng {
    u64 seed, stream, ctr;
    __device__ u64 raw() { return lob_rng(seed, stream, ctr++); }
    __device__ int rnd() { return (int)(raw() >> 33); }  // stands in for libc rand()
};

// Greedy::Sample (policy.cpp:37-55)
__device__ inline int greedy_sample(const f64* qs, Rng& g) {
    // (best value tracked in a register: no runtime-indexed array, which would live in scratch)
    int argmax = 0, n_ties = 1;
    f64 best = qs[0];
#pragma unroll
    for (int a = 1; a < LOB_N_ACTIONS; a++) {
        if (qs[a] > best) { argmax = a; best = qs[a]; }
        else if (qs[a] >= best) {
            n_ties++;
            if (0 == g.rnd() % n_ties) { argmax = a; best = qs[a]; }
        }
    }
    return argmax;
}
// EpsilonGreedy::Sample (policy.cpp:69-75)
__device__ inline int policy_sample(const f64* qs, f64 eps, bool greedy, Rng& g) {
    if (!greedy) {
        f64 u = (f64)(g.raw() >> 11) * (1.0 / 9007199254740992.0);
        if (u < eps) return (int)(((g.raw() >> 32) * 9ull) >> 32);
    }
    return greedy_sample(qs, g);
}
// Agent::argmaxQ (agent.cpp:144-169)
__device__ inline int argmax_ties(const f64* qs, Rng& g) {
    int index = 0, n_ties = 1;
    f64 cur = qs[0];
#pragma unroll
    for (int a = 1; a < LOB_N_ACTIONS; a++) {
        f64 val = qs[a];
        if (val >= cur) {
            if (val > cur) { cur = val; index = a; }
            else {
                n_ties++;
                if (0 == g.rnd() % n_ties) { cur = val; index = a; }
            }
        }
    }
    return index;
}

// Q(s, a) for all 9 actions of one state, one wave.
//   vars      : V floats of the state (LDS), ignored if `zero`
//   zero      : the rl::State still holds its constructor zeros (all tiles 0)
//   vals      : per-wave LDS scratch [9][LOB_QSTRIDE] doubles (one tile GROUP at a time)
//   out_q[9]  : every lane returns all nine Q values
// Lane l < 32 owns tiling l of groups 0 and 2, lane 32 + l owns tiling l of
// group 1: 13.5 gathers per lane, all issued before any is consumed.  The sum
// then follows the reference's sequential order term by term -- group 0 (w0),
// group 1 (w1), group 1 again and group 2 (w2): quirk Q3 -- on nine lanes (one
// per action), the 32 x 9 values of one group staged through LDS at a time, so
// that Q is bitwise the value Agent::getQ computes.
// `nz` is the "ever written" bitmap of theta (lob_state.h): weights start at +0.0
// and only group-0 tiles are ever updated (quirk Q4), so almost every group-1/2
// gather would fetch a 64-byte sector from HBM to read a zero.  One bit per
// weight (2.5 MB at M = 20 M: L2-resident) answers that without the fetch; the
// value used is bit-identical either way.
// vd_mode 0: look the bits up in the bitmap; 1: look up AND return them in *vd_bits (learn saves
// them); 2: take *vd_bits as the verdicts (act re-using learn's), OR-ed with the filter `newf` (LDS,
// 4096 bits, keyed like the map itself) of the map bits set for the first time since.  A filter
// false positive only costs a fetch of a weight that is still exactly 0.0.
__device__ inline void gather9(const DevParams& P, const f64* __restrict__ theta, const uint32_t* __restrict__ nz,
                               const f32* vars, bool zero, const uint32_t* rnd, const u64* act_terms, int g, int j,
                               f64* t, int vd_mode, uint32_t* vd_bits, const uint32_t* newf) {
    const int nf = g == 0 ? 3 : (g == 1 ? P.V - 3 : P.V);
    const f32* v = g == 1 ? vars + 3 : vars;
    const u64 base = zero ? 0 : tile_base(v, nf, j, rnd);
    i32 idx[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++)
        idx[a] = zero ? 0 : mod_m(base + act_terms[g * LOB_N_ACTIONS + a], P.M, P.inv_M);
    uint32_t bits;
    if (g == 0) {
        bits = 0x1ffu;  // group-0 tiles are the ones that get written: fetch them directly (one request, not two)
    } else if (vd_mode == 2) {
        bits = *vd_bits;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++)
            bits |= ((newf[LOB_NZ_WORD(idx[a]) & (LOB_NZ_FILTER - 1)] & LOB_NZ_BIT(idx[a])) ? 1u : 0u) << a;
    } else {
        uint32_t word[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) word[a] = nz[LOB_NZ_WORD(idx[a])];
        bits = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) bits |= ((word[a] & LOB_NZ_BIT(idx[a])) ? 1u : 0u) << a;
        if (vd_mode == 1) *vd_bits = bits;
    }
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        t[a] = 0.0;
        if ((bits >> a) & 1u) t[a] = theta[idx[a]];
    }
}

// vd_io: this lane's 9 verdict bits (lanes 32-63: group 1, lanes 0-31: group 2; row index lane ^ 32).
__device__ inline void q_values(const DevParams& P, const f64* __restrict__ theta, const uint32_t* __restrict__ nz,
                                const f32* vars, bool zero,
                                const uint32_t* rnd, const u64* act_terms /*[3][9] LDS*/, f64* vals, int lane,
                                f64* out_q, int vd_mode = 0, uint32_t* vd_io = nullptr, const uint32_t* newf = nullptr) {
    const int j = lane & 31, hi = lane >> 5;
    f64 ta[LOB_N_ACTIONS], tb[LOB_N_ACTIONS];
    uint32_t bits = 0;
    if (vd_mode == 2) bits = *vd_io;
    if (hi) {
        gather9(P, theta, nz, vars, zero, rnd, act_terms, 1, j, ta, vd_mode, &bits, newf);
    } else {
        gather9(P, theta, nz, vars, zero, rnd, act_terms, 0, j, ta, 0, nullptr, nullptr);
        gather9(P, theta, nz, vars, zero, rnd, act_terms, 2, j, tb, vd_mode, &bits, newf);
    }
    if (vd_mode == 1) *vd_io = bits;
    f64 q = 0.0;
    const f64* col = vals + lane * LOB_QSTRIDE;
    // ---- group 0 ----
    if (!hi) {
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) vals[a * LOB_QSTRIDE + j] = ta[a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        const f64 w = P.w0;
        for (int i = 0; i < 32; i++) q += w * col[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- group 1: once with w1, once more with w2 (quirk Q3) ----
    if (hi) {
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) vals[a * LOB_QSTRIDE + j] = ta[a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        f64 w = P.w1;
        for (int i = 0; i < 32; i++) q += w * col[i];
        w = P.w2;
        for (int i = 0; i < 32; i++) q += w * col[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- group 2 ----
    if (!hi) {
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) vals[a * LOB_QSTRIDE + j] = tb[a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        const f64 w = P.w2;
        for (int i = 0; i < 32; i++) q += w * col[i];
    }
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) out_q[a] = __shfl(q, a);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}


// ---- std::mt19937_64 (the reference's Agent::gen, include/rl/agent.h:37; C++ standard
// [rand.predef]: w=64 n=312 m=156 r=31 a=0xB5026F5AA96619E9 u=29 d=0x5555555555555555 s=17
// b=0x71D67FFFEDA60000 t=37 c=0xFFF7EEE000000000 l=43 f=6364136223846793005) ----------------
#define LOB_MT_N 312
#define LOB_MT_M 156
__host__ __device__ inline void mt64_seed(u64* x, u64 seed) {
    x[0] = seed;
    for (int i = 1; i < LOB_MT_N; i++) x[i] = 6364136223846793005ull * (x[i - 1] ^ (x[i - 1] >> 62)) + (u64)i;
}
__device__ inline u64 mt64_mix(u64 xi, u64 xi1, u64 xm) {
    const u64 y = (xi & 0xFFFFFFFF80000000ull) | (xi1 & 0x7FFFFFFFull);
    return xm ^ (y >> 1) ^ ((y & 1ull) ? 0xB5026F5AA96619E9ull : 0ull);
}
// Regenerate the 312-word block, one wave, state staged in LDS (`lds`, >= 312 u64).
// The sequential recurrence reads x[i+1] (old) and x[i+156] (old for i < 156, new
// afterwards): three phases, every phase reads all its inputs before writing.
__device__ inline void mt64_twist_wave(u64* gstate, u64* lds, int lane) {
    for (int i = lane; i < LOB_MT_N; i += 64) lds[i] = gstate[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    u64 r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = lane + 64 * k;
        r[k] = i < LOB_MT_M ? mt64_mix(lds[i], lds[i + 1], lds[i + LOB_MT_M]) : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = lane + 64 * k;
        if (i < LOB_MT_M) lds[i] = r[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = LOB_MT_M + lane + 64 * k;
        r[k] = i < LOB_MT_N - 1 ? mt64_mix(lds[i], lds[i + 1], lds[i - LOB_MT_M]) : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = LOB_MT_M + lane + 64 * k;
        if (i < LOB_MT_N - 1) lds[i] = r[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) lds[LOB_MT_N - 1] = mt64_mix(lds[LOB_MT_N - 1], lds[0], lds[LOB_MT_M - 1]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < LOB_MT_N; i += 64) gstate[i] = lds[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline u64 mt64_temper(u64 z) {
    z ^= (z >> 29) & 0x5555555555555555ull;
    z ^= (z << 17) & 0x71D67FFFEDA60000ull;
    z ^= (z << 37) & 0xFFF7EEE000000000ull;
    z ^= (z >> 43);
    return z;
}
// std::uniform_real_distribution<double>(0,1)(mt19937_64) in libstdc++ =
// generate_canonical<double,53>: one draw, double(u) / 2^64, clamped below 1.
__device__ inline f64 mt64_canonical(u64 u) {
    f64 r = (f64)u * 5.421010862427522e-20;  // 2^-64
    if (r >= 1.0) r = 0.99999999999999989;   // nextafter(1.0, 0.0)
    return r;
}

#endif
