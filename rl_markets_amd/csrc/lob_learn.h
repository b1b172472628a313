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
#include "lob_exp_table.h"

// Phase clocks of the wave-per-book kernels: -DLOB_PROF builds stamp clock64 at phase boundaries (lane 0
// of every wave adds the elapsed clocks to its book's row); a regular build compiles them away.
#ifdef LOB_PROF
struct Prof {
    long long t;
    i64* row;
    __device__ void start(i64* prof, int book, int lane) { row = (prof && lane == 0) ? prof + (size_t)book * LOB_PROF_N : nullptr; t = clock64(); }
    __device__ void mark(int i) {
        const long long n = clock64();
        if (row) row[i] += n - t;
        t = n;
    }
};
#else
struct Prof {
    __device__ void start(i64*, int, int) {}
    __device__ void mark(int) {}
};
#endif

#define LOB_QSTRIDE 33  // 32 tiles of one group + 1 pad double: lanes a=0..8 read column i without bank conflicts
#define LOB_HSLOTS 512  // per-wave LDS hash map (64-bit slots: tile index | rank) of the 288 "current" tiles

#include "lob_tiles.h"

// Wave form: `qv` holds, in lane i, the quantised variable i of the state (computed once per wave);
// the coordinates [first, first + nf) are read back as wave-uniform scalars.
template <int FIRST>
__device__ inline uint32_t tile_base_wave(uint32_t M, int qv, int nf, int j, const uint32_t* rndM) {
    uint32_t sum = 0;
    int base = j;
    for (int i = 0; i < nf; i++) {  // nf is wave-uniform
        const int q = __builtin_amdgcn_readlane(qv, FIRST + i);
        sum = mod_add(sum, rndM[(tile_coord(q, base) + 449 * i) & 2047], M);
        base += 2 * j;
    }
    return mod_add(sum, rndM[(j + 449 * nf) & 2047], M);
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
// std::exp(double) as the reference's libm computes it (glibc 2.35 x86-64, sysdeps/ieee754/dbl-64/e_exp.c, the FMA build its
// ifunc selects): exp(x) = 2^(k/128) * (1 + tail + r + r^2 (C2 + r C3) + r^4 (C4 + r C5)), 128-entry table, every a*b+c fused
// except inside the over/underflow special case.  Constants and table: lob_exp_table.h (the library's own __exp_data).
// tools/check_exp.c compares this restatement with libm's exp over 4e8 inputs (ordinary Q / tau, +-750, any bit pattern, near 0):
// 0 differences -- so a Boltzmann action, an index, is bit-exact by construction like every other action.
__device__ inline f64 exp_glibc(f64 x) {
    static const u64 T[256] = {LOB_EXP_TABLE_VALUES};
    const u64 ux = (u64)__double_as_longlong(x);
    uint32_t abstop = (uint32_t)(ux >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {  // top12(0x1p-54) = 0x3c9, top12(512.0) = 0x408
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;  // |x| < 2^-54
        if (abstop >= 0x409u) {                 // top12(1024.0)
            if (ux == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (ux >> 63) ? 0.0 : __longlong_as_double(0x7ff0000000000000ll);  // __math_uflow / __math_oflow
        }
        abstop = 0;  // large |x|: the special case below
    }
    const f64 z = LOB_EXP_INVLN2N * x;
    f64 kd = z + LOB_EXP_SHIFT;
    const u64 ki = (u64)__double_as_longlong(kd);
    kd -= LOB_EXP_SHIFT;
    const f64 r = fma(kd, LOB_EXP_NEGLN2LON, fma(kd, LOB_EXP_NEGLN2HIN, x));
    const u64 idx = 2 * (ki % 128), top = ki << (52 - 7);
    const f64 tail = __longlong_as_double((long long)T[idx]);
    u64 sbits = T[idx + 1] + top;
    const f64 r2 = r * r;
    const f64 p23 = fma(r, LOB_EXP_C3, LOB_EXP_C2), p45 = fma(r, LOB_EXP_C5, LOB_EXP_C4);
    const f64 tmp = fma(r2 * r2, p45, fma(r2, p23, tail + r));
    if (abstop == 0) {  // specialcase(): the result may over- or underflow
        if ((ki & 0x80000000ull) == 0) {  // k > 0
            sbits -= 1009ull << 52;
            const f64 scale = __longlong_as_double((long long)sbits);
            return 0x1p1009 * fma(scale, tmp, scale);
        }
        sbits += 1022ull << 52;  // k < 0: care in the subnormal range
        const f64 scale = __longlong_as_double((long long)sbits);
        f64 y = scale + scale * tmp;
        if (y < 1.0) {
            f64 lo = scale - y + scale * tmp;
            const f64 hi = 1.0 + y;
            lo = 1.0 - hi + y + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    const f64 scale = __longlong_as_double((long long)sbits);
    return fma(scale, tmp, scale);
}
// Boltzmann::Sample (policy.cpp:98-117): probabilities exp(Q / tau) / z, one uniform draw, first action whose
// cumulative probability exceeds it.
__device__ inline int boltzmann_sample(const f64* qs, f64 tau, Rng& g) {
    f64 p[LOB_N_ACTIONS], z = 0.0;
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        p[a] = exp_glibc(qs[a] / tau);
        z += p[a];
    }
    const f64 r = (f64)(g.raw() >> 11) * (1.0 / 9007199254740992.0);
    f64 acc = 0.0;
    int act = LOB_N_ACTIONS - 1;
    bool found = false;
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        acc += p[a] / z;
        if (!found && r < acc) { act = a; found = true; }
    }
    return act;
}
// The behaviour policy: EpsilonGreedy::Sample (policy.cpp:69-75) or Boltzmann; `greedy`: Agent::GoGreedy().
__device__ inline int policy_sample(const DevParams& P, const f64* qs, bool greedy, Rng& g) {
    if (!greedy) {
        if (P.policy == LOB_POLICY_BOLTZMANN) return boltzmann_sample(qs, P.tau, g);
        f64 u = (f64)(g.raw() >> 11) * (1.0 / 9007199254740992.0);
        if (u < P.epsilon) return (int)(((g.raw() >> 32) * 9ull) >> 32);
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

// ---- generation combining (lob_state.h) -------------------------------------
#define LOB_CB_EMPTY (~0ull)
#define LOB_CB_PROBES 64
#define LOB_CBS_VERIFIED (1 << 30) /* tr_cbslot: accumulate_kernel has compared the slot's identity with the generation's (cb_slots <= 2^24) */
__device__ inline u64 cb_hash(int q0, int q1, int q2, int code, uint32_t mask) {
    u64 h = lob_mix64((u64)(uint32_t)code ^ ((u64)mask << 32));
    h = lob_mix64(h + (u64)(uint32_t)q2);
    h = lob_mix64(h + (u64)(uint32_t)q1);
    h = lob_mix64(h + (u64)(uint32_t)q0);
    return h == LOB_CB_EMPTY ? 0 : h;
}
// One lane claims a slot for (signature, mask) unless somebody already has: the identity is written
// by the winner only and read by later kernels only, so no cross-wave publication is needed here.
// Slots PERSIST from step to step while some stepped book's generation adds to them.  The occupied slots are exactly those on
// the step's lists (cb_list[cb_par], one per segment of the table): the survivors of the previous step, then this step's new claims; apply_kernel adds the
// sums of the slots the step touched to theta (the tiles follow from the identity) and hands them on to the next step's list,
// and frees the others.  A generation whose (identity, mask) did not change keeps the slot it has on record (tr_cbslot)
// without asking again -- trace_sarsa_kernel; the wave-per-book kernels ask every step, and find the key there.  A freed slot
// cuts the probe sequences that ran through it, so an identity may come to hold two slots: both are summed and applied,
// which is the same update.
// The claim is split in two so that the CAS round trip overlaps the Q(s', .) evaluation:
// cb_claim_issue fires the first probe's CAS, cb_claim_finish (much later) looks at the answer,
// writes the identity if it won, and only then walks on along the probe sequence if it has to.
// A dense id for a slot just claimed (lob_state.h cb_dense): popped from this XCD's free list, -1 if that is empty.  Only claim
// winners pop (the trace / learn kernels), only apply_kernel pushes: the two never run together.
__device__ inline i32 cbd_pop(const DevState& S) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7u;
    const i32 have = atomicSub(&S.cb_free_n[2 * x], 1);
    if (have <= 0) { atomicAdd(&S.cb_free_n[2 * x], 1); return -1; }
    if (have - 1 < S.cb_free_n[2 * x + 1]) atomicMin(&S.cb_free_n[2 * x + 1], have - 1);  // (rare: only while the peak grows)
    return S.cb_free[(size_t)x * (LOB_CBD_CAP / 8) + have - 1];
}
struct CbPending {
    u64 h, old;
    uint32_t s, mask;
    int q0, q1, q2, code, src;
    bool active;
};
__device__ inline void cb_claim_issue(const DevState& S, CbPending& c, int q0, int q1, int q2, int code, uint32_t mask, int src) {
    c.q0 = q0; c.q1 = q1; c.q2 = q2; c.code = code; c.mask = mask; c.src = src; c.active = true;
    c.h = cb_hash(q0, q1, q2, code, mask);
    c.s = (uint32_t)c.h & (uint32_t)(S.cb_slots - 1);
    c.old = atomicCAS((unsigned long long*)&S.cb_key[c.s], (unsigned long long)LOB_CB_EMPTY, (unsigned long long)c.h);
}
// ... and records, for the generation that asked (`src` = book * trace_gens + ring slot), which slot carries its hash:
// accumulate_kernel goes straight there (it compares the identity in full: a slot taken by another identity with the same
// 64-bit hash, or no slot at all, sends the generation down the direct path).
__device__ inline void cb_claim_finish(const DevState& S, const CbPending& c) {
    if (!c.active) return;
    u64 old = c.old;
    uint32_t s = c.s;
    for (int probe = 0; probe < LOB_CB_PROBES; probe++) {
        if (old == LOB_CB_EMPTY) {
            i32* id = S.cb_ident + (size_t)s * 8;
            id[0] = c.q0; id[1] = c.q1; id[2] = c.q2; id[3] = c.code; id[4] = (i32)c.mask; id[5] = c.src;
            if (S.cb_dense_on) S.cb_dense[s] = cbd_pop(S);  // (off: the entry is -1 already -- apply_kernel leaves it so when it frees a slot)
            const int seg = S.cb_par * S.cb_segs + (int)(s & (uint32_t)(S.cb_segs - 1));
            const int pos = atomicAdd(&S.cb_count[seg], 1);
            S.cb_list[(size_t)seg * (S.cb_slots / S.cb_segs) + pos] = (i32)s;  // (an occupied slot is on its segment's list exactly once: it fits)
            S.tr_cbslot[c.src] = (i32)s;
            return;
        }
        if (old == c.h) { S.tr_cbslot[c.src] = (i32)s; return; }
        s = (s + 1) & (uint32_t)(S.cb_slots - 1);
        old = atomicCAS((unsigned long long*)&S.cb_key[s], (unsigned long long)LOB_CB_EMPTY, (unsigned long long)c.h);
    }
    S.tr_cbslot[c.src] = -1;  // table crowded
}


// Q(s, a) for all 9 actions of one state, one wave.
//   vars      : V floats of the state (LDS), ignored if `zero`
//   zero      : the rl::State still holds its constructor zeros (all tiles 0)
//   rnd       : LDS, the 2048-entry hash table reduced mod M
//   terms     : LDS, [3][9] table terms of the trailing (action code) coordinate, reduced mod M
//   vals      : per-wave LDS scratch [9][LOB_QSTRIDE] doubles (one tile GROUP at a time)
//   out_q[9]  : every lane returns all nine Q values
// The kernels that call this are VALU-issue bound (profiles/r01_pmc.csv: ~1 700-2 400 vector
// instructions per wave, the gathers themselves are a minor part), so the 96 tiles x 9 actions
// are spread over the 64 lanes without divergent halves:
//   pass A  lane l < 32: tiling l of group 0 (9 actions), lane 32 + l: tiling l of group 1;
//   pass B  tiling l of group 2, actions 0..4 on lanes 0-31 and 5..8 on lanes 32-63.
// The sum then follows the reference's sequential order term by term -- group 0 (w0),
// group 1 (w1), group 1 again and group 2 (w2): quirk Q3 -- on nine lanes (one per action),
// the products w * theta of one group staged through LDS at a time (the product is rounded
// before the add, as in the reference built without FMA), so that Q is bitwise the value
// Agent::getQ computes.
// `nz` is the "ever written" map of theta (lob_state.h): weights start at +0.0 and only
// group-0 tiles are ever updated (quirk Q4), so almost every group-1/2 gather would fetch a
// 64-byte sector from HBM to read a zero.  The map answers that from L2; the value used is
// bit-identical either way.  Group-0 tiles are fetched directly.
// vd_mode 0: look the map up; 1: look up AND return the verdicts in *vd_io (learn saves them:
// bits 0-8 pass A, bits 9-13 pass B); 2: take *vd_io as the verdicts (act re-using learn's),
// OR-ed with the filter `newf` (LDS, 4096 bits, keyed like the map itself) of the map bits set
// for the first time since.  A false positive only costs a fetch of a weight that is still +0.0.
__device__ inline void q_values(const DevParams& P, const f64* __restrict__ theta, const uint32_t* __restrict__ nz,
                                const f32* vars, bool zero, const uint32_t* rnd, const uint32_t* terms, f64* vals,
                                int lane, f64* out_q, int vd_mode = 0, uint32_t* vd_io = nullptr,
                                const uint32_t* newf = nullptr) {
    const int j = lane & 31;
    const bool hi = lane >= 32;
    const uint32_t M = (uint32_t)P.M;
    constexpr int NB = 5;  // pass-B actions per lane
    i32 iA[LOB_N_ACTIONS], iB[NB];
    {
        uint32_t baseA = 0, baseB = 0;
        if (!zero) {
            const int qv = tile_quant(vars[lane & 15]);  // lane i < V: quantised variable i (slots >= V hold 0)
            if (hi) baseA = tile_base_wave<3>(M, qv, P.V - 3, j, rnd);
            else baseA = tile_base_wave<0>(M, qv, 3, j, rnd);
            baseB = tile_base_wave<0>(M, qv, P.V, j, rnd);
        }
        const uint32_t* tA = terms + (hi ? LOB_N_ACTIONS : 0);
        const uint32_t* tB = terms + 2 * LOB_N_ACTIONS + (hi ? NB : 0);
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) iA[a] = zero ? 0 : tile_index(baseA, tA[a], M);
#pragma unroll
        for (int k = 0; k < NB; k++) iB[k] = zero ? 0 : tile_index(baseB, tB[k < 4 || !hi ? k : 0], M);
    }
    const uint32_t maskB = hi ? 0xfu : 0x1fu;  // lanes 32-63 own four actions, their fifth slot is idle
    uint32_t bA = 0x1ffu, bB;
    if (vd_mode == 2) {
        const uint32_t vd = *vd_io;
        bB = vd >> 9;
        uint32_t fA = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++)
            fA |= ((newf[LOB_NZ_WORD(iA[a]) & (LOB_NZ_FILTER - 1)] & LOB_NZ_BIT(iA[a])) ? 1u : 0u) << a;
#pragma unroll
        for (int k = 0; k < NB; k++)
            bB |= ((newf[LOB_NZ_WORD(iB[k]) & (LOB_NZ_FILTER - 1)] & LOB_NZ_BIT(iB[k])) ? 1u : 0u) << k;
        if (hi) bA = (vd & 0x1ffu) | fA;
    } else {
        uint32_t wA[LOB_N_ACTIONS], wB[NB];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) wA[a] = hi ? nz[LOB_NZ_WORD(iA[a])] : 0xffffffffu;  // group 0: fetched directly
#pragma unroll
        for (int k = 0; k < NB; k++) wB[k] = nz[LOB_NZ_WORD(iB[k])];
        bA = 0; bB = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) bA |= ((wA[a] & LOB_NZ_BIT(iA[a])) ? 1u : 0u) << a;
#pragma unroll
        for (int k = 0; k < NB; k++) bB |= ((wB[k] & LOB_NZ_BIT(iB[k])) ? 1u : 0u) << k;
        if (!hi) bA = 0x1ffu;
    }
    bB &= maskB;
    if (vd_mode == 1) *vd_io = (hi ? bA : 0u) | (bB << 9);
    f64 tA[LOB_N_ACTIONS], tB[NB];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        tA[a] = 0.0;
        if ((bA >> a) & 1u) tA[a] = theta[iA[a]];
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        tB[k] = 0.0;
        if ((bB >> k) & 1u) tB[k] = theta[iB[k]];
    }

    f64 q = 0.0;
    const f64* col = vals + lane * LOB_QSTRIDE;
    // ---- group 0 ----
    if (!hi) {
        const f64 w = P.w0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) vals[a * LOB_QSTRIDE + j] = w * tA[a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        for (int i = 0; i < 32; i++) q += col[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    // ---- group 1: once with w1, once more with w2 (quirk Q3) ----
    if (hi) {
        const f64 w = P.w1;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) vals[a * LOB_QSTRIDE + j] = w * tA[a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        for (int i = 0; i < 32; i++) q += col[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    if (hi) {
        const f64 w = P.w2;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) vals[a * LOB_QSTRIDE + j] = w * tA[a];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        for (int i = 0; i < 32; i++) q += col[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    // ---- group 2: both halves stage their actions ----
    {
        const f64 w = P.w2;
        const int a0 = hi ? NB : 0;
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (k < 4 || !hi) vals[(a0 + k) * LOB_QSTRIDE + j] = w * tB[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    if (lane < LOB_N_ACTIONS) {
        for (int i = 0; i < 32; i++) q += col[i];
    }
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) out_q[a] = __shfl(q, a);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}


// ---- group-0 memo (lob_state.h) ---------------------------------------------------------------------
#define LOB_MK_EMPTY (~0ull)
__device__ inline u64 mk_hash3(int q0, int q1, int q2) {
    u64 h = lob_mix64((u64)(uint32_t)q0 | ((u64)(uint32_t)q1 << 32));
    h = lob_mix64(h + (u64)(uint32_t)q2 * 0x9E3779B97F4A7C15ull);
    return h == LOB_MK_EMPTY ? 0 : h;
}
// env_kernel, one lane per book: find or make the slot of the book's new group-0 triple and put it on
// this step's list (once per slot and step).  The 64-bit hash picks and marks the slot; the identity
// is written by the claim winner only and compared in full by the readers in LATER kernels (no
// cross-wave publication inside this one).  -1: table crowded, the book takes the general path.
// `pre_k` / `pre_stamp`: mk_hash / mk_stamp of the triple's home slot, fetched by the caller ahead of time (together, and with
// other work in between: the claim is the last thing a book's step does, and looked up on the spot it is two dependent round
// trips at the tail of every wave).
__device__ inline int mk_claim(const DevState& S, int q0, int q1, int q2, int step_id, int par, u64 pre_k, int pre_stamp) {
    const u64 h = mk_hash3(q0, q1, q2);
    const uint32_t mask = (uint32_t)(S.mk_slots - 1);
    uint32_t s = (uint32_t)h & mask;
    if (pre_k == h && pre_stamp == step_id) return (int)s;  // claimed, and on this step's list already: the usual case
    for (int probe = 0; probe < LOB_MK_PROBES; probe++) {
        u64 k = probe == 0 ? pre_k : S.mk_hash[s];
        if (k == LOB_MK_EMPTY) {
            k = atomicCAS((unsigned long long*)&S.mk_hash[s], (unsigned long long)LOB_MK_EMPTY, (unsigned long long)h);
            if (k == LOB_MK_EMPTY) {
                *reinterpret_cast<int4*>(S.mk_ident + (size_t)s * 4) = make_int4(q0, q1, q2, 0);
                k = h;
            }
        }
        if (k == h) {
            if (S.mk_stamp[s] != step_id) {
                const int old = atomicExch(&S.mk_stamp[s], step_id);
                if (old != step_id) {
                    const int pos = atomicAdd(&S.mk_count[par], 1);
                    if (pos < S.mk_slots) S.mk_list[(size_t)par * S.mk_slots + pos] = (i32)s;
                }
            }
            return (int)s;
        }
        s = (s + 1) & mask;
    }
    return -1;
}

__device__ inline f64 readlane_f64(f64 x, int l) {  // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

struct MemoRec {  // one record of DevState::mk_rec
    f64 s0[LOB_N_ACTIONS];
    u64 ver;
};
static_assert(sizeof(MemoRec) == LOB_MK_REC * 8, "memo record layout");

// ---- std::mt19937_64 (the reference's Agent::gen, include/rl/agent.h:37; C++ standard
// [rand.predef]: w=64 n=312 m=156 r=31 a=0xB5026F5AA96619E9 u=29 d=0x5555555555555555 s=17
// b=0x71D67FFFEDA60000 t=37 c=0xFFF7EEE000000000 l=43 f=6364136223846793005) ----------------
#define LOB_MT_N 312
#define LOB_MT_M 156
__host__ __device__ inline void mt64_seed(u64* x, u64 seed) {
    x[0] = seed;
    for (int i = 1; i < LOB_MT_N; i++) x[i] = 6364136223846793005ull * (x[i - 1] ^ (x[i - 1] >> 62)) + (u64)i;
}
__host__ __device__ inline u64 mt64_mix(u64 xi, u64 xi1, u64 xm) {
    const u64 y = (xi & 0xFFFFFFFF80000000ull) | (xi1 & 0x7FFFFFFFull);
    return xm ^ (y >> 1) ^ ((y & 1ull) ? 0xB5026F5AA96619E9ull : 0ull);
}
// Regenerate the 312-word block, one wave, state staged in LDS (`lds`, >= 312 u64).
// The sequential recurrence reads x[i+1] (old) and x[i+156] (old for i < 156, new
// afterwards): three phases, every phase reads all its inputs before writing.
__device__ inline void mt64_twist_wave(u64* gstate, u64* lds, int lane) {
    for (int i = lane; i < LOB_MT_N; i += 64) lds[i] = gstate[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    u64 r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = lane + 64 * k;
        r[k] = i < LOB_MT_M ? mt64_mix(lds[i], lds[i + 1], lds[i + LOB_MT_M]) : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = lane + 64 * k;
        if (i < LOB_MT_M) lds[i] = r[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = LOB_MT_M + lane + 64 * k;
        r[k] = i < LOB_MT_N - 1 ? mt64_mix(lds[i], lds[i + 1], lds[i - LOB_MT_M]) : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = LOB_MT_M + lane + 64 * k;
        if (i < LOB_MT_N - 1) lds[i] = r[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) lds[LOB_MT_N - 1] = mt64_mix(lds[LOB_MT_N - 1], lds[0], lds[LOB_MT_M - 1]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < LOB_MT_N; i += 64) gstate[i] = lds[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}
__host__ __device__ inline u64 mt64_temper(u64 z) {
    z ^= (z >> 29) & 0x5555555555555555ull;
    z ^= (z << 17) & 0x71D67FFFEDA60000ull;
    z ^= (z << 37) & 0xFFF7EEE000000000ull;
    z ^= (z >> 43);
    return z;
}
// std::uniform_real_distribution<double>(0,1)(mt19937_64) in libstdc++ =
// generate_canonical<double,53>: one draw, double(u) / 2^64, clamped below 1.
__host__ __device__ inline f64 mt64_canonical(u64 u) {
    f64 r = (f64)u * 5.421010862427522e-20;  // 2^-64
    if (r >= 1.0) r = 0.99999999999999989;   // nextafter(1.0, 0.0)
    return r;
}

#endif
