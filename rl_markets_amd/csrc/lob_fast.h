// The fast learner path (shared theta, SARSA / Q(lambda)): persistent 16-wave blocks, one per CU.
//
// What bounds the per-book Q evaluation after the group-0 memo (lob_learn.h) is the 576 group-1/2
// "was this weight ever written" look-ups: divergent 4-byte gathers from the L2-resident map run
// at ~280 G lane-loads/s on the whole chip (tools/ubench/gather2: 37.7 M of them per launch =
// 135 us), LDS gathers at ~3 000 G/s.  So each CU keeps a COARSE image of the map in LDS -- one bit
// per 2^cshift consecutive weights, 78 KB at M = 20 M -- staged once per launch by a block that owns
// the CU's whole LDS and loops over books (which also amortises the 8 KB hash-table staging that
// cost the 4-wave blocks a third of their time).  Per tile:
//     LDS coarse bit   clear (~82 %)  -> weight is +0.0
//     exact map bit    (theta_nzx, one bit per weight, 2.5 MB, only for coarse hits)  clear -> +0.0
//     theta fetch      only for weights that really were written (a handful per book)
// and the 96 remaining terms of Agent::getQ are added to the memoised group-0 sum in the
// reference's order, skipping exact zeros.  Books without a valid memo record (first step of an
// episode, constructor-zero State, weights just loaded) are handed to the general kernels
// (lob_kernels.h) through a work list; they see exactly the state they would have seen.
#ifndef LOB_FAST_H
#define LOB_FAST_H

#include "lob_kernels.h"

#define LOB_FAST_WAVES 16
#define LOB_FAST_BLOCK (64 * LOB_FAST_WAVES)

// dynamic LDS image: [rnd 2048][act_terms 32][coarse cwords4 * 4][vars NW x 48 f32]([tab NW x LOB_HSLOTS u64])
__host__ __device__ inline size_t fast_lds_bytes(int cwords4, bool with_tab) {
    return (size_t)(2048 + 32 + cwords4 * 4) * 4 + (size_t)LOB_FAST_WAVES * 48 * 4 + (with_tab ? (size_t)LOB_FAST_WAVES * LOB_HSLOTS * 8 : 0);
}

struct FastLds {
    uint32_t* rnd;
    uint32_t* act_terms;
    uint32_t* coarse;
    f32* vars;  // this wave's row
    u64* tab;   // this wave's hash map (learn) or null
};

__device__ inline FastLds fast_stage(unsigned char* raw, const DevParams& P, const DevState& S, const uint32_t* __restrict__ rnd_g, bool with_tab) {
    FastLds L;
    L.rnd = reinterpret_cast<uint32_t*>(raw);
    L.act_terms = L.rnd + 2048;
    L.coarse = L.act_terms + 32;
    f32* vars_all = reinterpret_cast<f32*>(L.coarse + (size_t)P.cwords4 * 4);
    const int w = threadIdx.x >> 6;
    L.vars = vars_all + w * 48;
    L.tab = with_tab ? reinterpret_cast<u64*>(vars_all + LOB_FAST_WAVES * 48) + (size_t)w * LOB_HSLOTS : nullptr;
    {
        const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
        uint4* dst = reinterpret_cast<uint4*>(L.rnd);
        if (threadIdx.x < 512) dst[threadIdx.x] = src[threadIdx.x];
        if (threadIdx.x < 27) L.act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(S.theta_nzc);
        uint4* dst = reinterpret_cast<uint4*>(L.coarse);
        for (int i = threadIdx.x; i < P.cwords4; i += LOB_FAST_BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    return L;
}

// Action-independent hash sums of this lane's tiling: lane l < 32 tiling l of group 1 (state variables
// 3..V-1), lane 32 + l tiling l of group 2 (all V variables).  `qv`: lane i holds the quantised variable
// i.  The coordinates come through readlane (no LDS round trip), all table reads are in flight before the
// first is consumed; the reduced sum does not depend on the order of its terms.
__device__ __forceinline__ uint32_t fast_base(const DevParams& P, int qv, int lane, const uint32_t* rnd) {
    const int j = lane & 31;
    const bool hi = lane >= 32;
    const uint32_t M = (uint32_t)P.M;
    const int nf = hi ? P.V : P.V - 3;
    uint32_t t[LOB_MAX_VARS];
#pragma unroll
    for (int i = 0; i < LOB_MAX_VARS; i++) {
        t[i] = 0;
        if (i < P.V) {  // wave-uniform
            const int qa = __builtin_amdgcn_readlane(qv, i);                       // coordinate i of group 2
            const int qb = __builtin_amdgcn_readlane(qv, i + 3 < 16 ? i + 3 : 15);  // coordinate i of group 1 (unused from i = V - 3 on)
            t[i] = rnd[(tile_coord(hi ? qa : qb, j * (1 + 2 * i)) + 449 * i) & 2047];
        }
    }
    uint32_t sum = rnd[(j + 449 * nf) & 2047];
#pragma unroll
    for (int i = 0; i < LOB_MAX_VARS; i++) sum = mod_add(sum, i < nf ? t[i] : 0u, M);
    return sum;
}

// Q(s, .) for the nine actions, continued from the memoised group-0 sums `s0` (wave-uniform).
__device__ __forceinline__ void q_values_fast(const DevParams& P, const DevState& S, const FastLds& L, int qv, int lane, const f64* s0, f64* out_q,
                                     Prof& pf, int pf0) {
    const bool hi = lane >= 32;
    const uint32_t M = (uint32_t)P.M;
    const uint32_t sum = fast_base(P, qv, lane, L.rnd);
    pf.mark(pf0);  // tile hashing
    const uint32_t* tg = L.act_terms + (hi ? 2 * LOB_N_ACTIONS : LOB_N_ACTIONS);
    i32 idx[LOB_N_ACTIONS];
    uint32_t cw[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) idx[a] = tile_index(sum, tg[a], M);
    const int cs = P.cshift;
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) cw[a] = L.coarse[(uint32_t)idx[a] >> (cs + 5)];
    uint32_t maybe = 0;
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) maybe |= ((cw[a] >> (((uint32_t)idx[a] >> cs) & 31)) & 1u) << a;
    pf.mark(pf0 + 1);  // coarse filter (LDS)
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) out_q[a] = s0[a];
    if (__ballot(maybe != 0) == 0) { pf.mark(pf0 + 2); return; }
    uint32_t xw[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        xw[a] = 0;
        if ((maybe >> a) & 1u) xw[a] = S.theta_nzx[(uint32_t)idx[a] >> 5];
    }
    uint32_t hit = 0;
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) hit |= ((xw[a] >> ((uint32_t)idx[a] & 31)) & 1u) << a;
    const bool none = __ballot(hit != 0) == 0;
    pf.mark(pf0 + 2);  // exact map for the coarse hits
    if (none) return;
    f64 v[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        v[a] = 0.0;
        if ((hit >> a) & 1u) v[a] = S.theta[idx[a]];
    }
    const f64 w1 = P.w1, w2 = P.w2;
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) {
        const u64 m = __ballot(v[a] != 0.0);
        if (m == 0) continue;
        // group 1 with w1, group 1 again with w2 (quirk Q3), group 2 with w2: tilings in ascending order
        f64 q = out_q[a];
        for (uint32_t mm = (uint32_t)m; mm; mm &= mm - 1) q += w1 * readlane_f64(v[a], __builtin_ctz(mm));
        for (uint32_t mm = (uint32_t)m; mm; mm &= mm - 1) q += w2 * readlane_f64(v[a], __builtin_ctz(mm));
        for (uint32_t mm = (uint32_t)(m >> 32); mm; mm &= mm - 1) q += w2 * readlane_f64(v[a], 32 + __builtin_ctz(mm));
        out_q[a] = q;
    }
    pf.mark(pf0 + 3);  // written weights + ordered continuation
}

// Does the book's memo record (`which` 0: under theta_t, 1: after the last update) belong to the
// State whose quantised variables are in `qv`, and to the current weights?
__device__ inline bool fast_memo_ok(const DevState& S, int mslot, int which, u64 ver, int qv, MemoRec& rec) {
    const int ms = mslot >= 0 ? mslot : 0;
    const int4 mid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ms * 4);
    rec = *reinterpret_cast<const MemoRec*>(S.mk_rec + ((size_t)which * S.mk_slots + ms) * LOB_MK_REC);
    return mslot >= 0 && rec.ver == ver && mid.x == __builtin_amdgcn_readlane(qv, 0) && mid.y == __builtin_amdgcn_readlane(qv, 1) &&
           mid.z == __builtin_amdgcn_readlane(qv, 2);
}
__device__ inline void fast_hand_back(const DevState& S, int kind, int lpar, int b, int lane) {
    if (lane == 0) {
        const int pos = atomicAdd(&S.slow_n[lpar * 2 + kind], 1);
        S.slow_list[(size_t)kind * S.B + pos] = b;  // every book at most once per kernel: pos < B
    }
}

// Learner::_step / Backtester::_step prologue for every book (see act_book).  `lpar`: parity of the
// work lists of this step.
template <int ALGO>
__global__ void __launch_bounds__(LOB_FAST_BLOCK) act_fast_kernel(DevParams P, DevState S, const uint32_t* __restrict__ rnd_g, int mode,
                                                                  int par, int lpar, u64 ver) {
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S.cb_count[0] = 0;    // the previous step's apply_kernel has consumed the list
        S.mk_count[par] = 0;  // this step's env_kernel starts a new list of memo slots
        S.slow_n[(lpar ^ 1) * 2 + 0] = 0;  // the next step's work lists
        S.slow_n[(lpar ^ 1) * 2 + 1] = 0;
    }
    const FastLds L = fast_stage(fast_lds_raw, P, S, rnd_g, false);
    const int w = threadIdx.x >> 6;
    int lane_ = threadIdx.x & 63;
#pragma unroll 1
    for (int t = blockIdx.x * LOB_FAST_WAVES + w; t < S.B; t += gridDim.x * LOB_FAST_WAVES) {
        asm volatile("" : "+v"(lane_));  // nothing lane-dependent is carried (kept in registers) across books
        const int lane = lane_;
        const int b = __builtin_amdgcn_readfirstlane(t);
        const LHdr h = S.hdr[b];
        const int mslot = S.mk_slot[b];
        Prof pf;
        pf.start(S.prof, b, lane);
        learn_stage_vars(S.vars + (size_t)b * 48, L.vars, lane);
        LHdr* hp = S.hdr + b;
        if (h.done) { if (lane == 0) hp->stepped = 0; continue; }
        int cur = h.slot_cur;
        if (mode == 0) cur ^= 1;  // swap(state, last_state)
        if (!is_open(P, h.time_ms)) {  // environment.isTerminal()
            if (lane == 0) { hp->slot_cur = cur; hp->done = 1; hp->stepped = 0; S.done[b] = 1; }
            continue;
        }
        const int src = mode == 0 ? (cur ^ 1) : 2;
        const bool zero = mode == 0 && ((h.zero_mask >> src) & 1);
        const int qv = tile_quant(L.vars[src * 16 + (lane & 15)]);
        MemoRec rec;
        if (!fast_memo_ok(S, zero ? -1 : mslot, 1, ver, qv, rec)) { fast_hand_back(S, 0, lpar, b, lane); continue; }
        pf.mark(0);  // header, memo record, state variables
        f64 qs[LOB_N_ACTIONS];
        q_values_fast(P, S, L, qv, lane, rec.s0, qs, pf, 1);
        if (lane < LOB_N_ACTIONS) S.qs_last[(size_t)b * LOB_N_ACTIONS + lane] = qs[lane];
        Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
        const int action = policy_sample(qs, P.epsilon, mode == 1, g);
        if (lane == 0) {
            hp->slot_cur = cur;
            hp->action = action;
            hp->stepped = 1;
            hp->rng_ctr = g.ctr;
        }
        pf.mark(5);  // policy + stores
    }
}

// Agent::HandleTransition up to updateQ for every book that stepped (see learn_book).
template <int ALGO>
__global__ void __launch_bounds__(LOB_FAST_BLOCK) learn_fast_kernel(DevParams P, DevState S, const uint32_t* __restrict__ rnd_g, int par,
                                                                    int lpar, u64 ver) {
    static_assert(ALGO == LOB_ALGO_SARSA || ALGO == LOB_ALGO_QLAMBDA, "one weight vector");
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    // this step's update appends to nz_new[par]; the list the general act path reads is nz_new[par ^ 1]
    if (blockIdx.x == 0 && threadIdx.x < LOB_NZ_WORDS) {
        S.nz_new[par * LOB_NZ_WORDS + threadIdx.x] = 0;
        S.nz_new[(2 + par) * LOB_NZ_WORDS + threadIdx.x] = 0;
    }
    const FastLds L = fast_stage(fast_lds_raw, P, S, rnd_g, true);
    const int w = threadIdx.x >> 6;
    int lane_ = threadIdx.x & 63;
#pragma unroll 1
    for (int t = blockIdx.x * LOB_FAST_WAVES + w; t < S.B; t += gridDim.x * LOB_FAST_WAVES) {
        asm volatile("" : "+v"(lane_));
        const int lane = lane_;
        const int b = __builtin_amdgcn_readfirstlane(t);
        const LHdr h = S.hdr[b];
        const int mslot = S.mk_slot[b];
        f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = S.qs_last[(size_t)b * LOB_N_ACTIONS + a];
        if (!h.stepped) continue;
        Prof pf;
        pf.start(S.prof, b, lane);
        learn_stage_vars(S.vars + (size_t)b * 48, L.vars, lane);
        LHdr* hp = S.hdr + b;
        const int cur = h.slot_cur, last = cur ^ 1;
        const bool zero_last = (h.zero_mask >> last) & 1;
        const f32* vars_to = L.vars + cur * 16;
        const f32* vars_from = L.vars + last * 16;
        const int qv = tile_quant(vars_to[lane & 15]);
        MemoRec rec;
        if (!fast_memo_ok(S, mslot, 0, ver, qv, rec)) { fast_hand_back(S, 1, lpar, b, lane); continue; }  // before anything is modified
        pf.mark(8);  // header, Q(s, .), memo record, state variables
        Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
        CbPending pend;
        learn_traces<ALGO>(P, S, b, h, L.rnd, L.act_terms, L.tab, vars_from, zero_last, qs_last, g, lane, pend, pf);
        f64 qs_to[LOB_N_ACTIONS];
        q_values_fast(P, S, L, qv, lane, rec.s0, qs_to, pf, 13);
        learn_delta_single<ALGO>(P, hp, h, qs_to, qs_last, g, lane);
        pf.mark(17);  // argmax / delta / header stores
        cb_claim_finish(S, pend);
        pf.mark(18);  // claim finish
    }
}

#endif
