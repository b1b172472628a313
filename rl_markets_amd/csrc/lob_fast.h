// The fast learner path (shared theta, SARSA / Q(lambda)).  Kernels of one step, in launch order:
//     act_light_book          action selection from the hit list the previous step's learn kernel left (a lane per book), inside
//                             env_kernel<.., 1> (lob_kernels.h) or as act_light_kernel; act_fast_kernel, the full evaluation
//                             (a wave per book), when there is no valid list
//     [env_kernel, memo_kernel: lob_kernels.h]
//     learn_q_pair_kernel     Q(s', .), TD error, hit list -- two lanes per book -- and, Q(lambda), the trace step of the books
//                             whose step leaves no older generation; learn_q_lane_kernel (one lane per book) for tables of
//                             2^27 weights and more, learn_q_fast_kernel (a wave per book) for batches below 32 768 books
//     trace_fast_kernel       the remaining trace steps, a wave per book (all of them for SARSA(lambda));
//                             trace_light_kernel: the light trace step as its own kernel when the learn kernel is the wave one
//     [accumulate_kernel, apply_kernel, memo_kernel: lob_kernels.h]
//
// The wave-per-book Q evaluation (act_fast_kernel, learn_q_fast_kernel): persistent 16-wave blocks, one per CU.
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
/* trace_fast_kernel's block: waves per block, and the waves per SIMD its register budget is cut for.  The
 * kernel wants 79 VGPRs: at 8 waves per SIMD (64 VGPRs) it spilled 72 bytes per lane.  Measured, Q(lambda),
 * ms per launch: 16 waves x 2 blocks per CU (8 / SIMD) 0.115, 12 x 2 (6 / SIMD) 0.110, 8 x 3 (6 / SIMD) 0.105,
 * 4 x 6 (6 / SIMD) 0.105, 6 x 4 (6 / SIMD, waves not a multiple of the 4 SIMDs) 0.137, 4 x 7 0.133, 14 x 2 0.138. */
#ifndef LOB_TRACE_WAVES
#define LOB_TRACE_WAVES 8
#endif
#ifndef LOB_TRACE_OCC
#define LOB_TRACE_OCC 6
#endif
#define LOB_TRACE_BLOCK (64 * LOB_TRACE_WAVES)

#ifndef LOB_FAST_NB
#define LOB_FAST_NB 2  /* books a wave of the Q kernels takes through the stages together */
#endif
// dynamic LDS image: [rnd 2048][act_terms 32][coarse cwords4 * 4][vars NW x rows x 48 f32]([tab NW x LOB_HSLOTS u64])
__host__ __device__ inline size_t fast_lds_bytes(int cwords4, int rows, bool with_tab) {
    return (size_t)(2048 + 32 + cwords4 * 4) * 4 + (size_t)LOB_FAST_WAVES * rows * 48 * 4 + (with_tab ? (size_t)LOB_FAST_WAVES * LOB_HSLOTS * 8 : 0);
}

struct FastLds {
    uint32_t* rnd;
    uint32_t* act_terms;
    uint32_t* coarse;
    f32* vars;  // this wave's row
    u64* tab;   // this wave's hash map (learn) or null
};

__device__ inline FastLds fast_stage(unsigned char* raw, const DevParams& P, const DevState& S, const uint32_t* __restrict__ rnd_g, int rows, bool with_tab) {
    FastLds L;
    L.rnd = reinterpret_cast<uint32_t*>(raw);
    L.act_terms = L.rnd + 2048;
    L.coarse = L.act_terms + 32;
    f32* vars_all = reinterpret_cast<f32*>(L.coarse + (size_t)P.cwords4 * 4);
    const int w = threadIdx.x >> 6;
    L.vars = vars_all + w * rows * 48;
    L.tab = with_tab ? reinterpret_cast<u64*>(vars_all + LOB_FAST_WAVES * rows * 48) + (size_t)w * LOB_HSLOTS : nullptr;
    {
        const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
        uint4* dst = reinterpret_cast<uint4*>(L.rnd);
        if (threadIdx.x < 512) dst[threadIdx.x] = src[threadIdx.x];
        if (threadIdx.x < 27) L.act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(+S.theta_nzc);
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

// Q(s, .) for the nine actions of NB books at once, continued from the memoised group-0 sums (wave-uniform).
// The books of a batch go through every stage together, so that a stage's LDS reads / map words / weights
// of ALL of them are in flight before the first is consumed: the kernel is a chain of dependent
// look-ups per book and a CU holds only 16 waves (the LDS image), so the parallelism has to come from
// inside the wave.  No branches on per-book conditions in here: a book that is not `go` computes on
// whatever its (valid) inputs are and its result is ignored by the caller.
// WR: also put each book's hit list (lob_state.h) -- the additions made below, in their order -- into the wave's LDS
// rows `buf` (NB x LOB_HL_ROW u64: the staged state variables are not needed any more) and return the counts
// (`o_cnt[k]`, above LOB_HL_CAP: no list); hit_list_store sends them to memory.  Order of the additions for action a:
// group-1 tilings ascending with w1, the same again with w2, group-2 tilings ascending with w2 -- whether or not the
// weight is non-zero YET: it is marked, the next update may write it.
template <int NB, bool WR>
__device__ __forceinline__ void q_values_fast(const DevParams& P, const DevState& S, const FastLds& L, const int* qv, int lane,
                                              const MemoRec* rec, f64 (*out_q)[LOB_N_ACTIONS], Prof& pf, int pf0,
                                              u64* buf = nullptr, int* o_cnt = nullptr) {
    const bool hi = lane >= 32;
    const uint32_t M = (uint32_t)P.M;
    uint32_t sum[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) sum[k] = fast_base(P, qv[k], lane, L.rnd);
    pf.mark(pf0);  // tile hashing
    const uint32_t* tg = L.act_terms + (hi ? 2 * LOB_N_ACTIONS : LOB_N_ACTIONS);
    uint32_t term[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) term[a] = tg[a];
    i32 idx[NB][LOB_N_ACTIONS];
    uint32_t cw[NB][LOB_N_ACTIONS];
    const int cs = P.cshift;
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            idx[k][a] = tile_index(sum[k], term[a], M);
            cw[k][a] = L.coarse[(uint32_t)idx[k][a] >> (cs + 5)];
        }
    uint32_t maybe[NB];
    bool any_maybe = false;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        maybe[k] = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) maybe[k] |= ((cw[k][a] >> (((uint32_t)idx[k][a] >> cs) & 31)) & 1u) << a;
        any_maybe |= maybe[k] != 0;
    }
    pf.mark(pf0 + 1);  // coarse filter (LDS)
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) out_q[k][a] = rec[k].s0[a];
    if (WR) {
#pragma unroll
        for (int k = 0; k < NB; k++) o_cnt[k] = 0;
    }
    if (__ballot(any_maybe) == 0) { pf.mark(pf0 + 2); return; }
    uint32_t hit[NB];
    bool any_hit = false;
    {
        uint32_t xw[NB][LOB_N_ACTIONS];
#pragma unroll
        for (int k = 0; k < NB; k++)
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) {
                xw[k][a] = 0;
                if ((maybe[k] >> a) & 1u) xw[k][a] = S.theta_nzx[(uint32_t)idx[k][a] >> 5];
            }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            hit[k] = 0;
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) hit[k] |= ((xw[k][a] >> ((uint32_t)idx[k][a] & 31)) & 1u) << a;
            any_hit |= hit[k] != 0;
        }
    }
    const bool none = __ballot(any_hit) == 0;
    pf.mark(pf0 + 2);  // exact map for the coarse hits
    if (none) return;
    // A lane rarely has more than one written weight among its 9 x NB tiles: fetch the first two of each
    // book up front (in flight together), anything beyond that on demand.
    const f64 w1 = P.w1, w2 = P.w2;
    f64 v0[NB], v1[NB];
    int a0[NB], a1[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
        a0[k] = hit[k] ? __builtin_ctz(hit[k]) : -1;
        const uint32_t r = hit[k] & (hit[k] - 1);
        a1[k] = r ? __builtin_ctz(r) : -1;
        i32 i0 = 0, i1 = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) { i0 = a0[k] == a ? idx[k][a] : i0; i1 = a1[k] == a ? idx[k][a] : i1; }
        v0[k] = 0.0; v1[k] = 0.0;
        if (a0[k] >= 0) v0[k] = S.theta[i0];
        if (a1[k] >= 0) v1[k] = S.theta[i1];
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        if (__ballot(hit[k] != 0) == 0) continue;
        int cnt = 0;  // (wave-uniform) additions listed so far
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            const bool mine = (hit[k] >> a) & 1u;
            const u64 mb = __ballot(mine);
            if (mb == 0) continue;
            if (WR) {
                const uint32_t lo = (uint32_t)mb, hi_m = (uint32_t)(mb >> 32);
                const int nlo = __builtin_popcount(lo), nhi = __builtin_popcount(hi_m);
                // entries below this lane's among the action's: lanes < 32 come first (twice), then lanes >= 32
                const int below = (int)__builtin_amdgcn_mbcnt_hi(hi_m, __builtin_amdgcn_mbcnt_lo(lo, 0u));  // set bits of mb below this lane
                const int p = cnt + (hi ? nlo + below : below);
                if (mine) {
                    u64* row = buf + k * LOB_HL_ROW + 1;
                    const u64 ent = (u64)(uint32_t)idx[k][a] | ((u64)a << 32);
                    if (p < LOB_HL_CAP) row[p] = hi ? ent | (1ull << 36) : ent;
                    if (!hi && p + nlo < LOB_HL_CAP) row[p + nlo] = ent | (1ull << 36);
                }
                cnt += 2 * nlo + nhi;
            }
            f64 v = 0.0;
            if (mine) v = a0[k] == a ? v0[k] : (a1[k] == a ? v1[k] : S.theta[idx[k][a]]);
            const u64 m = __ballot(v != 0.0);
            if (m == 0) continue;
            // group 1 with w1, group 1 again with w2 (quirk Q3), group 2 with w2: tilings in ascending order
            f64 q = out_q[k][a];
            for (uint32_t mm = (uint32_t)m; mm; mm &= mm - 1) q += w1 * readlane_f64(v, __builtin_ctz(mm));
            for (uint32_t mm = (uint32_t)m; mm; mm &= mm - 1) q += w2 * readlane_f64(v, __builtin_ctz(mm));
            for (uint32_t mm = (uint32_t)(m >> 32); mm; mm &= mm - 1) q += w2 * readlane_f64(v, 32 + __builtin_ctz(mm));
            out_q[k][a] = q;
        }
        if (WR) o_cnt[k] = cnt;
    }
    pf.mark(pf0 + 3);  // written weights + ordered continuation
}

// The hit lists q_values_fast<NB, true> left in the wave's LDS rows leave as ONE coalesced store of at most 192 bytes
// per book; `wr_b[k]`: the book, or -1.
#define LOB_HL_NONE (~0ull)
template <int NB>
__device__ __forceinline__ void hit_list_store(const DevState& S, int lane, const int* cnt, const int* wr_b, const u64* buf) {
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NB; k++) {
        if (wr_b[k] < 0) continue;  // wave-uniform
        const int n = cnt[k] <= LOB_HL_CAP ? cnt[k] : -1;
        if (lane <= n || lane == 0) S.hl_rec[(size_t)wr_b[k] * LOB_HL_REC + lane] = lane == 0 ? (n < 0 ? LOB_HL_NONE : (u64)n) : buf[k * LOB_HL_ROW + lane];
    }
}

// Does the memo record `rec` of slot `mslot` (`ident` its triple) belong to the State whose quantised
// variables are in `qv`, and to the current weights?
__device__ inline bool fast_memo_ok(int mslot, const int4& mid, const MemoRec& rec, u64 ver, int qv) {
    return mslot >= 0 && rec.ver == ver && mid.x == __builtin_amdgcn_readlane(qv, 0) && mid.y == __builtin_amdgcn_readlane(qv, 1) &&
           mid.z == __builtin_amdgcn_readlane(qv, 2);
}
__device__ inline void fast_hand_back(const DevState& S, int kind, int lpar, int b, int lane) {
    if (lane == 0) {
        const int pos = atomicAdd(&S.slow_n[lpar * 2 + kind], 1);
        S.slow_list[(size_t)kind * S.B + pos] = b;  // every book at most once per kernel: pos < B
    }
}

// Learner::_step / Backtester::_step prologue for every book (see act_book), NB books of a wave at a time.
// `lpar`: parity of the work lists of this step.
template <int NB>
__global__ void __launch_bounds__(LOB_FAST_BLOCK) act_fast_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g, int mode,
                                                                  int par, int lpar, u64 ver) {
    LOB_PS_REFS
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S.mk_count[par] = 0;  // this step's env_kernel starts a new list of memo slots
        S.slow_n[(lpar ^ 1) * 2 + 0] = 0;  // the next step's work lists
        S.slow_n[(lpar ^ 1) * 2 + 1] = 0;
        S.tr_list_n[lpar ^ 1] = 0;
        S.tr_list2_n[lpar ^ 1] = 0;
        S.acc_list_n[lpar ^ 1] = 0;
    }
    const FastLds L = fast_stage(fast_lds_raw, P, S, rnd_g, NB, false);
    const int w = threadIdx.x >> 6;
    const int stride = gridDim.x * LOB_FAST_WAVES;
    int lane_ = threadIdx.x & 63;
#pragma unroll 1
    for (int t0 = blockIdx.x * LOB_FAST_WAVES + w; t0 < S.B; t0 += stride * NB) {
        asm volatile("" : "+v"(lane_));  // nothing lane-dependent is carried (kept in registers) across iterations
        const int lane = lane_;
        int b[NB], mslot[NB];
        bool go[NB];
        LHdr h[NB];
        f32 vv[NB];
        Prof pf;
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int t = __builtin_amdgcn_readfirstlane(t0 + k * stride);
            go[k] = t < S.B;
            b[k] = go[k] ? t : 0;
            h[k] = S.hdr[b[k]];
            mslot[k] = S.mk_slot[b[k]];
            vv[k] = lane < 48 ? S.vars[(size_t)b[k] * 48 + lane] : 0.0f;
        }
        pf.start(S.prof, b[0], lane);
        wave_lds_fence();  // the previous batch's readers are done with the rows
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (lane < 48) L.vars[k * 48 + lane] = vv[k];
        wave_lds_fence();
        int cur[NB], qv[NB];
        int4 mid[NB];
        MemoRec rec[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            LHdr* hp = S.hdr + b[k];
            cur[k] = h[k].slot_cur ^ (mode == 0 ? 1 : 0);  // swap(state, last_state)
            if (go[k] && h[k].done) { if (lane == 0) hp->stepped = 0; go[k] = false; }
            if (go[k] && !is_open(P, h[k].time_ms)) {  // environment.isTerminal()
                if (lane == 0) { hp->slot_cur = cur[k]; hp->done = 1; hp->stepped = 0; S.done[b[k]] = 1; }
                go[k] = false;
            }
            // the State the action is computed from: last_state (learner) / latest getState() (backtester)
            const int src = mode == 0 ? (cur[k] ^ 1) : 2;
            const bool zero = mode == 0 && ((h[k].zero_mask >> src) & 1);
            if (zero) mslot[k] = -1;
            qv[k] = tile_quant(L.vars[k * 48 + src * 16 + (lane & 15)]);
            const int ms = mslot[k] >= 0 ? mslot[k] : 0;
            mid[k] = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ms * 4);
            rec[k] = *reinterpret_cast<const MemoRec*>(S.mk_rec + ((size_t)S.mk_slots + ms) * LOB_MK_REC);  // [1]: after the last update
        }
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (go[k] && !fast_memo_ok(mslot[k], mid[k], rec[k], ver, qv[k])) { fast_hand_back(S, 0, lpar, b[k], lane); go[k] = false; }
        pf.mark(0);  // header, memo record, state variables
        f64 qs[NB][LOB_N_ACTIONS];
        q_values_fast<NB, false>(P, S, L, qv, lane, rec, qs, pf, 1);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (!go[k]) continue;
            if (lane < LOB_N_ACTIONS) S.qs_last[(size_t)b[k] * LOB_N_ACTIONS + lane] = sel9(qs[k], lane);
            Rng g{P.seed, P.book_id_offset + (u64)b[k], h[k].rng_ctr};
            const int action = policy_sample(P, qs[k], mode == 1, g);
            if (lane == 0) {
                LHdr* hp = S.hdr + b[k];
                hp->slot_cur = cur[k];
                hp->action = action;
                hp->stepped = 1;
                hp->rng_ctr = g.ctr;
            }
            // (as act_light_kernel: the chosen action's 32 group-0 tiles are marked before the learn kernel looks)
            if (mode == 0 && S.mk_tiles_ok[mslot[k]] && !((S.mk_marked[mslot[k]] >> action) & 1u)) {
                if (lane < 32) nzx_mark(P, S, S.mk_tiles[((size_t)mslot[k] * LOB_N_ACTIONS + action) * 32 + lane]);
                if (lane == 0) atomicOr(&S.mk_marked[mslot[k]], 1u << action);
            }
        }
        pf.mark(5);  // policy + stores
    }
}

// One lane marks the 32 group-0 tiles of (memo slot, action) in the written-weights maps (nzx_mark): all tile
// loads, then all map words, in flight together -- a lane alone would otherwise walk 32 dependent round trips.
__device__ __forceinline__ void mark_generation(const DevParams& P, const DevState& S, int slot, int action) {
    const int4* src = reinterpret_cast<const int4*>(S.mk_tiles + ((size_t)slot * LOB_N_ACTIONS + action) * 32);
    int4 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = src[i];
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        w[4 * i + 0] = S.theta_nzx[(uint32_t)t[i].x >> 5]; w[4 * i + 1] = S.theta_nzx[(uint32_t)t[i].y >> 5];
        w[4 * i + 2] = S.theta_nzx[(uint32_t)t[i].z >> 5]; w[4 * i + 3] = S.theta_nzx[(uint32_t)t[i].w >> 5];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const i32 f[4] = {t[i].x, t[i].y, t[i].z, t[i].w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (!((w[4 * i + k] >> ((uint32_t)f[k] & 31)) & 1u)) nzx_mark(P, S, f[k]);
    }
}

// The same prologue when every book still has the hit list the previous step's learn_q_fast_kernel left for this very
// State (lob_state.h): Q(s, a) = the memoised group-0 sum under the new weights + the listed additions, in the listed
// order.  No tile hashing, no map, no LDS image: one LANE per book.  A book without a (valid) list, whose memo record is
// not of this weight version, or -- never in the steady state -- every book after an update set a map bit that the trace
// kernel had not (`hl_dirty`), goes to the general kernel through the work list, like act_fast_kernel's hand-backs.
#define LOB_LIGHT_BLOCK 256
// One book's part of it (one lane): false if the book does not step now -- done, terminal, or handed to the general kernel
// through the act work list (then its header is left alone) --, else the action (header and Q(s, .) stored).
__device__ inline bool act_light_book(const DevParams& P, const DevState& S, int b, const LHdr& h, int lpar, u64 ver, bool dirty, int& action) {
    LHdr* hp = S.hdr + b;
    // the first 64 bytes of the book's list -- the count and seven entries, the usual case in full -- with the header
    const ulonglong2* lp = reinterpret_cast<const ulonglong2*>(S.hl_rec + (size_t)b * LOB_HL_REC);
    const ulonglong2 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];
    const int mslot = S.mk_slot[b];
    const int cur = h.slot_cur ^ 1;  // swap(state, last_state)
    action = 0;
    if (h.done) { hp->stepped = 0; return false; }
    if (!is_open(P, h.time_ms)) {  // environment.isTerminal()
        hp->slot_cur = cur; hp->done = 1; hp->stepped = 0; S.done[b] = 1;
        return false;
    }
    const int n = l0.x == LOB_HL_NONE ? -1 : (int)l0.x;
    bool ok = !dirty && !((h.zero_mask >> (cur ^ 1)) & 1) && mslot >= 0 && n >= 0;
    f64 q[LOB_N_ACTIONS];
    uint32_t marked = 0x1ffu;
    if (ok) {
        // (the memo record and the slot's mark bits leave together)
        const MemoRec rec = *reinterpret_cast<const MemoRec*>(S.mk_rec + ((size_t)S.mk_slots + mslot) * LOB_MK_REC);  // [1]: after the last update
        const int tiles_ok = S.mk_tiles_ok[mslot];
        const uint32_t mk = S.mk_marked[mslot];
        marked = tiles_ok ? mk : 0x1ffu;
        ok = rec.ver == ver;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) q[a] = rec.s0[a];
    }
    if (!ok) {
        const int pos = atomicAdd(&S.slow_n[lpar * 2 + 0], 1);
        S.slow_list[pos] = b;
        return false;
    }
    const f64 w1 = P.w1, w2 = P.w2;
#define LOB_LIGHT_ADD(ENT, V)                                                              \
    if ((V) != 0.0) { /* (+0.0 added to a sum that is never -0.0) */                       \
        const int a_ = (int)((ENT) >> 32) & 15;                                            \
        const f64 x_ = (((ENT) >> 36) & 1ull ? w2 : w1) * (V);                             \
        _Pragma("unroll") for (int c = 0; c < LOB_N_ACTIONS; c++) q[c] = a_ == c ? q[c] + x_ : q[c]; \
    }
    {
        f64 v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0, v6 = 0.0;
        if (0 < n) v0 = S.theta[(uint32_t)l0.y];
        if (1 < n) v1 = S.theta[(uint32_t)l1.x];
        if (2 < n) v2 = S.theta[(uint32_t)l1.y];
        if (3 < n) v3 = S.theta[(uint32_t)l2.x];
        if (4 < n) v4 = S.theta[(uint32_t)l2.y];
        if (5 < n) v5 = S.theta[(uint32_t)l3.x];
        if (6 < n) v6 = S.theta[(uint32_t)l3.y];
        LOB_LIGHT_ADD(l0.y, v0) LOB_LIGHT_ADD(l1.x, v1) LOB_LIGHT_ADD(l1.y, v2) LOB_LIGHT_ADD(l2.x, v3)
        LOB_LIGHT_ADD(l2.y, v4) LOB_LIGHT_ADD(l3.x, v5) LOB_LIGHT_ADD(l3.y, v6)
    }
    for (int i0 = 7; i0 < n; i0 += 2) {
        const u64 e0 = S.hl_rec[(size_t)b * LOB_HL_REC + 1 + i0], e1 = i0 + 1 < n ? S.hl_rec[(size_t)b * LOB_HL_REC + 2 + i0] : 0ull;
        const f64 v0 = S.theta[(uint32_t)e0];
        f64 v1 = 0.0;
        if (i0 + 1 < n) v1 = S.theta[(uint32_t)e1];
        LOB_LIGHT_ADD(e0, v0) LOB_LIGHT_ADD(e1, v1)
    }
#undef LOB_LIGHT_ADD
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) S.qs_last[(size_t)b * LOB_N_ACTIONS + a] = q[a];
    Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
    action = policy_sample(P, q, false, g);
    hp->slot_cur = cur;
    hp->action = action;
    hp->stepped = 1;
    hp->rng_ctr = g.ctr;
    // The tiles the trace generation of (this state, this action) will write are marked in the written-weights maps NOW,
    // before this step's learn kernel looks at the maps (see learn_traces): once per (triple, action), remembered in the slot.
    // (by memo_kernel, a lane per tile: a lane here would hold its whole wave up over a few round trips)
    if (!((marked >> action) & 1u)) {
        const int pos = atomicAdd(&S.mk_markcount[0], 1);
        if (pos < S.mk_slots) S.mk_marklist[pos] = mslot * 16 + action;
        else mark_generation(P, S, mslot, action);
        atomicOr(&S.mk_marked[mslot], 1u << action);
    }
    const u64 act = __ballot(1);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) cnt_add(S, 5, (unsigned long long)__builtin_popcountll(act));
    return true;
}
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_LIGHT_BLOCK) act_light_kernel(LOB_PS_ARGS, int par, int lpar, u64 ver, int sid_prev) {
    LOB_PS_REFS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S.mk_count[par] = 0;
        S.slow_n[(lpar ^ 1) * 2 + 0] = 0;
        S.slow_n[(lpar ^ 1) * 2 + 1] = 0;
        S.tr_list_n[lpar ^ 1] = 0;
        S.tr_list2_n[lpar ^ 1] = 0;
        S.acc_list_n[lpar ^ 1] = 0;
    }
    const int b = blockIdx.x * LOB_LIGHT_BLOCK + threadIdx.x;
    if (b >= S.B) return;
    const LHdr h = S.hdr[b];
    int action;
    act_light_book(P, S, b, h, lpar, ver, S.hl_dirty[0] == sid_prev, action);
}
#endif

// Agent::UpdateTraces for every book that stepped (learn_traces), the first half of learn_book; leaves
// Q(s, a) in LHdr::td and the RNG counter after its draws in LHdr::rng_ctr for the second half
// (learn_q_fast_kernel).  Persistent 16-wave blocks, two per CU (75 KB of LDS each: the hash table and a
// 4 KB tile map per wave).  A book's slot claims are resolved one book later, so that their CAS round
// trips overlap the next book's work.
// LIST 1: only the books the lane-per-book kernel (trace_light_kernel, below) left on the list `tr_list`.
// LIST 2: the same list left by learn_q_lane_kernel<.., TR = true>, which has ALREADY run for this step: the entries
// carry argmax Q(s, .) (its draws are made), Q(s, a) and the RNG counter are the learn kernel's business, marks are late.
#define LOB_TRL_BOOK(x) ((x) & 0x7ffffff)
#define LOB_TRL_AMAX(x) ((int)((uint32_t)(x) >> 27))
template <int ALGO, int LIST>
__global__ void __launch_bounds__(LOB_TRACE_BLOCK, LOB_TRACE_OCC) trace_fast_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g, int par, int lpar,
                                                                                    int sid) {
    LOB_PS_REFS
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    // this step's update appends to nz_new[par]; the list the general act path reads is nz_new[par ^ 1]
    if (blockIdx.x == 0 && threadIdx.x < LOB_NZ_WORDS) {
        S.nz_new[par * LOB_NZ_WORDS + threadIdx.x] = 0;
        S.nz_new[(2 + par) * LOB_NZ_WORDS + threadIdx.x] = 0;
    }
    // (with the lane-per-generation kernel on, this one serves what that kernel hands on: SARSA `tr_list`, Q(lambda) `tr_list2`)
    const i32* list = (LIST == 2 && P.sarsa_lanes) ? S.tr_list2 : S.tr_list;
    const int list_n = LIST ? ((LIST == 2 && P.sarsa_lanes) ? S.tr_list2_n[lpar] : S.tr_list_n[lpar]) : 0;
    if (LIST && P.sarsa_lanes && list_n == 0) return;  // nothing handed on: the usual case
    uint32_t* rnd = reinterpret_cast<uint32_t*>(fast_lds_raw);
    uint32_t* act_terms = rnd + 2048;
    const int w = threadIdx.x >> 6;
    f32* vars = reinterpret_cast<f32*>(act_terms + 32) + w * 48;
    u64* tab = reinterpret_cast<u64*>(reinterpret_cast<f32*>(act_terms + 32) + LOB_TRACE_WAVES * 48) + (size_t)w * LOB_HSLOTS;
    for (int i = threadIdx.x; i < 512; i += LOB_TRACE_BLOCK) reinterpret_cast<uint4*>(rnd)[i] = reinterpret_cast<const uint4*>(rnd_g)[i];
    if (threadIdx.x < 27) act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
    for (int i = threadIdx.x & 63; i < LOB_TSLOTS / 4; i += 64)  // every wave's tile set starts (and is handed on) empty
        reinterpret_cast<uint4*>(tab)[i] = make_uint4(LOB_NOTILE, LOB_NOTILE, LOB_NOTILE, LOB_NOTILE);
    __syncthreads();
    CbPending prev;
    prev.active = false;
    int lane_ = threadIdx.x & 63;
    const int n_todo = LIST ? list_n : S.B;
#pragma unroll 1
    for (int t = blockIdx.x * LOB_TRACE_WAVES + w; t < n_todo; t += gridDim.x * LOB_TRACE_WAVES) {
        asm volatile("" : "+v"(lane_));
        const int lane = lane_;
        const int ent = __builtin_amdgcn_readfirstlane(LIST ? list[t] : t);
        const int b = LIST == 2 ? LOB_TRL_BOOK(ent) : ent;
        const LHdr h = S.hdr[b];
        const int lslot = P.memo ? S.mk_slot_last[b] : -1;  // memo slot of last_state's group-0 triple (checked below)
        f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = S.qs_last[(size_t)b * LOB_N_ACTIONS + a];
        if (!h.stepped) continue;
        const int4 lid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)(lslot >= 0 ? lslot : 0) * 4);
        Prof pf;
        pf.start(S.prof, b, lane);
        learn_stage_vars(S.vars + (size_t)b * 48, vars, lane);
        pf.mark(8);  // header, Q(s, .), state variables
        const int last = h.slot_cur ^ 1;
        const bool zero_last = (h.zero_mask >> last) & 1;
        // is that slot really last_state's triple?  then its "288 tiles distinct" flag may be used / learnt
        const int qvl = tile_quant(vars[last * 16 + (lane & 15)]);
        const bool lmatch = lslot >= 0 && !zero_last && lid.x == __builtin_amdgcn_readlane(qvl, 0) && lid.y == __builtin_amdgcn_readlane(qvl, 1) &&
                            lid.z == __builtin_amdgcn_readlane(qvl, 2);
        Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
        CbPending pend;
        bool dup = false;
        int mtag = -1;
        if (P.sarsa_lanes && lmatch && S.mk_tiles_ok[lslot] == 3) mtag = lslot | ((P.epi_epoch & 0x7fff) << 16);
        learn_traces<ALGO>(P, S, b, h, rnd, act_terms, tab, false, vars + last * 16, zero_last, qs_last, g, lane, pend, pf, lmatch ? lid.w : 0, &dup,
                           LIST == 2 ? LOB_TRL_AMAX(ent) : -1, LIST == 2 ? sid : -1, mtag);
        if (lmatch && lid.w == 0 && lane == 0) S.mk_ident[(size_t)lslot * 4 + 3] = dup ? 2 : 1;  // (every wave that gets here writes the same value)
        if (LIST != 2 && lane == 0) {
            LHdr* hp = S.hdr + b;
            if (!LOB_TD_KEEP(P)) hp->td = sel9(qs_last, h.action);  // Q(s, a), for the TD error
            hp->rng_ctr = g.ctr;
        }
        cb_claim_finish(S, prev);
        prev = pend;
        pf.mark(18);  // header stores, previous book's claims
    }
    cb_claim_finish(S, prev);
}
// trace_rest_kernel: the near-empty launches around trace_lane_kernel in the fused Q(lambda) / double Q flow -- learn_q_rest_kernel
// over the books the lane learn kernel hands back (the `learn` work list), trace_fast_kernel<., 2> over what the lane trace
// kernel hands on (`tr_list2`) and accumulate_kernel over what the fused accumulation left (`acc_list`) -- as ONE: in most steps
// the three lists hold nothing or a handful of books, and every such launch is a dependency of the step whatever its grid
// (NOTES.md "Round 4", "Round 5").  The orderings they had between them move inside a wave: a book that needs more than one of the
// three gets them from ONE wave, in order -- TD error (learn_q_book), trace step (learn_traces), then its generations added up
// (accumulate_generations) -- and nobody else touches it: the learn kernel flags a handed-back book in acc_pend (bit 0) instead
// of listing it for the accumulation, trace_lane_kernel (acc_fuse = 2) flags a handed-on book (bit 1) and lists neither kind on
// `acc_list`.  A wave's learn_q_book READS theta while other waves are adding generations up: nothing in this kernel may write
// theta -- a generation without a slot, which accumulate_kernel applies tile by tile, goes on `dir_list` for apply_kernel
// (accumulate_generations, `defer`).  ALGO: the learn side's (Q(lambda) or double Q); the trace step is Watkins's in both.
// `hint`: as learn_q_rest_kernel's (the hand-back count for the host).
template <int ALGO>
__global__ void __launch_bounds__(LOB_TRACE_BLOCK) trace_rest_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g, int par, int lpar, int sid,
                                                                     int lpb_shift, u64* hint, uint32_t hint_tag) {
    LOB_PS_REFS
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    __shared__ LearnLds L;  // (learn_q_book: waves 0-3 of the block)
    // this step's update appends to nz_new[par]; the list the general act path reads is nz_new[par ^ 1] (as trace_fast_kernel)
    if (blockIdx.x == 0 && threadIdx.x < LOB_NZ_WORDS) {
        S.nz_new[par * LOB_NZ_WORDS + threadIdx.x] = 0;
        S.nz_new[(2 + par) * LOB_NZ_WORDS + threadIdx.x] = 0;
    }
    const int n_learn = S.slow_n[lpar * 2 + 1], n_tr = S.tr_list2_n[lpar], n_acc = S.acc_list_n[lpar];
    if (hint && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(hint, ((u64)hint_tag << 32) | (u64)(uint32_t)n_learn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (n_learn == 0 && n_tr == 0 && n_acc == 0) return;  // the usual case
    if (n_learn > 0 && blockIdx.x == 0 && threadIdx.x == 0) cnt_add(S, 4, (u64)n_learn);  // (lob_get_path_stats [7])
    const int w = threadIdx.x >> 6;
    const int wave = blockIdx.x * LOB_TRACE_WAVES + w, n_waves = gridDim.x * LOB_TRACE_WAVES;
    const int xcd = acc_copy(S, wave);
    int lane_ = threadIdx.x & 63;
    const i32* learn_list = S.slow_list + S.B;
    if (n_learn > 0) learn_stage_table(rnd_g, L);  // (block-uniform; ends in a block barrier)
    // ---- the books the learn kernel handed back (and that the lane trace kernel did not hand on as well): TD error, then their sums ----
    if (n_learn > 0 && w < LOB_WAVES_PER_BLOCK) {
#pragma unroll 1
        for (int t = blockIdx.x * LOB_WAVES_PER_BLOCK + w; t < n_learn; t += gridDim.x * LOB_WAVES_PER_BLOCK) {
            asm volatile("" : "+v"(lane_));
            const int lane = lane_;
            const int b = __builtin_amdgcn_readfirstlane(learn_list[t]);
            if (S.acc_pend[b] & 2) continue;  // (its trace step comes first: below)
            learn_q_book<ALGO>(P, S, L, w, lane, b);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            __builtin_amdgcn_wave_barrier();
            accumulate_generations(P, S, par, sid, xcd, 64, lane, b, false, lane, true);
        }
    }
    if (n_tr > 0) {  // (block-uniform)
        uint32_t* rnd = reinterpret_cast<uint32_t*>(fast_lds_raw);
        uint32_t* act_terms = rnd + 2048;
        f32* vars = reinterpret_cast<f32*>(act_terms + 32) + w * 48;
        u64* tab = reinterpret_cast<u64*>(reinterpret_cast<f32*>(act_terms + 32) + LOB_TRACE_WAVES * 48) + (size_t)w * LOB_HSLOTS;
        for (int i = threadIdx.x; i < 512; i += LOB_TRACE_BLOCK) reinterpret_cast<uint4*>(rnd)[i] = reinterpret_cast<const uint4*>(rnd_g)[i];
        if (threadIdx.x < 27) act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
        for (int i = threadIdx.x & 63; i < LOB_TSLOTS / 4; i += 64)  // every wave's tile set starts (and is handed on) empty
            reinterpret_cast<uint4*>(tab)[i] = make_uint4(LOB_NOTILE, LOB_NOTILE, LOB_NOTILE, LOB_NOTILE);
        __syncthreads();
        // ---- the books the lane trace kernel handed on: (TD error if that is pending too,) trace step, then their sums ----
        if (w < LOB_WAVES_PER_BLOCK) {
#pragma unroll 1
            for (int t = blockIdx.x * LOB_WAVES_PER_BLOCK + w; t < n_tr; t += gridDim.x * LOB_WAVES_PER_BLOCK) {
                asm volatile("" : "+v"(lane_));
                const int lane = lane_;
                const int ent = __builtin_amdgcn_readfirstlane(S.tr_list2[t]);
                const int b = LOB_TRL_BOOK(ent);
                if (S.acc_pend[b] & 1) {  // (handed back by the learn kernel as well: n_learn > 0, the table is staged)
                    learn_q_book<ALGO>(P, S, L, w, lane, b);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                    __builtin_amdgcn_wave_barrier();
                }
                const LHdr h = S.hdr[b];
                const int lslot = P.memo ? S.mk_slot_last[b] : -1;
                f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = S.qs_last[(size_t)b * LOB_N_ACTIONS + a];
                if (!h.stepped) continue;
                const int4 lid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)(lslot >= 0 ? lslot : 0) * 4);
                Prof pf;
                pf.start(S.prof, b, lane);
                learn_stage_vars(S.vars + (size_t)b * 48, vars, lane);
                const int last = h.slot_cur ^ 1;
                const bool zero_last = (h.zero_mask >> last) & 1;
                const int qvl = tile_quant(vars[last * 16 + (lane & 15)]);
                const bool lmatch = lslot >= 0 && !zero_last && lid.x == __builtin_amdgcn_readlane(qvl, 0) && lid.y == __builtin_amdgcn_readlane(qvl, 1) &&
                                    lid.z == __builtin_amdgcn_readlane(qvl, 2);
                Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
                CbPending pend;
                bool dup = false;
                int mtag = -1;
                if (P.sarsa_lanes && lmatch && S.mk_tiles_ok[lslot] == 3) mtag = lslot | ((P.epi_epoch & 0x7fff) << 16);
                learn_traces<LOB_ALGO_QLAMBDA>(P, S, b, h, rnd, act_terms, tab, false, vars + last * 16, zero_last, qs_last, g, lane, pend, pf, lmatch ? lid.w : 0, &dup,
                                               LOB_TRL_AMAX(ent), sid, mtag);
                if (lmatch && lid.w == 0 && lane == 0) S.mk_ident[(size_t)lslot * 4 + 3] = dup ? 2 : 1;
                cb_claim_finish(S, pend);  // (at once: the generation's slot is looked up right below)
                // the book's generations as the trace step has left them (header, masks, slots: this wave's own stores)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                __builtin_amdgcn_wave_barrier();
                accumulate_generations(P, S, par, sid, xcd, 64, lane, b, false, lane, true);
            }
        }
    }
    // ---- what the fused accumulation left (accumulate_kernel over `acc_list`) ----
    const int lane = threadIdx.x & 63;
    const int lpb = 1 << lpb_shift, sub = lane & (lpb - 1);
#pragma unroll 1
    for (int wv = wave; (wv << (6 - lpb_shift)) < n_acc; wv += n_waves) {
        const int i = (wv << (6 - lpb_shift)) + (lane >> lpb_shift);
        const bool have = i < n_acc;
        const uint32_t ent = have ? (uint32_t)S.acc_list[i] : 0u;
        accumulate_generations(P, S, par, sid, xcd, lpb, sub, have ? (int)(ent & 0x7fffffffu) : S.B, have && (ent >> 31) != 0, lane, true);
    }
}
// Agent::UpdateTraces with one LANE per book, for the books whose step leaves no older generation behind -- Watkins's
// cut after an exploratory action (QLearn::UpdateTraces, agent.cpp:272-280: traces.decay(0.0)), or no traces yet --
// and whose last_state has a memo slot with its 288 group-0 tiles on record and known to be distinct: the new
// generation is then the chosen action's 32 tiles, all alive, copied from the slot's record (memo_kernel wrote them),
// and nothing is looked up in any set.  At an exploration rate of 0.8 that is 7 books in 10; the others go on the list
// of the wave-per-book kernel.  Same draws, same stores as learn_traces for these books.
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_LIGHT_BLOCK) trace_light_kernel(LOB_PS_ARGS, int lpar) {
    LOB_PS_REFS
    // Slot claims of the combined update: thousands of books hold the very same generation, and compare-and-swaps on one
    // address queue up behind each other.  The block elects one claimant per distinct generation first (LDS).
    __shared__ u64 claimed[512];
    for (int i = threadIdx.x; i < 512; i += LOB_LIGHT_BLOCK) claimed[i] = LOB_CB_EMPTY;
    __syncthreads();
    const int b = blockIdx.x * LOB_LIGHT_BLOCK + threadIdx.x;
    if (b >= S.B) return;
    const LHdr h = S.hdr[b];
    if (!h.stepped) return;
    LHdr* hp = S.hdr + b;
    const int lslot = S.mk_slot_last[b];
    f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = S.qs_last[(size_t)b * LOB_N_ACTIONS + a];
    const int last = h.slot_cur ^ 1;
    const bool zero_last = (h.zero_mask >> last) & 1;
    const float4 vl = *reinterpret_cast<const float4*>(S.vars + (size_t)b * 48 + last * 16);
    const int q0 = tile_quant(vl.x), q1 = tile_quant(vl.y), q2 = tile_quant(vl.z);
    const int ls = lslot >= 0 ? lslot : 0;
    const int4 lid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ls * 4);
    const int tiles_ok = S.mk_tiles_ok[ls];
    Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
    const int action = h.action;
    const int amax = argmax_ties(qs_last, g);
    int n_old = h.tr_n;
    int kmax = P.trace_kmax;
    if (action != amax) kmax = 1;
    if (n_old > kmax - 1) n_old = kmax - 1;
    const bool light = n_old == 0 && lslot >= 0 && !zero_last && lid.x == q0 && lid.y == q1 && lid.z == q2 && lid.w == 1 && tiles_ok != 0;
    {   // (one atomic per wave: tens of thousands of lanes adding to one counter queue up behind each other)
        const u64 mb = __ballot(!light);
        if (mb) {
            int base = 0;
            const int leader = __builtin_ctzll(mb);
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&S.tr_list_n[lpar], __builtin_popcountll(mb));
            base = __shfl(base, leader);
            if (!light) {
                S.tr_list[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u))] = b;
                return;  // (nothing of the book has been touched)
            }
        }
    }
    const int G = P.trace_gens;
    const int nh = (h.tr_head + 1) & (G - 1);
    const int4* src = reinterpret_cast<const int4*>(S.mk_tiles + ((size_t)lslot * LOB_N_ACTIONS + action) * 32);
    int4* dst = reinterpret_cast<int4*>(S.tr_idx + ((size_t)b * G + nh) * 32);
    int4 tl[8];
#pragma unroll
    for (int i = 0; i < 8; i++) tl[i] = src[i];
    // the generation's tiles are marked in the written-weights maps before the learn kernel looks (learn_traces): once
    // per (triple, action), remembered in the slot
    if (!((S.mk_marked[lslot] >> action) & 1u)) {
#pragma unroll
        for (int i = 0; i < 8; i++) { nzx_mark(P, S, tl[i].x); nzx_mark(P, S, tl[i].y); nzx_mark(P, S, tl[i].z); nzx_mark(P, S, tl[i].w); }
        atomicOr(&S.mk_marked[lslot], 1u << action);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = tl[i];
    S.tr_alive[(size_t)b * G + nh] = 0xffffffffu;
    hp->tr_head = nh;
    hp->tr_n = 1;
    if (!LOB_TD_KEEP(P)) hp->td = sel9(qs_last, action);  // Q(s, a), for the TD error
    hp->rng_ctr = g.ctr;
    if (P.combine) {
        *reinterpret_cast<int4*>(S.tr_sig + ((size_t)b * G + nh) * 4) = make_int4(q0, q1, q2, action);
        // (thousands of books hold this very generation: look before the compare-and-swap)
        const u64 ch = cb_hash(q0, q1, q2, action, 0xffffffffu);
        const u64 seen = atomicCAS((unsigned long long*)&claimed[(ch >> 40) & 511], (unsigned long long)LOB_CB_EMPTY, (unsigned long long)ch);
        if (seen != ch && S.cb_key[(uint32_t)ch & (uint32_t)(S.cb_slots - 1)] != ch) {  // (not: another lane of the block claims it / it is claimed)
            CbPending pend;
            cb_claim_issue(S, pend, q0, q1, q2, action, 0xffffffffu, b * G + nh);
            cb_claim_finish(S, pend);
        } else S.tr_cbslot[(size_t)b * G + nh] = (i32)((uint32_t)ch & (uint32_t)(S.cb_slots - 1));
    }
}
#endif

// Agent::UpdateTraces with a LANE per trace generation, 32 lanes per book, for the books whose older generations survive the
// step: every book of SARSA(lambda) (no Watkins cut: a book keeps its trace_kmax - 1 latest generations), the books of Watkins's
// Q(lambda) whose action was the greedy one (ALGO Q(lambda): the entries learn_q_*_kernel<.., TR = true> left on `tr_list`,
// which carry argmax Q(s, .); that kernel has run, so Q(s, a) and the RNG counter are its business and the marks are late).
// What Traces::update (traces.cpp:40-50) does to an old generation -- clear, or re-set to 1 and so hand over to the new
// generation, every tile that is also one of last_state's 288 -- is decided without the tile indices: tile j of generation
// (triple T, action a) is the tile of (T', a) in tiling j exactly when T and T' fall in the same cell of that tiling (integer
// arithmetic on the six quantised coordinates), and any OTHER coincidence of indices goes through an index the tile registry has
// marked ambiguous (lob_state.h ow_tab; mk_amb: per memo slot and action, which tilings) -- those few are compared index by
// index from the slots' tile records.  The new generation is the chosen action's 32 tiles copied from the slot's record, as in
// trace_light_kernel.  A book the path cannot serve exactly (no memo slot / unregistered tiles / a constructor-zero or NaN
// state among its generations / duplicate tiles inside last_state / too many ambiguous pairs) is handed on untouched to the
// wave-per-book kernel: SARSA on `tr_list` (trace_fast_kernel<SARSA, 1>), Q(lambda) on `tr_list2` (trace_fast_kernel<.., 2>).
// Same stores as learn_traces for the books it takes.
#define LOB_TS_BLOCK 256
#define LOB_TS_BLOCK_WAVES (LOB_TS_BLOCK / 64)
// Tile registry (lob_state.h ow_tab): the memo slots on this step's list whose tiles are not registered yet -- new triples, a few
// per step, dozens in an episode's first steps -- enter their 288 tiles, learn which of them lie on an index another tile uses
// (mk_amb) and whether two of their own coincide (mk_ident[3]).  One wave per slot.  Runs as the FIRST LOB_REG_BLOCKS blocks of
// trace_lane_kernel's launch, beside the blocks that read what it writes: nothing of it is needed before the next step's trace
// kernel (a slot new in step t is nobody's last_state before step t + 1; bits that appear early in mk_amb of older slots only
// widen the set of tile pairs the lane kernel compares index by index, and equal indices are the ground truth; a slot counts
// as registered -- mk_tiles_ok bit 1 -- behind a fence, last).  Until round 6 a kernel of its own on the engine's second
// stream, whose fork and join were the main stream's only cross-stream operations: two gaps of 5-6 us per step between kernels
// that otherwise follow each other within 0.0 us -- and 40-70 us long, a chain of round trips: a lane's five compare-and-swaps
// one after the other, each followed by a load (now: all ten in flight at once, tile_register_ask), and 1 600 compares per lane
// to find two equal tiles among the 288 (now: a 512-entry hash set per wave in LDS, filled while the swaps are under way).
#define LOB_REG_BLOCKS 128   /* blocks of trace_lane_kernel's launch that run registry_block */
#define LOB_SCAN_BLOCKS 256  /* blocks of apply_kernel's launch that run registry_scan_block (lob_kernels.h) */
// (a wave's hash set of the tiles it has met: true if `v` was there already)
__device__ __forceinline__ bool registry_seen(uint32_t* seen /* LDS, 512 entries, 0xffffffff = free */, uint32_t v) {
    uint32_t hh = (v * 2654435761u) >> 23;
    for (int probe = 0; probe < 512; probe++) {
        const uint32_t old = atomicCAS(&seen[hh], 0xffffffffu, v);
        if (old == 0xffffffffu) return false;
        if (old == v) return true;
        hh = (hh + 1) & 511u;
    }
    return false;
}
struct RegistryLds {
    uint32_t rnd[2048 + 32];
    uint32_t seen[LOB_TS_BLOCK_WAVES][512];
};
// (always inlined: as a called function -- shared by the kernel's two instantiations -- it gave the whole kernel a stack, 96
// registers instead of 82, and itself generic pointers into LDS)
__device__ __forceinline__ void registry_block(const DevParams& P, const DevState& S, const uint32_t* __restrict__ rnd_g, RegistryLds& L, int par, int apar, int blk, int nblk) {
    // (a short chain of dependent look-ups -- count -> list entry -> registered? -> identity: the first entry is asked for with
    // the count, as memo_kernel does, and an entry's identity with its flags)
    const int wave0 = blk * LOB_TS_BLOCK_WAVES + (int)(threadIdx.x >> 6);
    const int s_first = S.mk_list[(size_t)par * S.mk_slots + (wave0 < S.mk_slots ? wave0 : 0)];
    const int wave1 = wave0 + nblk * LOB_TS_BLOCK_WAVES;   // (a step's list is longer than the launch has waves: the second entry too)
    const int s_second = S.mk_list[(size_t)par * S.mk_slots + (wave1 < S.mk_slots ? wave1 : 0)];
    int count = S.mk_count[par];
    {
        const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
        uint4* dst = reinterpret_cast<uint4*>(L.rnd);
        const uint4 r0 = src[threadIdx.x], r1 = src[threadIdx.x + 256];
        dst[threadIdx.x] = r0; dst[threadIdx.x + 256] = r1;
        if (threadIdx.x < 27) L.rnd[2048 + threadIdx.x] = rnd_g[2048 + threadIdx.x];
    }
    __syncthreads();
    const uint32_t* rnd = L.rnd;
    const int lane = threadIdx.x & 63, j = lane & 31;
    const bool hi = lane >= 32;
    const int wave = blk * LOB_TS_BLOCK_WAVES + (threadIdx.x >> 6), n_waves = nblk * LOB_TS_BLOCK_WAVES;
    uint32_t* seen = L.seen[threadIdx.x >> 6];
    const uint32_t M = (uint32_t)P.M;
    if (count > S.mk_slots) count = S.mk_slots;
    for (int i = wave; i < count; i += n_waves) {
        const int s = i == wave ? s_first : i == wave + n_waves ? s_second : S.mk_list[(size_t)par * S.mk_slots + i];
        const int ok_bits = S.mk_tiles_ok[s];
        const int4 id = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)s * 4);
        if (ok_bits & 2) continue;
#pragma unroll
        for (int x = 0; x < 8; x++) seen[x * 64 + lane] = 0xffffffffu;
        uint32_t sum = 0;
        {
            int base = j;
            sum = mod_add(sum, rnd[(tile_coord(id.x, base) + 449 * 0) & 2047], M); base += 2 * j;
            sum = mod_add(sum, rnd[(tile_coord(id.y, base) + 449 * 1) & 2047], M); base += 2 * j;
            sum = mod_add(sum, rnd[(tile_coord(id.z, base) + 449 * 2) & 2047], M);
            sum = mod_add(sum, rnd[(j + 449 * 3) & 2047], M);
        }
        // a lane's five tiles (action (hi ? 5 : 0) + k of tiling j; lanes 32-63 have four): every first probe under way at once
        i32 tl[5];
        u64 first_old[5];
        uint32_t first_word[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int a = (hi ? 5 : 0) + k;
            const i32 tile = tile_index(sum, rnd[2048 + (a < LOB_N_ACTIONS ? a : 0)], M);
            tl[k] = tile;
            first_old[k] = 0; first_word[k] = 0;
            if (a < LOB_N_ACTIONS) {
                S.mk_tiles[((size_t)s * LOB_N_ACTIONS + a) * 32 + j] = tile;  // (memo_kernel writes the same values: registry_scan_block must find them)
                tile_register_ask(S, s, a, j, tile, first_old[k], first_word[k]);
            }
        }
        // do two of the triple's 288 tiles coincide?  (mk_ident[3]: the lane trace kernel takes only triples known to be free of
        // that; the wave-per-book kernel would find out the first time it builds the set -- a step later, for every book that
        // starts from this triple.)  Every tile goes into the wave's hash set; the second of two equal ones finds the first.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        bool dupl = false;
#pragma unroll
        for (int k = 0; k < 5; k++)
            if ((hi ? 5 : 0) + k < LOB_N_ACTIONS) dupl |= registry_seen(seen, (uint32_t)tl[k]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        bool reg_fail = false;
        uint32_t my_amb = 0;  // bit k: this lane's tile of action (hi ? 5 : 0) + k lies on an ambiguous index
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int a = (hi ? 5 : 0) + k;
            if (a < LOB_N_ACTIONS) {
                const int r = tile_register_impl<true>(S, id, s, a, j, tl[k], apar, first_old[k], first_word[k]);
                reg_fail |= r < 0;
                if (r > 0) my_amb |= 1u << k;
            }
        }
        const bool any_dup = __ballot(dupl) != 0;
        const bool failed = __ballot(reg_fail) != 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const u64 mb = __ballot((my_amb >> k) & 1u);
            if (lane == 0) {
                S.mk_amb[(size_t)s * LOB_N_ACTIONS + k] = (uint32_t)mb;
                if (k < 4) S.mk_amb[(size_t)s * LOB_N_ACTIONS + 5 + k] = (uint32_t)(mb >> 32);
            }
        }
        if (lane == 0) {
            S.mk_ident[(size_t)s * 4 + 3] = any_dup ? 2 : 1;
            // (its place on the list of registered slots is asked for in front of the fence: one round trip instead of two.  The
            // list is read by the scan blocks of a later launch.)
            const int pos = failed ? 0 : atomicAdd(S.mk_all_n, 1);
            __threadfence();
            if (!failed) {
                S.mk_all[pos] = s;  // (pos < mk_slots: a slot registers once per episode)
                atomicOr(&S.mk_tiles_ok[s], 2);
            }
        }
    }
}

// acc_fuse (Q(lambda)): every generation's update is added to its slot as soon as the slot is known (acc_generation); a book
// the learn kernel handed back (its TD error is not known yet: acc_pend) or this kernel hands on goes on acc_list.
template <int ALGO>
__global__ void __launch_bounds__(LOB_TS_BLOCK, 5) trace_lane_kernel(LOB_PS_ARGS, int lpar, int sid, int acc_fuse, const uint32_t* __restrict__ rnd_g, int par, int apar,
                                                                     int reg_blocks) {
    LOB_PS_REFS
    if ((int)blockIdx.x < reg_blocks) {  // the launch's first blocks: the tile registry
        __shared__ RegistryLds reg_lds;
        registry_block(P, S, rnd_g, reg_lds, par, apar, (int)blockIdx.x, reg_blocks);
        return;
    }
    const int bid = (int)blockIdx.x - reg_blocks, n_blocks = (int)gridDim.x - reg_blocks;
    constexpr bool QL = ALGO == LOB_ALGO_QLAMBDA;
    const int lane = threadIdx.x & 63, k = lane & 31, half = lane >> 5, grp = threadIdx.x >> 5;
    const int n_todo = QL ? S.tr_list_n[lpar] : S.B;
    const int G = P.trace_gens;  // 32
    const int tag_epoch = (P.epi_epoch & 0x7fff) << 16;
    const int lim = LOB_TILE_PLAIN_MIN;  // (below: tile_coord's wrap-around branch -- a NaN variable)
    const bool reg_ok = S.amb_flag[0] == 0;
#pragma unroll 1
    for (int t0 = bid * (LOB_TS_BLOCK / 32); t0 < n_todo; t0 += n_blocks * (LOB_TS_BLOCK / 32)) {
        const int t = t0 + grp;
        const bool in = t < n_todo;
        const int ent = QL ? S.tr_list[in ? t : 0] : t;
        const int b = QL ? LOB_TRL_BOOK(ent) : ent;
        const int bb = in ? b : 0;
        // Everything addressed by the book alone is asked for in ONE round trip: the header, the memo slot, BOTH State rows (which
        // one is last_state is in the header) and this lane's generation -- the lane is the generation's RING SLOT, not its age
        // (the age follows from the header's ring head afterwards), so its words do not wait for the header.
        const size_t gi = (size_t)bb * G + k;
        const LHdr h = S.hdr[bb];
        const int lslot = S.mk_slot_last[bb];
        const float4 v0 = *reinterpret_cast<const float4*>(S.vars + (size_t)bb * 48), v1 = *reinterpret_cast<const float4*>(S.vars + (size_t)bb * 48 + 16);
        const uint32_t m_raw = S.tr_alive[gi];
        const int4 sg_raw = *reinterpret_cast<const int4*>(S.tr_sig + gi * 4);
        const int tag_raw = S.tr_mslot[gi], cs_raw = S.tr_cbslot[gi];
        const bool stepped = in && h.stepped != 0;
        const int ls = lslot >= 0 ? lslot : 0;
        const int last = h.slot_cur ^ 1;
        const bool zero_last = (h.zero_mask >> last) & 1;
        const float4 vl = last ? v1 : v0;
        const int q0 = tile_quant(vl.x), q1 = tile_quant(vl.y), q2 = tile_quant(vl.z);
        const int4 lid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ls * 4);
        const int action = h.action;
        const bool fuse_acc = QL && acc_fuse != 0;
        const bool td_pending = fuse_acc && (S.acc_pend[bb] & 1) != 0;  // (handed back by the learn kernel)
        int n_old = h.tr_n;
        if (n_old > P.trace_kmax - 1) n_old = P.trace_kmax - 1;
        if (QL && action != LOB_TRL_AMAX(ent)) n_old = 0;  // Watkins's cut (QLearn::UpdateTraces, agent.cpp:272-280: traces.decay(0.0))
        bool ok = lslot >= 0 && !zero_last && lid.x == q0 && lid.y == q1 && lid.z == q2 && lid.w == 1 && S.mk_tiles_ok[ls] == 3 && reg_ok &&
                  q0 >= lim && q1 >= lim && q2 >= lim;
        // ---- this lane's generation: ring slot k, age (head - k) mod G; age G - 1 is the slot the new generation goes to ----
        const int age = (h.tr_head - k) & (G - 1);
        const bool has_old = stepped && age < n_old;
        const uint32_t m = has_old ? m_raw : 0u;
        const int4 sg = has_old ? sg_raw : make_int4(0, 0, 0, 0);
        const int tag = has_old ? tag_raw : -1;
        int cs = has_old ? cs_raw : -1;
        // (the new generation's tile, ahead of its use: the chain of dependent look-ups is what this kernel's time is made of)
        const i32 N = S.mk_tiles[((size_t)ls * LOB_N_ACTIONS + action) * 32 + k];
        const uint32_t marked = S.mk_marked[ls];
        const int so = tag & 0xffff;  // the generation's memo slot
        uint32_t amb_old = 0;
        if (m) {
            const bool gen_ok = tag >= 0 && (tag & 0x7fff0000) == tag_epoch && !(sg.w & 256) && sg.x >= lim && sg.y >= lim && sg.z >= lim;
            ok = ok && gen_ok;
            if (gen_ok) amb_old = S.mk_amb[(size_t)so * LOB_N_ACTIONS + (sg.w & 15)];
        }
        // ambiguous tiles of last_state: lane a < 9 of the half holds the word of action a
        const uint32_t amb_w = k < LOB_N_ACTIONS ? S.mk_amb[(size_t)ls * LOB_N_ACTIONS + k] : 0u;
        int n_amb_new = __popc(amb_w);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) n_amb_new += __shfl_xor(n_amb_new, o);  // (lanes 0-15 of the half: all nine words)
        n_amb_new = __shfl(n_amb_new, half * 32);
        // ---- same cell, tiling by tiling (lob_tiles.h) ----
        uint32_t hit = 0;
        if (m) hit = tile_same_cell_mask(q0, q1, q2, sg.x, sg.y, sg.z);
        const uint32_t cand = m & ~hit & amb_old;  // live tiles that may meet last_state's through an ambiguous index
        const bool heavy = cand != 0 && n_amb_new != 0 && __popc(cand) * n_amb_new > 96;
        {
            const u64 bad = __ballot(stepped && (!ok || heavy));
            if ((uint32_t)(bad >> (half * 32))) {
                if (k == 0 && stepped) {  // (nothing of the book has been touched)
                    if (QL) S.tr_list2[atomicAdd(&S.tr_list2_n[lpar], 1)] = ent;
                    else S.tr_list[atomicAdd(&S.tr_list_n[lpar], 1)] = b;
                    cnt_add(S, 6, 1ull);
                    // (the wave-per-book kernel does its traces: all its generations are accumulate_kernel's -- or, acc_fuse = 2,
                    // trace_rest_kernel's own wave adds them up right behind the trace step)
                    if (fuse_acc && acc_fuse != 2) S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = b;
                    if (acc_fuse == 2) S.acc_pend[b] = (uint8_t)((td_pending ? 1 : 0) | 2);  // (bit 1: its trace step is trace_rest_kernel's too)
                }
                continue;
            }
        }
        if (!stepped) continue;
        if (td_pending && k == 0 && acc_fuse != 2) S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = b;  // (acc_fuse 2: the work list names it already)
        const bool add_here = fuse_acc && !td_pending;
        const f64 upd32 = h.upd / (f64)LOB_N_TILINGS;
        const int vec = h.stepped == 2 ? 1 : 0;  // (double Q: the learn kernel's coin)
        const int xcd = add_here ? acc_copy(S, (int)(blockIdx.x * (LOB_TS_BLOCK / 64) + (threadIdx.x >> 6))) : 0;
        bool acc_failed = false;
        if (cand != 0 && n_amb_new != 0) {
            const i32* to = S.mk_tiles + ((size_t)so * LOB_N_ACTIONS + (sg.w & 15)) * 32;
            uint32_t c = cand;
            while (c) {
                const int j = __builtin_ctz(c);
                c &= c - 1;
                const i32 f = to[j];
                for (int a = 0; a < LOB_N_ACTIONS; a++) {
                    uint32_t w = S.mk_amb[(size_t)ls * LOB_N_ACTIONS + a];
                    while (w) {
                        const int j2 = __builtin_ctz(w);
                        w &= w - 1;
                        if (S.mk_tiles[((size_t)ls * LOB_N_ACTIONS + a) * 32 + j2] == f) hit |= 1u << j;
                    }
                }
            }
        }
        // ---- old generations: new masks, slot claims of the combined update ----
        if (m) {
            const uint32_t m2 = m & ~hit;
            if (m2 != m) S.tr_alive[gi] = m2;
            bool cs_here = true;  // (`cs` is what tr_cbslot[gi] holds)
            if (m2 && (m2 != m || cs < 0)) {  // (an unchanged generation keeps the slot it has: the table persists, lob_learn.h)
                const u64 ch = cb_hash(sg.x, sg.y, sg.z, sg.w, m2);
                const uint32_t home = (uint32_t)ch & (uint32_t)(S.cb_slots - 1);
                if (S.cb_key[home] == ch) { cs = (i32)home; S.tr_cbslot[gi] = cs; }  // (thousands of books hold this very generation: look before the compare-and-swap)
                else {
                    CbPending pend;
                    cb_claim_issue(S, pend, sg.x, sg.y, sg.z, sg.w, m2, (int)gi);
                    cb_claim_finish(S, pend);
                    cs_here = false;
                }
            }
            // (old age `age`: age + 1 after this step's decay)
            if (add_here && m2) {
                const f64 val = upd32 * (f64)P.trace_pow[age + 1];
                if (!(cs_here ? acc_generation_at(S, gi, cs, sg, m2, val, xcd, vec) : acc_generation(S, gi, m2, val, xcd, vec))) acc_failed = true;
            }
        }
        // ---- the new generation: the chosen action's 32 tiles, all alive ----
        const int nh = (h.tr_head + 1) & (G - 1);
        const size_t ni = (size_t)b * G + nh;
        if (!((marked >> action) & 1u)) {  // marked in the written-weights maps before the learn kernel looks (learn_traces) -- or late
            if (QL) nzx_mark_late(P, S, N, sid);
            else nzx_mark(P, S, N);
            if (k == 0) atomicOr(&S.mk_marked[lslot], 1u << action);
        }
        S.tr_idx[ni * 32 + k] = N;
        if (age == G - 1) {  // (ring slot nh; n_old <= G - 1: this lane has no old generation)
            LHdr* hp = S.hdr + b;
            S.tr_alive[ni] = 0xffffffffu;
            S.tr_mslot[ni] = lslot | tag_epoch;
            *reinterpret_cast<int4*>(S.tr_sig + ni * 4) = make_int4(q0, q1, q2, action);
            hp->tr_head = nh;
            hp->tr_n = n_old + 1;
            if (!QL && !LOB_TD_KEEP(P)) hp->td = S.qs_last[(size_t)b * LOB_N_ACTIONS + action];  // Q(s, a), for the TD error
            const u64 ch = cb_hash(q0, q1, q2, action, 0xffffffffu);
            const uint32_t home = (uint32_t)ch & (uint32_t)(S.cb_slots - 1);
            bool at_home = false;
            if (S.cb_key[home] == ch) { S.tr_cbslot[ni] = (i32)home; at_home = true; }
            else {
                CbPending pend;
                cb_claim_issue(S, pend, q0, q1, q2, action, 0xffffffffu, (int)ni);
                cb_claim_finish(S, pend);
            }
            if (add_here) {
                const f64 val = upd32 * (f64)P.trace_pow[0];
                if (!(at_home ? acc_generation_at(S, ni, (int)home, make_int4(q0, q1, q2, action), 0xffffffffu, val, xcd, vec) : acc_generation(S, ni, 0xffffffffu, val, xcd, vec)))
                    acc_failed = true;
            }
        }
        if (add_here) {  // a generation without a slot: the book goes on the list once, for the direct path only
            const u64 fl = __ballot(acc_failed);
            if ((uint32_t)(fl >> (half * 32)) && k == 0) S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = (i32)((uint32_t)b | 0x80000000u);
        }
    }
}

__host__ __device__ inline size_t trace_lds_bytes() { return (size_t)(2048 + 32) * 4 + (size_t)LOB_TRACE_WAVES * 48 * 4 + (size_t)LOB_TRACE_WAVES * LOB_HSLOTS * 8; }

// What a wave of the persistent kernels fetches of a book one loop iteration AHEAD (the loads of the next batch are
// issued before the current one is worked on, so their latency is not exposed at the top of the next iteration):
// the 64-byte header spread over lanes (lane i & 15 holds dword i: one register for the lot, fields come back through
// readlane), the memo slot, the three State rows (lane < 48).
struct FastPre {
    int hdw, mslot;
    f32 vv;
};
template <int NB>
__device__ __forceinline__ void fast_prefetch(const DevState& S, int t0, int stride, int lane, FastPre* p) {
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int t = __builtin_amdgcn_readfirstlane(t0 + k * stride);
        const int bb = t < S.B ? t : 0;
        p[k].hdw = reinterpret_cast<const int*>(S.hdr + bb)[lane & 15];
        p[k].mslot = S.mk_slot[bb];
        p[k].vv = lane < 48 ? S.vars[(size_t)bb * 48 + lane] : 0.0f;
    }
}
__device__ __forceinline__ u64 hdw_u64(int hdw, int i) {
    return (u64)(uint32_t)__builtin_amdgcn_readlane(hdw, i) | ((u64)(uint32_t)__builtin_amdgcn_readlane(hdw, i + 1) << 32);
}
// (dword offsets of the LHdr fields: lob_state.h)
static_assert(offsetof(LHdr, slot_cur) == 8 && offsetof(LHdr, stepped) == 20 && offsetof(LHdr, rng_ctr) == 32 && offsetof(LHdr, reward) == 40 &&
              offsetof(LHdr, td) == 48, "LHdr layout");

// The second half of learn_book: Q(to_state, .), the TD error of SARSA / QLearn::UpdateWeights (agent.cpp:282-311).
template <int ALGO, int NB>
__global__ void __launch_bounds__(LOB_FAST_BLOCK) learn_q_fast_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g, int lpar,
                                                                      u64 ver) {
    LOB_PS_REFS
    static_assert(ALGO == LOB_ALGO_SARSA || ALGO == LOB_ALGO_QLAMBDA, "one weight vector");
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    const FastLds L = fast_stage(fast_lds_raw, P, S, rnd_g, NB, false);
    const int w = threadIdx.x >> 6;
    const int stride = gridDim.x * LOB_FAST_WAVES;
    int lane_ = threadIdx.x & 63;
    FastPre cur[NB];
    fast_prefetch<NB>(S, blockIdx.x * LOB_FAST_WAVES + w, stride, lane_, cur);
#pragma unroll 1
    for (int t0 = blockIdx.x * LOB_FAST_WAVES + w; t0 < S.B; t0 += stride * NB) {
        asm volatile("" : "+v"(lane_));
        const int lane = lane_;
        int b[NB], mslot[NB];
        bool go[NB], real[NB];
        LHdr h[NB];
        Prof pf;
        int4 mid[NB];
        MemoRec rec[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int t = __builtin_amdgcn_readfirstlane(t0 + k * stride);
            go[k] = real[k] = t < S.B;
            b[k] = go[k] ? t : 0;
            h[k].slot_cur = __builtin_amdgcn_readlane(cur[k].hdw, 2);
            h[k].stepped = __builtin_amdgcn_readlane(cur[k].hdw, 5);
            h[k].rng_ctr = hdw_u64(cur[k].hdw, 8);
            h[k].reward = __longlong_as_double((long long)hdw_u64(cur[k].hdw, 10));
            h[k].td = __longlong_as_double((long long)hdw_u64(cur[k].hdw, 12));
            mslot[k] = __builtin_amdgcn_readfirstlane(cur[k].mslot);
            go[k] = go[k] && h[k].stepped != 0;
            // this batch's memo records first, then the next batch's rows: a wait for the former does not wait for the latter
            const int ms = mslot[k] >= 0 ? mslot[k] : 0;
            mid[k] = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ms * 4);
            rec[k] = *reinterpret_cast<const MemoRec*>(S.mk_rec + (size_t)ms * LOB_MK_REC);  // [0]: under theta_t
        }
        pf.start(S.prof, b[0], lane);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (lane < 48) L.vars[k * 48 + lane] = cur[k].vv;
        fast_prefetch<NB>(S, t0 + stride * NB, stride, lane, cur);
        wave_lds_fence();
        int qv[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) qv[k] = tile_quant(L.vars[k * 48 + h[k].slot_cur * 16 + (lane & 15)]);
        pf.mark(13);  // header, state variables
        f64 qs[NB][LOB_N_ACTIONS];
        int hcnt[NB];
        u64* hbuf = reinterpret_cast<u64*>(L.vars);
        // (a book that turns out to have no valid memo record is evaluated on whatever its -- valid -- inputs are)
        q_values_fast<NB, true>(P, S, L, qv, lane, rec, qs, pf, 14, hbuf, hcnt);
        int wr_b[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (go[k] && !fast_memo_ok(mslot[k], mid[k], rec[k], ver, qv[k])) { fast_hand_back(S, 1, lpar, b[k], lane); go[k] = false; }
            wr_b[k] = go[k] ? b[k] : -1;
            if (!go[k] && real[k] && lane == 0) S.hl_rec[(size_t)b[k] * LOB_HL_REC] = LOB_HL_NONE;  // no list: the next act takes the general path for this book
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            if (!go[k]) continue;
            Rng g{P.seed, P.book_id_offset + (u64)b[k], h[k].rng_ctr};
            learn_delta_single<ALGO>(P, S.hdr + b[k], h[k], qs[k], LOB_QSA(P, S, b[k], h[k], ALGO), g, lane);
        }
        hit_list_store<NB>(S, lane, hcnt, wr_b, hbuf);
        pf.mark(19);  // argmax / delta / header stores
    }
}


// ---- Q(to_state, .) + TD error, one LANE per book ------------------------------------------------------------------
// The wave-per-book kernel above spends most of its instructions on work that is the same in all 64 lanes (headers,
// memo checks, ballots, the ordered continuation, argmax, the TD error); only the tile hashing itself is spread over
// the lanes.  With a lane per book every instruction does 64 books' worth: a lane walks its book's 64 group-1/2
// tilings (hash sum from the LDS table, nine tile indices, nine coarse-map bits from the LDS image), looks the coarse
// hits up in the exact map, and appends the tiles on marked weights to the book's hit list in the order Agent::getQ
// adds them (per action: group-1 tilings ascending with w1 | the same again with w2 | group-2 tilings ascending with
// w2 -- the list holds [all group-1 hits, w1][the same, w2][all group-2 hits], which is that order for every action).
// Q is then the memoised group-0 sum + the listed additions, exactly as act_light_kernel replays them.
// The exact-map look-ups are the only global loads inside the walk: they are issued for LOB_QL_CHUNK tilings at a
// time, unconditionally (a tile that is not a coarse hit reads word 0: one broadcast line) and consumed one stage
// later, so their latency hides behind the next stage's hashing -- a CU runs one 4-wave block (the LDS image).
#define LOB_QL_BLOCK 256
#define LOB_QL_ROW 25   /* u64 per lane in LDS: its hit list (LOB_HL_REC) + 1 pad */
#ifndef LOB_QL_CHUNK
#define LOB_QL_CHUNK 4  /* tilings per pipeline stage */
#endif
static_assert(32 % (2 * LOB_QL_CHUNK) == 0, "two stages per loop iteration");
__host__ __device__ inline size_t qlane_lds_bytes(int cwords4) { return (size_t)(2048 + 32 + cwords4 * 4) * 4 + (size_t)LOB_QL_BLOCK * LOB_QL_ROW * 8; }

template <int CH>
struct QlStage {
    i32 idx[CH][LOB_N_ACTIONS];
    uint32_t xw[CH][LOB_N_ACTIONS];  // exact-map word of the tile (word 0 if it is not a coarse hit)
    uint32_t maybe[CH];
};
// tilings j0 .. j0 + LOB_QL_CHUNK - 1 of group G (1: state variables 3..V-1, 2: all V): tile indices, coarse bits,
// exact-map loads issued
// VT: the number of state variables when it is the default 8 (every loop bound static: the table reads of a tiling
// leave together), 0: any (P.V; reads beyond it are made and discarded rather than branched around).
template <int G, int VT, int CH>
__device__ __forceinline__ void ql_issue(const DevParams& P, const DevState& S, const uint32_t* rnd, const uint32_t* terms, const uint32_t* coarse,
                                         const int* q, int j0, QlStage<CH>& st) {
    const uint32_t M = (uint32_t)P.M;
    const int V = VT ? VT : P.V;
    const int nf = G == 1 ? V - 3 : V;
    const int cs = P.cshift;
    constexpr int NI = VT ? (G == 1 ? VT - 3 : VT) : (G == 1 ? LOB_MAX_VARS - 3 : LOB_MAX_VARS);
#pragma unroll
    for (int u = 0; u < CH; u++) {
        const int j = j0 + u;
        uint32_t t[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int qi = G == 1 ? q[i + 3] : q[i];
            const int base = j * (1 + 2 * i);
            t[i] = rnd[(base + ((qi - base) & ~31) + 449 * i) & 2047];  // tile_coord without the wrap-around case (such books are not here)
        }
        uint32_t sum = rnd[(j + 449 * nf) & 2047];
#pragma unroll
        for (int i = 0; i < NI; i++) sum = mod_add(sum, (VT || i < nf) ? t[i] : 0u, M);
        uint32_t mb = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            const i32 x = tile_index(sum, terms[a], M);
            st.idx[u][a] = x;
            mb |= ((coarse[(uint32_t)x >> (cs + 5)] >> (((uint32_t)x >> cs) & 31)) & 1u) << a;
        }
        st.maybe[u] = mb;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) st.xw[u][a] = S.theta_nzx[((mb >> a) & 1u) ? (uint32_t)st.idx[u][a] >> 5 : 0u];
    }
}
// the tiles of a stage that fall on marked weights join the lane's list (`row[1 + n]`, n counts on beyond the capacity)
// `row`: u64 entries as in the record (tile index | action << 32 | w2 << 36), or -- learn_q_pair_kernel, tables below 2^27
// weights -- packed in 32 bits (tile index | action << 27 | w2 << 31)
__device__ __forceinline__ void ql_put(u64* row, int i, i32 tile, int a, bool g2) { row[i] = (u64)(uint32_t)tile | ((u64)a << 32) | (g2 ? 1ull << 36 : 0ull); }
__device__ __forceinline__ void ql_put(uint32_t* row, int i, i32 tile, int a, bool g2) { row[i] = (uint32_t)tile | ((uint32_t)a << 27) | (g2 ? 1u << 31 : 0u); }
__device__ __forceinline__ u64 ql_unpack(uint32_t e) { return (u64)(e & 0x7ffffffu) | ((u64)((e >> 27) & 15u) << 32) | ((u64)(e >> 31) << 36); }
template <int G, int CH, int CAP, class E>
__device__ __forceinline__ void ql_consume(const QlStage<CH>& st, E* row, int& n) {
#pragma unroll
    for (int u = 0; u < CH; u++) {
        uint32_t hit = 0;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) hit |= ((st.xw[u][a] >> ((uint32_t)st.idx[u][a] & 31)) & (st.maybe[u] >> a) & 1u) << a;
        if (hit) {
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) {
                if ((hit >> a) & 1u) {
                    if (n < CAP) ql_put(row, 1 + n, st.idx[u][a], a, G == 2);
                    n++;
                }
            }
        }
    }
}
template <int G, int VT, int CH = LOB_QL_CHUNK, int CAP = LOB_HL_CAP, class E = u64>
__device__ __forceinline__ void ql_group(const DevParams& P, const DevState& S, const uint32_t* rnd, const uint32_t* act_terms, const uint32_t* coarse,
                                         const int* q, E* row, int& n) {
    static_assert(32 % (2 * CH) == 0, "two stages per loop iteration");
    const uint32_t* terms = act_terms + G * LOB_N_ACTIONS;
    QlStage<CH> A, B;
    ql_issue<G, VT, CH>(P, S, rnd, terms, coarse, q, 0, A);
#pragma unroll 1
    for (int j0 = CH; j0 < 32; j0 += 2 * CH) {
        ql_issue<G, VT, CH>(P, S, rnd, terms, coarse, q, j0, B);
        ql_consume<G, CH, CAP, E>(A, row, n);
        if (j0 + CH < 32) ql_issue<G, VT, CH>(P, S, rnd, terms, coarse, q, j0 + CH, A);
        ql_consume<G, CH, CAP, E>(B, row, n);
    }
}

// ---- the same walk over the map folded over the actions (lob_state.h theta_nzd) -----------------------------------------
// One look-up per tiling instead of nine coarse-map reads: bit s of the group's theta_nzd map says whether ANY of the nine tiles
// (s + term[a]) mod M of the tiling with hash sum s lies on a written weight.  The walk only hashes and tests that bit (the
// words are requested a stage ahead); the few tilings that pass (7 % at 160 k written weights of 20 M) leave their sums in a
// short per-lane list in LDS and are resolved afterwards, two at a time: nine exact-map words each, then the entries in
// Agent::getQ's order (tilings ascending, actions ascending inside a tiling: what ql_consume produces).  No coarse image in
// LDS: the block is small and several share a CU.
#define LOB_QD_HCAP 12   /* tilings of one group that may pass per book (more: the general kernel takes the book) */
template <int G, int VT, int CH>
__device__ __forceinline__ void qd_issue(const DevParams& P, const DevState& S, const uint32_t* rnd, const int* q, int j0, uint32_t* sum_out, uint32_t* dw_out) {
    const uint32_t M = (uint32_t)P.M;
    const int V = VT ? VT : P.V;
    const int nf = G == 1 ? V - 3 : V;
    constexpr int NI = VT ? (G == 1 ? VT - 3 : VT) : (G == 1 ? LOB_MAX_VARS - 3 : LOB_MAX_VARS);
#pragma unroll
    for (int u = 0; u < CH; u++) {
        const int j = j0 + u;
        uint32_t t[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int qi = G == 1 ? q[i + 3] : q[i];
            const int base = j * (1 + 2 * i);
            t[i] = rnd[(base + ((qi - base) & ~31) + 449 * i) & 2047];  // tile_coord without the wrap-around case (such books are not here)
        }
        uint32_t sum = rnd[(j + 449 * nf) & 2047];
#pragma unroll
        for (int i = 0; i < NI; i++) sum = mod_add(sum, (VT || i < nf) ? t[i] : 0u, M);
        sum_out[u] = sum;
        dw_out[u] = S.theta_nzd[(G == 2 ? (size_t)P.M / 32 + 1 : 0) + (sum >> 5)];
    }
}
template <int G, int VT, int CAP, class E>
__device__ __forceinline__ void ql_group_d(const DevParams& P, const DevState& S, const uint32_t* rnd, const uint32_t* act_terms, const int* q,
                                           uint32_t* hits, E* row, int& n) {
#ifndef LOB_QD_CHUNK
#define LOB_QD_CHUNK 4   /* tilings per pipeline stage: their map words are in flight together (8: no faster, 0.0569 vs 0.0567 ms) */
#endif
    constexpr int CH = LOB_QD_CHUNK;
    static_assert(32 % (2 * CH) == 0, "two stages per loop iteration");
    const uint32_t* terms = act_terms + G * LOB_N_ACTIONS;
    const uint32_t M = (uint32_t)P.M;
    int nh = 0;
    uint32_t sA[CH], dA[CH], sB[CH], dB[CH];
#define LOB_QD_TAKE(SUM, DW)                                                        \
    _Pragma("unroll") for (int u = 0; u < CH; u++) {                                \
        if ((DW[u] >> (SUM[u] & 31)) & 1u) {                                        \
            if (nh < LOB_QD_HCAP) hits[nh] = SUM[u];                                \
            nh++;                                                                   \
        }                                                                           \
    }
    qd_issue<G, VT, CH>(P, S, rnd, q, 0, sA, dA);
#pragma unroll 1
    for (int j0 = CH; j0 < 32; j0 += 2 * CH) {
        qd_issue<G, VT, CH>(P, S, rnd, q, j0, sB, dB);
        LOB_QD_TAKE(sA, dA)
        if (j0 + CH < 32) qd_issue<G, VT, CH>(P, S, rnd, q, j0 + CH, sA, dA);
        LOB_QD_TAKE(sB, dB)
    }
#undef LOB_QD_TAKE
    if (nh > LOB_QD_HCAP) { n = CAP + 1; return; }  // (more tilings than the list holds: the general kernel)
    // the tilings that passed, two per round: nine exact-map words each in flight together
#pragma unroll 1
    for (int k = 0; __any(k < nh); k += 2) {
        const bool v0 = k < nh, v1 = k + 1 < nh;
        const uint32_t s0 = v0 ? hits[k] : 0u, s1 = v1 ? hits[k + 1] : 0u;
        i32 x0[LOB_N_ACTIONS], x1[LOB_N_ACTIONS];
        uint32_t w0[LOB_N_ACTIONS], w1[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            x0[a] = tile_index(s0, terms[a], M);
            x1[a] = tile_index(s1, terms[a], M);
            w0[a] = S.theta_nzx[v0 ? (uint32_t)x0[a] >> 5 : 0u];
            w1[a] = S.theta_nzx[v1 ? (uint32_t)x1[a] >> 5 : 0u];
        }
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            if (v0 && ((w0[a] >> ((uint32_t)x0[a] & 31)) & 1u)) {
                if (n < CAP) ql_put(row, 1 + n, x0[a], a, G == 2);
                n++;
            }
        }
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            if (v1 && ((w1[a] >> ((uint32_t)x1[a] & 31)) & 1u)) {
                if (n < CAP) ql_put(row, 1 + n, x1[a], a, G == 2);
                n++;
            }
        }
    }
}

// TR (Q(lambda)): the kernel also runs Agent::UpdateTraces for its books, BEFORE the Q evaluation as the reference does
// (its argmax draws come first): the light case of trace_light_kernel right here, the others through the list
// `tr_list` to trace_fast_kernel<.., 2>, which runs after this kernel (the entry carries argmax Q(s, .); Q(s, a) and
// the RNG counter are settled here).  The look-ups of the trace part are issued at the top and consumed after the
// tile walk, whose arithmetic hides them.
template <int ALGO, int VT, bool TR>
// acc_fuse (TR): as learn_q_pair_kernel -- the update of a book whose step leaves one new generation is added to that generation's
// slot here (double Q: in the sums of the vector the coin picked).
__global__ void __launch_bounds__(LOB_QL_BLOCK) learn_q_lane_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g, int lpar, u64 ver, int sid, int acc_fuse) {
    LOB_PS_REFS
    // (LOB_ALGO_DOUBLE_Q: DoubleQLearn on the fast path -- both weight vectors share the triples, the tiles, the maps and the
    // hit list; Q_a and Q_b continue from the memo's two records; its trace step is Watkins's, argmax over Q_a)
    static_assert(!TR || ALGO == LOB_ALGO_QLAMBDA || ALGO == LOB_ALGO_DOUBLE_Q, "the fused trace step is Watkins's");
    static_assert(ALGO != LOB_ALGO_DOUBLE_Q || TR, "double Q: with the fused trace step only");
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    __shared__ u64 claimed[512];  // (as trace_light_kernel)
    uint32_t* rnd = reinterpret_cast<uint32_t*>(fast_lds_raw);
    uint32_t* act_terms = rnd + 2048;
    uint32_t* coarse = act_terms + 32;
    u64* row = reinterpret_cast<u64*>(coarse + (size_t)P.cwords4 * 4) + (size_t)threadIdx.x * LOB_QL_ROW;
    for (int i = threadIdx.x; i < 512; i += LOB_QL_BLOCK) reinterpret_cast<uint4*>(rnd)[i] = reinterpret_cast<const uint4*>(rnd_g)[i];
    if (threadIdx.x < 27) act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
    for (int i = threadIdx.x; i < P.cwords4; i += LOB_QL_BLOCK) reinterpret_cast<uint4*>(coarse)[i] = reinterpret_cast<const uint4*>(+S.theta_nzc)[i];
    if (TR) for (int i = threadIdx.x; i < 512; i += LOB_QL_BLOCK) claimed[i] = LOB_CB_EMPTY;
    __syncthreads();
#pragma unroll 1
    for (int b = blockIdx.x * LOB_QL_BLOCK + threadIdx.x; b < S.B; b += gridDim.x * LOB_QL_BLOCK) {
        // everything whose address does not depend on the header leaves with it (both State rows: which is which comes with the header)
        const LHdr h = S.hdr[b];
        const int mslot = S.mk_slot[b];
        const int lslot_ = TR ? S.mk_slot_last[b] : -1;
        float4 vr[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) vr[r][i] = reinterpret_cast<const float4*>(S.vars + (size_t)b * 48 + r * 16)[i];
        f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = TR ? S.qs_last[(size_t)b * LOB_N_ACTIONS + a] : 0.0;
        LHdr* hp = S.hdr + b;
        u64* recp = S.hl_rec + (size_t)b * LOB_HL_REC;
        if (!h.stepped) { recp[0] = LOB_HL_NONE; continue; }
        Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
        // ---- UpdateTraces, first half: the decisions (QLearn::UpdateTraces, agent.cpp:272-280) ----
        f64 q_sa = LOB_QSA(P, S, b, h, ALGO);
        int amax = 0, lslot = -1, tq0 = 0, tq1 = 0, tq2 = 0;
        bool tlight = false;
        uint32_t tmarked = 0;
        int ttag = -1;  // tr_mslot of the generation the light step creates (trace_lane_kernel)
        int4 tl[8];  // the new generation's tiles (light case), fetched before the tile walk, stored after it
        if (TR) {
            lslot = lslot_;
            const int last = h.slot_cur ^ 1;
            const bool zero_last = (h.zero_mask >> last) & 1;
            const float4 vl = last ? vr[1][0] : vr[0][0];
            tq0 = tile_quant(vl.x); tq1 = tile_quant(vl.y); tq2 = tile_quant(vl.z);
            const int ls = lslot >= 0 ? lslot : 0;
            const int4 lid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ls * 4);
            const int tiles_ok = S.mk_tiles_ok[ls];
            tmarked = S.mk_marked[ls];
            if (tiles_ok == 3) ttag = ls | ((P.epi_epoch & 0x7fff) << 16);
            amax = argmax_ties(qs_last, g);
            int n_old = h.tr_n, kmax = P.trace_kmax;
            if (h.action != amax) kmax = 1;
            if (n_old > kmax - 1) n_old = kmax - 1;
            tlight = n_old == 0 && lslot >= 0 && !zero_last && lid.x == tq0 && lid.y == tq1 && lid.z == tq2 && lid.w == 1 && tiles_ok != 0;
            q_sa = sel9(qs_last, h.action);
            if (tlight) {
                const int4* src = reinterpret_cast<const int4*>(S.mk_tiles + ((size_t)lslot * LOB_N_ACTIONS + h.action) * 32);
#pragma unroll
                for (int i = 0; i < 8; i++) tl[i] = src[i];
            }
        }
        int q[LOB_MAX_VARS];
        {
            const bool c1 = h.slot_cur != 0;
            const float4 v0 = c1 ? vr[1][0] : vr[0][0], v1 = c1 ? vr[1][1] : vr[0][1], v2 = c1 ? vr[1][2] : vr[0][2], v3 = c1 ? vr[1][3] : vr[0][3];
            const f32 v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int i = 0; i < LOB_MAX_VARS; i++) q[i] = tile_quant(v[i]);
        }
        // a quantised variable within 1024 of INT_MIN (in practice: a NaN state variable) takes tile_coord's
        // wrap-around branch: left to the general kernel
        bool plain = true;
#pragma unroll
        for (int i = 0; i < LOB_MAX_VARS; i++) plain = plain && (i >= P.V || q[i] >= (int)0x80000400);
        const int ms = mslot >= 0 ? mslot : 0;
        const int4 mid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ms * 4);
        const f64* recm = S.mk_rec + (size_t)ms * LOB_MK_REC;  // [0]: under theta_t
        const u64 rver = reinterpret_cast<const u64*>(recm)[LOB_N_ACTIONS];
        int n = 0;
        bool ok = plain && mslot >= 0 && rver == ver && mid.x == q[0] && mid.y == q[1] && mid.z == q[2];
        if (ok) {
            ql_group<1, VT>(P, S, rnd, act_terms, coarse, q, row, n);
            // the group-1 additions once more, with w2 (quirk Q3)
            const int n1 = n;
            for (int i = 0; i < n1; i++) {
                if (n < LOB_HL_CAP) row[1 + n] = row[1 + i] | (1ull << 36);
                n++;
            }
            if (n <= LOB_HL_CAP) ql_group<2, VT>(P, S, rnd, act_terms, coarse, q, row, n);
            ok = n <= LOB_HL_CAP;
        }
        // ---- UpdateTraces, second half ----
        CbPending pend;
        pend.active = false;
        if (TR) {
            const u64 mb = __ballot(!tlight);  // (one atomic per wave for the list)
            if (mb) {
                int base = 0;
                const int leader = __builtin_ctzll(mb);
                if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&S.tr_list_n[lpar], __builtin_popcountll(mb));
                base = __shfl(base, leader);
                if (!tlight) S.tr_list[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u))] = b | (amax << 27);
            }
            if (tlight) {  // the new generation = the chosen action's 32 tiles, all alive, from the memo slot's record
                const int action = h.action;
                const int G = P.trace_gens;
                const int nh = (h.tr_head + 1) & (G - 1);
                int4* dst = reinterpret_cast<int4*>(S.tr_idx + ((size_t)b * G + nh) * 32);
                if (!((tmarked >> action) & 1u)) {  // (the act kernel marks them; if it could not, the marks are late)
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        nzx_mark_late(P, S, tl[i].x, sid); nzx_mark_late(P, S, tl[i].y, sid); nzx_mark_late(P, S, tl[i].z, sid); nzx_mark_late(P, S, tl[i].w, sid);
                    }
                    atomicOr(&S.mk_marked[lslot], 1u << action);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) dst[i] = tl[i];
                S.tr_alive[(size_t)b * G + nh] = 0xffffffffu;
                hp->tr_head = nh;
                hp->tr_n = 1;
                if (P.combine) {
                    *reinterpret_cast<int4*>(S.tr_sig + ((size_t)b * G + nh) * 4) = make_int4(tq0, tq1, tq2, action);
                    if (P.sarsa_lanes) S.tr_mslot[(size_t)b * G + nh] = ttag;
                    const u64 ch = cb_hash(tq0, tq1, tq2, action, 0xffffffffu);
                    const u64 seen = atomicCAS((unsigned long long*)&claimed[(ch >> 40) & 511], (unsigned long long)LOB_CB_EMPTY, (unsigned long long)ch);
                    if (seen != ch && S.cb_key[(uint32_t)ch & (uint32_t)(S.cb_slots - 1)] != ch)
                        cb_claim_issue(S, pend, tq0, tq1, tq2, action, 0xffffffffu, b * G + nh);  // (its answer is looked at last)
                    else S.tr_cbslot[(size_t)b * G + nh] = (i32)((uint32_t)ch & (uint32_t)(S.cb_slots - 1));  // (the hash's home slot: where accumulate_kernel looks first)
                }
            }
        }
        if (!ok) {  // no (valid) memo record, or a list longer than a record: the general kernel takes the book
            if (TR) { hp->td = q_sa; hp->rng_ctr = g.ctr; }  // (what it expects of the trace step)
            const int pos = atomicAdd(&S.slow_n[lpar * 2 + 1], 1);
            S.slow_list[(size_t)S.B + pos] = b;
            recp[0] = LOB_HL_NONE;
            cb_claim_finish(S, pend);
            if (TR && acc_fuse == 2) S.acc_pend[b] = 1;  // (trace_rest_kernel computes its TD error and adds its generations up: the work list names it)
            else if (TR && acc_fuse) {  // (its TD error comes later: accumulate_kernel takes every generation of the book)
                if (tlight) S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = b;
                else S.acc_pend[b] = 1;
            }
            continue;
        }
        f64 qs[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) qs[a] = recm[a];
        const f64 w1 = P.w1, w2 = P.w2;
        for (int i0 = 0; i0 < n; i0 += 4) {
            u64 ent[4];
            f64 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) ent[u] = i0 + u < n ? row[1 + i0 + u] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = i0 + u < n ? S.theta[(uint32_t)ent[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (v[u] == 0.0) continue;  // (+0.0 added to a sum that is never -0.0)
                const int a = (int)(ent[u] >> 32) & 15;
                const f64 x = ((ent[u] >> 36) & 1ull ? w2 : w1) * v[u];
#pragma unroll
                for (int c = 0; c < LOB_N_ACTIONS; c++) qs[c] = a == c ? qs[c] + x : qs[c];
            }
        }
        f64 delta;
        int vec = 0;  // (double Q: the vector the update goes into)
        if (ALGO == LOB_ALGO_DOUBLE_Q) {
            // Q_b(s', .) the same way: the memo's record under theta_b (written by the same memo_kernel launch: same version) + the
            // same additions with theta_b's weights
            const f64* recb = S.mk_rec_b + (size_t)ms * LOB_MK_REC;
            f64 qb[LOB_N_ACTIONS];
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) qb[a] = recb[a];
            for (int i0 = 0; i0 < n; i0 += 4) {
                u64 ent[4];
                f64 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) ent[u] = i0 + u < n ? row[1 + i0 + u] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = i0 + u < n ? S.theta_b[(uint32_t)ent[u]] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (v[u] == 0.0) continue;
                    const int a = (int)(ent[u] >> 32) & 15;
                    const f64 x = ((ent[u] >> 36) & 1ull ? w2 : w1) * v[u];
#pragma unroll
                    for (int c = 0; c < LOB_N_ACTIONS; c++) qb[c] = a == c ? qb[c] + x : qb[c];
                }
            }
            delta = learn_delta_double<false>(P, S, hp, h, b, qs, qb, q_sa, g, 0, nullptr, &vec);
        } else {
            delta = learn_delta_single<ALGO == LOB_ALGO_DOUBLE_Q ? LOB_ALGO_SARSA : ALGO>(P, hp, h, qs, q_sa, g, 0);
        }
        recp[0] = (u64)n;
        for (int i = 0; i < n; i++) recp[1 + i] = row[1 + i];
        cb_claim_finish(S, pend);
        if (TR && acc_fuse) {
            if (tlight) {
                // the book's one generation (age 0, all 32 tiles alive): alpha delta / 32 x e(0) into its slot
                const size_t gi = (size_t)b * P.trace_gens + ((h.tr_head + 1) & (P.trace_gens - 1));
                const f64 val = (P.alpha * delta) / (f64)LOB_N_TILINGS * (f64)P.trace_pow[0];
                if (!acc_generation(S, gi, 0xffffffffu, val, acc_copy(S, (int)(blockIdx.x * (LOB_QL_BLOCK / 64) + (threadIdx.x >> 6))), vec))
                    S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = (i32)((uint32_t)b | 0x80000000u);
            } else {
                S.acc_pend[b] = 0;
            }
            if (acc_fuse == 2 && tlight) S.acc_pend[b] = 0;  // (every stepped book's flag is current: trace_rest_kernel reads it for whatever list names the book)
        }
    }
}


// ---- the same with TWO lanes per book --------------------------------------------------------------------------------
// learn_q_lane_kernel runs one wave per SIMD (65 536 books are 1 024 waves) and waits half of the time: on the LDS table
// reads, on the exact-map words, on the dependent look-ups before and after the walk.  Here a 256-book block is 8 waves:
// waves 0-3 walk the books' group-1 tilings (and run the trace step), waves 4-7 the same books' group-2 tilings, then --
// after one block barrier -- finish: Q = (S0 + the group-1 additions, summed by the group-1 lane) + the group-2
// additions, argmax, TD error, the hit list.  Two waves per SIMD, the same instructions in total.
#ifndef LOB_QP_BOOKS
#define LOB_QP_BOOKS 128   /* books per block: 4 waves, two blocks per CU (57 KB of LDS each: no coarse map image any more) */
#endif
#define LOB_QP_BLOCK (2 * LOB_QP_BOOKS)
#define LOB_QP_OCC 2      /* blocks per CU the launch counts on */
#define LOB_QP_CAP1 22  /* entries of the group-1 half (both passes: 11 tiles on marked weights) */
#define LOB_QP_CAP2 14  /* entries of the group-2 half */
#define LOB_QP_ROW1 23  /* u32 per lane in LDS (entries packed in 32 bits: tables below 2^27 weights) */
#define LOB_QP_ROW2 15
#define LOB_QP_XCH 21   /* f64 per book handed from the group-1 lane to the group-2 lane: S0 + group-1 additions (9), entries (-1: none), Q(s, a), RNG counter; double Q: the same nine sums under theta_b */
__host__ __device__ inline size_t qpair_lds_bytes(int /*cwords4*/) {
    return (size_t)(2048 + 32) * 4 + (size_t)LOB_QP_BOOKS * (LOB_QP_ROW1 + LOB_QP_ROW2 + 2) * 4 + (size_t)LOB_QP_BOOKS * LOB_QP_XCH * 8 +
           (size_t)LOB_QP_BLOCK * LOB_QD_HCAP * 4;
}
template <int ALGO, int VT, bool TR>
// acc_fuse (Q(lambda), TR): the update of a book whose step leaves ONE new generation is added to that generation's slot right
// here, by the lane that has just computed the TD error (acc_generation); the listed books' by trace_lane_kernel; what neither
// can finish goes on acc_list for accumulate_kernel.
// LOB_ALGO_DOUBLE_Q (DoubleQLearn, agent.cpp:185-264,315-353; as learn_q_lane_kernel<LOB_ALGO_DOUBLE_Q>): both weight vectors share the
// triples, the tiles, the folded map and the hit list -- each lane walks its group ONCE and fetches theta AND theta_b for its
// hits; Q_a and Q_b continue from the memo's two records; the coin of the book's mt19937_64, the TD error and the addition to
// the slot (in the sums of the vector the coin picked) are the group-2 lane's; the trace step is Watkins's over Q_a.
__global__ void __launch_bounds__(LOB_QP_BLOCK) learn_q_pair_kernel(const DevParams* __restrict__ Pp, const DevState* __restrict__ Sp, const uint32_t* __restrict__ rnd_g, int lpar, u64 ver, int sid, int acc_fuse) {
    // parameters and state through their device-resident copies (3 KB of by-value arguments before: 145 spilled scalar registers
    // -> 10, 170 -> 166 vector registers; 0.0638 -> 0.0608 ms at 65 536 books, round 6.  Two blocks per CU either way: 61 KB of LDS each)
    const DevParams& P = *Pp;
    const DevState& S = *Sp;
    constexpr bool DQ = ALGO == LOB_ALGO_DOUBLE_Q;
    static_assert(!TR || ALGO == LOB_ALGO_QLAMBDA || DQ, "the fused trace step is Watkins's");
    static_assert(!DQ || TR, "double Q: with the fused trace step only");
    extern __shared__ __align__(16) unsigned char fast_lds_raw[];
    __shared__ u64 claimed[512];  // (as trace_light_kernel)
    uint32_t* rnd = reinterpret_cast<uint32_t*>(fast_lds_raw);
    uint32_t* act_terms = rnd + 2048;
    uint32_t* rows1 = act_terms + 32;                                        // [books][LOB_QP_ROW1]
    uint32_t* rows2 = rows1 + (size_t)LOB_QP_BOOKS * LOB_QP_ROW1;           // [books][LOB_QP_ROW2]
    f64* xch = reinterpret_cast<f64*>(rows2 + (size_t)LOB_QP_BOOKS * LOB_QP_ROW2 + ((LOB_QP_BOOKS * (LOB_QP_ROW1 + LOB_QP_ROW2)) & 1) + 0);
    uint32_t* hits = reinterpret_cast<uint32_t*>(xch + (size_t)LOB_QP_BOOKS * LOB_QP_XCH) + (size_t)threadIdx.x * LOB_QD_HCAP;  // this lane's passed tilings (ql_group_d)
    for (int i = threadIdx.x; i < 512; i += LOB_QP_BLOCK) reinterpret_cast<uint4*>(rnd)[i] = reinterpret_cast<const uint4*>(rnd_g)[i];
    if (threadIdx.x < 27) act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
    if (TR) for (int i = threadIdx.x; i < 512; i += LOB_QP_BLOCK) claimed[i] = LOB_CB_EMPTY;
    __syncthreads();
    const bool second = threadIdx.x >= LOB_QP_BOOKS;  // wave-uniform: the group-2 half
    const int lb = threadIdx.x - (second ? LOB_QP_BOOKS : 0);
    uint32_t* row = second ? rows2 + (size_t)lb * LOB_QP_ROW2 : rows1 + (size_t)lb * LOB_QP_ROW1;  // this lane's entries: row[1 + i]
    f64* xc = xch + (size_t)lb * LOB_QP_XCH;
    const f64 w1 = P.w1, w2 = P.w2;
#pragma unroll 1
    for (int base = blockIdx.x * LOB_QP_BOOKS; base < S.B; base += gridDim.x * LOB_QP_BOOKS) {
        const int b = base + lb;
        const bool real = b < S.B;
        const int bb = real ? b : 0;
        // everything whose address does not depend on the header leaves with it (both State rows: which is which comes with the header)
        const LHdr h = S.hdr[bb];
        const int mslot = S.mk_slot[bb];
        float4 vr[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) vr[r][i] = reinterpret_cast<const float4*>(S.vars + (size_t)bb * 48 + r * 16)[i];
        LHdr* hp = S.hdr + bb;
        u64* recp = S.hl_rec + (size_t)bb * LOB_HL_REC;
        const bool stepped = real && h.stepped != 0;
        int q[LOB_MAX_VARS];
        {
            const bool c1 = h.slot_cur != 0;
            const float4 v0 = c1 ? vr[1][0] : vr[0][0], v1 = c1 ? vr[1][1] : vr[0][1], v2 = c1 ? vr[1][2] : vr[0][2], v3 = c1 ? vr[1][3] : vr[0][3];
            const f32 v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int i = 0; i < LOB_MAX_VARS; i++) q[i] = tile_quant(v[i]);
        }
        bool plain = true;  // (see learn_q_lane_kernel)
#pragma unroll
        for (int i = 0; i < LOB_MAX_VARS; i++) plain = plain && (i >= P.V || q[i] >= (int)0x80000400);
        const int ms = mslot >= 0 ? mslot : 0;
        const int4 mid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ms * 4);
        const f64* recm = S.mk_rec + (size_t)ms * LOB_MK_REC;  // [0]: under theta_t
        const u64 rver = reinterpret_cast<const u64*>(recm)[LOB_N_ACTIONS];
        // both lanes of a book come to the same verdict
        const bool walk = stepped && plain && mslot >= 0 && rver == ver && mid.x == q[0] && mid.y == q[1] && mid.z == q[2];
        int n = 0;
        CbPending pend;
        pend.active = false;
        f64 v2[LOB_QP_CAP2];  // (group-2 lane) the weights of its entries
        f64 v2b[DQ ? LOB_QP_CAP2 : 1];  // ... under theta_b
        if (!second) {
            // ---- group-1 lane: the trace step's decisions, the group-1 walk, S0 + its additions, the trace step's stores ----
            Rng g{P.seed, P.book_id_offset + (u64)bb, h.rng_ctr};
            f64 q_sa = LOB_QSA(P, S, b, h, ALGO);
            int amax = 0, lslot = -1, tq0 = 0, tq1 = 0, tq2 = 0;
            bool tlight = false;
            uint32_t tmarked = 0;
            int ttag = -1;  // tr_mslot of the generation the light step creates (trace_lane_kernel)
            if (TR && stepped) {
                f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = S.qs_last[(size_t)bb * LOB_N_ACTIONS + a];
                lslot = S.mk_slot_last[bb];
                const int last = h.slot_cur ^ 1;
                const bool zero_last = (h.zero_mask >> last) & 1;
                const float4 vl = last ? vr[1][0] : vr[0][0];
                tq0 = tile_quant(vl.x); tq1 = tile_quant(vl.y); tq2 = tile_quant(vl.z);
                const int ls = lslot >= 0 ? lslot : 0;
                const int4 lid = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ls * 4);
                const int tiles_ok = S.mk_tiles_ok[ls];
                tmarked = S.mk_marked[ls];
                if (tiles_ok == 3) ttag = ls | ((P.epi_epoch & 0x7fff) << 16);
                amax = argmax_ties(qs_last, g);
                int n_old = h.tr_n, kmax = P.trace_kmax;
                if (h.action != amax) kmax = 1;
                if (n_old > kmax - 1) n_old = kmax - 1;
                tlight = n_old == 0 && lslot >= 0 && !zero_last && lid.x == tq0 && lid.y == tq1 && lid.z == tq2 && lid.w == 1 && tiles_ok != 0;
                q_sa = sel9(qs_last, h.action);
            }
            if (walk) {
                ql_group_d<1, VT, LOB_QP_CAP1, uint32_t>(P, S, rnd, act_terms, q, hits, row, n);
                const int n1 = n;  // the group-1 additions once more, with w2 (quirk Q3)
                for (int i = 0; i < n1; i++) {
                    if (n < LOB_QP_CAP1) row[1 + n] = row[1 + i] | (1u << 31);
                    n++;
                }
            }
            // S0 + the group-1 additions, in their order
            f64 qs[LOB_N_ACTIONS];
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) qs[a] = walk ? recm[a] : 0.0;
            if (walk && n <= LOB_QP_CAP1) {
                for (int i0 = 0; i0 < n; i0 += 4) {
                    u64 ent[4];
                    f64 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) ent[u] = i0 + u < n ? ql_unpack(row[1 + i0 + u]) : 0ull;
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = i0 + u < n ? S.theta[(uint32_t)ent[u]] : 0.0;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (v[u] == 0.0) continue;  // (+0.0 added to a sum that is never -0.0)
                        const int a = (int)(ent[u] >> 32) & 15;
                        const f64 x = ((ent[u] >> 36) & 1ull ? w2 : w1) * v[u];
#pragma unroll
                        for (int c = 0; c < LOB_N_ACTIONS; c++) qs[c] = a == c ? qs[c] + x : qs[c];
                    }
                }
                for (int i = 0; i < n && i < LOB_HL_MAX; i++) recp[1 + i] = ql_unpack(row[1 + i]);  // its part of the book's hit list
            }
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) xc[a] = qs[a];
            if (DQ) {
                // Q_b the same way: the memo's record under theta_b (the same memo_kernel launch: same version) + the same additions
                // with theta_b's weights
                const f64* recb = S.mk_rec_b + (size_t)ms * LOB_MK_REC;
                f64 qb[LOB_N_ACTIONS];
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) qb[a] = walk ? recb[a] : 0.0;
                if (walk && n <= LOB_QP_CAP1) {
                    for (int i0 = 0; i0 < n; i0 += 4) {
                        u64 ent[4];
                        f64 v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) ent[u] = i0 + u < n ? ql_unpack(row[1 + i0 + u]) : 0ull;
#pragma unroll
                        for (int u = 0; u < 4; u++) v[u] = i0 + u < n ? S.theta_b[(uint32_t)ent[u]] : 0.0;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (v[u] == 0.0) continue;
                            const int a = (int)(ent[u] >> 32) & 15;
                            const f64 x = ((ent[u] >> 36) & 1ull ? w2 : w1) * v[u];
#pragma unroll
                            for (int c = 0; c < LOB_N_ACTIONS; c++) qb[c] = a == c ? qb[c] + x : qb[c];
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) xc[12 + a] = qb[a];
            }
            reinterpret_cast<int*>(xc + LOB_N_ACTIONS)[0] = (walk && n <= LOB_QP_CAP1) ? n : -1;
            // for the group-2 lane's addition to the slot of the generation a light step creates -- 0: listed, else bits 0-5: 1 + the
            // generation's ring slot, bits 6-30: 1 + the combine slot this lane has put on record for it (0: a claim is on its way)
            uint32_t light_code = (TR && stepped && tlight) ? 1u + (uint32_t)((h.tr_head + 1) & (P.trace_gens - 1)) : 0u;
            // ---- UpdateTraces, second half (see learn_q_lane_kernel) ----
            if (TR) {
                const bool listed = stepped && !tlight;
                const u64 mb = __ballot(listed);  // (one atomic per wave for the list)
                if (mb) {
                    int lbase = 0;
                    const int leader = __builtin_ctzll(mb);
                    if ((int)(threadIdx.x & 63) == leader) lbase = atomicAdd(&S.tr_list_n[lpar], __builtin_popcountll(mb));
                    lbase = __shfl(lbase, leader);
                    if (listed) S.tr_list[lbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u))] = b | (amax << 27);
                }
                if (stepped && tlight) {  // the new generation = the chosen action's 32 tiles, all alive, from the memo slot's record
                    const int action = h.action;
                    const int G = P.trace_gens;
                    const int nh = (h.tr_head + 1) & (G - 1);
                    const int4* src = reinterpret_cast<const int4*>(S.mk_tiles + ((size_t)lslot * LOB_N_ACTIONS + action) * 32);
                    int4* dst = reinterpret_cast<int4*>(S.tr_idx + ((size_t)b * G + nh) * 32);
                    int4 tl[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) tl[i] = src[i];
                    if (!((tmarked >> action) & 1u)) {  // (the act kernel marks them; if it could not, the marks are late)
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            nzx_mark_late(P, S, tl[i].x, sid); nzx_mark_late(P, S, tl[i].y, sid); nzx_mark_late(P, S, tl[i].z, sid); nzx_mark_late(P, S, tl[i].w, sid);
                        }
                        atomicOr(&S.mk_marked[lslot], 1u << action);
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) dst[i] = tl[i];
                    S.tr_alive[(size_t)b * G + nh] = 0xffffffffu;
                    hp->tr_head = nh;
                    hp->tr_n = 1;
                    if (P.combine) {
                        *reinterpret_cast<int4*>(S.tr_sig + ((size_t)b * G + nh) * 4) = make_int4(tq0, tq1, tq2, action);
                        if (P.sarsa_lanes) S.tr_mslot[(size_t)b * G + nh] = ttag;
                        const u64 ch = cb_hash(tq0, tq1, tq2, action, 0xffffffffu);
                        const u64 seen = atomicCAS((unsigned long long*)&claimed[(ch >> 40) & 511], (unsigned long long)LOB_CB_EMPTY, (unsigned long long)ch);
                        if (seen != ch && S.cb_key[(uint32_t)ch & (uint32_t)(S.cb_slots - 1)] != ch)
                            cb_claim_issue(S, pend, tq0, tq1, tq2, action, 0xffffffffu, b * G + nh);
                        else {
                            const uint32_t home = (uint32_t)ch & (uint32_t)(S.cb_slots - 1);
                            S.tr_cbslot[(size_t)b * G + nh] = (i32)home;
                            light_code |= (home + 1u) << 6;
                        }
                    }
                }
                // Q(s, a) and the RNG counter after the trace step's draws: for the general kernel if the book is handed back
                if (stepped) { hp->td = q_sa; hp->rng_ctr = g.ctr; }
            }
            // ... and for the group-2 lane
            reinterpret_cast<uint32_t*>(xc + LOB_N_ACTIONS)[1] = light_code;
            xc[LOB_N_ACTIONS + 1] = q_sa;
            reinterpret_cast<u64*>(xc)[LOB_N_ACTIONS + 2] = g.ctr;
        } else {
            // ---- group-2 lane: the group-2 walk (its weights can be fetched before the other half is in) ----
            if (walk) ql_group_d<2, VT, LOB_QP_CAP2, uint32_t>(P, S, rnd, act_terms, q, hits, row, n);
#pragma unroll
            for (int i = 0; i < LOB_QP_CAP2; i++) v2[i] = (walk && n <= LOB_QP_CAP2 && i < n) ? S.theta[row[1 + i] & 0x7ffffffu] : 0.0;  // (n beyond the row: entries not written)
            if (DQ) {
#pragma unroll
                for (int i = 0; i < LOB_QP_CAP2; i++) v2b[DQ ? i : 0] = (walk && n <= LOB_QP_CAP2 && i < n) ? S.theta_b[row[1 + i] & 0x7ffffffu] : 0.0;
            }
        }
        if (TR && acc_fuse && !second) { cb_claim_finish(S, pend); pend.active = false; }  // (the group-2 lane adds to the slot after the barrier)
        __syncthreads();
        if (!second) {
            cb_claim_finish(S, pend);
        } else {
            // ---- group-2 lane: Q, argmax, the TD error, the rest of the hit list ----
            const int n1 = reinterpret_cast<const int*>(xc + LOB_N_ACTIONS)[0];
            const uint32_t light_code = TR ? reinterpret_cast<const uint32_t*>(xc + LOB_N_ACTIONS)[1] : 0u;
            const bool light = light_code != 0;
            if (stepped) {
                if (!(walk && n1 >= 0 && n <= LOB_QP_CAP2 && n1 + n <= LOB_HL_MAX)) {
                    // no (valid) memo record, or a half-list longer than its row: the general kernel takes the book
                    const int pos = atomicAdd(&S.slow_n[lpar * 2 + 1], 1);
                    S.slow_list[(size_t)S.B + pos] = b;
                    recp[0] = LOB_HL_NONE;
                    // (its TD error comes later: every generation of the book is accumulate_kernel's -- a light book goes on the
                    // list here, a listed one when trace_lane_kernel meets it)
                    if (TR && acc_fuse == 2) S.acc_pend[b] = 1;  // (trace_rest_kernel: its TD error and its sums, from the work list)
                    else if (TR && acc_fuse) {
                        if (light) S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = b;
                        else S.acc_pend[b] = 1;
                    }
                } else {
                    f64 qs[LOB_N_ACTIONS];
#pragma unroll
                    for (int a = 0; a < LOB_N_ACTIONS; a++) qs[a] = xc[a];
#pragma unroll
                    for (int i = 0; i < LOB_QP_CAP2; i++) {
                        if (i < n && v2[i] != 0.0) {
                            const int a = (int)(row[1 + i] >> 27) & 15;
                            const f64 x = w2 * v2[i];
#pragma unroll
                            for (int c = 0; c < LOB_N_ACTIONS; c++) qs[c] = a == c ? qs[c] + x : qs[c];
                        }
                    }
                    // Q(s, a) / the RNG counter after the trace step, as the group-1 lane left them
                    const f64 q_sa = xc[LOB_N_ACTIONS + 1];
                    Rng g{P.seed, P.book_id_offset + (u64)b, reinterpret_cast<const u64*>(xc)[LOB_N_ACTIONS + 2]};
                    f64 delta;
                    int vec = 0;  // (double Q: the vector the update goes into)
                    if (DQ) {
                        f64 qb[LOB_N_ACTIONS];
#pragma unroll
                        for (int a = 0; a < LOB_N_ACTIONS; a++) qb[a] = xc[12 + a];
#pragma unroll
                        for (int i = 0; i < LOB_QP_CAP2; i++) {
                            if (i < n && v2b[DQ ? i : 0] != 0.0) {
                                const int a = (int)(row[1 + i] >> 27) & 15;
                                const f64 x = w2 * v2b[DQ ? i : 0];
#pragma unroll
                                for (int c = 0; c < LOB_N_ACTIONS; c++) qb[c] = a == c ? qb[c] + x : qb[c];
                            }
                        }
                        delta = learn_delta_double<false>(P, S, hp, h, b, qs, qb, q_sa, g, 0, nullptr, &vec);
                    } else {
                        delta = learn_delta_single<DQ ? LOB_ALGO_SARSA : ALGO>(P, hp, h, qs, q_sa, g, 0);
                    }
                    recp[0] = (u64)(n1 + n);
                    for (int i = 0; i < n; i++) recp[1 + n1 + i] = ql_unpack(row[1 + i]);
                    if (TR && acc_fuse && (!light || acc_fuse == 2)) S.acc_pend[b] = 0;  // (acc_fuse 2: every stepped book's flag is current)
                    if (TR && acc_fuse && light) {
                        // the book's one generation (age 0, all 32 tiles alive): alpha delta / 32 x e(0) into its slot
                        const size_t gi = (size_t)b * P.trace_gens + ((light_code & 63u) - 1u);
                        const f64 val = (P.alpha * delta) / (f64)LOB_N_TILINGS * (f64)P.trace_pow[0];
                        const int xcd = acc_copy(S, (int)(blockIdx.x * (LOB_QP_BLOCK / 64) + (threadIdx.x >> 6)));
                        const uint32_t hs = (light_code >> 6) & 0x1ffffffu;
                        bool added;
                        if (hs) {  // (the slot the group-1 lane recorded; the generation's signature = last_state's triple + the action)
                            const float4 vl = (h.slot_cur ^ 1) ? vr[1][0] : vr[0][0];
                            added = acc_generation_at(S, gi, (int)(hs - 1u), make_int4(tile_quant(vl.x), tile_quant(vl.y), tile_quant(vl.z), h.action), 0xffffffffu, val, xcd, vec);
                        } else added = acc_generation(S, gi, 0xffffffffu, val, xcd, vec);
                        if (!added) S.acc_list[atomicAdd(&S.acc_list_n[lpar], 1)] = (i32)((uint32_t)b | 0x80000000u);
                    }
                }
            } else if (real) {
                recp[0] = LOB_HL_NONE;
            }
        }
        __syncthreads();  // (the rows and the hand-over area are reused by the block's next batch of books)
    }
}

#endif
