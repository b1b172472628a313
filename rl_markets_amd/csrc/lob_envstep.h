// env_step_kernel: the learner step's action selection + performAction of every book, a lane per book -- the same work as
// env_kernel<64, 2, 1> (lob_kernels.h: act_light_book + perform_action_fast), re-ordered around what actually bounds it.
//
// Measured (tools/exp_prof.py, tools/ubench/rowload): with one wave per SIMD and 65 536 lanes asking at once, a dependent
// memory round trip costs ~6 600 clocks (2.7 us) whatever its size, and env_kernel<64, 2, 1> was a chain of ~40 of them --
// header -> hit list -> memo record -> weights -> (more list entries -> more weights) -> agent scalars -> track entry of the
// quotes -> BookMeta -> [per pass: row] -> window state -> window slots -> track entry of the state -> memo slot -> ... --
// 173 000 of its 256 000 clocks outside the event loop.  Nothing in that chain is a real dependency beyond two levels:
//   round 1 (needs the book id only):   header, k, rec_cur, memo slot, the WHOLE hit list (192 B), all agent scalars,
//                                       the PnL windows' registers, BookMeta, the tick table;
//   round 2 (needs round 1's indices):  memo record + mark bits, every listed weight, the track entries of events k - 1
//                                       (quotes) and k (first pass), the current snapshot's rows and the first row the
//                                       step applies, the two window slots that fall out;
//   event loop:                         entry k + 1 and its first row requested one pass ahead (event_loop_fast);
//   round 3 (needs the new state):      the memo slot's hash / stamp.
// The arithmetic is act_light_book's and perform_action_fast's, call for call.  Books the hit-list replay cannot serve go
// on the act work list exactly as before (general act kernel + env_kernel<64, 2, 2>).
#ifndef LOB_ENVSTEP_H
#define LOB_ENVSTEP_H

#include "lob_fast.h"

__device__ inline Track track_load(const Track* p) {
    // eight 16-byte loads, in flight together
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = q[i];
    Track t;
    uint4* d = reinterpret_cast<uint4*>(&t);
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = w[i];
    return t;
}

// INLINE_GENERAL: a book without a usable hit list (a list longer than a record, a State still at its constructor zeros, lists
// voided by a late map bit) gets its action from the general evaluation right here -- act_book, the whole wave on one book at a
// time -- instead of going through the work list to act_kernel + env_kernel<64, 2, 2>: two launches that find their list empty
// in all but a handful of steps per episode no longer exist.  (false: the work-list version; the engine uses it for the one
// step per episode in which EVERY book takes the general path.)
// DQ: DoubleQLearn on the fast path (DoubleAgent::action, agent.cpp:196-204): Q_a and Q_b continue from the memo's two records
// with the same listed additions under theta / theta_b, the policy sees (Q_a + Q_b) / 2, both vectors' values are kept for the
// learn kernel (qs_last, qs_last_b).
// LANES: books per wave.  64; or 32 with the register budget of TWO waves per SIMD (an experiment, LOB_ENV_STEP_LANES=32: the
// kernel is one dependent instruction stream per wave at one wave per SIMD -- two half-full waves per SIMD fill each other's
// stalls if they fit; they only fit by spilling, see NOTES).
template <int LANES> struct EnvStepOcc { static constexpr int waves = LANES == 64 ? 1 : 2; };
template <bool INLINE_GENERAL, bool DQ = false, int LANES = 64>
__global__ void __launch_bounds__(64, EnvStepOcc<LANES>::waves) env_step_kernel(const DevParams* __restrict__ Pp, const DevState* __restrict__ Sp, int step_id, int par, EnvFuse F,
                                                                                  const uint32_t* __restrict__ rnd_g) {
    static_assert(LANES == 64 || !INLINE_GENERAL, "the half-full variant leaves the books without a list to the work list");
    const DevParams& P = *Pp;
    // The state through its device-resident copy (lob_state.h DevState::self), not as 1.7 KB of by-value arguments that are all
    // loaded at the kernel's entry and live across it: 104 spilled scalar registers and 160 vector registers parked in AGPRs ->
    // 13 and 26 (round 6).  The kernel's time did not follow (0.101 -> 0.099-0.103 ms: it is not bound by those instructions);
    // double Q's variant, which spilled more, gained 4 us.
    const DevState& S = *Sp;
    __shared__ TickLds tick_lds;
    __shared__ EnvSlot lds_env[LANES == 64 ? 64 : LANES + 1];  // (32-book waves: one more slot, for the idle upper half -- below)
    __shared__ LearnLds1 lds_learn;
#ifdef LOB_PROF
    const long long t_entry = clock64();
#else
    const long long t_entry = 0;
#endif
    const int t = blockIdx.x * LANES + (int)threadIdx.x;
    const bool valid = (int)threadIdx.x < LANES && t < S.B;
    const int b = valid ? t : S.B - 1;  // (lanes past the batch fetch the last book's inputs and do nothing with them)
    const int B = S.B;

    // ---- round 1: everything addressed by the book id ------------------------------------------------------------------
    f64 tk_lb = 0.0, tk_tick = 0.0, tk_pp = 0.0;
    i64 tk_cum = 0;
    i32 tk_pt = 0;
    if (threadIdx.x < LOB_MAX_BANDS) {
        const int i = threadIdx.x;
        tk_lb = P.band_lb[i]; tk_tick = P.band_tick[i]; tk_cum = P.band_cum[i]; tk_pp = P.band_pp[i]; tk_pt = P.band_pt[i];
    }
    EnvCtx c(P, S, b, &tick_lds);  // (a replayed stream: the book's phase is a round-1 load too)
    LHdr* hp = S.hdr + b;
    const LHdr h0 = *hp;
    const int k0 = S.k[b], rc0 = S.rec_cur[b];
    const int mslot = S.mk_slot[b];
    const bool dirty = S.hl_dirty[0] == F.sid_prev;
    // the hit list: its first 128 bytes (the count and 15 entries: all of it but for a book in a thousand) now, entries 15-22
    // below, the ones beyond LOB_HL_CAP (a book in ten thousand: up to LOB_HL_MAX) in a pass of their own behind the replay
    constexpr int HLQ = LOB_HL_ROW / 2;
    ulonglong2 hl[HLQ];
    const ulonglong2* hl_p = reinterpret_cast<const ulonglong2*>(S.hl_rec + (size_t)b * LOB_HL_REC);
#pragma unroll
    for (int i = 0; i < HLQ; i++) hl[i] = i < 8 ? hl_p[i] : make_ulonglong2(0ull, 0ull);
    EnvR er;
    env_load(S, b, er);
    RMReg wu, wd;
    rm_load(S.pnl_ups, b, wu);
    rm_load(S.pnl_downs, b, wd);
    const int m_n_track = S.meta[b].n_track, m_complete = S.meta[b].complete;

    // (tick table and the book's scalars to LDS)
    if (threadIdx.x < LOB_MAX_BANDS) {
        const int i = threadIdx.x;
        tick_lds.lb[i] = tk_lb; tick_lds.tick[i] = tk_tick; tick_lds.cum[i] = tk_cum; tick_lds.pp[i] = tk_pp; tick_lds.pt[i] = tk_pt;
    }
    if (threadIdx.x == 0) tick_lds.n = P.n_bands;
    // (the idle upper half of a 32-book wave -- the experiments build's LOB_ENV_STEP_LANES=32 -- shares ONE slot of its own: whatever those
    // lanes may come to store through `e`, no live lane's registers are there)
    EnvR& e = lds_env[LANES == 64 || threadIdx.x < LANES ? (threadIdx.x & (LANES - 1)) : LANES].e;
    if (LANES == 64 || threadIdx.x < LANES) e = er;
    __syncthreads();
#ifdef LOB_PROF
    const long long t_r1 = clock64();  // (round 1 is in: its scalars have gone to LDS)
#endif

    // ---- who steps, and from which list (act_light_book) ----------------------------------------------------------------
    const int cur_slot = h0.slot_cur ^ 1;  // swap(state, last_state)
    const bool alive = valid && !h0.done;
    const bool open = is_open(P, h0.time_ms);
    const int n_list = hl[0].x == LOB_HL_NONE ? -1 : (int)hl[0].x;
    bool ok = alive && open && !dirty && !((h0.zero_mask >> (cur_slot ^ 1)) & 1) && mslot >= 0 && n_list >= 0;
    if (ok && n_list > 15) {  // (a round trip of its own, for the waves that hold such a book)
#pragma unroll
        for (int i = 8; i < HLQ; i++) hl[i] = hl_p[i];
    }

    // ---- round 2: everything addressed by what round 1 brought ----------------------------------------------------------
    const int ms = mslot >= 0 ? mslot : 0;
    const MemoRec rec = *reinterpret_cast<const MemoRec*>(S.mk_rec + ((size_t)S.mk_slots + ms) * LOB_MK_REC);  // [1]: after the last update
    const int tiles_ok = S.mk_tiles_ok[ms];
    const uint32_t mk_bits = S.mk_marked[ms];
    // entry i of the list = word 1 + i of the record
#define LOB_HL_ENT(i) (((1 + (i)) & 1) ? hl[(1 + (i)) >> 1].y : hl[(1 + (i)) >> 1].x)
    f64 wv[LOB_HL_CAP];
#pragma unroll
    for (int i = 0; i < LOB_HL_CAP; i++) {
        wv[i] = 0.0;
        if (ok && i < n_list) wv[i] = S.theta[(uint32_t)LOB_HL_ENT(i)];
    }
    f64 wvb[DQ ? LOB_HL_CAP : 1];
    f64 s0b[DQ ? LOB_N_ACTIONS : 1];
    if (DQ) {
        const f64* recb = S.mk_rec_b + ((size_t)S.mk_slots + ms) * LOB_MK_REC;  // (written with `rec` by one memo_kernel launch: its version)
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) s0b[DQ ? a : 0] = recb[a];
#pragma unroll
        for (int i = 0; i < LOB_HL_CAP; i++) {
            wvb[DQ ? i : 0] = 0.0;
            if (ok && i < n_list) wvb[DQ ? i : 0] = S.theta_b[(uint32_t)LOB_HL_ENT(i)];
        }
    }
    const int kk = k0 > 0 ? k0 : 1;  // (a live book has consumed its warm-up: k0 >= 1)
    // (of the previous event's entry the quotes and the market order read tp_val, spread_mean, a_tv, b_tv: bytes 64-95)
    Track tprev;
    {
        const uint4* q = reinterpret_cast<const uint4*>(&c.track(kk - 1));
        const uint4 w4 = q[4], w5 = q[5];
        uint4* d = reinterpret_cast<uint4*>(&tprev);
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = make_uint4(0, 0, 0, 0);
        d[4] = w4; d[5] = w5;
    }
    Track tcur = track_load(&c.track(k0));
    RowFull cur, first;
    row_full_load(c, rc0, cur);
    {
        const int last_row = S.n_events - 1;
        row_full_load(c, rc0 + 1 < last_row ? rc0 + 1 : last_row, first);
    }
    rm_prep(S.pnl_ups, B, b, wu);
    rm_prep(S.pnl_downs, B, b, wd);
#ifdef LOB_PROF
    __builtin_amdgcn_s_waitcnt(0);     // (the instrumented build waits for round 2 here, to time it apart from the arithmetic that follows)
    const long long t_r2 = clock64();
#endif

    // ---- the action (act_light_book, from here on) ------------------------------------------------------------------------
    int action = 0;
    bool go = false;
    if (valid) {
        if (h0.done) {
            hp->stepped = 0;
        } else if (!open) {  // environment.isTerminal()
            hp->slot_cur = cur_slot; hp->done = 1; hp->stepped = 0; S.done[b] = 1;
        } else {
            ok = ok && rec.ver == F.ver;
            if (!ok) {
                if (!INLINE_GENERAL) {
                    const int pos = atomicAdd(&S.slow_n[F.lpar * 2 + 0], 1);
                    S.slow_list[pos] = b;
                }
            } else {
                f64 q[LOB_N_ACTIONS];
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) q[a] = rec.s0[a];
                const f64 w1 = P.w1, w2 = P.w2;
#pragma unroll
                for (int i = 0; i < LOB_HL_CAP; i++) {
                    const u64 ent = LOB_HL_ENT(i);
                    const f64 v = wv[i];
                    if (i < n_list && v != 0.0) {  // (+0.0 added to a sum that is never -0.0)
                        const int a_ = (int)(ent >> 32) & 15;
                        const f64 x_ = ((ent >> 36) & 1ull ? w2 : w1) * v;
#pragma unroll
                        for (int a = 0; a < LOB_N_ACTIONS; a++) q[a] = a_ == a ? q[a] + x_ : q[a];
                    }
                }
                // the additions beyond LOB_HL_CAP, in their order behind the others: entries and weights fetched here (two round
                // trips for the wave that holds such a book -- instead of a whole-wave evaluation of all 128 terms for the book)
                constexpr int NT = LOB_HL_MAX - LOB_HL_CAP;
                u64 te[NT];
                if (n_list > LOB_HL_CAP) {
                    const u64* tp = S.hl_rec + (size_t)b * LOB_HL_REC + 1 + LOB_HL_CAP;
#pragma unroll
                    for (int j = 0; j < NT; j++) te[j] = LOB_HL_CAP + j < n_list ? tp[j] : 0ull;
                    f64 tv[NT];
#pragma unroll
                    for (int j = 0; j < NT; j++) tv[j] = LOB_HL_CAP + j < n_list ? S.theta[(uint32_t)te[j]] : 0.0;
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        if (LOB_HL_CAP + j < n_list && tv[j] != 0.0) {
                            const int a_ = (int)(te[j] >> 32) & 15;
                            const f64 x_ = ((te[j] >> 36) & 1ull ? w2 : w1) * tv[j];
#pragma unroll
                            for (int a = 0; a < LOB_N_ACTIONS; a++) q[a] = a_ == a ? q[a] + x_ : q[a];
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) S.qs_last[(size_t)b * LOB_N_ACTIONS + a] = q[a];
                if (DQ) {
                    f64 qb[LOB_N_ACTIONS];
#pragma unroll
                    for (int a = 0; a < LOB_N_ACTIONS; a++) qb[a] = s0b[DQ ? a : 0];
#pragma unroll
                    for (int i = 0; i < LOB_HL_CAP; i++) {
                        const u64 ent = LOB_HL_ENT(i);
                        const f64 v = wvb[DQ ? i : 0];
                        if (i < n_list && v != 0.0) {
                            const int a_ = (int)(ent >> 32) & 15;
                            const f64 x_ = ((ent >> 36) & 1ull ? w2 : w1) * v;
#pragma unroll
                            for (int a = 0; a < LOB_N_ACTIONS; a++) qb[a] = a_ == a ? qb[a] + x_ : qb[a];
                        }
                    }
                    if (n_list > LOB_HL_CAP) {  // (the same additions under theta_b)
                        f64 tv[NT];
#pragma unroll
                        for (int j = 0; j < NT; j++) tv[j] = LOB_HL_CAP + j < n_list ? S.theta_b[(uint32_t)te[j]] : 0.0;
#pragma unroll
                        for (int j = 0; j < NT; j++) {
                            if (LOB_HL_CAP + j < n_list && tv[j] != 0.0) {
                                const int a_ = (int)(te[j] >> 32) & 15;
                                const f64 x_ = ((te[j] >> 36) & 1ull ? w2 : w1) * tv[j];
#pragma unroll
                                for (int a = 0; a < LOB_N_ACTIONS; a++) qb[a] = a_ == a ? qb[a] + x_ : qb[a];
                            }
                        }
                    }
#pragma unroll
                    for (int a = 0; a < LOB_N_ACTIONS; a++) {
                        S.qs_last_b[(size_t)b * LOB_N_ACTIONS + a] = qb[a];
                        q[a] = (q[a] + qb[a]) / 2.0;  // qs[a] = (getQ + getQb) / 2.0f
                    }
                }
                Rng g{P.seed, P.book_id_offset + (u64)b, h0.rng_ctr};
                action = policy_sample(P, q, false, g);
                hp->slot_cur = cur_slot;
                hp->action = action;
                hp->stepped = 1;
                hp->rng_ctr = g.ctr;
                // the tiles of (this state, this action)'s trace generation are marked in the written-weights maps before
                // this step's learn kernel looks (by memo_kernel, a lane per tile), once per (triple, action)
                const uint32_t marked = tiles_ok ? mk_bits : 0x1ffu;
                if (!((marked >> action) & 1u)) {
                    const int pos = atomicAdd(&S.mk_markcount[0], 1);
                    if (pos < S.mk_slots) S.mk_marklist[pos] = mslot * 16 + action;
                    else mark_generation(P, S, mslot, action);
                    atomicOr(&S.mk_marked[mslot], 1u << action);
                }
                const u64 act = __ballot(1);
                if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) cnt_add(S, 5, (unsigned long long)__builtin_popcountll(act));
                go = true;
            }
        }
    }
#undef LOB_HL_ENT
    if (INLINE_GENERAL) {
        // the books the replay could not serve: Agent::action in full (all 128 terms of every Q), one book at a time, the wave
        // on it as in act_kernel (act_book: swap, terminal check, Q(last_state, .), policy, header, tile marks)
        u64 todo = __ballot(valid && alive && open && !ok);
        if (todo) {
            if (threadIdx.x == 0) cnt_add(S, 2, (u64)__builtin_popcountll(todo));  // (lob_get_path_stats [6])
            const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
            uint4* dst = reinterpret_cast<uint4*>(lds_learn.rnd);
            for (int i = threadIdx.x; i < 512; i += 64) dst[i] = src[i];
            if (threadIdx.x < 27) lds_learn.act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
            wave_lds_fence();
            while (todo) {
                const int src_lane = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int b_src = __builtin_amdgcn_readlane(b, src_lane);
                int act = -1;
                act_book<DQ ? LOB_ALGO_DOUBLE_Q : LOB_ALGO_SARSA, LearnLds1>(P, S, lds_learn, 0, (int)threadIdx.x, b_src, 0, par, &act);
                if ((int)threadIdx.x == src_lane && act >= 0) { action = act; go = true; }
            }
        }
    }

    // ---- performAction -----------------------------------------------------------------------------------------------------
    i64 d_steps = 0, d_events = 0;
    if (go) {
        c.pre_prev = &tprev;
        c.pre_n_track = m_n_track;
        c.pre_complete = m_complete;
        c.prof_start(S.prof, threadIdx.x & 63, t_entry);
#ifdef LOB_PROF
        c.acc_[0] += t_r1 - t_entry;   // [20] launch, round 1, scalars to LDS
        c.acc_[11] += t_r2 - t_r1;     // [31] round 2 issued and in
        c.pt_ = t_r2;
#endif
        c.mark(30);  // action selection (uninstrumented: from the launch, rounds 1 and 2 included)
        const i64 ev0 = e.events;
        StepAgg g;
        step_prologue(c, e, action, g, cur);
        const int st = event_loop_fast(c, e, g, tcur, first);
        bool claim = false;
        u64 claim_k = 0;
        int claim_stamp = 0, claim_slot_prev = -1, claim_q0 = 0, claim_q1 = 0, claim_q2 = 0;
        d_events = e.events - ev0;
        if (st != 2) {
            step_epilogue_pre(c, e, g, wu, wd);
            const int cs = cur_slot;
            f32* v = S.vars + ((size_t)b * 3 + cs) * 16;
            f32* vf = S.vars + ((size_t)b * 3 + 2) * 16;
            int qg[3] = {0, 0, 0};
            for (int i = 0; i < P.V; i++) {
                v[i] = (f32)get_variable(c, e, P.vars[i], tcur);  // tcur = the entry of the last completed event (state_track)
                vf[i] = v[i];
                if (i < 3) qg[i] = tile_quant(v[i]);
            }
            if (P.memo) {
                const uint32_t s0 = (uint32_t)mk_hash3(qg[0], qg[1], qg[2]) & (uint32_t)(S.mk_slots - 1);
                claim_k = S.mk_hash[s0];
                claim_stamp = S.mk_stamp[s0];
                claim_slot_prev = mslot;
                claim_q0 = qg[0]; claim_q1 = qg[1]; claim_q2 = qg[2];
                claim = true;
            }
            hp->zero_mask = h0.zero_mask & ~(1 << cs);
            S.verdict[(size_t)b * LOB_VD_STRIDE + 67] = 0;  // a State changed: saved verdicts are void until learn saves new ones
            hp->reward = get_reward(c, e, &tcur.spread_mean);
            hp->stepped = 1;
            d_steps = 1;
        } else {
            hp->stepped = 0;
        }
        c.mark(28);  // state variables
        hp->done = e.done;
        hp->time_ms = e.time_ms;
        env_store(S, b, e);
        if (claim) {
            S.mk_slot_last[b] = claim_slot_prev;
            S.mk_slot[b] = mk_claim(S, claim_q0, claim_q1, claim_q2, step_id, par, claim_k, claim_stamp);
        }
        c.mark(29);  // agent scalars out, memo claim
        c.flush();
    }
    // one atomic per wave for the counters
    for (int off = 32; off > 0; off >>= 1) {
        d_steps += __shfl_down(d_steps, off);
        d_events += __shfl_down(d_events, off);
    }
    if ((threadIdx.x & 63) == 0 && (d_steps | d_events)) {
        cnt_add(S, 0, (u64)d_steps);
        cnt_add(S, 1, (u64)d_events);
        cnt_add(S, 3, (u64)d_steps);  // every stepped book gets one TD update
    }
}


// ---- the same step with a book's LEVELS ACROSS LANES: 16 lanes per book, four books per wave (RowLev, lob_env.h) -----------------
// For the batches that leave the chip idle under the lane-per-book kernel (4 096 books: 64 of its waves on 1 024 SIMDs): 16 x the
// waves, each on four books instead of 64 -- a wave waits for the longest of four steps, not of 64, and executes the branches of
// four books, not the union of 64.  What is per level is wave arithmetic:
//   * a record is one coalesced 16-byte load per lane (14 lanes x 16 B = the 224-byte row), turned into "lane l = level l" through
//     a 64-word LDS row; Book::volume / WalkTheBook by ballot and prefix sum (lob_env.h, RowLev);
//   * the hit list is a lane-parallel gather -- lane i fetches entries i, 16 + i, 32 + i and their weights: three loads instead of
//     35 -- followed by the ORDERED reduction Agent::getQ prescribes (entry j's product is handed round by lane j & 15);
//   * everything that is per book (orders, inventory, PnL, reward, state variables, the memo claim) is replicated in the 16 lanes
//     of the group -- the same instructions on the same values; the group's first lane stores.
// The arithmetic is env_step_kernel's, call for call (the row-consuming routines are templates over the row type).  One weight
// vector, two trade slots per record.
#define LOB_ENV16_BOOKS 4
template <bool INLINE_GENERAL>
__global__ void __launch_bounds__(64) env_step16_kernel(const DevParams* __restrict__ Pp, const DevState* __restrict__ Sp, int step_id, int par, EnvFuse F,
                                                        const uint32_t* __restrict__ rnd_g) {
    const DevParams& P = *Pp;
    const DevState& S = *Sp;
    __shared__ TickLds tick_lds;
    __shared__ __align__(16) uint32_t lds_row[LOB_ENV16_BOOKS][64];  // RowLev's staging rows
    __shared__ LearnLds1 lds_learn;
    const int grp = (int)threadIdx.x >> 4, li = (int)threadIdx.x & 15;
    const bool lead = li == 0;
    const int t = blockIdx.x * LOB_ENV16_BOOKS + grp;
    const bool valid = t < S.B;
    const int b = valid ? t : S.B - 1;
    const int B = S.B;

    // ---- round 1: everything addressed by the book id (replicated: the 16 lanes of a group ask for the same words) ----------
    if (threadIdx.x < LOB_MAX_BANDS) {
        const int i = threadIdx.x;
        tick_lds.lb[i] = P.band_lb[i]; tick_lds.tick[i] = P.band_tick[i]; tick_lds.cum[i] = P.band_cum[i]; tick_lds.pp[i] = P.band_pp[i]; tick_lds.pt[i] = P.band_pt[i];
    }
    if (threadIdx.x == 0) tick_lds.n = P.n_bands;
    EnvCtx c(P, S, b, &tick_lds);
    LHdr* hp = S.hdr + b;
    const LHdr h0 = *hp;
    const int k0 = S.k[b], rc0 = S.rec_cur[b];
    const int mslot = S.mk_slot[b];
    const bool dirty = S.hl_dirty[0] == F.sid_prev;
    // the hit list, lane-parallel: the count for everybody, entries li, 16 + li, 32 + li for this lane
    const u64* hl_p = S.hl_rec + (size_t)b * LOB_HL_REC;
    const u64 hl_n = hl_p[0];
    u64 ent[3];
#pragma unroll
    for (int j = 0; j < 3; j++) ent[j] = 16 * j + li < LOB_HL_MAX ? hl_p[1 + 16 * j + li] : 0ull;
    EnvR er;
    env_load(S, b, er);
    RMReg wu, wd;
    rm_load(S.pnl_ups, b, wu);
    rm_load(S.pnl_downs, b, wd);
    const int m_n_track = S.meta[b].n_track, m_complete = S.meta[b].complete;
    EnvR e = er;   // (in registers: this kernel holds no 56-word rows -- and 16 lanes writing one LDS word serialise)
    __syncthreads();

    // ---- who steps, and from which list ---------------------------------------------------------------------------------------
    const int cur_slot = h0.slot_cur ^ 1;  // swap(state, last_state)
    const bool alive = valid && !h0.done;
    const bool open = is_open(P, h0.time_ms);
    const int n_list = hl_n == LOB_HL_NONE ? -1 : (int)hl_n;
    bool ok = alive && open && !dirty && !((h0.zero_mask >> (cur_slot ^ 1)) & 1) && mslot >= 0 && n_list >= 0;

    // ---- round 2: everything addressed by what round 1 brought ----------------------------------------------------------------
    const int ms = mslot >= 0 ? mslot : 0;
    const MemoRec rec = *reinterpret_cast<const MemoRec*>(S.mk_rec + ((size_t)S.mk_slots + ms) * LOB_MK_REC);  // [1]: after the last update
    const int tiles_ok = S.mk_tiles_ok[ms];
    const uint32_t mk_bits = S.mk_marked[ms];
    f64 wv[3];
#pragma unroll
    for (int j = 0; j < 3; j++) wv[j] = (ok && 16 * j + li < n_list) ? S.theta[(uint32_t)ent[j]] : 0.0;
    const int kk = k0 > 0 ? k0 : 1;
    Track tprev;
    {
        const uint4* q = reinterpret_cast<const uint4*>(&c.track(kk - 1));
        const uint4 w4 = q[4], w5 = q[5];
        uint4* d = reinterpret_cast<uint4*>(&tprev);
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = make_uint4(0, 0, 0, 0);
        d[4] = w4; d[5] = w5;
    }
    Track tcur = track_load(&c.track(k0));
    RowLev cur, first;
    cur.stage = first.stage = lds_row[grp];
    cur.li = first.li = li;
    cur.pending = first.pending = false;
    row_full_load(c, rc0, cur);
    {
        const int last_row = S.n_events - 1;
        row_full_load(c, rc0 + 1 < last_row ? rc0 + 1 : last_row, first);
    }
    rm_prep(S.pnl_ups, B, b, wu);
    rm_prep(S.pnl_downs, B, b, wd);

    // ---- the action ------------------------------------------------------------------------------------------------------------
    int action = 0;
    bool go = false;
    if (valid) {
        if (h0.done) {
            if (lead) hp->stepped = 0;
        } else if (!open) {  // environment.isTerminal()
            if (lead) { hp->slot_cur = cur_slot; hp->done = 1; hp->stepped = 0; S.done[b] = 1; }
        } else {
            ok = ok && rec.ver == F.ver;
            if (!ok) {
                if (!INLINE_GENERAL && lead) {
                    const int pos = atomicAdd(&S.slow_n[F.lpar * 2 + 0], 1);
                    S.slow_list[pos] = b;
                }
            } else {
                f64 q[LOB_N_ACTIONS];
#pragma unroll
                for (int a = 0; a < LOB_N_ACTIONS; a++) q[a] = rec.s0[a];
                const f64 w1 = P.w1, w2 = P.w2;
                // the ordered reduction: addition j is entry j's, wherever it was fetched (n_list is the group's: its lanes loop together)
                for (int j = 0; j < n_list; j++) {
                    const int jj = j >> 4;
                    const u64 en = __shfl(jj == 0 ? ent[0] : jj == 1 ? ent[1] : ent[2], j & 15, 16);
                    const f64 v = __shfl(jj == 0 ? wv[0] : jj == 1 ? wv[1] : wv[2], j & 15, 16);
                    if (v != 0.0) {  // (+0.0 added to a sum that is never -0.0)
                        const int a_ = (int)(en >> 32) & 15;
                        const f64 x_ = ((en >> 36) & 1ull ? w2 : w1) * v;
#pragma unroll
                        for (int a = 0; a < LOB_N_ACTIONS; a++) q[a] = a_ == a ? q[a] + x_ : q[a];
                    }
                }
                if (li < LOB_N_ACTIONS) S.qs_last[(size_t)b * LOB_N_ACTIONS + li] = sel9(q, li);
                Rng g{P.seed, P.book_id_offset + (u64)b, h0.rng_ctr};
                action = policy_sample(P, q, false, g);
                if (lead) {
                    hp->slot_cur = cur_slot;
                    hp->action = action;
                    hp->stepped = 1;
                    hp->rng_ctr = g.ctr;
                    const uint32_t marked = tiles_ok ? mk_bits : 0x1ffu;
                    if (!((marked >> action) & 1u)) {
                        const int pos = atomicAdd(&S.mk_markcount[0], 1);
                        if (pos < S.mk_slots) S.mk_marklist[pos] = mslot * 16 + action;
                        else mark_generation(P, S, mslot, action);
                        atomicOr(&S.mk_marked[mslot], 1u << action);
                    }
                }
                go = true;
            }
        }
    }
    {   // (lob_get_path_stats [1]: books acted on from their hit list)
        const u64 act = __ballot(go && lead);
        if (threadIdx.x == 0 && act) cnt_add(S, 5, (unsigned long long)__builtin_popcountll(act));
    }
    if (INLINE_GENERAL) {
        // the books the replay could not serve: Agent::action in full, one book at a time, the whole wave on it (act_book)
        u64 todo = __ballot(lead && valid && alive && open && !ok);
        if (todo) {
            if (threadIdx.x == 0) cnt_add(S, 2, (u64)__builtin_popcountll(todo));  // (lob_get_path_stats [6])
            const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
            uint4* dst = reinterpret_cast<uint4*>(lds_learn.rnd);
            for (int i = threadIdx.x; i < 512; i += 64) dst[i] = src[i];
            if (threadIdx.x < 27) lds_learn.act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
            wave_lds_fence();
            while (todo) {
                const int src_lane = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int b_src = __builtin_amdgcn_readlane(b, src_lane);
                int act = -1;
                act_book<LOB_ALGO_SARSA, LearnLds1>(P, S, lds_learn, 0, (int)threadIdx.x, b_src, 0, par, &act);
                if (grp == (src_lane >> 4) && act >= 0) { action = act; go = true; }
            }
        }
    }

    // ---- performAction (replicated in the group; the row look-ups are the group's) ----------------------------------------------
    i64 d_steps = 0, d_events = 0;
    if (go) {
        c.pre_prev = &tprev;
        c.pre_n_track = m_n_track;
        c.pre_complete = m_complete;
        const i64 ev0 = e.events;
        StepAgg g;
        step_prologue(c, e, action, g, cur);
        const int st = event_loop_fast(c, e, g, tcur, first);
        bool claim = false;
        u64 claim_k = 0;
        int claim_stamp = 0, claim_q0 = 0, claim_q1 = 0, claim_q2 = 0;
        d_events = e.events - ev0;
        if (st != 2) {
            step_epilogue_pre(c, e, g, wu, wd);
            const int cs = cur_slot;
            f32* v = S.vars + ((size_t)b * 3 + cs) * 16;
            f32* vf = S.vars + ((size_t)b * 3 + 2) * 16;
            int qg[3] = {0, 0, 0};
            for (int i = 0; i < P.V; i++) {
                const f32 x = (f32)get_variable(c, e, P.vars[i], tcur);  // tcur = the entry of the last completed event (state_track)
                if (lead) { v[i] = x; vf[i] = x; }
                if (i < 3) qg[i] = tile_quant(x);
            }
            if (P.memo) {
                const uint32_t s0 = (uint32_t)mk_hash3(qg[0], qg[1], qg[2]) & (uint32_t)(S.mk_slots - 1);
                claim_k = S.mk_hash[s0];
                claim_stamp = S.mk_stamp[s0];
                claim_q0 = qg[0]; claim_q1 = qg[1]; claim_q2 = qg[2];
                claim = true;
            }
            const f64 rw = get_reward(c, e, &tcur.spread_mean);
            if (lead) {
                hp->zero_mask = h0.zero_mask & ~(1 << cs);
                S.verdict[(size_t)b * LOB_VD_STRIDE + 67] = 0;  // a State changed: saved verdicts are void until learn saves new ones
                hp->reward = rw;
                hp->stepped = 1;
            }
            d_steps = 1;
        } else {
            if (lead) hp->stepped = 0;
        }
        if (lead) {
            hp->done = e.done;
            hp->time_ms = e.time_ms;
            env_store(S, b, e);
            if (claim) {
                S.mk_slot_last[b] = mslot;
                S.mk_slot[b] = mk_claim(S, claim_q0, claim_q1, claim_q2, step_id, par, claim_k, claim_stamp);
            }
        }
    }
    // one atomic per wave for the counters (every book counted once: by its first lane)
    if (!lead) { d_steps = 0; d_events = 0; }
    for (int off = 32; off > 0; off >>= 1) {
        d_steps += __shfl_down(d_steps, off);
        d_events += __shfl_down(d_events, off);
    }
    if ((threadIdx.x & 63) == 0 && (d_steps | d_events)) {
        cnt_add(S, 0, (u64)d_steps);
        cnt_add(S, 1, (u64)d_events);
        cnt_add(S, 3, (u64)d_steps);  // every stepped book gets one TD update
    }
}

#endif
