// HIP kernels of the batched engine (gfx950).  One env-step of every book =
//   act_kernel    (wave / book)  swap states, Q(last_state,.), policy -> action
//   env_kernel    (lane / book)  performAction: book/order/event loop, reward, new state vars
//   learn_kernel  (wave / book)  traces, Q(state,.), TD error
//   update_kernel (wave / book)  theta[f] += alpha*delta/32 * e[f]   (f64 atomics)
// act/learn only READ theta, update only WRITES it, so within one step every
// book sees the same theta_t ("synchronous batch" semantic); with
// one book this is exactly the reference's Learner::_step order
// (src/experiment/serial.cpp:53-70).
#ifndef LOB_KERNELS_H
#define LOB_KERNELS_H

#include <hip/hip_runtime.h>

#include "lob_state.h"
#include "lob_stream.h"
#include "lob_launch.h"

// The engine library is built from four translation units (lob_launch.h).  The kernel TEMPLATES of these headers are only
// compiled where they are launched; the plain kernels are compiled in the unit that launches them: LOB_TU_SPLIT + one of
// LOB_TU_MAIN / LOB_TU_ENV / LOB_TU_PREPASS / LOB_TU_LEARN says which unit this is (neither: one unit holds everything, as the
// experiment builds of tools/ and the host-side tests of the device headers do).
#if !defined(LOB_TU_SPLIT)
#define LOB_IN_MAIN 1
#define LOB_IN_ENV 1
#define LOB_IN_PREPASS 1
#else
#if defined(LOB_TU_MAIN)
#define LOB_IN_MAIN 1
#else
#define LOB_IN_MAIN 0
#endif
#if defined(LOB_TU_ENV)
#define LOB_IN_ENV 1
#else
#define LOB_IN_ENV 0
#endif
#if defined(LOB_TU_PREPASS)
#define LOB_IN_PREPASS 1
#else
#define LOB_IN_PREPASS 0
#endif
#endif

// The venue's tick table for the lane-per-book kernels, in LDS: Market::ToTicks / ToPrice index it with a
// per-lane band, and a chain of such look-ups from global memory (six conversions per DoAction, four or
// five dependent loads each) was a sixth of env_kernel's time.
struct TickLds {
    int n;
    f64 lb[LOB_MAX_BANDS];
    f64 tick[LOB_MAX_BANDS];
    i64 cum[LOB_MAX_BANDS];
    f64 pp[LOB_MAX_BANDS];
    i32 pt[LOB_MAX_BANDS];
};
// all threads of the block call this (ends with a block barrier)
__device__ inline void stage_ticks(const DevParams& P, TickLds& t) {
    for (int i = threadIdx.x; i < LOB_MAX_BANDS; i += blockDim.x) {
        t.lb[i] = P.band_lb[i]; t.tick[i] = P.band_tick[i]; t.cum[i] = P.band_cum[i]; t.pp[i] = P.band_pp[i]; t.pt[i] = P.band_pt[i];
    }
    if (threadIdx.x == 0) t.n = P.n_bands;
    __syncthreads();
}

#define LOB_TRACK_MARGIN 4
// the lean event pass of env_kernel for streams with at most two trade slots per record (lob_env.h pass_fast);
// -DLOB_FAST_PASS=0 builds the general pass everywhere (A/B runs, tools/exp_variants.sh)
#ifndef LOB_FAST_PASS
#define LOB_FAST_PASS 1
#endif
#include "lob_env.h"
#include "lob_learn.h"

// The per-book environment registers of the lane-per-book kernels live in LDS
// (one padded slot per lane) instead of VGPRs: the fully inlined event loop
// otherwise spills ~1 KB per lane to scratch, and every spill reload is a
// vector-memory round trip on the critical path of a latency-bound kernel.
// Slot stride = sizeof(EnvR) rounded up to an odd multiple of 8 bytes, so a
// wave's 64-bit accesses to one field are at most 2-way bank conflicted.
struct EnvSlot {
    EnvR e;
    char pad[((sizeof(EnvR) / 8) % 2 == 0) ? 8 : 16];
};
static_assert(sizeof(EnvSlot) % 8 == 0 && (sizeof(EnvSlot) / 8) % 2 == 1, "odd 8-byte stride");
static_assert(sizeof(EnvSlot) * 256 <= 160 * 1024, "one 256-lane block per CU must fit in LDS");

// 4 books (waves) per block: 26 KB of LDS -> 6 waves per SIMD.  16 per block (75 KB, hash table
// staged once per 16 books, 8 waves per SIMD) measured the same for act and slower for learn
// (register pressure at the higher occupancy target): these kernels are issue-bound, not
// occupancy-bound.  Any value >= 8 builds (-DLOB_WAVES_PER_BLOCK=..); 4 needs the two-load staging.
#ifndef LOB_WAVES_PER_BLOCK
#define LOB_WAVES_PER_BLOCK 4
#endif
#define LOB_BLOCK (64 * LOB_WAVES_PER_BLOCK)
static_assert(LOB_BLOCK >= LOB_NZ_WORDS && 512 % LOB_BLOCK == 0 || LOB_BLOCK % 512 == 0, "block 0 clears one nz_new buffer; the block stages the 512 x 16 B hash table");

// ---------------------------------------------------------------------------
#if LOB_IN_PREPASS
__global__ void gen_events_kernel(lob_gen_params g, int D, int T, u64 first_book, int B, uint32_t* out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int Wd = drec_words(D, T);
    lob_gen_state s;
    lob_gen_init(g, s);
    uint32_t rec[64], drec[80];
    for (int e = 0; e < g.n_events; e++) {
        lob_gen_event(g, D, T, first_book + (u64)b, e, s, rec);
        drec_from_abi(rec, D, T, drec);  // straight into the device layout (lob_env.h)
        uint32_t* dst = out + ((size_t)b * g.n_events + e) * Wd;
        for (int i = 0; i < Wd; i += 4)
            *reinterpret_cast<uint4*>(dst + i) = make_uint4(drec[i], drec[i + 1], drec[i + 2], drec[i + 3]);
    }
}
#endif
// Uploaded streams: ABI records (lob_engine.h) -> device records, one thread per record.
#if LOB_IN_PREPASS
__global__ void repack_kernel(const uint32_t* __restrict__ src, int D, int T, size_t n_records, uint32_t* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_records) return;
    const int W = lob_rec_words(D, T), Wd = drec_words(D, T);
    uint32_t rec[64], drec[80];
    for (int k = 0; k < W; k++) rec[k] = src[i * W + k];
    drec_from_abi(rec, D, T, drec);
    for (int k = 0; k < Wd; k += 4)
        *reinterpret_cast<uint4*>(dst + i * Wd + k) = make_uint4(drec[k], drec[k + 1], drec[k + 2], drec[k + 3]);
}
#endif

// ---------------------------------------------------------------------------
// Base::Initialise + Intraday::Initialise (base.cpp:123-135, intraday.cpp:103-138)
// followed by the Runner prologue `last_state->newState(environment)`
// (serial.cpp:25).  Lane per book.  The market pre-pass (lob_env.h) walks the
// whole stream once and leaves the per-event Track; the agent side of
// Initialise is then: zero the books' agent state, jump to the end of the
// warm-up, place the (1,1) quotes.
// RB books per block (= per wave when RB <= 64): the kernel needs the whole register file of a lane,
// so at most two waves share a SIMD; with 32 books per wave those two hide each other's latency.
template <int RB, int TM>
// (the state BY VALUE here: read through its device-resident copy the kernel spills nothing to scratch -- 176 spilled scalar
// registers -> 10, 180 bytes of scratch per lane -> 0 -- and takes the same 16.5-16.7 ms: round 6)
#ifdef LOB_RESET_REGCAP  // probe only (tools/exp_background_prepass.py): 112 = 224 registers in all, a wave that fits beside env_step_kernel's
__attribute__((amdgpu_num_vgpr(LOB_RESET_REGCAP)))
#endif
__global__ void __launch_bounds__(RB) reset_kernel(const DevParams* __restrict__ Pp, DevState S) {
    const DevParams& P = *Pp;
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    EnvCtx c(P, S, b, &tick_lds);
    __shared__ EnvSlot lds_env[RB];
    EnvR& e = lds_env[threadIdx.x].e;
    env_load(S, b, e);  // position, pnl_step, quote levels etc. persist across episodes
    const i64 ev0 = e.events;
    persist_io(S, b, true);  // sums as of this episode's start
    BookMeta M;
    {
        // the whole stream when its track is resident, else the first ring-full (prepass_extend_kernel goes on from there)
        PrepState st;
        prepass_begin(c, st, M);
        prepass_run<TM>(c, st, M, S.track_mask == 0x7fffffff ? 0x7fffffff : S.track_len - LOB_TRACK_MARGIN, true);
        S.meta[b] = M;
        S.prep[b] = st;
    }
    e.done = 0;
    e.ask_quote = 0.0; e.bid_quote = 0.0;
    e.a_ntr = 0; e.a_on = 0; e.b_ntr = 0; e.b_on = 0;
    e.ep_reward = e.ep_pnl = e.ep_bandh = 0.0;
    e.total_ticks = e.market_buys = e.market_sells = 0;
    e.tick_ab = e.tick_pos = 0; e.tick_both = 0;
    e.ntr_snap = 0;
    if (M.init_ok) {
        const int k = M.k_warm;
        const Track t1 = c.track(k - 1);
        e.k = k;
        e.pf = t1.rec_last;   // Initialise's closing SkipUntil(market time), intraday.cpp:130 (== rec_first unless the event ran through invalid rows)
        e.rec_cur = t1.rec_last;
        e.mid = t1.mid;
        e.time_ms = t1.time_ms;
        if (k >= 2) { const Track t0 = c.track(k - 2); e.rec_last = t0.rec_last; e.mid_prev = t0.mid; }
        else { e.rec_last = M.rec_cur0; e.mid_prev = M.mid0; }
        e.events += (i64)(t1.rec_last + 1);  // records 0..rec_last were consumed by Initialise
        place_orders(c, e, 1, 1);
        // last_state->newState(env).  Slot 2 always mirrors the latest getState():
        // Backtester::_step extracts the state itself before acting (serial.cpp:124-137).
        LHdr& h = S.hdr[b];
        const int last = h.slot_cur ^ 1;
        f32* v = S.vars + ((size_t)b * 3 + last) * 16;
        f32* vf = S.vars + ((size_t)b * 3 + 2) * 16;
        const Track tk = state_track(c, e);
        for (int i = 0; i < P.V; i++) {
            v[i] = (f32)get_variable(c, e, P.vars[i], tk);
            vf[i] = v[i];
        }
        h.zero_mask &= ~(1 << last);
        S.verdict[(size_t)b * LOB_VD_STRIDE + 67] = 0;
    } else {
        // Initialise() == false: out of data before the windows filled
        e.k = M.n_track;
        e.pf = M.rec_cur0;
        e.rec_cur = M.ex_cur; e.rec_last = M.ex_last; e.time_ms = M.ex_time;
        e.mid = 0.0; e.mid_prev = 0.0;
        e.events += (i64)(S.n_events > 0 ? S.n_events - 1 : 0);
        e.done = 2;
    }
    S.hdr[b].stepped = 0;
    S.hdr[b].done = e.done;
    S.hdr[b].time_ms = e.time_ms;
    S.mk_slot[b] = -1;  // the memo table is emptied at every reset: the first act of the episode takes the general path
    S.mk_slot_last[b] = -1;
    env_store(S, b, e);
    cnt_add(S, 1, (u64)(e.events - ev0));  // warm-up events count as consumed
}

// reset_kernel<64, TM> with the pre-pass on two waves per 64 books (lob_env.h prepass_run2): wave 0 of the block is the books'
// row / trade / tick side and, after the loop, everything reset_kernel does with the result; wave 1 the windows' side.
// OPT-IN (LOB_PREPASS_ROLES=1), bit-exact, measured SLOWER than the one-wave kernel: 20.6 ms against 16.4 -- the roles alone take
// 14.2 and 9.2 ms and together almost their sum, although nothing they share (LDS, scalar unit, VALU) is near saturation by the
// counters of the one-wave kernel; what doubles is the number of waves fetching instructions from a 54 KB kernel body.
template <int TM>
__global__ void __launch_bounds__(128, 2) reset2_kernel(const DevParams* __restrict__ Pp, DevState S) {
    const DevParams& P = *Pp;
    __shared__ TickLds tick_lds;
    __shared__ PreXch xch[2];
    __shared__ EnvSlot lds_env[64];
    stage_ticks(P, tick_lds);
    const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 64 + lane;
    const bool on = t < S.B;
    const int b = on ? t : S.B - 1;
    EnvCtx c(P, S, b, &tick_lds);
    EnvR& e = lds_env[lane].e;
    i64 ev0 = 0;
    BookMeta M;
    PrepState st;
    memset(&st, 0, sizeof st);
    memset(&M, 0, sizeof M);
    if (role == 0) {
        if (on) env_load(S, b, e);  // position, pnl_step, quote levels etc. persist across episodes
        ev0 = e.events;
        if (on) prepass_begin_books(c, st, M);
        else M.complete = 1;
    } else if (on) {
        persist_io(S, b, true);  // sums as of this episode's start
        // ClearWindows (base.cpp:145-163): deques emptied, running sums kept (quirk Q7)
#define X(n) S.n.cnt[b] = 0;
        LOB_ROLLING_MEANS(X)
        LOB_ACCUMULATORS(X)
#undef X
        st.ewma_up = S.ewma_up[b]; st.ewma_down = S.ewma_down[b]; st.tp_val = S.tp_val[b];
        st.k = 0;
    }
    prepass_run2<TM>(c, st, M, S.track_mask == 0x7fffffff ? 0x7fffffff : S.track_len - LOB_TRACK_MARGIN, on, role, xch, lane);
    // the window role's last words (ewma, target price) belong in the resumable state the book role's lanes write
    f64* hand = reinterpret_cast<f64*>(&xch[0]);
    __syncthreads();
    if (role == 1) { hand[lane] = st.ewma_up; hand[64 + lane] = st.ewma_down; hand[128 + lane] = st.tp_val; }
    __syncthreads();   // (also: both halves of every track entry are written -- the same CU's L1 serves both waves)
    if (role != 0 || !on) return;
    st.ewma_up = hand[lane]; st.ewma_down = hand[64 + lane]; st.tp_val = hand[128 + lane];
    S.meta[b] = M;
    S.prep[b] = st;
    e.done = 0;
    e.ask_quote = 0.0; e.bid_quote = 0.0;
    e.a_ntr = 0; e.a_on = 0; e.b_ntr = 0; e.b_on = 0;
    e.ep_reward = e.ep_pnl = e.ep_bandh = 0.0;
    e.total_ticks = e.market_buys = e.market_sells = 0;
    e.tick_ab = e.tick_pos = 0; e.tick_both = 0;
    e.ntr_snap = 0;
    if (M.init_ok) {
        const int k = M.k_warm;
        const Track t1 = c.track(k - 1);
        e.k = k;
        e.pf = t1.rec_last;   // Initialise's closing SkipUntil(market time), intraday.cpp:130
        e.rec_cur = t1.rec_last;
        e.mid = t1.mid;
        e.time_ms = t1.time_ms;
        if (k >= 2) { const Track t0 = c.track(k - 2); e.rec_last = t0.rec_last; e.mid_prev = t0.mid; }
        else { e.rec_last = M.rec_cur0; e.mid_prev = M.mid0; }
        e.events += (i64)(t1.rec_last + 1);  // records 0..rec_last were consumed by Initialise
        place_orders(c, e, 1, 1);
        LHdr& h = S.hdr[b];
        const int last = h.slot_cur ^ 1;
        f32* v = S.vars + ((size_t)b * 3 + last) * 16;
        f32* vf = S.vars + ((size_t)b * 3 + 2) * 16;
        const Track tk = state_track(c, e);
        for (int i = 0; i < P.V; i++) {
            v[i] = (f32)get_variable(c, e, P.vars[i], tk);
            vf[i] = v[i];
        }
        h.zero_mask &= ~(1 << last);
        S.verdict[(size_t)b * LOB_VD_STRIDE + 67] = 0;
    } else {
        // Initialise() == false: out of data before the windows filled
        e.k = M.n_track;
        e.pf = M.rec_cur0;
        e.rec_cur = M.ex_cur; e.rec_last = M.ex_last; e.time_ms = M.ex_time;
        e.mid = 0.0; e.mid_prev = 0.0;
        e.events += (i64)(S.n_events > 0 ? S.n_events - 1 : 0);
        e.done = 2;
    }
    S.hdr[b].stepped = 0;
    S.hdr[b].done = e.done;
    S.hdr[b].time_ms = e.time_ms;
    S.mk_slot[b] = -1;  // the memo table is emptied at every reset: the first act of the episode takes the general path
    S.mk_slot_last[b] = -1;
    env_store(S, b, e);
    cnt_add(S, 1, (u64)(e.events - ev0));  // warm-up events count as consumed
}

// End of an episode: the pre-pass has walked the window arithmetic to the END of
// the stream, but an episode may stop earlier (market close).  Regenerate the
// window sums exactly as they stood after the `k` events the episode consumed
// (restore the episode-start sums, replay k events) -- they are what the next
// episode inherits (quirk Q7).
template <int TM>
__global__ void __launch_bounds__(256, 1) finalize_kernel(const DevParams* __restrict__ Pp, DevState S) {
    const DevParams& P = *Pp;
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    EnvCtx c(P, S, b, &tick_lds);
    persist_io(S, b, false);
    const int k = S.k[b];
    if (k > 0) {
        PrepState st;
        BookMeta M;
        prepass_begin(c, st, M);
        M.k_warm = S.meta[b].k_warm;   // where Initialise's warm-up ended (its closing SkipUntil drops trades there)
        prepass_run<TM>(c, st, M, k, false);
    }
}

// Long streams (a recorded day: tens of thousands of events per book, more than a resident track of
// 96 B per event and book leaves room for at 65 536 books): the track is a ring of the latest
// `track_len` entries per book, and every few steps this kernel lets every book's pre-pass run on from
// where it stopped until the ring is full again -- up to LOB_TRACK_MARGIN entries short of overwriting
// what the agent side may still look at (events k - 2 .. k).
template <int TM>
__global__ void __launch_bounds__(64) prepass_extend_kernel(const DevParams* __restrict__ Pp, DevState S) {
    const DevParams& P = *Pp;
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    BookMeta M = S.meta[b];
    if (M.complete) return;
    const int k_stop = S.k[b] + S.track_len - LOB_TRACK_MARGIN;
    PrepState st = S.prep[b];
    if (st.k >= k_stop) return;
    EnvCtx c(P, S, b, &tick_lds);
    prepass_run<TM>(c, st, M, k_stop, true);
    S.meta[b] = M;
    S.prep[b] = st;
}

// ... the same on two waves per 64 books
template <int TM>
__global__ void __launch_bounds__(128, 2) prepass_extend2_kernel(const DevParams* __restrict__ Pp, DevState S) {
    const DevParams& P = *Pp;
    __shared__ TickLds tick_lds;
    __shared__ PreXch xch[2];
    stage_ticks(P, tick_lds);
    const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 64 + lane;
    const bool in = t < S.B;
    const int b = in ? t : S.B - 1;
    BookMeta M = S.meta[b];
    PrepState st = S.prep[b];
    const int k_stop = S.k[b] + S.track_len - LOB_TRACK_MARGIN;
    const bool on = in && !M.complete && st.k < k_stop;
    EnvCtx c(P, S, b, &tick_lds);
    prepass_run2<TM>(c, st, M, k_stop, on, role, xch, lane);
    f64* hand = reinterpret_cast<f64*>(&xch[0]);
    __syncthreads();
    if (role == 1) { hand[lane] = st.ewma_up; hand[64 + lane] = st.ewma_down; hand[128 + lane] = st.tp_val; }
    __syncthreads();
    if (role != 0 || !on) return;
    st.ewma_up = hand[lane]; st.ewma_down = hand[64 + lane]; st.tp_val = hand[128 + lane];
    S.meta[b] = M;
    S.prep[b] = st;
}

// Evaluate getState() for every book (lob_get_state): lane per book.
#if LOB_IN_ENV
__global__ void __launch_bounds__(256) get_state_kernel(const DevParams* __restrict__ Pp, DevState S, f32* out /*[B][V]*/, f64* reward /*[B] or null*/) {
    const DevParams& P = *Pp;  // parameters read through the scalar cache, never copied to scratch
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    EnvCtx c(P, S, b, &tick_lds);
    EnvR e;
    env_load(S, b, e);
    if (out) {
        const Track tk = state_track(c, e);
        for (int i = 0; i < P.V; i++) out[(size_t)b * P.V + i] = (f32)get_variable(c, e, P.vars[i], tk);
    }
    if (reward) reward[b] = get_reward(c, e);
}
#endif

// performAction for every book that has an action pending (S.stepped).
// mode 0: learner step (new vars go to `state` = slot_cur); mode 1: host
// supplied actions (lob_step).
// LOB_ENV_BLOCK books per block = per wave.  64: 23.5 KB of LDS slots, small enough to share a CU with
// learner blocks.  The kernel is a serial chain per lane and a wave pays for the longest of its
// lanes' event loops, so when the batch cannot fill the chip anyway (<= 16 384 books: at most one
// wave per SIMD) 16 books per wave spread the work over 4x the waves: 0.139 -> see DESIGN.md at 4 096
// books; at 65 536 books the same choice is slower (0.31 vs 0.19 ms).
// TM = compile-time bound of the trade slots per record (lob_params.max_trades <= TM): the merged trade
// list of a pass lives in registers, and the engine's default of 2 slots should not carry arrays of 8.
// MODE 0: the books b0 .. b0 + nb - 1, their actions chosen by an act kernel (or given by the host).
// MODE 1: the same, the action chosen HERE from the book's hit list (act_light_book, lob_fast.h): the look-ups of the
//         action selection and of the step's first loads share their round trips, and the step is one launch shorter.  A
//         book without a valid list goes on the act work list and is skipped; after the general act kernel has served the
//         list, MODE 2 takes its books' steps.
// MODE 2: the books of the work list `list` (`*list_n` entries).
// (struct EnvFuse: lob_launch.h)
__device__ inline bool act_light_book(const DevParams& P, const DevState& S, int b, const LHdr& h, int lpar, u64 ver, bool dirty, int& action);
template <int LOB_ENV_BLOCK, int TM, int MODE = 0>
__global__ void __launch_bounds__(LOB_ENV_BLOCK) env_kernel(const DevParams* __restrict__ Pp, DevState S, const i32* host_actions, int count_updates, int b0, int nb,
                                                            int step_id, int par, EnvFuse F = EnvFuse()) {
    const DevParams& P = *Pp;  // parameters read through the scalar cache, never copied to scratch
    if (MODE == 2 && (int)(blockIdx.x * blockDim.x) >= *F.list_n) return;  // (the usual case: nothing on the list)
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    __shared__ EnvSlot lds_env[LOB_ENV_BLOCK];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_here = MODE == 2 ? *F.list_n : nb;
    const int b = MODE == 2 ? (t < n_here ? F.list[t] : 0) : b0 + t;
    i64 d_steps = 0, d_events = 0;
#ifdef LOB_PROF
    const long long t_entry = clock64();
#else
    const long long t_entry = 0;
#endif
    if (t < n_here) {
        const int k0 = S.k[b], rc0 = S.rec_cur[b];  // in flight together with the header
        LHdr& h = S.hdr[b];
        bool go;
        int action;
        if (MODE == 1) {
            const LHdr h0 = h;
            go = act_light_book(P, S, b, h0, F.lpar, F.ver, S.hl_dirty[0] == F.sid_prev, action);
        } else if (host_actions) {
            go = S.done[b] != 2;
            action = host_actions[b];
            h.action = action;
        } else {
            go = h.stepped != 0;
            action = h.action;
        }
        if (go) {
            EnvCtx c(P, S, b, &tick_lds);
            c.prof_start(S.prof, threadIdx.x & 63, t_entry);
            c.mark(30);  // action selection
            // the first event's track entry, the current snapshot and (fast pass) the first row the step will apply are
            // requested before the bulk of the state
            const TrackHead64 t0 = c.track_head64(k0);
            RowFull cur;
            row_full_load(c, rc0, cur);
            RowFull first;
            if (LOB_FAST_PASS && TM == 2) row_full_load(c, rc0 + 1 < S.n_events - 1 ? rc0 + 1 : S.n_events - 1, first);
            EnvR& e = lds_env[threadIdx.x].e;
            env_load(S, b, e);
            c.mark(20);  // agent scalars in
            i64 ev0 = e.events;
            bool ok;
            if (LOB_FAST_PASS && TM == 2) ok = perform_action_fast(c, e, action, t0, cur, first);
            else {
                TrackHead t32;
                t32.rec_first = t0.rec_first; t32.rec_last = t0.rec_last; t32.time_ms = t0.time_ms; t32.tick_ap0 = t0.tick_ap0;
                t32.tick_bp0 = t0.tick_bp0; t32.info = t0.info; t32.mid = t0.mid;
                ok = perform_action<TM>(c, e, action, t32, cur);
            }
            d_events = e.events - ev0;
            bool claim = false;
            u64 claim_k = 0;
            int claim_stamp = 0, claim_slot_prev = -1, claim_q0 = 0, claim_q1 = 0, claim_q2 = 0;
            if (ok) {
                const int cur = h.slot_cur;
                f32* v = S.vars + ((size_t)b * 3 + cur) * 16;
                f32* vf = S.vars + ((size_t)b * 3 + 2) * 16;
                const Track tk = state_track(c, e);
                int qg[3] = {0, 0, 0};
                for (int i = 0; i < P.V; i++) {
                    v[i] = (f32)get_variable(c, e, P.vars[i], tk);
                    vf[i] = v[i];
                    if (i < 3) qg[i] = tile_quant(v[i]);
                }
                // group-0 memo: the new state's triple gets (or finds) its slot and goes on this step's list -- at the very end
                // (below): its home slot's hash and stamp are requested here, the answer is looked at after everything else
                if (P.memo) {
                    const uint32_t s0 = (uint32_t)mk_hash3(qg[0], qg[1], qg[2]) & (uint32_t)(S.mk_slots - 1);
                    claim_k = S.mk_hash[s0];
                    claim_stamp = S.mk_stamp[s0];
                    claim_slot_prev = S.mk_slot[b];
                    claim_q0 = qg[0]; claim_q1 = qg[1]; claim_q2 = qg[2];
                    claim = true;
                }
                h.zero_mask &= ~(1 << cur);
                S.verdict[(size_t)b * LOB_VD_STRIDE + 67] = 0;  // a State changed: saved verdicts are void until learn saves new ones
                h.reward = get_reward(c, e);
                h.stepped = 1;
                d_steps = 1;
            } else {
                h.stepped = 0;
            }
            c.mark(28);  // state variables
            h.done = e.done;
            h.time_ms = e.time_ms;
            env_store(S, b, e);
            if (claim) {
                S.mk_slot_last[b] = claim_slot_prev;
                S.mk_slot[b] = mk_claim(S, claim_q0, claim_q1, claim_q2, step_id, par, claim_k, claim_stamp);
            }
            c.mark(29);  // agent scalars out, memo claim
            c.flush();
        } else if (MODE != 1) {  // (MODE 1: act_light_book has set the header of a book that does not step, or left it to the work list)
            h.stepped = 0;
        }
    }
    // one atomic per wave for the counters
    for (int off = LOB_ENV_BLOCK / 2; off > 0; off >>= 1) {
        d_steps += __shfl_down(d_steps, off);
        d_events += __shfl_down(d_events, off);
    }
    if ((threadIdx.x & 63) == 0 && (d_steps | d_events)) {
        cnt_add(S, 0, (u64)d_steps);
        cnt_add(S, 1, (u64)d_events);
        if (count_updates) cnt_add(S, 3, (u64)d_steps);  // every stepped book gets one TD update
    }
}

// The same step with the event loop COMPACTED across a 192-book block.  How many market events a step
// consumes is data dependent (it ends when the midprice has moved: 1.68 events on average, 5-6 for the
// unluckiest of 64 books), and in env_kernel a wave iterates until its slowest lane is done -- 70 % of
// its loop time is spent waiting for that lane.  Here a book's whole step state lives in its
// LDS slot (EnvSlot + the running sums of the loop), so ANY lane can carry it through its next event:
// after every pass the books that need another event are packed (wave ballot + prefix sum, one LDS
// atomic per wave for the block offset) into a work list, lanes 0..n-1 take entry n each, and waves
// beyond the list go idle at the block barrier instead of spinning.  Prologue and epilogue are
// lane = book as before.  Same arithmetic in the same order per book: results cannot differ.
#define LOB_ENVC_BLOCK 192
struct EnvCompactSlot {
    EnvSlot s;
    StepAgg agg;
    const uint32_t* rows;  // EnvCtx::rows of the book
};
static_assert(sizeof(EnvCompactSlot) * LOB_ENVC_BLOCK + 2 * 2 * LOB_ENVC_BLOCK + LOB_ENVC_BLOCK + 64 <= 80 * 1024, "two blocks per CU");

#if LOB_IN_ENV && defined(LOB_EXPERIMENTS)  // (opt-in LOB_ENV_LANES=256: measured no faster than env_kernel<64>, NOTES.md)
__global__ void __launch_bounds__(LOB_ENVC_BLOCK) env_compact_kernel(const DevParams* __restrict__ Pp, DevState S, int count_updates, int b0, int nb,
                                                                     int step_id, int par) {
    const DevParams& P = *Pp;  // parameters read through the scalar cache, never copied to scratch
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    __shared__ EnvCompactSlot slots[LOB_ENVC_BLOCK];
    __shared__ uint16_t work[2][LOB_ENVC_BLOCK];
    __shared__ int cnt[3];
    __shared__ uint8_t status[LOB_ENVC_BLOCK];  // 0: still in the loop, 1: step complete, 2: out of data, 3: no action pending
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * LOB_ENVC_BLOCK + threadIdx.x;
    const int b = b0 + t;
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    // ---- prologue: lane = book ----
    bool go = false;
    i64 ev0 = 0;
    if (t < nb) {
        LHdr& h = S.hdr[b];
        go = h.stepped != 0;
        if (go) {
            EnvCtx c(P, S, b, &tick_lds);
            EnvCompactSlot& sl = slots[threadIdx.x];
            EnvR& e = sl.s.e;
            env_load(S, b, e);
            ev0 = e.events;
            sl.rows = c.rows;
            RowFull cur;
            row_full_load(c, e.rec_cur, cur);
            step_prologue(c, e, h.action, sl.agg, cur);
        } else {
            h.stepped = 0;
        }
    }
    status[threadIdx.x] = go ? 0 : 3;
    {
        const u64 m = __ballot(go);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&cnt[0], __popcll(m));
        base = __shfl(base, 0);
        if (go) work[0][base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)threadIdx.x;
    }
    __syncthreads();
    // ---- the event loop, one event of every pending book per pass ----
    for (int it = 0;; it++) {
        const int n = cnt[it % 3];
        if (n == 0) break;
        if (threadIdx.x == 0) cnt[(it + 2) % 3] = 0;
        bool again = false;
        int slot = 0;
        if ((int)threadIdx.x < n) {
            slot = work[it & 1][threadIdx.x];
            EnvCompactSlot& sl = slots[slot];
            EnvCtx c(P, S, b0 + blockIdx.x * LOB_ENVC_BLOCK + slot, &tick_lds, sl.rows);
            const TrackHead t = c.track_head(sl.s.e.k);
            const int st = step_event<LOB_MAX_TRADES>(c, sl.s.e, sl.agg, t);
            again = st == 0;
            if (!again) status[slot] = (uint8_t)st;
        }
        const u64 m = __ballot(again);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&cnt[(it + 1) % 3], __popcll(m));
        base = __shfl(base, 0);
        if (again) work[(it + 1) & 1][base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)slot;
        __syncthreads();
    }
    // ---- epilogue: lane = book ----
    i64 d_steps = 0, d_events = 0;
    if (go) {
        LHdr& h = S.hdr[b];
        EnvCompactSlot& sl = slots[threadIdx.x];
        EnvR& e = sl.s.e;
        EnvCtx c(P, S, b, &tick_lds, sl.rows);
        const bool ok = status[threadIdx.x] == 1;
        if (ok) {
            step_epilogue(c, e, sl.agg);
            const int cur = h.slot_cur;
            f32* v = S.vars + ((size_t)b * 3 + cur) * 16;
            f32* vf = S.vars + ((size_t)b * 3 + 2) * 16;
            const Track tk = state_track(c, e);
            int qg[3] = {0, 0, 0};
            for (int i = 0; i < P.V; i++) {
                v[i] = (f32)get_variable(c, e, P.vars[i], tk);
                vf[i] = v[i];
                if (i < 3) qg[i] = tile_quant(v[i]);
            }
            if (P.memo) {
                const uint32_t s0 = (uint32_t)mk_hash3(qg[0], qg[1], qg[2]) & (uint32_t)(S.mk_slots - 1);
                const u64 pk = S.mk_hash[s0];
                const int ps = S.mk_stamp[s0];
                S.mk_slot_last[b] = S.mk_slot[b];
                S.mk_slot[b] = mk_claim(S, qg[0], qg[1], qg[2], step_id, par, pk, ps);
            }
            h.zero_mask &= ~(1 << cur);
            S.verdict[(size_t)b * LOB_VD_STRIDE + 67] = 0;
            h.reward = get_reward(c, e);
            h.stepped = 1;
            d_steps = 1;
        } else {
            h.stepped = 0;
        }
        d_events = e.events - ev0;
        h.done = e.done;
        h.time_ms = e.time_ms;
        env_store(S, b, e);
    }
    for (int off = 32; off > 0; off >>= 1) {
        d_steps += __shfl_down(d_steps, off);
        d_events += __shfl_down(d_events, off);
    }
    if (lane == 0 && (d_steps | d_events)) {
        cnt_add(S, 0, (u64)d_steps);
        cnt_add(S, 1, (u64)d_events);
        if (count_updates) cnt_add(S, 3, (u64)d_steps);  // every stepped book gets one TD update
    }
}
#endif

// Base::ClearInventory for every book (Runner::RunEpisode epilogue, serial.cpp:31)
#if LOB_IN_ENV
__global__ void __launch_bounds__(256) clear_inventory_kernel(const DevParams* __restrict__ Pp, DevState S) {
    const DevParams& P = *Pp;  // parameters read through the scalar cache, never copied to scratch
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    EnvCtx c(P, S, b, &tick_lds);
    EnvR e;
    env_load(S, b, e);
    clear_inventory(c, e);
    env_store(S, b, e);
}
#endif

// ---------------------------------------------------------------------------
// Shared LDS image of a learner block.
struct LearnLds {
    uint32_t rnd[2048];                                   // hash_UNH table, every entry reduced mod M
    uint32_t act_terms[32];                               // trailing-coordinate terms [group][action] mod M (27 used)
    f64 vals[LOB_WAVES_PER_BLOCK][LOB_HSLOTS];      // 4 KB per wave: one group's gathered theta (9 x 33 f64) / trace hash set (aliased)
    f32 vars[LOB_WAVES_PER_BLOCK][3][16];
    uint32_t newf[2][LOB_NZ_FILTER];                      // act only: filters of the map bits first set by the previous update (theta, theta_b)
};

// The same for a block of ONE wave (env_step_kernel's in-kernel general action selection)
struct LearnLds1 {
    uint32_t rnd[2048];
    uint32_t act_terms[32];
    f64 vals[1][LOB_HSLOTS];
    f32 vars[1][3][16];
    uint32_t newf[2][LOB_NZ_FILTER];  // (unused on the memo path: no verdict carry-over there)
};

// Stage the hash table (8 KB, two 16-byte loads per thread), the 27 action terms and, for act, the
// carry-over filters; ONE block barrier.
__device__ inline void learn_stage_table(const uint32_t* __restrict__ rnd_g, LearnLds& L, const i32* __restrict__ nz_buf = nullptr) {
    const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
    uint4* dst = reinterpret_cast<uint4*>(L.rnd);
    constexpr int PER = LOB_BLOCK >= 512 ? 1 : 512 / LOB_BLOCK;  // 512 x 16 B = the 8 KB table
    uint4 r[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) r[i] = (threadIdx.x + i * LOB_BLOCK < 512) ? src[threadIdx.x + i * LOB_BLOCK] : uint4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (threadIdx.x + i * LOB_BLOCK < 512) dst[threadIdx.x + i * LOB_BLOCK] = r[i];
    if (threadIdx.x < 27) L.act_terms[threadIdx.x] = rnd_g[2048 + threadIdx.x];
    if (nz_buf && threadIdx.x < 2 * LOB_NZ_FILTER) {  // [target][parity][LOB_NZ_WORDS]: the two targets are 2 * LOB_NZ_WORDS apart
        const int tg = threadIdx.x / LOB_NZ_FILTER, i = threadIdx.x % LOB_NZ_FILTER;
        L.newf[tg][i] = (uint32_t)nz_buf[tg * 2 * LOB_NZ_WORDS + LOB_NZ_FILTER + i];
    }
    __syncthreads();
}
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}
// This wave's three state-variable slots (the two rl::State objects + the latest getState()) into its
// own LDS row; wave-level hand-over only.
__device__ inline void learn_stage_vars(const f32* __restrict__ vars_b, f32* dst, int lane) {
    const f32 vv = lane < 48 ? vars_b[lane] : 0.0f;
    wave_lds_fence();  // the previous book's readers are done with the row
    if (lane < 48) dst[lane] = vv;
    wave_lds_fence();
}

// tag of a verdict row: theta epoch | State slot | valid (u16 words 64..67 of the row)
__device__ inline u64 vd_tag(uint32_t epoch, int slot) { return (u64)epoch | ((u64)(uint16_t)slot << 32) | (1ull << 48); }

// ---- the general learner path: every term of Q per book (q_values).  It serves private theta, double Q,
// two book groups and LOB_NO_MEMO=1 over the whole batch, and -- over a work list -- the few books the
// fast path (lob_fast.h) hands back (no valid memo record: first step of an episode, constructor-zero
// State, weights just loaded).
// mode 0: Learner::_step prologue (swap, terminal check, epsilon-greedy action)
// mode 1: Backtester::_step prologue (no swap, greedy action on the current state)
// ALGO is a compile-time parameter: the double-Q path needs a second weight vector and more
// registers; keeping it out of the SARSA / Q(lambda) instantiations keeps them small.
__device__ inline bool nzx_mark(const DevParams& P, const DevState& S, i32 f);
// LDS = LearnLds, or the one-wave image env_step_kernel keeps for the books its hit-list replay cannot serve; `out_action`
// (optional): the action chosen, or -1 if the book does not step.
template <int ALGO, class LDS = LearnLds>
__device__ __forceinline__ void act_book(const DevParams& P, const DevState& S, LDS& L, int w, int lane, int b, int mode, int par, int* out_action = nullptr) {
    if (out_action) *out_action = -1;
    const LHdr h = S.hdr[b];  // one scalar 64-byte load
    const i32* nz_new = S.nz_new + (par ^ 1) * LOB_NZ_WORDS;  // written by the previous step's update
    const uint16_t* vd = S.verdict + (size_t)b * LOB_VD_STRIDE;
    const int n_new = ALGO == LOB_ALGO_DOUBLE_Q ? max(nz_new[0], nz_new[2 * LOB_NZ_WORDS]) : nz_new[0];
    const uint32_t ep = (uint32_t)S.nz_epoch[0];
    const u64 tag = *(const u64*)(vd + 64);  // {epoch, slot, valid} in one load
    uint32_t my_vd = vd[lane];
    uint32_t my_vd_b = ALGO == LOB_ALGO_DOUBLE_Q ? S.verdict_b[(size_t)b * 64 + lane] : 0;
    learn_stage_vars(S.vars + (size_t)b * 48, &L.vars[w][0][0], lane);
    LHdr* hp = S.hdr + b;
    if (h.done) { if (lane == 0) hp->stepped = 0; return; }
    int cur = h.slot_cur;
    if (mode == 0) cur ^= 1;  // swap(state, last_state)
    if (!is_open(P, h.time_ms)) {  // environment.isTerminal()
        if (lane == 0) { hp->slot_cur = cur; hp->done = 1; hp->stepped = 0; S.done[b] = 1; }
        return;
    }
    // the State the action is computed from: last_state (learner) / latest getState() (backtester)
    const int src = mode == 0 ? (cur ^ 1) : 2;
    const bool zero = mode == 0 && ((h.zero_mask >> src) & 1);
    const f64* theta = S.theta + (P.theta_private ? (size_t)b * (size_t)P.M : 0);
    f64 qs[LOB_N_ACTIONS];
    const size_t nz_off = P.theta_private ? (size_t)b * LOB_NZ_NWORDS(P.M) : 0;
    const uint32_t* nz = S.theta_nz + nz_off;
    // learn(t) of this book evaluated the very same State: take over its "weight is zero"
    // verdicts if nothing but update(t) touched theta since (epoch) and that update set only a
    // few new bits (kept in a 4096-bit filter, staged in LDS).
    const bool reuse = mode == 0 && !zero && !P.theta_private && P.carry_verdicts && n_new <= LOB_NZ_NEW_MAX && tag == vd_tag(ep, src);
    if (reuse) q_values(P, theta, nz, L.vars[w][src], zero, L.rnd, L.act_terms, L.vals[w], lane, qs, 2, &my_vd, L.newf[0]);
    else q_values(P, theta, nz, L.vars[w][src], zero, L.rnd, L.act_terms, L.vals[w], lane, qs);
    if (lane < LOB_N_ACTIONS) S.qs_last[(size_t)b * LOB_N_ACTIONS + lane] = qs[lane];
    if (ALGO == LOB_ALGO_DOUBLE_Q) {
        // DoubleAgent::action (agent.cpp:196-204): qs[a] = (getQ + getQb) / 2.0f
        f64 qb[LOB_N_ACTIONS];
        if (reuse)
            q_values(P, S.theta_b, S.theta_b_nz, L.vars[w][src], zero, L.rnd, L.act_terms, L.vals[w], lane, qb, 2, &my_vd_b, L.newf[1]);
        else
            q_values(P, S.theta_b + (P.theta_private ? (size_t)b * (size_t)P.M : 0), S.theta_b_nz + nz_off, L.vars[w][src], zero,
                     L.rnd, L.act_terms, L.vals[w], lane, qb);
        if (lane < LOB_N_ACTIONS) S.qs_last_b[(size_t)b * LOB_N_ACTIONS + lane] = qb[lane];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) qs[a] = (qs[a] + qb[a]) / 2.0;
    }
    Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
    const int action = policy_sample(P, qs, mode == 1, g);
    if (out_action) *out_action = action;
    if (lane == 0) {
        hp->slot_cur = cur;
        hp->action = action;
        hp->stepped = 1;
        hp->rng_ctr = g.ctr;
    }
    // fast path (this kernel then serves the books handed back by its act kernels): the chosen action's 32 group-0
    // tiles are marked in the written-weights maps before the learn kernel looks, as those kernels do
    if (P.memo && mode == 0 && !zero) {
        const int ms = S.mk_slot[b];
        if (ms >= 0 && S.mk_tiles_ok[ms] && !((S.mk_marked[ms] >> action) & 1u)) {
            const int4 id = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)ms * 4);
            const f32* v = L.vars[w][src];
            if (id.x == tile_quant(v[0]) && id.y == tile_quant(v[1]) && id.z == tile_quant(v[2])) {
                if (lane < 32) nzx_mark(P, S, S.mk_tiles[((size_t)ms * LOB_N_ACTIONS + action) * 32 + lane]);
                if (lane == 0) atomicOr(&S.mk_marked[ms], 1u << action);
            }
        }
    }
}

// Whole batch (`list` null: wave t handles book b0 + t) or a work list of book ids (`list`, `*list_n`
// entries; a fixed grid strides over it).
template <int ALGO, bool LIST>
__global__ void __launch_bounds__(LOB_BLOCK) act_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g,
                                                        int mode, int b0, int nb, int par, const i32* __restrict__ list,
                                                        const i32* __restrict__ list_n) {
    LOB_PS_REFS
    __shared__ LearnLds L;
    if (LIST && *list_n == 0) return;  // nothing handed back by the fast path: the usual case
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!LIST && blockIdx.x == 0 && threadIdx.x == 0 && b0 == 0) {
        S.mk_count[par] = 0;  // this step's env_kernel starts a new list of memo slots
    }
    learn_stage_table(rnd_g, L, S.nz_new + (par ^ 1) * LOB_NZ_WORDS);
    const int t0 = __builtin_amdgcn_readfirstlane(blockIdx.x * LOB_WAVES_PER_BLOCK + w);
    if (!LIST) {
        if (t0 < nb) act_book<ALGO>(P, S, L, w, lane, b0 + t0, mode, par);
    } else {
        const int n = *list_n;
#pragma unroll 1
        for (int i = t0; i < n; i += gridDim.x * LOB_WAVES_PER_BLOCK)
            act_book<ALGO>(P, S, L, w, lane, __builtin_amdgcn_readfirstlane(list[i]), mode, par);
    }
}

// q[k] / f[k] for a run-time k without run-time indexing.  Each element passes through an opaque
// register copy first: left alone, the optimiser folds the select chain back into ONE load at a
// computed address, which pins the whole array in scratch memory.
__device__ inline f64 sel9(const f64* q, int k) {
    f64 r = q[0];
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int a = 1; a < LOB_N_ACTIONS; a++) {
        f64 x = q[a];
        asm volatile("" : "+v"(x));
        r = k == a ? x : r;
    }
    return r;
}
__device__ inline i32 sel5(const i32* f, int k) {
    i32 r = f[0];
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int a = 1; a < 5; a++) {
        i32 x = f[a];
        asm volatile("" : "+v"(x));
        r = k == a ? x : r;
    }
    return r;
}

// Weight f of the shared theta is (about to be) written for the first time: the exact map and its coarse image
// (lob_fast.h); monotone bits.  True if this call set the exact bit.
// (the map folded over the actions, lob_state.h theta_nzd: the 18 hash sums whose tilings contain weight f)
__device__ inline void nzd_mark(uint32_t* nzd, const uint32_t* terms18, uint32_t M, uint32_t f) {
    const size_t words = (size_t)M / 32 + 1;              // one map per tile group: [2][words]
    // 18 independent read-modify-writes nobody waits for (the caller has just flipped the exact bit: once per weight)
    uint32_t t[18];
#pragma unroll
    for (int i = 0; i < 18; i++) t[i] = terms18[i];       // < M
#pragma unroll
    for (int i = 0; i < 18; i++) {
        const uint32_t s = f >= t[i] ? f - t[i] : f + (M - t[i]);  // (s + t) mod M == f
        __hip_atomic_fetch_or(nzd + (i >= LOB_N_ACTIONS ? words : 0) + (s >> 5), 1u << (s & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ inline bool nzx_mark(const DevParams& P, const DevState& S, i32 f) {
    const uint32_t xb = 1u << ((uint32_t)f & 31);
    if (S.theta_nzx[(uint32_t)f >> 5] & xb) return false;
    const uint32_t old = atomicOr(&S.theta_nzx[(uint32_t)f >> 5], xb);
    if (!(old & xb) && S.theta_nzd) nzd_mark(S.theta_nzd, S.nzd_terms, (uint32_t)P.M, (uint32_t)f);
    const uint32_t c = (uint32_t)f >> P.cshift;
    const uint32_t cb = 1u << (c & 31);
    if (!(S.theta_nzc[c >> 5] & cb)) atomicOr(&S.theta_nzc[c >> 5], cb);
    return !(old & xb);
}
// ... after the learn kernels of step `sid` have looked at the maps: the hit lists they left are void (lob_state.h)
__device__ inline void nzx_mark_late(const DevParams& P, const DevState& S, i32 f, int sid) {
    if (nzx_mark(P, S, f)) S.hl_dirty[0] = sid;
}

// per-wave set of the current group-0 tiles (learn_traces): LOB_TSLOTS 32-bit slots in the wave's 4 KB of LDS
#define LOB_TSLOTS 1024
#define LOB_NOTILE 0xffffffffu /* tile indices are < M < 2^31 */
static_assert(LOB_TSLOTS * 4 == LOB_HSLOTS * 8, "the tile set aliases the wave's Q staging area");
__device__ inline unsigned trace_slot(uint32_t x) { return (x * 2654435761u) >> 22; }
__device__ inline unsigned trace_step(uint32_t x) { return ((x >> 9) ^ (x << 3) | 1u) & (LOB_TSLOTS - 1); }  // odd: visits every slot

// Agent::UpdateTraces (agent.cpp:86-101 -> traces.cpp:30-50; QLearn / DoubleQLearn: 272-280, 319-327)
// for one book, one wave: decay, clear / replace against the 288 group-0 tiles of last_state, new
// generation, and the slot claims of the combined update (issued here, resolved by the caller at the
// end of its kernel).  `tab`: this wave's 4 KB of LDS for the tile set (`init_tab`: it may hold anything
// on entry; otherwise it is all-NOTILE on entry and on exit); `vars_from`: the State acted on (LDS);
// shared by the general and the fast learner kernel.
template <int ALGO>
__device__ __forceinline__ void learn_traces(const DevParams& P, const DevState& S, int b, const LHdr& h, const uint32_t* rnd, const uint32_t* act_terms,
                                    u64* tab, bool init_tab, const f32* vars_from, bool zero_last, const f64* qs_last, Rng& g, int lane,
                                    CbPending& pend, Prof& pf, int dup_flag = 0, bool* dup_out = nullptr, int amax_given = -1, int late_sid = -1,
                                    int mslot_tag = -1) {
    LHdr* hp = S.hdr + b;
    const int action = h.action;
    // ---- group-0 tiles of last_state for all nine actions: lane -> (a = half + 2k, j) ----
    const int j = lane & 31, half = lane >> 5;
    i32 F[5];
    {
        const uint32_t base = zero_last ? 0 : tile_base_wave<0>((uint32_t)P.M, tile_quant(vars_from[lane & 15]), 3, j, rnd);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int a = half + 2 * k;
            F[k] = (a < LOB_N_ACTIONS && !zero_last) ? tile_index(base, act_terms[a < LOB_N_ACTIONS ? a : 0], (uint32_t)P.M) : 0;
        }
    }
    pf.mark(9);  // group-0 tiles of s

    // ---- Traces::decay (traces.cpp:30-38) ----
    int n_old = h.tr_n;
    const int head = h.tr_head;
    int kmax = P.trace_kmax;
    if (ALGO == LOB_ALGO_QLAMBDA || ALGO == LOB_ALGO_DOUBLE_Q) {
        // QLearn / DoubleQLearn::UpdateTraces (agent.cpp:272-280, 319-327); `amax_given`: the learn kernel ran first and made the draws
        const int amax = amax_given >= 0 ? amax_given : argmax_ties(qs_last, g);
        if (action != amax) kmax = 1;              // traces.decay(0.0)
    }
    if (n_old > kmax - 1) n_old = kmax - 1;        // generations whose eligibility fell below tolerance

    // ---- Traces::update (traces.cpp:40-50) ----
    // Set of the 288 current tiles: a live older entry that appears in it is either cleared (other
    // action) or re-set to 1 (chosen action): in both cases it leaves its old generation.
    // 1024 32-bit slots (load 0.28, double hashing: probe sequences stay short -- the kernel is bound by
    // the number of LDS instructions a wave issues, and a wave walks on until its unluckiest lane is
    // done), 32-bit compare-and-swap, tile index only.  The table is all-NOTILE between books: every key
    // clears its slot at the end (5 stores instead of re-initialising 4 KB).
    uint32_t* tt = reinterpret_cast<uint32_t*>(tab);
    // No older generation survives this step (Watkins cut, or the first step) and this triple's 288 tiles are
    // known to be distinct (`dup_flag` 1, learnt the first time the triple's set was built): nothing would
    // be looked up in the set -- the new generation is the chosen action's 32 tiles, all alive.
    const bool skip_set = !init_tab && n_old == 0 && dup_flag == 1;
    if (init_tab) {
        for (int i = lane; i < LOB_TSLOTS / 4; i += 64) reinterpret_cast<uint4*>(tt)[i] = make_uint4(LOB_NOTILE, LOB_NOTILE, LOB_NOTILE, LOB_NOTILE);
        wave_lds_fence();
    }
    unsigned sl[5];
    bool dupl = false;  // two of the 288 (action, tiling) pairs produce the same tile: only through a hash collision
    if (!skip_set) {
        bool todo[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sl[k] = trace_slot((uint32_t)F[k]);
            todo[k] = half + 2 * k < LOB_N_ACTIONS;
        }
        while (true) {
            uint32_t old[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                old[k] = 0;
                if (todo[k]) old[k] = atomicCAS(&tt[sl[k]], LOB_NOTILE, (uint32_t)F[k]);  // a round's CAS are in flight together
            }
            bool more = false;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                if (!todo[k]) continue;
                if (old[k] == LOB_NOTILE) todo[k] = false;                                // placed
                else if (old[k] == (uint32_t)F[k]) { todo[k] = false; dupl = true; }      // ANOTHER pair put this very tile there
                else { sl[k] = (sl[k] + trace_step((uint32_t)F[k])) & (LOB_TSLOTS - 1); more = true; }
            }
            if (__ballot(more) == 0) break;
        }
    }
    const bool any_dup = __ballot(dupl) != 0;
    if (dup_out) *dup_out = any_dup;
    wave_lds_fence();
    pf.mark(10);  // argmax(qs_last), LDS map: init + 288 inserts
    const int G = P.trace_gens;  // ring size, a power of two
    i32* tr_idx = S.tr_idx + (size_t)b * G * 32;
    uint32_t* tr_alive = S.tr_alive + (size_t)b * G;
    i32* tr_sig = S.tr_sig + (size_t)b * G * 4;
    const bool combine = P.combine != 0;
    // Old generations.  Lane l keeps the alive mask of the generation of (old) age l: loaded in one go,
    // rewritten in registers as the scan proceeds, stored and claimed (one lane per generation, all
    // CAS in flight together) after the scan.  The scan itself takes 8 generations per round: four
    // index loads per lane in flight, then the LDS probes.
    uint32_t my_mask = 0;   // lane = age
    int my_slot = 0;
    if (lane < n_old) {
        my_slot = (head - lane + G) & (G - 1);
        my_mask = tr_alive[my_slot];
    }
    for (int k0 = 0; k0 < n_old; k0 += 8) {
        i32 xs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = k0 + 2 * u + half;  // old age (before this step's decay) this half-wave scans
            const uint32_t am = __shfl(my_mask, k & 63);
            const int slot = (head - k + G) & (G - 1);
            xs[u] = (k < n_old && ((am >> j) & 1u)) ? tr_idx[slot * 32 + j] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (k0 + 2 * u >= n_old) break;  // wave-uniform
            bool alive = xs[u] >= 0;
            if (alive) {
                const uint32_t x = (uint32_t)xs[u];
                unsigned s = trace_slot(x);
                while (true) {
                    const uint32_t tv = tt[s];
                    if (tv == x) { alive = false; break; }
                    if (tv == LOB_NOTILE) break;
                    s = (s + trace_step(x)) & (LOB_TSLOTS - 1);
                }
            }
            const u64 m = __ballot(alive);
            if (lane == k0 + 2 * u) my_mask = (uint32_t)m;
            if (lane == k0 + 2 * u + 1) my_mask = (uint32_t)(m >> 32);
        }
    }
    pend.active = false;
    if (lane < n_old) {
        tr_alive[my_slot] = my_mask;
        if (combine && my_mask) {  // the generation keeps live tiles: make sure its (identity, mask) has a slot
            const int4 sg = *reinterpret_cast<const int4*>(tr_sig + my_slot * 4);
            cb_claim_issue(S, pend, sg.x, sg.y, sg.z, sg.w, my_mask, b * G + my_slot);
        }
    }
    pf.mark(11);  // old generations: scan, stores, claim issue
    // new generation: the chosen action's tiles, minus those a later action
    // clears again and minus duplicates inside the list (set() of a live tile
    // only overwrites its eligibility).
    {
        const i32 N = __shfl(sel5(F, action >> 1), (action & 1) * 32 + j);
        // dead <=> some (a', j') with the same tile outranks (action, j), rank(a, j) = 32 a + (31 - j): a later
        // action clears it again, or an earlier tiling of the chosen action already holds it.  Only
        // possible when two of the 288 pairs share a tile, which the inserts above have detected; then --
        // rarely: M is 20 M -- every pair is compared with every new tile.
        bool dead = false;
        if (any_dup) {
            const uint32_t myrank = (uint32_t)(action * 32 + 31 - j);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                for (int l = 0; l < 64; l++) {
                    const int a2 = (l >> 5) + 2 * k;
                    if (a2 >= LOB_N_ACTIONS) break;
                    const i32 t2 = __builtin_amdgcn_readlane(F[k], l);
                    dead |= t2 == N && (uint32_t)(a2 * 32 + 31 - (l & 31)) > myrank;
                }
            }
        }
        const int nh = (head + 1) & (G - 1);
        const u64 m = __ballot(!dead && half == 0);
        if (half == 0) tr_idx[nh * 32 + j] = N;
        // fast path: the tiles this generation will write are marked in the written-weights maps NOW, before the
        // learn kernel evaluates Q(s', .) -- the set of group-1/2 tiles of s' that fall on a marked weight is then
        // the same for that evaluation and for the next step's action selection (hit lists, lob_state.h)
        // (`late_sid`: this kernel runs AFTER the learn kernel of step late_sid -- the act kernel has marked the tiles, and a
        // bit that still flips here voids the hit lists)
        if (P.memo && half == 0 && !dead) {
            if (late_sid >= 0) nzx_mark_late(P, S, N, late_sid);
            else nzx_mark(P, S, N);
        }
        if (lane == 0) {
            tr_alive[nh] = (uint32_t)m;
            hp->tr_head = nh;
            hp->tr_n = n_old + 1;
        }
        if (lane == 63) {  // n_old <= 63: this lane has no old generation to claim for
            if (combine) {
                // identity of the generation: the quantised group-0 coordinates + the action fix all 32 tiles
                const int q0 = zero_last ? 0 : tile_quant(vars_from[0]), q1 = zero_last ? 0 : tile_quant(vars_from[1]),
                          q2 = zero_last ? 0 : tile_quant(vars_from[2]);
                const int code = action | (zero_last ? 256 : 0);
                *reinterpret_cast<int4*>(tr_sig + nh * 4) = make_int4(q0, q1, q2, code);
                if (P.sarsa_lanes) S.tr_mslot[(size_t)b * G + nh] = mslot_tag;  // (trace_sarsa_kernel: which memo slot holds the generation's tiles)
                if ((uint32_t)m) cb_claim_issue(S, pend, q0, q1, q2, code, (uint32_t)m, b * G + nh);
            }
        }
    }
    if (!init_tab && !skip_set) {  // leave the set empty for the next book
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (half + 2 * k < LOB_N_ACTIONS) tt[sl[k]] = LOB_NOTILE;
    }
    wave_lds_fence();
    pf.mark(12);  // new generation + claim issue
}

// UpdateWeights of SARSA / QLearn (agent.cpp:282-311) once Q(to_state, .) is known: the TD error and
// the header stores.
template <int ALGO>
__device__ __forceinline__ f64 learn_delta_single(const DevParams& P, LHdr* hp, const LHdr& h, const f64* qs_to, f64 q_sa, Rng& g, int lane,
                                                   const f64* rho = nullptr, f64* rl_t = nullptr) {
    static_assert(ALGO != LOB_ALGO_DOUBLE_Q, "two weight vectors: see learn_book");
    const f64 F_term = P.gamma * 0.0 - 0.0;  // potentials are identically 0 (base.cpp:239-242)
    f64 tq;  // the bootstrap value: maxQ(to_state) (QLearn, RLearn) / Q(to_state, this->action(to_state)) (SARSA -- quirk Q9 --, OnlineRLearn)
    if (ALGO == LOB_ALGO_QLAMBDA) {
        const int am2 = argmax_ties(qs_to, g);
        tq = sel9(qs_to, am2);
    } else {
        const int a2 = policy_sample(P, qs_to, false, g);
        tq = sel9(qs_to, a2);
    }
    f64 delta;
    if (rho) {  // RLearn / OnlineRLearn::UpdateWeights (agent.cpp:373-380, 398-405): delta = reward - rho + target - Q
        delta = h.reward - *rho + tq - q_sa;
        if (lane == 0) *rl_t = tq;
    } else {
        delta = h.reward + F_term + P.gamma * tq - q_sa;
    }
    if (lane == 0) {
        hp->td = delta;
        hp->upd = P.alpha * delta;
        hp->rng_ctr = g.ctr;
    }
    return delta;
}

// Agent::HandleTransition up to (not including) updateQ: UpdateTraces +
// the TD error of UpdateWeights (agent.cpp:86-115, 268-311).
template <int ALGO>
__device__ __forceinline__ void learn_book(const DevParams& P, const DevState& S, LearnLds& L, int w, int lane, int b) {
    const LHdr h = S.hdr[b];
    f64 qs_last[LOB_N_ACTIONS];
#pragma unroll
    for (int a = 0; a < LOB_N_ACTIONS; a++) qs_last[a] = S.qs_last[(size_t)b * LOB_N_ACTIONS + a];
    if (!h.stepped) return;
    learn_stage_vars(S.vars + (size_t)b * 48, &L.vars[w][0][0], lane);
    Prof pf;
    pf.start(S.prof, b, lane);
    LHdr* hp = S.hdr + b;
    const int cur = h.slot_cur, last = cur ^ 1;
    const bool zero_last = (h.zero_mask >> last) & 1;
    const f32* vars_to = L.vars[w][cur];
    const f32* vars_from = L.vars[w][last];
    const int action = h.action;
    Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
    CbPending pend;
    learn_traces<ALGO>(P, S, b, h, L.rnd, L.act_terms, reinterpret_cast<u64*>(L.vals[w]), true, vars_from, zero_last, qs_last, g, lane, pend, pf);

    // ---- UpdateWeights: TD error under theta_t ----
    const f64* theta = S.theta + (P.theta_private ? (size_t)b * (size_t)P.M : 0);
    const size_t nz_off = P.theta_private ? (size_t)b * LOB_NZ_NWORDS(P.M) : 0;
    const uint32_t* nz = S.theta_nz + nz_off;
    f64 qs_to[LOB_N_ACTIONS];
    {
        uint16_t* vd = S.verdict + (size_t)b * LOB_VD_STRIDE;
        const uint32_t ep = (uint32_t)S.nz_epoch[0];
        uint32_t my_vd = 0;
        q_values(P, theta, nz, vars_to, false, L.rnd, L.act_terms, L.vals[w], lane, qs_to, 1, &my_vd);
        vd[lane] = (uint16_t)my_vd;
        if (lane == 0) *(u64*)(vd + 64) = vd_tag(ep, cur);
    }
    if (ALGO == LOB_ALGO_DOUBLE_Q) {
        // DoubleQLearn::UpdateWeights (agent.cpp:329-353)
        const f64 reward = h.reward;
        const f64 F_term = P.gamma * 0.0 - 0.0;  // potentials are identically 0 (base.cpp:239-242)
        f64 delta;
        int target = 1;
        f64 qb_to[LOB_N_ACTIONS];
        uint32_t my_vd_b = 0;
        q_values(P, S.theta_b + (P.theta_private ? (size_t)b * (size_t)P.M : 0), S.theta_b_nz + nz_off, vars_to, false, L.rnd,
                 L.act_terms, L.vals[w], lane, qb_to, 1, &my_vd_b);
        S.verdict_b[(size_t)b * 64 + lane] = (uint16_t)my_vd_b;
        // the coin: unif_dist(gen) > 0.5 on the agent's own std::mt19937_64
        u64* mt = S.mt_state + (size_t)b * LOB_MT_N;
        int mi = S.mt_idx[b];
        if (mi >= LOB_MT_N) { mt64_twist_wave(mt, reinterpret_cast<u64*>(L.vals[w]), lane); mi = 0; }
        const f64 coin = mt64_canonical(mt64_temper(mt[mi]));
        if (lane == 0) S.mt_idx[b] = mi + 1;
        // (DoubleRLearn::UpdateWeights, agent.cpp:432-451: the same two branches with delta = reward - rho + mQ - Q)
        const f64 rho = P.r_learn ? S.rho[P.theta_private ? b : 0] : 0.0;
        if (coin > 0.5) {  // UPDATE(A)
            const int am = argmax_ties(qs_to, g);
            delta = P.r_learn ? reward - rho + sel9(qb_to, am) - sel9(qs_last, action)
                              : reward + F_term + P.gamma * sel9(qb_to, am) - sel9(qs_last, action);
        } else {           // UPDATE(B)
            f64 qb_last[LOB_N_ACTIONS];
#pragma unroll
            for (int a = 0; a < LOB_N_ACTIONS; a++) qb_last[a] = S.qs_last_b[(size_t)b * LOB_N_ACTIONS + a];
            const int am = argmax_ties(qb_to, g);
            delta = P.r_learn ? reward - rho + sel9(qs_to, am) - sel9(qb_last, action)
                              : reward + F_term + P.gamma * sel9(qs_to, am) - sel9(qb_last, action);
            target = 2;
        }
        if (lane == 0 && target == 2) hp->stepped = 2;  // update_kernel scatters into theta_b
        if (lane == 0) {
            hp->td = delta;
            hp->upd = P.alpha * delta;
            hp->rng_ctr = g.ctr;
        }
    } else {
        const bool r_learn = P.r_learn != 0;  // RLearn / OnlineRLearn: the Q(lambda) / SARSA instantiation with the average-reward TD error
        learn_delta_single<ALGO == LOB_ALGO_DOUBLE_Q ? LOB_ALGO_SARSA : ALGO>(P, hp, h, qs_to, sel9(qs_last, action), g, lane,
                                                                              r_learn ? S.rho + (P.theta_private ? b : 0) : nullptr, S.rl_t + b);
    }
    pf.mark(17);  // argmax / delta / header stores
    cb_claim_finish(S, pend);  // the CAS was issued before Q(s', .): its answer has long arrived
    pf.mark(18);  // claim finish
}

// The second half of learn_book alone, for the books the fast path's learn_q kernel hands back: its trace
// kernel has already run UpdateTraces for them, left Q(s, a) in LHdr::td and the RNG counter after the
// trace step's draws in LHdr::rng_ctr.
// DoubleQLearn::UpdateWeights (agent.cpp:329-353) once Q(to_state, .) is known under both vectors: the coin of the agent's own
// std::mt19937_64, the TD error against the other vector's value at this vector's argmax, the header stores.  `q_sa` = Q_a(from, a).
// WAVE: one wave per book (the 312-word block is regenerated by the wave through `lds`); else one LANE per book (every lane its
// own book: the block is regenerated in place, once per 312 steps).
__device__ inline void mt64_twist_lane(u64* x) {
    for (int i = 0; i < LOB_MT_M; i++) x[i] = mt64_mix(x[i], x[i + 1], x[i + LOB_MT_M]);
    for (int i = LOB_MT_M; i < LOB_MT_N - 1; i++) x[i] = mt64_mix(x[i], x[i + 1], x[i - LOB_MT_M]);
    x[LOB_MT_N - 1] = mt64_mix(x[LOB_MT_N - 1], x[0], x[LOB_MT_M - 1]);
}
template <bool WAVE>
__device__ __forceinline__ f64 learn_delta_double(const DevParams& P, const DevState& S, LHdr* hp, const LHdr& h, int b, const f64* qs_to, const f64* qb_to,
                                                  f64 q_sa, Rng& g, int lane, u64* lds, int* vector_out = nullptr) {
    const f64 reward = h.reward;
    const f64 F_term = P.gamma * 0.0 - 0.0;  // potentials are identically 0 (base.cpp:239-242)
    const int action = h.action;
    u64* mt = S.mt_state + (size_t)b * LOB_MT_N;
    int mi = S.mt_idx[b];
    if (mi >= LOB_MT_N) {
        if (WAVE) mt64_twist_wave(mt, lds, lane);
        else mt64_twist_lane(mt);
        mi = 0;
    }
    const f64 coin = mt64_canonical(mt64_temper(mt[mi]));
    if (lane == 0) S.mt_idx[b] = mi + 1;
    f64 delta;
    int target = 1;
    if (coin > 0.5) {  // UPDATE(A)
        const int am = argmax_ties(qs_to, g);
        delta = reward + F_term + P.gamma * sel9(qb_to, am) - q_sa;
    } else {           // UPDATE(B)
        f64 qb_last[LOB_N_ACTIONS];
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) qb_last[a] = S.qs_last_b[(size_t)b * LOB_N_ACTIONS + a];
        const int am = argmax_ties(qb_to, g);
        delta = reward + F_term + P.gamma * sel9(qs_to, am) - sel9(qb_last, action);
        target = 2;
    }
    if (lane == 0) {
        if (target == 2) hp->stepped = 2;  // the update scatters into theta_b
        hp->td = delta;
        hp->upd = P.alpha * delta;
        hp->rng_ctr = g.ctr;
    }
    if (vector_out) *vector_out = target - 1;  // 0: theta, 1: theta_b
    return delta;
}
template <int ALGO>
__device__ __forceinline__ void learn_q_book(const DevParams& P, const DevState& S, LearnLds& L, int w, int lane, int b) {
    const LHdr h = S.hdr[b];
    if (!h.stepped) return;
    learn_stage_vars(S.vars + (size_t)b * 48, &L.vars[w][0][0], lane);
    LHdr* hp = S.hdr + b;
    const int cur = h.slot_cur;
    Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
    f64 qs_to[LOB_N_ACTIONS];
    uint16_t* vd = S.verdict + (size_t)b * LOB_VD_STRIDE;
    const uint32_t ep = (uint32_t)S.nz_epoch[0];
    uint32_t my_vd = 0;
    q_values(P, S.theta, S.theta_nz, L.vars[w][cur], false, L.rnd, L.act_terms, L.vals[w], lane, qs_to, 1, &my_vd);
    vd[lane] = (uint16_t)my_vd;
    if (lane == 0) *(u64*)(vd + 64) = vd_tag(ep, cur);
    if (ALGO == LOB_ALGO_DOUBLE_Q) {
        f64 qb_to[LOB_N_ACTIONS];
        uint32_t my_vd_b = 0;
        q_values(P, S.theta_b, S.theta_b_nz, L.vars[w][cur], false, L.rnd, L.act_terms, L.vals[w], lane, qb_to, 1, &my_vd_b);
        S.verdict_b[(size_t)b * 64 + lane] = (uint16_t)my_vd_b;
        learn_delta_double<true>(P, S, hp, h, b, qs_to, qb_to, h.td, g, lane, reinterpret_cast<u64*>(L.vals[w]));
    } else {
        learn_delta_single<ALGO == LOB_ALGO_DOUBLE_Q ? LOB_ALGO_SARSA : ALGO>(P, hp, h, qs_to, LOB_QSA(P, S, b, h, ALGO), g, lane);
    }
}
template <int ALGO>
// `hint` (host-mapped memory, or null): the list's length tagged with this launch's serial number, for the host to read
// LOB_HINT_LAG steps later (lob_engine.hip: which act / accumulate path a step takes, whether this launch runs beside the trace kernels).
__global__ void __launch_bounds__(LOB_BLOCK) learn_q_rest_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g,
                                                                 const i32* __restrict__ list, const i32* __restrict__ list_n, u64* hint, uint32_t hint_tag) {
    LOB_PS_REFS
    __shared__ LearnLds L;
    if (hint && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(hint, ((u64)hint_tag << 32) | (u64)(uint32_t)*list_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (*list_n == 0) return;  // nothing handed back: the usual case
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    learn_stage_table(rnd_g, L);
    const int n = *list_n;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt_add(S, 4, (u64)n);  // (lob_get_path_stats [7])
#pragma unroll 1
    for (int i = __builtin_amdgcn_readfirstlane(blockIdx.x * LOB_WAVES_PER_BLOCK + w); i < n; i += gridDim.x * LOB_WAVES_PER_BLOCK)
        learn_q_book<ALGO>(P, S, L, w, lane, __builtin_amdgcn_readfirstlane(list[i]));
}

template <int ALGO, bool LIST>
__global__ void __launch_bounds__(LOB_BLOCK) learn_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g,
                                                          int b0, int nb, int par, const i32* __restrict__ list, const i32* __restrict__ list_n) {
    LOB_PS_REFS
    __shared__ LearnLds L;
    if (LIST && *list_n == 0) return;
    // this step's update appends to nz_new[par]; the list act reads is nz_new[par ^ 1]
    if (!LIST && blockIdx.x == 0 && threadIdx.x < LOB_NZ_WORDS) {
        S.nz_new[par * LOB_NZ_WORDS + threadIdx.x] = 0;
        S.nz_new[(2 + par) * LOB_NZ_WORDS + threadIdx.x] = 0;  // theta_b's (double Q)
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    learn_stage_table(rnd_g, L);
    const int t0 = __builtin_amdgcn_readfirstlane(blockIdx.x * LOB_WAVES_PER_BLOCK + w);
    if (!LIST) {
        if (t0 < nb) learn_book<ALGO>(P, S, L, w, lane, b0 + t0);
    } else {
        const int n = *list_n;
#pragma unroll 1
        for (int i = t0; i < n; i += gridDim.x * LOB_WAVES_PER_BLOCK)
            learn_book<ALGO>(P, S, L, w, lane, __builtin_amdgcn_readfirstlane(list[i]));
    }
}

// Agent::updateQ (agent.cpp:137-142): theta[f] += (alpha*delta / N_TILINGS) * e[f]
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_BLOCK) update_kernel(LOB_PS_ARGS, int par, int sid) {
    LOB_PS_REFS
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * LOB_WAVES_PER_BLOCK + w);
    if (b >= S.B) return;
    const LHdr h = S.hdr[b];
    if (!h.stepped) return;
    const int n = h.tr_n, head = h.tr_head;
    const f64 scaled = h.upd / (f64)LOB_N_TILINGS;
    f64* theta = (h.stepped == 2 ? S.theta_b : S.theta) + (P.theta_private ? (size_t)b * (size_t)P.M : 0);
    uint32_t* nz = (h.stepped == 2 ? S.theta_b_nz : S.theta_nz) + (P.theta_private ? (size_t)b * LOB_NZ_NWORDS(P.M) : 0);
    const int G = P.trace_gens;
    const i32* tr_idx = S.tr_idx + (size_t)b * G * 32;
    const uint32_t* tr_alive = S.tr_alive + (size_t)b * G;
    const int j = lane & 31, half = lane >> 5;
    // 32 generations per round (one round unless gamma*lambda > 0.86), three phases per round, each
    // with all of its loads in flight at once: alive masks -> trace indices -> adds + map words.
    for (int c0 = 0; c0 < n; c0 += 32) {
        const int ka = c0 + j;  // lane j (and j + 32) holds the mask of the generation of age c0 + j
        const uint32_t my_alive = ka < n ? tr_alive[(head - ka + G) & (G - 1)] : 0u;
        constexpr int NIT = 16;
        i32 f[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int k = c0 + 2 * it + half;  // age
            const int slot = (head - k + G) & (G - 1);
            const uint32_t alive = __shfl(my_alive, 2 * it + half);
            f[it] = (k < n && ((alive >> j) & 1u)) ? tr_idx[slot * 32 + j] : -1;
        }
        uint32_t word[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            word[it] = 0xffffffffu;
            if (f[it] >= 0) {
                const f64 val = scaled * (f64)P.trace_pow[c0 + 2 * it + half];
                __hip_atomic_fetch_add(&theta[f[it]], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                word[it] = nz[LOB_NZ_WORD(f[it])];
                if (P.memo) nzx_mark_late(P, S, f[it], sid);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            if (f[it] < 0) continue;
            const uint32_t bit = LOB_NZ_BIT(f[it]);
            if (!(word[it] & bit)) {  // monotone: set once, then a plain L2 hit
                const uint32_t old = atomicOr(&nz[LOB_NZ_WORD(f[it])], bit);
                if (!(old & bit) && P.carry_verdicts) {  // this lane flipped it: tell the next act_kernel
                    i32* nz_new = S.nz_new + ((h.stepped == 2 ? 2 : 0) + par) * LOB_NZ_WORDS;
                    atomicAdd(&nz_new[0], 1);
                    atomicOr((uint32_t*)&nz_new[LOB_NZ_FILTER + (LOB_NZ_WORD(f[it]) & (LOB_NZ_FILTER - 1))], bit);  // keyed like the map
                }
            }
        }
    }
}
#endif

// RLearn / OnlineRLearn::UpdateWeights after updateQ (agent.cpp:382-385, 407-410):
//     nQ = Q + update;  if (nQ - maxQ(from_state) < 1e-7) rho += beta * (reward - rho + target - nQ)
// with maxQ(from_state) under the weights the update has just written (one more Q evaluation of last_state; its argmax
// draws are the last of the book's step).  Every book reads rho_t; the increments are summed (rho_inc) and folded in by
// rho_fold_kernel -- the batch semantic of theta.  One wave per book, as learn_kernel.
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_BLOCK) rho_kernel(DevParams P, DevState S, const uint32_t* __restrict__ rnd_g) {
    __shared__ LearnLds L;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    learn_stage_table(rnd_g, L);
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * LOB_WAVES_PER_BLOCK + w);
    if (b >= S.B) return;
    const LHdr h = S.hdr[b];
    if (!h.stepped) return;
    learn_stage_vars(S.vars + (size_t)b * 48, &L.vars[w][0][0], lane);
    const int last = h.slot_cur ^ 1;
    const bool zero_last = (h.zero_mask >> last) & 1;
    const f64* theta = S.theta + (P.theta_private ? (size_t)b * (size_t)P.M : 0);
    const uint32_t* nz = S.theta_nz + (P.theta_private ? (size_t)b * LOB_NZ_NWORDS(P.M) : 0);
    f64 qs[LOB_N_ACTIONS];
    q_values(P, theta, nz, L.vars[w][last], zero_last, L.rnd, L.act_terms, L.vals[w], lane, qs);
    const int ri = P.theta_private ? b : 0;
    const f64 rho = S.rho[ri];
    if (P.algo == LOB_ALGO_DOUBLE_Q) {
        // DoubleRLearn (agent.cpp:453-464): mQ = max over the actions of (getQ + getQb) / 2.0 of from_state (first maximum: no
        // draws), Q = the vector's that was updated, and the increment uses THIS mQ, not the bootstrap value
        f64 qb[LOB_N_ACTIONS];
        q_values(P, S.theta_b + (P.theta_private ? (size_t)b * (size_t)P.M : 0), S.theta_b_nz + (P.theta_private ? (size_t)b * LOB_NZ_NWORDS(P.M) : 0),
                 L.vars[w][last], zero_last, L.rnd, L.act_terms, L.vals[w], lane, qb);
        f64 mQ = -1.7976931348623157e308;
#pragma unroll
        for (int a = 0; a < LOB_N_ACTIONS; a++) {
            const f64 val = (qs[a] + qb[a]) / 2.0;
            if (val > mQ) mQ = val;
        }
        const f64 Q = (h.stepped == 2 ? S.qs_last_b : S.qs_last)[(size_t)b * LOB_N_ACTIONS + h.action];
        const f64 nQ = Q + h.upd;
        if (lane == 0 && nQ - mQ < 1e-7) {
            __hip_atomic_fetch_add(&S.rho_inc[ri], P.beta * (h.reward - rho + mQ - nQ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&S.rho_cnt[ri], 1);
        }
        return;
    }
    Rng g{P.seed, P.book_id_offset + (u64)b, h.rng_ctr};
    const int am = argmax_ties(qs, g);
    const f64 mq_from = sel9(qs, am);
    const f64 nQ = S.qs_last[(size_t)b * LOB_N_ACTIONS + h.action] + h.upd;
    if (lane == 0) {
        S.hdr[b].rng_ctr = g.ctr;
        if (nQ - mq_from < 1e-7) {
            __hip_atomic_fetch_add(&S.rho_inc[ri], P.beta * (h.reward - rho + S.rl_t[b] - nQ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&S.rho_cnt[ri], 1);
        }
    }
}
#endif
// rho_{t+1} = rho_t + the MEAN of the step's increments: the reference's shared Agent applies them one after the other
// (each relative to the rho the one before left: a contraction), which a sum of n increments all relative to rho_t is
// not -- beta * n > 2 diverges.  One contributor (one book, or private weights): the increment itself, bit for bit.
#if LOB_IN_MAIN
__global__ void rho_fold_kernel(DevState S, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = S.rho_cnt[i];
    if (c > 0) S.rho[i] += S.rho_inc[i] / (f64)c;
    S.rho_inc[i] = 0.0;
    S.rho_cnt[i] = 0;
}
#endif
// Multi-GPU exchange of the shared rho (two extra slots behind the weights' delta: [rho - rho_sync, 1.0]; the all-reduce
// sums both, so the second is the number of ranks and rho <- rho_sync + mean of the ranks' changes)
#if LOB_IN_MAIN
__global__ void rho_delta_begin_kernel(const f64* rho, const f64* sync_slot, f64* delta_slots) {
    delta_slots[0] = rho[0] - sync_slot[0];
    delta_slots[1] = 1.0;
}
#endif
#if LOB_IN_MAIN
__global__ void rho_delta_apply_kernel(f64* rho, f64* sync_slot, const f64* delta_slots) {
    const f64 n = delta_slots[1];
    const f64 r = sync_slot[0] + (n > 0.0 ? delta_slots[0] / n : 0.0);
    rho[0] = r;
    sync_slot[0] = r;
}
#endif

// ---- combined update (shared theta) --------------------------------------------------------------
// accumulate_kernel: ONE LANE per trace generation, `1 << lpb_shift` lanes per book (a book with more
// generations than that walks them in rounds).  Each live generation finds the slot learn_kernel claimed
// for its (identity, alive mask) and adds its update alpha*delta/32 * e there: one atomic per generation
// instead of one per trace (32x fewer, and the slot array is small).  A generation without a slot (table
// crowded, or a 64-bit hash shared by two identities) is applied directly, tile by tile, like
// update_kernel does.  Watkins's Q(lambda) leaves 1-2 live generations per book at exploration rates
// near 1, so a whole wave per book is a chain of four dependent look-ups run 65 536 times for two lanes of
// work; 8 books per wave run the same chain 8 192 times.
// which copy of the slots' sums this wave adds to (one per XCD, DevState::cb_reps)
__device__ inline int acc_copy(const DevState& S, int wave) {
    int xcd = 0;
    if (S.cb_reps > 1) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcd = (int)(x & (unsigned)(S.cb_reps - 1));
        if (S.cb_reps > 8) xcd = (int)(x & 7u) * (S.cb_reps >> 3) + (wave & ((S.cb_reps >> 3) - 1));
    }
    return xcd;
}
// ONE generation's update (`val` = alpha delta / 32 x its eligibility) added to the slot of its (identity, mask), as a lane of
// accumulate_kernel does it: the slot on record, its identity compared in full once (LOB_CBS_VERIFIED), the probe sequence
// if it was displaced.  False: no slot (table crowded / hash shared by two identities) -- tr_cbslot is set to -1 and the caller
// leaves the generation to accumulate_kernel's direct path, which must wait until nobody reads theta.
// (`cs` = tr_cbslot[gi] and `sg` = the generation's signature as the caller holds them in registers: nothing is loaded again)
// (`target` 0: theta, 1: theta_b -- double Q, the vector the book's coin picked)
__device__ inline bool acc_generation_at(const DevState& S, size_t gi, int cs, int4 sg, uint32_t mask, f64 val, int xcd, int target = 0) {
    const bool known = cs >= 0 && (cs & LOB_CBS_VERIFIED);
    uint32_t s = (uint32_t)cs & (uint32_t)(S.cb_slots - 1);
    bool found = known;
    uint32_t touch = 0;
    if (cs >= 0) touch = S.cb_touch[s];  // (beside the identity's words)
    if (!known) {
        if (cs >= 0) {
            const int4 id = *reinterpret_cast<const int4*>(S.cb_ident + (size_t)s * 8);
            const uint32_t idm = (uint32_t)S.cb_ident[(size_t)s * 8 + 4];
            found = id.x == sg.x && id.y == sg.y && id.z == sg.z && id.w == sg.w && idm == mask;
        }
        if (!found) {
            const u64 hsh = cb_hash(sg.x, sg.y, sg.z, sg.w, mask);
            s = (uint32_t)hsh & (uint32_t)(S.cb_slots - 1);
            for (int probe = 0; probe < LOB_CB_PROBES; probe++) {
                const u64 kk = S.cb_key[s];
                if (kk == hsh) {
                    const i32* id = S.cb_ident + (size_t)s * 8;
                    found = id[0] == sg.x && id[1] == sg.y && id[2] == sg.z && id[3] == sg.w && (uint32_t)id[4] == mask;
                    if (found) break;
                }
                if (kk == LOB_CB_EMPTY) break;
                s = (s + 1) & (uint32_t)(S.cb_slots - 1);
            }
            if (found) touch = S.cb_touch[s];
        }
        S.tr_cbslot[gi] = found ? (i32)(s | LOB_CBS_VERIFIED) : -1;
        // (accumulate_dense_kernel's record of the slot's dense id, tr_cbd, is not kept up here: this path only runs while the
        // dense sums are off, and the host voids every record when it turns them on -- lob_engine.hip run_steps)
    }
    if (!found) return false;
    __hip_atomic_fetch_add(&S.cb_acc[((size_t)xcd * S.cb_slots + s) * 2 + target], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(touch & (1u << target))) atomicOr(&S.cb_touch[s], 1u << target);
    return true;
}
__device__ inline bool acc_generation(const DevState& S, size_t gi, uint32_t mask, f64 val, int xcd, int target = 0) {
    const int cs = S.tr_cbslot[gi];
    int4 sg = make_int4(0, 0, 0, 0);
    if (!(cs >= 0 && (cs & LOB_CBS_VERIFIED))) sg = *reinterpret_cast<const int4*>(S.tr_sig + gi * 4);
    return acc_generation_at(S, gi, cs, sg, mask, val, xcd, target);
}
// ONE generation without a slot, the whole wave: its update (`scaled` x e(age)) onto the weights of its live tiles (`m`), with
// the written-weights maps kept up as apply_kernel does for a slot's tiles.  Only while nobody reads theta.
__device__ inline void apply_generation_directly(const DevParams& P, const DevState& S, int par, int sid, int b, int sl, int age, uint32_t m, f64 scaled, int target, int lane) {
    f64* theta = target ? S.theta_b : S.theta;
    uint32_t* nz = target ? S.theta_b_nz : S.theta_nz;
    const int j = lane & 31;
    if (lane < 32 && ((m >> j) & 1u)) {
        const i32 f = S.tr_idx[((size_t)b * P.trace_gens + sl) * 32 + j];
        __hip_atomic_fetch_add(&theta[f], scaled * (f64)P.trace_pow[age], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (P.memo) nzx_mark_late(P, S, f, sid);  // (one map for both weight vectors of double Q: written in either)
        const uint32_t bit = LOB_NZ_BIT(f);
        if (!(nz[LOB_NZ_WORD(f)] & bit)) {
            const uint32_t old = atomicOr(&nz[LOB_NZ_WORD(f)], bit);
            if (!(old & bit) && P.carry_verdicts) {
                i32* nz_new = S.nz_new + (2 * target + par) * LOB_NZ_WORDS;
                atomicAdd(&nz_new[0], 1);
                atomicOr((uint32_t*)&nz_new[LOB_NZ_FILTER + (LOB_NZ_WORD(f) & (LOB_NZ_FILTER - 1))], bit);
            }
        }
    }
}
// The generations of ONE book (per lane group of `lpb` lanes: lane `sub` of the group takes the ages sub, sub + lpb, ...) added to
// their slots; `direct_only`: only those without a slot.  Called by whole waves (the direct path ballots): accumulate_kernel with
// 64 >> lpb_shift books per wave, trace_rest_kernel with one.
// `defer` (trace_rest_kernel, where other waves read theta meanwhile): a generation without a slot is not applied here but put on
// `dir_list` for apply_kernel (apply_deferred_generations).
__device__ __forceinline__ void accumulate_generations(const DevParams& P, const DevState& S, int par, int sid, int xcd, int lpb, int sub, int b, bool direct_only, int lane,
                                                       bool defer = false) {
    int n = 0, head = 0, target = 0;
    f64 scaled = 0.0;
    if (b < S.B) {
        const LHdr& h = S.hdr[b];
        if (h.stepped) {
            n = h.tr_n;
            head = h.tr_head;
            scaled = h.upd / (f64)LOB_N_TILINGS;
            target = h.stepped == 2 ? 1 : 0;
        }
    }
    const int G = P.trace_gens;
    const int bb = b < S.B ? b : 0;
    const uint32_t* tr_alive = S.tr_alive + (size_t)bb * G;
    const i32* tr_sig = S.tr_sig + (size_t)bb * G * 4;
    for (int base = 0; base < G; base += lpb) {
        const int age = base + sub;
        if (!__any(age < n)) break;
        const int slot = (head - age + G) & (G - 1);
        uint32_t mask = 0;
        int4 sg = make_int4(0, 0, 0, 0);
        int cs = -1;
        if (age < n) {
            mask = tr_alive[slot];
            cs = S.tr_cbslot[(size_t)bb * G + slot];  // where the generation's claim ended (cb_claim_finish)
        }
        // LOB_CBS_VERIFIED: an earlier step has compared this slot's identity with the generation's, and neither has changed since
        // (a claim rewrites tr_cbslot; the slot cannot have been freed: the generation added to it in every step in between)
        if (direct_only && cs >= 0) mask = 0;  // (its update is in its slot already: acc_generation)
        const bool known = mask != 0 && cs >= 0 && (cs & LOB_CBS_VERIFIED);
        if (mask != 0 && !known) sg = *reinterpret_cast<const int4*>(tr_sig + slot * 4);
        bool direct = false;
        if (mask) {
            bool found = known;
            uint32_t s = (uint32_t)cs & (uint32_t)(S.cb_slots - 1);
            if (!known && cs >= 0) {  // the slot the generation's claim ended on (or, for a claim another lane made, the hash's home slot)
                const int4 id = *reinterpret_cast<const int4*>(S.cb_ident + (size_t)s * 8);
                const uint32_t idm = (uint32_t)S.cb_ident[(size_t)s * 8 + 4];  // (0 in a free slot; mask != 0 here)
                found = id.x == sg.x && id.y == sg.y && id.z == sg.z && id.w == sg.w && idm == mask;
            }
            if (!found) {  // (a displaced slot: walk the probe sequence)
                const u64 hsh = cb_hash(sg.x, sg.y, sg.z, sg.w, mask);
                s = (uint32_t)hsh & (uint32_t)(S.cb_slots - 1);
                for (int probe = 0; probe < LOB_CB_PROBES; probe++) {
                    const u64 kk = S.cb_key[s];
                    if (kk == hsh) {
                        const i32* id = S.cb_ident + (size_t)s * 8;
                        found = id[0] == sg.x && id[1] == sg.y && id[2] == sg.z && id[3] == sg.w && (uint32_t)id[4] == mask;
                        if (found) break;  // (another identity with this hash: the claim may have walked on past it)
                    }
                    if (kk == LOB_CB_EMPTY) break;
                    s = (s + 1) & (uint32_t)(S.cb_slots - 1);
                }
            }
            if (found && !known) S.tr_cbslot[(size_t)bb * G + slot] = (i32)(s | LOB_CBS_VERIFIED);
            if (found) {
                __hip_atomic_fetch_add(&S.cb_acc[((size_t)xcd * S.cb_slots + s) * 2 + target], scaled * (f64)P.trace_pow[age], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(S.cb_touch[s] & (1u << target))) atomicOr(&S.cb_touch[s], 1u << target);
            } else {
                direct = true;
            }
        }
        if (defer) {  // (wave-uniform)
            if (direct) S.dir_list[atomicAdd(&S.dir_list_n[par], 1)] = bb * G + slot;
            continue;
        }
        u64 todo = __ballot(direct);
        while (todo) {  // rare: apply these generations tile by tile, the whole wave per generation
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int d_b = __shfl(bb, src), d_age = __shfl(age, src), d_head = __shfl(head, src), d_t = __shfl(target, src);
            const uint32_t m = __shfl(mask, src);
            const f64 d_scaled = readlane_f64(scaled, src);
            const int sl = (d_head - d_age + G) & (G - 1);
            apply_generation_directly(P, S, par, sid, d_b, sl, d_age, m, d_scaled, d_t, lane);
        }
    }
}
// apply_kernel's share of trace_rest_kernel's work: the generations it found without a slot (`dir_list`: book x G + ring slot),
// applied tile by tile now that nobody reads theta -- the grid's waves stride over the list (empty in most steps).  The list of
// the other parity, consumed a step ago, is emptied for the next step.
__device__ inline void apply_deferred_generations(const DevParams& P, const DevState& S, int par, int sid, int wave, int n_waves, int lane) {
    if (!S.dir_list) return;
    if (wave == 0 && lane == 0) S.dir_list_n[par ^ 1] = 0;
    const int n = S.dir_list_n[par];
    if (n > 0 && wave == 0 && lane == 0) cnt_add(S, 8, (u64)n);  // (lob_debug_deferred)
    const int G = P.trace_gens;
    for (int i = wave; i < n; i += n_waves) {
        const int ent = S.dir_list[i];
        const int b = ent / G, sl = ent & (G - 1);
        const LHdr& h = S.hdr[b];
        const int age = (h.tr_head - sl) & (G - 1);
        apply_generation_directly(P, S, par, sid, b, sl, age, S.tr_alive[(size_t)b * G + sl], h.upd / (f64)LOB_N_TILINGS, h.stepped == 2 ? 1 : 0, lane);
    }
}
// `list` (or null: every book): accumulate_kernel over the books the fused accumulation left (lob_state.h acc_list); an entry
// with bit 31 takes only the book's generations without a slot.
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_BLOCK) accumulate_kernel(LOB_PS_ARGS, int par, int lpb_shift, int sid, const i32* __restrict__ list = nullptr,
                                                               const i32* __restrict__ list_n = nullptr) {
    LOB_PS_REFS
    const int lane = threadIdx.x & 63;
    const int lpb = 1 << lpb_shift, sub = lane & (lpb - 1);
    const int wave = blockIdx.x * LOB_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * LOB_WAVES_PER_BLOCK;
    const int nl = list ? *list_n : 0;
    // the sums are kept in S.cb_reps copies, one per XCD (apply_kernel adds them up): the additions to a popular generation's
    // slot queue up behind each other at one address
    const int xcd = acc_copy(S, wave);  // (more than one copy per XCD -- LOB_ACC_REPS=16|32|64, an experiment: the waves of an XCD spread over cb_reps / 8 copies)
    // (the list: the grid's waves stride over it; every book: one pass)
#pragma unroll 1
    for (int wv = wave;; wv += n_waves) {
        int b = (wv << (6 - lpb_shift)) + (lane >> lpb_shift);
        bool direct_only = false;
        if (list) {
            if ((wv << (6 - lpb_shift)) >= nl) break;  // (wave-uniform)
            const bool have = b < nl;
            const uint32_t ent = have ? (uint32_t)list[b] : 0u;
            direct_only = have && (ent >> 31) != 0;
            b = have ? (int)(ent & 0x7fffffffu) : S.B;
        }
        accumulate_generations(P, S, par, sid, xcd, lpb, sub, b, direct_only, lane);
        if (!list) break;
    }
}
#endif

// accumulate_block_kernel: the same sums for SARSA(lambda), whose every book keeps ~25 live generations -- 1.6 M additions per step
// landing on fewer than 10 k slots, bound by the rate of f64 atomics (11 G/s) however the addresses are spread.  A block takes
// ONE age of 1 024 consecutive books (generations created in one step: the books that were in the same state then hold the very
// same generation, and a popular one is held by a few per cent of all books), sums the updates per slot in an LDS table first
// and sends one global addition per distinct slot: the heavy slots, which carry most of the additions, shrink by the number of
// their holders among the block's books.  Slot look-up, verification and the direct path as in accumulate_kernel.
#define LOB_ACB_BLOCK 1024
#define LOB_ACB_TAB 2048
#ifndef LOB_ACB_K
#define LOB_ACB_K 4      /* batches of LOB_ACB_BLOCK books per block */
#endif
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_ACB_BLOCK) accumulate_block_kernel(LOB_PS_ARGS, int par, int sid, int n_batches) {
    LOB_PS_REFS
    __shared__ i32 keys[LOB_ACB_TAB];
    __shared__ f64 sums[LOB_ACB_TAB];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < LOB_ACB_TAB; i += LOB_ACB_BLOCK) { keys[i] = -1; sums[i] = 0.0; }
    __syncthreads();
    const int age = blockIdx.y;
    int xcd = 0;
    if (S.cb_reps > 1) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcd = (int)(x & (unsigned)(S.cb_reps - 1));
        if (S.cb_reps > 8) xcd = (int)(x & 7u) * (S.cb_reps >> 3) + (int)(blockIdx.x & ((S.cb_reps >> 3) - 1));
    }
    // n_batches (up to LOB_ACB_K, fewer for a small batch of books: the batches are a chain) of LOB_ACB_BLOCK books share the block's table: what a block sends to memory at the end is one addition per
    // DISTINCT slot among its books, and the distinct slots grow far slower than the books (the popular generations are held by
    // a few per cent of all books each)
#pragma unroll 1
    for (int kb = 0; kb < n_batches; kb++) {
        const int b = (blockIdx.x * n_batches + kb) * LOB_ACB_BLOCK + threadIdx.x;
        int n = 0, head = 0;
        f64 scaled = 0.0;
        if (b < S.B) {
            const LHdr& h = S.hdr[b];
            if (h.stepped) { n = h.tr_n; head = h.tr_head; scaled = h.upd / (f64)LOB_N_TILINGS; }
        }
        const bool mine = age < n;
        if (!__syncthreads_or(mine ? 1 : 0)) continue;  // (block-uniform: no book of the batch has a generation this old)
        const int G = P.trace_gens;
        const int bb = b < S.B ? b : 0;
        const int slot = (head - age + G) & (G - 1);
        uint32_t mask = 0;
        int cs = -1;
        if (mine) {
            mask = S.tr_alive[(size_t)bb * G + slot];
            cs = S.tr_cbslot[(size_t)bb * G + slot];
        }
        const bool known = mask != 0 && cs >= 0 && (cs & LOB_CBS_VERIFIED);
        int4 sg = make_int4(0, 0, 0, 0);
        if (mask != 0 && !known) sg = *reinterpret_cast<const int4*>(S.tr_sig + ((size_t)bb * G + slot) * 4);
        bool direct = false, found = false;
        uint32_t s = 0;
        if (mask) {
            found = known;
            s = (uint32_t)cs & (uint32_t)(S.cb_slots - 1);
            if (!known && cs >= 0) {
                const int4 id = *reinterpret_cast<const int4*>(S.cb_ident + (size_t)s * 8);
                const uint32_t idm = (uint32_t)S.cb_ident[(size_t)s * 8 + 4];
                found = id.x == sg.x && id.y == sg.y && id.z == sg.z && id.w == sg.w && idm == mask;
            }
            if (!found) {
                const u64 hsh = cb_hash(sg.x, sg.y, sg.z, sg.w, mask);
                s = (uint32_t)hsh & (uint32_t)(S.cb_slots - 1);
                for (int probe = 0; probe < LOB_CB_PROBES; probe++) {
                    const u64 kk = S.cb_key[s];
                    if (kk == hsh) {
                        const i32* id = S.cb_ident + (size_t)s * 8;
                        found = id[0] == sg.x && id[1] == sg.y && id[2] == sg.z && id[3] == sg.w && (uint32_t)id[4] == mask;
                        if (found) break;
                    }
                    if (kk == LOB_CB_EMPTY) break;
                    s = (s + 1) & (uint32_t)(S.cb_slots - 1);
                }
            }
            if (found && !known) S.tr_cbslot[(size_t)bb * G + slot] = (i32)(s | LOB_CBS_VERIFIED);
            direct = !found;
        }
        // the block's sums per slot
        bool spilled = false;
        const f64 val = scaled * (f64)P.trace_pow[age];
        if (found) {
            uint32_t hh = (s * 2654435761u) >> 21;  // 11 bits
            bool placed = false;
            for (int probe = 0; probe < 16 && !placed; probe++) {
                const i32 old = atomicCAS(&keys[hh], -1, (i32)s);
                if (old == -1 || old == (i32)s) { unsafeAtomicAdd(&sums[hh], val); placed = true; }  // (LDS: ds_add_f64)
                else hh = (hh + 1) & (LOB_ACB_TAB - 1);
            }
            spilled = !placed;
        }
        if (spilled) {  // (a crowded block table: straight to the slot)
            __hip_atomic_fetch_add(&S.cb_acc[((size_t)xcd * S.cb_slots + s) * 2], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(S.cb_touch[s] & 1u)) atomicOr(&S.cb_touch[s], 1u);
        }
        u64 todo = __ballot(direct);
        while (todo) {  // rare: a generation without a slot, applied tile by tile by the whole wave (as accumulate_kernel)
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int d_b = __shfl(bb, src), d_head = __shfl(head, src);
            const uint32_t m = __shfl(mask, src);
            const f64 d_val = readlane_f64(val, src);
            const int sl = (d_head - age + G) & (G - 1);
            const int j = lane & 31;
            if (lane < 32 && ((m >> j) & 1u)) {
                const i32 f = S.tr_idx[((size_t)d_b * G + sl) * 32 + j];
                __hip_atomic_fetch_add(&S.theta[f], d_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P.memo) nzx_mark_late(P, S, f, sid);
                const uint32_t bit = LOB_NZ_BIT(f);
                if (!(S.theta_nz[LOB_NZ_WORD(f)] & bit)) {
                    const uint32_t old = atomicOr(&S.theta_nz[LOB_NZ_WORD(f)], bit);
                    if (!(old & bit) && P.carry_verdicts) {
                        i32* nz_new = S.nz_new + par * LOB_NZ_WORDS;
                        atomicAdd(&nz_new[0], 1);
                        atomicOr((uint32_t*)&nz_new[LOB_NZ_FILTER + (LOB_NZ_WORD(f) & (LOB_NZ_FILTER - 1))], bit);
                    }
                }
            }
        }
    }  // (batches)
    __syncthreads();
    for (int i = threadIdx.x; i < LOB_ACB_TAB; i += LOB_ACB_BLOCK) {
        const i32 k = keys[i];
        if (k < 0) continue;
        __hip_atomic_fetch_add(&S.cb_acc[((size_t)xcd * S.cb_slots + (uint32_t)k) * 2], sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(S.cb_touch[(uint32_t)k] & 1u)) atomicOr(&S.cb_touch[(uint32_t)k], 1u);
    }
}
#endif

// accumulate_dense_kernel: the same sums once more, without the table.  A block of accumulate_block_kernel is bound by its
// insertions -- a compare-and-swap on the key, then the addition, popular slots queueing -- not by what it sends to memory
// (NOTES.md "Round 4").  Here every occupied slot has a DENSE id below LOB_CBD_CAP (lob_state.h cb_dense: handed out when the
// slot is claimed), a generation keeps the id of its slot beside the slot (tr_cbd, valid while tr_cbslot names the same slot,
// verified), and a block adds the terms of its books into a direct-indexed LDS array of doubles: one ds_add_f64 and one
// ds_or_b32 (which ids got a term: a sum of 0.0 must still count as a touch, or apply_kernel would free a slot somebody holds)
// per generation, nothing returned, nothing compared.  The block then writes its array out as its row of cb_part -- plain
// stores; an id without a term carries LOB_ACD_MARK -- and apply_kernel, which walks the occupied slots anyway, adds the rows
// of a slot's id up: no atomic reaches memory for a slot with an id.  Mapping: 32 lanes per book, lane = RING SLOT (not age: the
// three per-generation loads do not wait for the header then), four books' worth of loads in flight per lane.  A generation
// whose record is stale takes accumulate_block_kernel's look-up (identity compared in full once, probe sequence if displaced)
// and renews it; a slot without an id, or a generation without a slot, goes the old way (atomics on cb_acc / tile by tile).
#define LOB_ACD_BLOCK 1024
#define LOB_ACD_UNROLL 4
__host__ __device__ inline size_t acd_lds_bytes() { return (size_t)LOB_CBD_CAP * 8 + (size_t)LOB_CBD_CAP / 8 + (size_t)(LOB_TRACE_GENS + 1) * 4; }
// ids below this were (or may have been) handed out: the deepest any free list has been drained
// (list x starts with ids x, 8 + x, ... lowest on top: position p from the bottom holds id (depth - 1 - p) * 8 + x, so a list
// drained down to p entries has handed out ids below (depth - p) * 8; `cb_ids` = 8 * depth)
__device__ inline int cbd_high_water(const DevState& S) {
    int deepest = S.cb_ids / 8;
    for (int x = 0; x < 8; x++) deepest = min(deepest, S.cb_free_n[2 * x + 1]);
    return (S.cb_ids / 8 - deepest) * 8;
}
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_ACD_BLOCK) accumulate_dense_kernel(LOB_PS_ARGS, int par, int sid, int books_per_block) {
    LOB_PS_REFS
    extern __shared__ __attribute__((aligned(16))) unsigned char acd_raw[];
    f64* sums = reinterpret_cast<f64*>(acd_raw);
    uint32_t* touched = reinterpret_cast<uint32_t*>(acd_raw + (size_t)LOB_CBD_CAP * 8);
    f32* pw = reinterpret_cast<f32*>(acd_raw + (size_t)LOB_CBD_CAP * 8 + (size_t)LOB_CBD_CAP / 8);
    const int tid = threadIdx.x, lane = tid & 63, l = tid & 31, grp = tid >> 5;
    const int n_hi = cbd_high_water(S);
    for (int i = tid; i < n_hi; i += LOB_ACD_BLOCK) sums[i] = 0.0;
    for (int i = tid; i < (n_hi + 31) / 32; i += LOB_ACD_BLOCK) touched[i] = 0u;
    if (tid <= LOB_TRACE_GENS) pw[tid] = P.trace_pow[tid];
    __syncthreads();
    const int G = P.trace_gens, gmask = G - 1;
    const int xcd = acc_copy(S, (int)blockIdx.x);
    const int b_lo = blockIdx.x * books_per_block, b_hi = min(b_lo + books_per_block, S.B);
    const uint32_t smask = (uint32_t)(S.cb_slots - 1);
    // LOB_ACD_BLOCK / 32 books per pass, LOB_ACD_UNROLL passes' loads in flight together; G / 32 ring slots per lane
#pragma unroll 1
    for (int b0 = b_lo; b0 < b_hi; b0 += (LOB_ACD_BLOCK / 32) * LOB_ACD_UNROLL) {
#pragma unroll 1
        for (int ks = l; ks < G; ks += 32) {
            uint32_t m_[LOB_ACD_UNROLL];
            i32 cs_[LOB_ACD_UNROLL], n_[LOB_ACD_UNROLL], head_[LOB_ACD_UNROLL], st_[LOB_ACD_UNROLL];
            u64 pd_[LOB_ACD_UNROLL];
            f64 upd_[LOB_ACD_UNROLL];
#pragma unroll
            for (int u = 0; u < LOB_ACD_UNROLL; u++) {
                const int b = b0 + u * (LOB_ACD_BLOCK / 32) + grp;
                const bool in = b < b_hi;
                const size_t gi = (size_t)(in ? b : b_lo) * G + ks;
                const LHdr& h = S.hdr[in ? b : b_lo];
                st_[u] = in ? h.stepped : 0;
                n_[u] = h.tr_n;
                head_[u] = h.tr_head;
                upd_[u] = h.upd;
                m_[u] = S.tr_alive[gi];
                cs_[u] = S.tr_cbslot[gi];
                pd_[u] = S.tr_cbd[gi];
            }
            // A generation whose record is stale (every generation a step has created or changed: a tenth of them) needs its
            // signature, the slot's identity and the slot's id: asked for together for all the passes in flight, before any is
            // looked at -- one more round trip per batch of passes, not three dependent ones per pass.
            int4 sg_[LOB_ACD_UNROLL], id_[LOB_ACD_UNROLL];
            uint32_t idm_[LOB_ACD_UNROLL], mask_[LOB_ACD_UNROLL];
            i32 dd_[LOB_ACD_UNROLL];
            bool fast_[LOB_ACD_UNROLL];
#pragma unroll
            for (int u = 0; u < LOB_ACD_UNROLL; u++) {
                const int b = b0 + u * (LOB_ACD_BLOCK / 32) + grp;
                const int bb = b < b_hi ? b : b_lo;
                const size_t gi = (size_t)bb * G + ks;
                const int age = (head_[u] - ks) & gmask;
                const bool mine = st_[u] != 0 && age < n_[u];
                mask_[u] = mine ? m_[u] : 0u;
                const int cs = cs_[u];
                const uint32_t s = (uint32_t)cs & smask;
                const bool known = cs >= 0 && (cs & LOB_CBS_VERIFIED);
                fast_[u] = mask_[u] != 0 && known && (uint32_t)(pd_[u] >> 32) == s;
                sg_[u] = make_int4(0, 0, 0, 0); id_[u] = make_int4(0, 0, 0, 0); idm_[u] = 0u; dd_[u] = -1;
                if (mask_[u] != 0 && !fast_[u]) {
                    if (!known) sg_[u] = *reinterpret_cast<const int4*>(S.tr_sig + gi * 4);
                    if (cs >= 0) {
                        if (!known) {
                            id_[u] = *reinterpret_cast<const int4*>(S.cb_ident + (size_t)s * 8);
                            idm_[u] = (uint32_t)S.cb_ident[(size_t)s * 8 + 4];  // (0 in a free slot; mask != 0 here)
                        }
                        dd_[u] = S.cb_dense[s];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < LOB_ACD_UNROLL; u++) {
                const int b = b0 + u * (LOB_ACD_BLOCK / 32) + grp;
                const int bb = b < b_hi ? b : b_lo;
                const size_t gi = (size_t)bb * G + ks;
                const int age = (head_[u] - ks) & gmask;
                const uint32_t mask = mask_[u];
                const int cs = cs_[u];
                const f64 val = upd_[u] / (f64)LOB_N_TILINGS * (f64)pw[age];
                uint32_t s = (uint32_t)cs & smask;
                bool found = false, direct = false;
                i32 d = -1;
                if (mask) {
                    if (fast_[u]) { found = true; d = (i32)(uint32_t)pd_[u]; }  // (the usual case)
                    else {
                        const bool known = cs >= 0 && (cs & LOB_CBS_VERIFIED);
                        const int4 sg = sg_[u];
                        found = known;
                        d = dd_[u];
                        if (!known) {
                            // the slot the generation's claim ended on (or, for a claim another lane made, the hash's home slot)
                            if (cs >= 0) found = id_[u].x == sg.x && id_[u].y == sg.y && id_[u].z == sg.z && id_[u].w == sg.w && idm_[u] == mask;
                            if (!found) {  // (a displaced slot: walk the probe sequence)
                                const u64 hsh = cb_hash(sg.x, sg.y, sg.z, sg.w, mask);
                                s = (uint32_t)hsh & smask;
                                for (int probe = 0; probe < LOB_CB_PROBES; probe++) {
                                    const u64 kk = S.cb_key[s];
                                    if (kk == hsh) {
                                        const i32* id = S.cb_ident + (size_t)s * 8;
                                        found = id[0] == sg.x && id[1] == sg.y && id[2] == sg.z && id[3] == sg.w && (uint32_t)id[4] == mask;
                                        if (found) break;  // (another identity with this hash: the claim may have walked on past it)
                                    }
                                    if (kk == LOB_CB_EMPTY) break;
                                    s = (s + 1) & smask;
                                }
                                if (found) d = S.cb_dense[s];
                            }
                            if (found) S.tr_cbslot[gi] = (i32)(s | LOB_CBS_VERIFIED);
                        }
                        if (found) S.tr_cbd[gi] = ((u64)s << 32) | (u64)(uint32_t)d;
                        direct = !found;
                    }
                }
                if (found) {
                    if (d >= 0) {
                        unsafeAtomicAdd(&sums[d], val);                  // (LDS: ds_add_f64, nothing returned)
                        atomicOr(&touched[d >> 5], 1u << (d & 31));      // (ds_or_b32)
                    } else {  // (a slot without an id: straight to its sums)
                        __hip_atomic_fetch_add(&S.cb_acc[((size_t)xcd * S.cb_slots + s) * 2], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!(S.cb_touch[s] & 1u)) atomicOr(&S.cb_touch[s], 1u);
                    }
                }
                u64 todo = __ballot(direct);
                while (todo) {  // rare: a generation without a slot, applied tile by tile by the whole wave (as accumulate_kernel)
                    const int src = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int d_b = __shfl(bb, src), d_ks = __shfl(ks, src);
                    const uint32_t m = __shfl(mask, src);
                    const f64 d_val = readlane_f64(val, src);
                    const int j = lane & 31;
                    if (lane < 32 && ((m >> j) & 1u)) {
                        const i32 f = S.tr_idx[((size_t)d_b * G + d_ks) * 32 + j];
                        __hip_atomic_fetch_add(&S.theta[f], d_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (P.memo) nzx_mark_late(P, S, f, sid);
                        const uint32_t bit = LOB_NZ_BIT(f);
                        if (!(S.theta_nz[LOB_NZ_WORD(f)] & bit)) {
                            const uint32_t old = atomicOr(&S.theta_nz[LOB_NZ_WORD(f)], bit);
                            if (!(old & bit) && P.carry_verdicts) {
                                i32* nz_new = S.nz_new + par * LOB_NZ_WORDS;
                                atomicAdd(&nz_new[0], 1);
                                atomicOr((uint32_t*)&nz_new[LOB_NZ_FILTER + (LOB_NZ_WORD(f) & (LOB_NZ_FILTER - 1))], bit);
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // the block's row: every id that may be in use, whether this block met it or not (apply_kernel reads all rows of a slot's id)
    u64* row = reinterpret_cast<u64*>(+S.cb_part) + (size_t)blockIdx.x * LOB_CBD_CAP;
    for (int i = tid; i < n_hi; i += LOB_ACD_BLOCK)
        row[i] = ((touched[i >> 5] >> (i & 31)) & 1u) ? (u64)__double_as_longlong(sums[i]) : LOB_ACD_MARK;
}
#endif

// reduce_dense_kernel: the rows of accumulate_dense_kernel's blocks added up by id, LOB_ACD_GROUPS partial rows out (group g = the
// blocks g * per .. (g + 1) * per - 1, in that order) -- consecutive threads take consecutive ids, so every load is one
// contiguous line; apply_kernel then reads LOB_ACD_GROUPS values per slot the way it reads the per-XCD copies of cb_acc.  (The
// first version let every wave of apply_kernel read its slot's id from all 256 rows: four loads of 64 scattered lines per slot,
// apply_kernel 0.026 -> 0.059 ms.)
#define LOB_ACD_GROUPS 8
#if LOB_IN_MAIN
__global__ void __launch_bounds__(256) reduce_dense_kernel(DevState S, int dense_blocks) {
    const int n_hi = cbd_high_water(S);
    const int id = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (id >= n_hi) return;
    const int per = (dense_blocks + LOB_ACD_GROUPS - 1) / LOB_ACD_GROUPS;
    const int r0 = g * per, r1 = min(r0 + per, dense_blocks);
    const u64* part = reinterpret_cast<const u64*>(+S.cb_part) + id;
    f64 acc = 0.0;
    bool any = false;
#pragma unroll 8
    for (int r = r0; r < r1; r++) {
        const u64 raw = part[(size_t)r * LOB_CBD_CAP];
        if (raw != LOB_ACD_MARK) { any = true; acc += __longlong_as_double((long long)raw); }
    }
    reinterpret_cast<u64*>(+S.cb_red)[(size_t)g * LOB_CBD_CAP + id] = any ? (u64)__double_as_longlong(acc) : LOB_ACD_MARK;
}
#endif

// ... and the indices that launch found ambiguous for the first time are marked in every registered slot that holds them (the
// slots registered since then know already: tile_register looks at the bitmap).  LOB_SCAN_BLOCKS extra blocks of apply_kernel's
// launch, behind the launch that registered (trace_lane_kernel's: registry_block, lob_fast.h).
#if LOB_IN_MAIN
__device__ inline void registry_scan_block(const DevState& S, int par /* of the registry launch, not the step's */, int blk, int nblk) {
    int n_new = S.amb_new_n[par];
    if (n_new > S.amb_cap) n_new = S.amb_cap;
    if (blk == 0 && threadIdx.x == 0) S.amb_new_n[par ^ 1] = 0;  // (consumed by the step before this one)
    if (n_new == 0) return;
    const int lane = threadIdx.x & 63, j = lane & 31;
    const bool hi = lane >= 32;
    const int wave = blk * 4 + (threadIdx.x >> 6), n_waves = nblk * 4;
    const int n_all = S.mk_all_n[0];
    const i32* nw = S.amb_new + (size_t)par * S.amb_cap;
    for (int i = wave; i < n_all; i += n_waves) {
        const int s = S.mk_all[i];
        i32 tl[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int a = (hi ? 5 : 0) + k;
            tl[k] = a < LOB_N_ACTIONS ? S.mk_tiles[((size_t)s * LOB_N_ACTIONS + a) * 32 + j] : -1;
        }
        for (int e = 0; e < n_new; e++) {
            const i32 f = nw[e];  // (wave-uniform)
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (tl[k] == f) atomicOr(&S.mk_amb[(size_t)s * LOB_N_ACTIONS + (hi ? 5 : 0) + k], 1u << j);
        }
    }
}
#endif

// apply_kernel: one wave per occupied slot (the step's list, lob_learn.h).  Touched this step: theta[tile] += the summed update
// for the live tiles of the slot's generation -- the 32 indices follow from its identity (quantised triple + action; all 0 for
// a constructor-zero State: learn_traces) --, maintain the written-weights map and the carry-over filter, hand the slot on to
// the next step's list.  Not touched: no stepped book holds the generation any more, free the slot.
#if LOB_IN_MAIN
// `dense_blocks` > 0: accumulate_dense_kernel + reduce_dense_kernel ran this step -- a slot with a dense id also gets the
// LOB_ACD_GROUPS partial sums of cb_red at its id (lane x reads group x's; LOB_ACD_MARK = no block of the group had a term for
// it), and counts as touched if any holds a term.  A slot that is freed hands its id back to the list it came from.  (< 0: no
// slot has ever been given an id -- nothing is looked up.)
__global__ void __launch_bounds__(256) apply_kernel(LOB_PS_ARGS, const uint32_t* __restrict__ rnd_g, int par, int sid, int dense_blocks, int apar) {
    LOB_PS_REFS
    __shared__ uint32_t rnd[2048 + 32];
    __shared__ int n_surv;
    const int segs = S.cb_segs, scan_blocks = (int)gridDim.x - segs;
    if ((int)blockIdx.x < scan_blocks) {  // the launch's FIRST blocks (they start with it, not behind its 2 048 others): registry_scan_block above
        registry_scan_block(S, apar, (int)blockIdx.x, scan_blocks);
        return;
    }
    // one block per segment of the table: its list, its survivors -- no counter shared between blocks
    const int sblk = (int)blockIdx.x - scan_blocks;
    const int seg = par * segs + sblk, seg_next = (par ^ 1) * segs + sblk, cap = S.cb_slots / segs;
    apply_deferred_generations(P, S, par, sid, sblk * 4 + (int)(threadIdx.x >> 6), segs * 4, (int)(threadIdx.x & 63));
    const int count = S.cb_count[seg];
    if (count == 0) return;  // (block-uniform)
    {
        const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
        uint4* dst = reinterpret_cast<uint4*>(rnd);
        const uint4 r0 = src[threadIdx.x], r1 = src[threadIdx.x + 256];
        dst[threadIdx.x] = r0; dst[threadIdx.x + 256] = r1;
        if (threadIdx.x < 27) rnd[2048 + threadIdx.x] = rnd_g[2048 + threadIdx.x];
        if (threadIdx.x == 0) n_surv = 0;
    }
    const int lane = threadIdx.x & 63, j = lane & 31;
    const int wave = threadIdx.x >> 6, n_waves = 4;
    const i32* list = S.cb_list + (size_t)seg * cap;
    i32* list_next = S.cb_list + (size_t)seg_next * cap;
    const uint32_t M = (uint32_t)P.M;
    __syncthreads();
    for (int i = wave; i < count; i += n_waves) {
        const int s = list[i];
        const uint32_t touch = S.cb_touch[s];
        const int4 id = *reinterpret_cast<const int4*>(S.cb_ident + (size_t)s * 8);
        const uint32_t mask = (uint32_t)S.cb_ident[(size_t)s * 8 + 4];
        // (the copies of the sums, one per XCD: lane x fetches copy x, added up in the order of the copies)
        f64 v0 = 0.0, v1 = 0.0;
        {
            f64 c0 = 0.0, c1 = 0.0;
            if (lane < S.cb_reps) {
                c0 = S.cb_acc[((size_t)lane * S.cb_slots + s) * 2];
                c1 = S.cb_acc[((size_t)lane * S.cb_slots + s) * 2 + 1];
            }
            for (int x = 0; x < S.cb_reps; x++) {
                v0 += readlane_f64(c0, x);
                v1 += readlane_f64(c1, x);
            }
        }
        const i32 did = (S.cb_dense && dense_blocks >= 0) ? S.cb_dense[s] : -1;
        bool dense_touch = false;
        if (dense_blocks > 0 && did >= 0) {
            u64 raw = LOB_ACD_MARK;
            if (lane < LOB_ACD_GROUPS) raw = reinterpret_cast<const u64*>(+S.cb_red)[(size_t)lane * LOB_CBD_CAP + did];
            const bool have = raw != LOB_ACD_MARK;
            const f64 x = have ? __longlong_as_double((long long)raw) : 0.0;
            dense_touch = __any(have);
            for (int g = 0; g < LOB_ACD_GROUPS; g++) v0 += readlane_f64(x, g);  // (in the order of the groups)
        }
        if (touch == 0 && !dense_touch) {  // (wave-uniform)
            if (lane == 0) {
                S.cb_key[s] = LOB_CB_EMPTY;
                S.cb_ident[(size_t)s * 8 + 4] = 0;
                if (did >= 0) {  // the id goes back on the list it came from (ids = x mod 8 belong to list x)
                    S.cb_dense[s] = -1;
                    const int x = did & 7;
                    S.cb_free[(size_t)x * (LOB_CBD_CAP / 8) + atomicAdd(&S.cb_free_n[2 * x], 1)] = did;
                }
            }
            continue;
        }
        i32 f = 0;
        if (!(id.w & 256)) {
            uint32_t sum = 0;
            int base = j;
            sum = mod_add(sum, rnd[(tile_coord(id.x, base) + 449 * 0) & 2047], M); base += 2 * j;
            sum = mod_add(sum, rnd[(tile_coord(id.y, base) + 449 * 1) & 2047], M); base += 2 * j;
            sum = mod_add(sum, rnd[(tile_coord(id.z, base) + 449 * 2) & 2047], M);
            sum = mod_add(sum, rnd[(j + 449 * 3) & 2047], M);
            f = tile_index(sum, rnd[2048 + (id.w & 15)], M);
        }
        // lanes 0-31 serve theta, lanes 32-63 theta_b (double Q)
        const int t = lane >> 5;
        if ((((touch | (dense_touch ? 1u : 0u)) >> t) & 1u) && ((mask >> j) & 1u)) {
            f64* theta = t ? S.theta_b : S.theta;
            uint32_t* nz = t ? S.theta_b_nz : S.theta_nz;
            __hip_atomic_fetch_add(&theta[f], t ? v1 : v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (P.memo) nzx_mark_late(P, S, f, sid);  // (one map for both weight vectors of double Q: written in either)
            const uint32_t bit = LOB_NZ_BIT(f);
            if (!(nz[LOB_NZ_WORD(f)] & bit)) {
                const uint32_t old = atomicOr(&nz[LOB_NZ_WORD(f)], bit);
                if (!(old & bit) && P.carry_verdicts) {  // tell the next act_kernel (verdict carry-over)
                    i32* nz_new = S.nz_new + (2 * t + par) * LOB_NZ_WORDS;
                    atomicAdd(&nz_new[0], 1);
                    atomicOr((uint32_t*)&nz_new[LOB_NZ_FILTER + (LOB_NZ_WORD(f) & (LOB_NZ_FILTER - 1))], bit);
                }
            }
        }
        if (lane < S.cb_reps) {
            S.cb_acc[((size_t)lane * S.cb_slots + s) * 2] = 0.0;
            S.cb_acc[((size_t)lane * S.cb_slots + s) * 2 + 1] = 0.0;
        }
        if (lane == 0) {
            S.cb_touch[s] = 0;
            list_next[atomicAdd(&n_surv, 1)] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S.cb_count[seg_next] = n_surv;  // (0 until now: the previous step's launch consumed that list; this step's claims are over)
        S.cb_count[seg] = 0;
    }
}
#endif

// Group-0 memo: S0(a) = the first 32 terms of Agent::getQ (agent.cpp:117-135), sum over the tilings of
// w0 * theta[group-0 tile], in the reference's order, for every triple on this step's list.  One wave
// per triple: lanes 0-31 (tiling j) fetch actions 0-4, lanes 32-63 actions 5-8; the products go
// through LDS and nine lanes add them up sequentially, exactly as q_values does.  `which` 0: under
// theta_t, read by learn_kernel; 1: after the update, read by the next act_kernel.  A few hundred
// triples per step: the hash table is read from global memory (8 KB, cache-resident), no staging.
// `reset_lpar` >= 0 (the step's last launch, which 1): also what an act kernel does before anything else -- empty the lists this
// step has consumed -- so that the next step may start with env_kernel<.., 1>, whose blocks append to them from the first
// instruction on.
#if LOB_IN_MAIN
__global__ void __launch_bounds__(256) memo_kernel(DevParams P, DevState S, const uint32_t* __restrict__ rnd_g, int par, int which, u64 ver, int reset_lpar) {
    if (reset_lpar >= 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        S.mk_count[par ^ 1] = 0;               // the next step's list of memo slots
        S.slow_n[reset_lpar * 2 + 0] = 0;      // this step's work lists: the step after the next fills them again
        S.slow_n[reset_lpar * 2 + 1] = 0;
        S.tr_list_n[reset_lpar] = 0;
        S.tr_list2_n[reset_lpar] = 0;
        S.acc_list_n[reset_lpar] = 0;
    }
    // the two words of the state a step changes, for the kernels that read the state through its device-resident copy (S.self,
    // lob_engine.hip sync_state): this launch (which 0) is in front of every kernel of the step that claims combine slots
    if (which == 0 && blockIdx.x == 0 && threadIdx.x == 0 && S.self) {
        DevState* w = const_cast<DevState*>(S.self);
        w->cb_par = S.cb_par;
        w->cb_dense_on = S.cb_dense_on;
    }
    __shared__ f64 vals[4][LOB_N_ACTIONS * LOB_QSTRIDE];
    __shared__ uint32_t rnd[2048 + 32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31;
    const bool hi = lane >= 32;
    const int wave = blockIdx.x * 4 + w, n_waves = gridDim.x * 4;
    const uint32_t M = (uint32_t)P.M;
    // The kernel is a short chain of dependent look-ups (count -> list entry -> identity -> hash table -> weights) run by a few
    // hundred waves: the first list entry is requested with the count (wave i takes entry i first; the list is always
    // allocated in full), and the hash table goes to LDS meanwhile -- two round trips fewer.
    const int s_first = S.mk_list[(size_t)par * S.mk_slots + (wave < S.mk_slots ? wave : 0)];
    int count = S.mk_count[par];
    {
        const uint4* src = reinterpret_cast<const uint4*>(rnd_g);
        uint4* dst = reinterpret_cast<uint4*>(rnd);
        const uint4 r0 = src[threadIdx.x], r1 = src[threadIdx.x + 256];
        dst[threadIdx.x] = r0; dst[threadIdx.x + 256] = r1;
        if (threadIdx.x < 27) rnd[2048 + threadIdx.x] = rnd_g[2048 + threadIdx.x];
    }
    __syncthreads();
    if (count > S.mk_slots) count = S.mk_slots;
    f64* v = vals[w];
    for (int i = wave; i < count; i += n_waves) {
        const int s = i == wave ? s_first : S.mk_list[(size_t)par * S.mk_slots + i];
        const int4 id = *reinterpret_cast<const int4*>(S.mk_ident + (size_t)s * 4);
        uint32_t sum = 0;
        {
            int base = j;
            sum = mod_add(sum, rnd[(tile_coord(id.x, base) + 449 * 0) & 2047], M); base += 2 * j;
            sum = mod_add(sum, rnd[(tile_coord(id.y, base) + 449 * 1) & 2047], M); base += 2 * j;
            sum = mod_add(sum, rnd[(tile_coord(id.z, base) + 449 * 2) & 2047], M);
            sum = mod_add(sum, rnd[(j + 449 * 3) & 2047], M);
        }
        f64 t[5], tb[5];
        const bool fill = (S.mk_tiles_ok[s] & 1) == 0;  // first time on a list: leave the tile indices for the trace kernel
        const bool two = S.mk_rec_b != nullptr;         // double Q: the same 288 tiles under theta_b
        if (which == 0 && !fill) {
            // theta_t of this step is theta_{t+1} of the step before: a triple that was on THAT step's list has its S0 under these very
            // weights in the other record already (memo_kernel which 1, same version) -- 80 bytes copied instead of 288 gathers
            // (most triples of a step: the books move among a few hundred of them)
            const f64* r1 = S.mk_rec + ((size_t)S.mk_slots + s) * LOB_MK_REC;
            if (reinterpret_cast<const u64*>(r1)[LOB_N_ACTIONS] == ver) {   // (wave-uniform)
                f64* r0 = S.mk_rec + (size_t)s * LOB_MK_REC;
                if (lane <= LOB_N_ACTIONS) reinterpret_cast<u64*>(r0)[lane] = reinterpret_cast<const u64*>(r1)[lane];
                if (two) {
                    const f64* b1 = S.mk_rec_b + ((size_t)S.mk_slots + s) * LOB_MK_REC;
                    f64* b0 = S.mk_rec_b + (size_t)s * LOB_MK_REC;
                    if (lane <= LOB_N_ACTIONS) reinterpret_cast<u64*>(b0)[lane] = reinterpret_cast<const u64*>(b1)[lane];
                }
                continue;
            }
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int a = (hi ? 5 : 0) + k;
            const i32 tile = tile_index(sum, rnd[2048 + (a < LOB_N_ACTIONS ? a : 0)], M);
            t[k] = a < LOB_N_ACTIONS ? S.theta[tile] : 0.0;
            tb[k] = (two && a < LOB_N_ACTIONS) ? S.theta_b[tile] : 0.0;
            if (fill && a < LOB_N_ACTIONS) S.mk_tiles[((size_t)s * LOB_N_ACTIONS + a) * 32 + j] = tile;
        }
        if (fill && lane == 0) {
            __threadfence();  // (read by a LATER kernel only; the flag just must not precede the tiles of another wave's view: one wave per slot)
            atomicOr(&S.mk_tiles_ok[s], 1);  // (bit 1 belongs to registry_block)
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int a = (hi ? 5 : 0) + k;
            if (a < LOB_N_ACTIONS) v[a * LOB_QSTRIDE + j] = P.w0 * t[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        f64 q = 0.0;
        if (lane < LOB_N_ACTIONS) {
            for (int k = 0; k < 32; k++) q += v[lane * LOB_QSTRIDE + k];
        }
        f64* rec = S.mk_rec + ((size_t)which * S.mk_slots + s) * LOB_MK_REC;
        if (lane < LOB_N_ACTIONS) rec[lane] = q;
        if (lane == LOB_N_ACTIONS) reinterpret_cast<u64*>(rec)[LOB_N_ACTIONS] = ver;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        if (two) {  // (wave-uniform) DoubleAgent::getQb's first 32 terms, the same way
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int a = (hi ? 5 : 0) + k;
                if (a < LOB_N_ACTIONS) v[a * LOB_QSTRIDE + j] = P.w0 * tb[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            f64 qb = 0.0;
            if (lane < LOB_N_ACTIONS) {
                for (int k = 0; k < 32; k++) qb += v[lane * LOB_QSTRIDE + k];
            }
            f64* recb = S.mk_rec_b + ((size_t)which * S.mk_slots + s) * LOB_MK_REC;
            if (lane < LOB_N_ACTIONS) recb[lane] = qb;
            if (lane == LOB_N_ACTIONS) reinterpret_cast<u64*>(recb)[LOB_N_ACTIONS] = ver;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the (triple, action) pairs this step's act_light_kernel met for the first time: their 32 tiles are marked in the
    // written-weights maps here, a lane per tile, before the learn kernel looks (see learn_traces)
    if (which == 0) {
        int nm = S.mk_markcount[0];
        if (nm > S.mk_slots) nm = S.mk_slots;
        for (int i = wave; i < nm; i += n_waves) {
            const int e = S.mk_marklist[i];
            if (lane < 32) nzx_mark(P, S, S.mk_tiles[((size_t)(e >> 4) * LOB_N_ACTIONS + (e & 15)) * 32 + lane]);
        }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        S.mk_markcount[0] = 0;
    }
}
#endif

// Agent::gen(seed) for every book: std::mt19937_64 seeded with (unsigned)(seed + global book id)
#if LOB_IN_MAIN
__global__ void mt_init_kernel(DevParams P, DevState S) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    mt64_seed(S.mt_state + (size_t)b * LOB_MT_N, (u64)(uint32_t)(P.seed + P.book_id_offset + (u64)b));
    S.mt_idx[b] = LOB_MT_N;  // first draw regenerates the block
}
#endif

// model_log (lob_state.h ml_*): sum of |delta| over the stepped books of this step, a partial per block (thread t of block k takes
// the books k * per + t, + 256, ...; the block adds its threads' sums up in a fixed tree) ...
#if LOB_IN_MAIN
__global__ void __launch_bounds__(256) td_stats_kernel(DevState S, int per) {
    __shared__ f64 red[256];
    __shared__ i32 cnt[256];
    const int lo = blockIdx.x * per, hi = min(lo + per, S.B);
    f64 a = 0.0;
    i32 n = 0;
    for (int b = lo + (int)threadIdx.x; b < hi; b += 256) {
        const LHdr& h = S.hdr[b];
        if (h.stepped) { a += fabs(h.td); n++; }
    }
    red[threadIdx.x] = a;
    cnt[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { red[threadIdx.x] += red[threadIdx.x + off]; cnt[threadIdx.x] += cnt[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { S.ml_part[blockIdx.x] = red[0]; S.ml_npart[blockIdx.x] = cnt[0]; }
}
// ... and the partials in order into the running aggregate; a row once it holds 1000 updates or more (agent.cpp:95-99)
__global__ void td_stats_fold_kernel(DevState S, int n_blocks) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    f64 agg = S.ml_agg[0];
    i64 cnt = S.ml_cnt[0];
    for (int k = 0; k < n_blocks; k++) { agg += S.ml_part[k]; cnt += S.ml_npart[k]; }
    if (cnt >= 1000) {
        const i64 row = S.ml_cnt[1];
        if (row < LOB_ML_ROWS) S.ml_rows[row] = agg / (f64)cnt;
        S.ml_cnt[1] = row + 1;
        agg = 0.0;
        cnt = 0;
    }
    S.ml_agg[0] = agg;
    S.ml_cnt[0] = cnt;
}
#endif

// The striped device counters (lob_state.h cnt_add) added up for the host: out[i] = sum over the stripes of counter i.
#if LOB_IN_MAIN
// ... and, in out[15] (no counter lives there), the books that are still live (done == 0): what Runner::RunEpisode's loop asks after
// every chunk of steps -- counted here instead of copying every book's flag to the host
__global__ void __launch_bounds__(LOB_CNT_STRIPES) counters_fold_kernel(const i64* __restrict__ cnt, i64* out, const i32* __restrict__ done, int B) {
    __shared__ i64 part[LOB_CNT_STRIPES][16];
    __shared__ i32 live[LOB_CNT_STRIPES];
    for (int i = 0; i < 16; i++) part[threadIdx.x][i] = cnt[(size_t)threadIdx.x * LOB_CNT_STRIDE + i];
    i32 n = 0;
    for (int b = threadIdx.x; b < B; b += LOB_CNT_STRIPES) n += done[b] == 0;
    live[threadIdx.x] = n;
    __syncthreads();
    if (threadIdx.x < 15) {
        i64 acc = 0;
        for (int st = 0; st < LOB_CNT_STRIPES; st++) acc += part[st][threadIdx.x];
        out[threadIdx.x] = acc;
    } else if (threadIdx.x == 15) {
        i64 acc = 0;
        for (int st = 0; st < LOB_CNT_STRIPES; st++) acc += live[st];
        out[15] = acc;
    }
}
__global__ void counters_zero_kernel(i64* cnt, int idx) { cnt[(size_t)threadIdx.x * LOB_CNT_STRIDE + idx] = 0; }
#endif

// The two words of the state a learner step changes, into its device-resident copy (lob_engine.hip sync_state) -- for the flows
// without a memo_kernel launch in front of the kernels that claim combine slots (memo_kernel does this itself).
#if LOB_IN_MAIN
__global__ void step_words_kernel(DevState* self, int cb_par, int cb_dense_on) {
    self->cb_par = cb_par;
    self->cb_dense_on = cb_dense_on;
}
#endif

// Agent::HandleTerminal: traces.decay(0.0) (agent.cpp:103-109)
#if LOB_IN_MAIN
__global__ void clear_traces_kernel(DevState S) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < S.B) S.hdr[b].tr_n = 0;
}
#endif

// State::newState(vector<float>&) + getFeatures / Agent::getQ for n free-standing
// states (lob_features / lob_q_values).  Wave per state.
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_BLOCK) features_kernel(DevParams P, const f64* __restrict__ theta,
                                                             const uint32_t* __restrict__ nz, const uint32_t* __restrict__ rnd_g, const f32* vars,
                                                             int n, i32* out_idx, f64* out_q) {
    __shared__ LearnLds L;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * LOB_WAVES_PER_BLOCK + w;
    learn_stage_table(rnd_g, L);
    if (s >= n) return;
    if (lane < 16) L.vars[w][0][lane] = lane < P.V ? vars[(size_t)s * P.V + lane] : 0.0f;
    wave_lds_fence();
    if (out_idx) {
        for (int p = lane; p < 96; p += 64) {
            const int g = p >> 5, j = p & 31;
            const int nf = g == 0 ? 3 : (g == 1 ? P.V - 3 : P.V);
            const f32* v = g == 1 ? L.vars[w][0] + 3 : L.vars[w][0];
            const uint32_t base = tile_base_m((uint32_t)P.M, v, nf, j, L.rnd);
            for (int a = 0; a < LOB_N_ACTIONS; a++)
                out_idx[((size_t)s * LOB_N_ACTIONS + a) * 96 + p] = tile_index(base, L.act_terms[g * LOB_N_ACTIONS + a], (uint32_t)P.M);
        }
    }
    if (out_q) {
        f64 qs[LOB_N_ACTIONS];
        q_values(P, theta, nz, L.vars[w][0], false, L.rnd, L.act_terms, L.vals[w], lane, qs);
        if (lane < LOB_N_ACTIONS) out_q[(size_t)s * LOB_N_ACTIONS + lane] = qs[lane];
    }
}
#endif

// ---- multi-GPU weight exchange --------------------------------------------
#if LOB_IN_MAIN
__global__ void delta_begin_kernel(const f64* __restrict__ theta, const f64* __restrict__ sync, f64* delta, i64 M) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < M; i += stride) delta[i] = theta[i] - sync[i];
}
#endif
#if LOB_IN_MAIN
__global__ void delta_apply_kernel(f64* theta, f64* sync, const f64* __restrict__ delta, uint32_t* nz, i32* nz_epoch, i64 M,
                                   uint32_t* nzx, uint32_t* nzc, int cshift, uint32_t* nzd, const uint32_t* nzd_terms) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) atomicAdd(nz_epoch, 1);  // verdicts saved before this exchange are stale
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < M; i += stride) {
        const f64 d = delta[i];
        const f64 t = sync[i] + d;
        theta[i] = t;
        sync[i] = t;
        if (d != 0.0) {  // written on some rank: from now on the weight must be fetched
            const uint32_t bit = LOB_NZ_BIT(i);
            if (!(nz[LOB_NZ_WORD(i)] & bit)) atomicOr(&nz[LOB_NZ_WORD(i)], bit);
            if (nzx) {  // the fast path's maps (theta only)
                const uint32_t xb = 1u << ((uint32_t)i & 31);
                if (!(nzx[(uint32_t)i >> 5] & xb)) {
                    const uint32_t old = atomicOr(&nzx[(uint32_t)i >> 5], xb);
                    if (!(old & xb) && nzd) nzd_mark(nzd, nzd_terms, (uint32_t)M, (uint32_t)i);
                }
                const uint32_t c = (uint32_t)i >> cshift;
                if (!(nzc[c >> 5] & (1u << (c & 31)))) atomicOr(&nzc[c >> 5], 1u << (c & 31));
            }
        }
    }
}
#endif
// ---- sparse weight exchange (shared theta on the fast path) ---------------------------------------------------------------
// The exact written-weights map theta_nzx (one bit per weight, 2.5 MB at M = 20 M) enumerates every weight a step of this
// rank has touched or is about to: a few hundred thousand of 20 M.  The ranks all-gather their maps; the UNION, in index
// order, defines one compact vector layout common to all ranks; each rank packs theta - theta_sync of those weights into
// it, ONE all-reduce (f64, SUM) of |union| doubles replaces the dense one of M, and the result is scattered back.  Every
// weight whose delta is non-zero on some rank has its bit set there, so the sum equals the dense exchange's term by term.
#define LOB_SPX_BLOCK 256
// union of the gathered maps + set bits per block
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_SPX_BLOCK) sparse_union_kernel(const uint32_t* __restrict__ gathered, int world, i64 words, uint32_t* u_map, i32* block_cnt) {
    __shared__ i32 red[LOB_SPX_BLOCK / 64];
    const i64 w = (i64)blockIdx.x * LOB_SPX_BLOCK + threadIdx.x;
    uint32_t u = 0;
    if (w < words) {
        for (int r = 0; r < world; r++) u |= gathered[(size_t)r * words + w];
        u_map[w] = u;
    }
    i32 c = __builtin_popcount(u);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        i32 t = 0;
        for (int i = 0; i < LOB_SPX_BLOCK / 64; i++) t += red[i];
        block_cnt[blockIdx.x] = t;
    }
}
#endif
// exclusive scan of the block counts (one block; a few thousand entries) + the total
#if LOB_IN_MAIN
__global__ void __launch_bounds__(1024) sparse_scan_kernel(const i32* __restrict__ block_cnt, int n_blocks, i64* block_off, i64* total) {
    __shared__ i64 part[1024];
    const int per = (n_blocks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    i64 s = 0;
    for (int i = lo; i < hi; i++) s += block_cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        i64 run = 0;
        for (int i = 0; i < 1024; i++) { const i64 v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    i64 run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { block_off[i] = run; run += block_cnt[i]; }
}
#endif
// position of a word's first set bit in the compact vector: block offset + exclusive prefix of the popcounts inside the block
__device__ inline i64 sparse_word_base(uint32_t u, const i64* block_off, i32* lds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const i32 c = __builtin_popcount(u);
    i32 incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const i32 v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) lds[wv] = incl;
    __syncthreads();
    i32 before = 0;
    for (int i = 0; i < wv; i++) before += lds[i];
    return block_off[blockIdx.x] + before + incl - c;
}
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_SPX_BLOCK) sparse_pack_kernel(const uint32_t* __restrict__ u_map, i64 words, const i64* __restrict__ block_off,
                                                                    const f64* __restrict__ theta, const f64* __restrict__ sync, f64* buf, i64 cap) {
    __shared__ i32 lds[LOB_SPX_BLOCK / 64];
    const i64 w = (i64)blockIdx.x * LOB_SPX_BLOCK + threadIdx.x;
    uint32_t u = w < words ? u_map[w] : 0u;
    i64 p = sparse_word_base(u, block_off, lds);
    while (u) {
        const i64 f = (w << 5) + __builtin_ctz(u);
        u &= u - 1;
        if (p < cap) buf[p] = theta[f] - sync[f];   // (an entry beyond the exchanged count keeps its delta for the next exchange)
        p++;
    }
}
// the exchanged vector has a FIXED length (agreed without asking the device): the entries behind the union's are zero
__global__ void __launch_bounds__(256) sparse_tail_kernel(f64* buf, const i64* __restrict__ total, i64 cap) {
    const i64 n = *total;
    for (i64 i = n + (i64)blockIdx.x * 256 + threadIdx.x; i < cap; i += (i64)gridDim.x * 256) buf[i] = 0.0;
}
#endif
// theta = theta_sync + sum(delta), theta_sync = theta for the weights of the union; the maps take the union's bits (a set bit
// only means "fetch the weight": a weight another rank marked but has not written yet reads as +0.0)
#if LOB_IN_MAIN
__global__ void __launch_bounds__(LOB_SPX_BLOCK) sparse_apply_kernel(const uint32_t* __restrict__ u_map, i64 words, const i64* __restrict__ block_off,
                                                                     f64* theta, f64* sync, const f64* __restrict__ buf, uint32_t* nz, i32* nz_epoch,
                                                                     uint32_t* nzx, uint32_t* nzc, int cshift, uint32_t* nzd, const uint32_t* nzd_terms, i64 M, i64 cap) {
    __shared__ i32 lds[LOB_SPX_BLOCK / 64];
    const i64 w = (i64)blockIdx.x * LOB_SPX_BLOCK + threadIdx.x;
    if (w == 0) atomicAdd(nz_epoch, 1);  // verdicts saved before this exchange are stale
    const uint32_t u0 = w < words ? u_map[w] : 0u;
    uint32_t u = u0;
    i64 p = sparse_word_base(u, block_off, lds);
    uint32_t nzm = 0;
    while (u) {
        const i64 f = (w << 5) + __builtin_ctz(u);
        u &= u - 1;
        if (p < cap) {   // (beyond the exchanged count: theta and theta_sync stay as they are, the delta travels next time)
            const f64 t = sync[f] + buf[p];
            theta[f] = t;
            sync[f] = t;
        }
        p++;
        nzm |= LOB_NZ_BIT(f);
    }
    if (u0) {
        const uint32_t have = nzx[w];
        if (have != (have | u0)) {  // (this thread owns the word; the rank's own bits are part of the union)
            nzx[w] = have | u0;
            if (nzd) {
                uint32_t fresh = u0 & ~have;  // weights only other ranks have written so far
                while (fresh) {
                    nzd_mark(nzd, nzd_terms, (uint32_t)M, (uint32_t)((w << 5) + __builtin_ctz(fresh)));
                    fresh &= fresh - 1;
                }
            }
        }
        const i64 f0 = w << 5;  // the 32 weights of a word share their coarse bit (cshift >= 5) and their word of the 1-in-8 map
        const uint32_t c = (uint32_t)(f0 >> cshift);
        if (!(nzc[c >> 5] & (1u << (c & 31)))) atomicOr(&nzc[c >> 5], 1u << (c & 31));
        if ((nz[LOB_NZ_WORD(f0)] & nzm) != nzm) atomicOr(&nz[LOB_NZ_WORD(f0)], nzm);
    }
}
#endif
// the fast path's maps after lob_theta_set (both cleared by the caller first): bit = (theta != +0.0 bitwise)
#if LOB_IN_MAIN
__global__ void rebuild_nzx_kernel(const f64* __restrict__ theta, uint32_t* nzx, uint32_t* nzc, int cshift, i64 M) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < M; i += stride) {
        if (__double_as_longlong(theta[i]) != 0) {
            atomicOr(&nzx[(uint32_t)i >> 5], 1u << ((uint32_t)i & 31));
            const uint32_t c = (uint32_t)i >> cshift;
            atomicOr(&nzc[c >> 5], 1u << (c & 31));
        }
    }
}
#endif
// ... and the map folded over the actions from the exact one (gather form: no atomics; monotone like the exact map)
#if LOB_IN_MAIN
__global__ void rebuild_nzd_kernel(const uint32_t* __restrict__ nzx, uint32_t* nzd, const uint32_t* __restrict__ terms18, i64 M) {
    const i64 words = M / 32 + 1;
    uint32_t t[18];
#pragma unroll
    for (int i = 0; i < 18; i++) t[i] = terms18[i];
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (i64)gridDim.x * blockDim.x) {
        uint32_t out1 = 0, out2 = 0;
        for (int k = 0; k < 32; k++) {
            const i64 s = (w << 5) + k;
            if (s >= M) break;
            bool any1 = false, any2 = false;
#pragma unroll
            for (int i = 0; i < 18; i++) {
                uint32_t f = (uint32_t)s + t[i];
                if (f >= (uint32_t)M) f -= (uint32_t)M;   // (s, t < M < 2^31: no wrap of the 32-bit sum)
                const bool hit = (nzx[f >> 5] >> (f & 31)) & 1u;
                if (i < LOB_N_ACTIONS) any1 |= hit; else any2 |= hit;
            }
            out1 |= any1 ? 1u << k : 0u;
            out2 |= any2 ? 1u << k : 0u;
        }
        if (out1 & ~nzd[w]) nzd[w] |= out1;   // (this thread owns the words)
        if (out2 & ~nzd[words + w]) nzd[words + w] |= out2;
    }
}
#endif
// rebuild the bitmap after lob_theta_set: bit = (theta != +0.0 bitwise)
#if LOB_IN_MAIN
__global__ void rebuild_nz_kernel(const f64* __restrict__ theta, uint32_t* nz, i32* nz_epoch, i64 M) {
    i64 wi = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi == 0) atomicAdd(nz_epoch, 1);
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 words = (i64)LOB_NZ_NWORDS(M);
    for (; wi < words; wi += stride) {
        uint32_t m = 0;
        for (int k = 0; k < (32 << LOB_NZ_SHIFT); k++) {
            const i64 i = (wi << (LOB_NZ_SHIFT + 5)) + k;
            if (i < M && __double_as_longlong(theta[i]) != 0) m |= LOB_NZ_BIT(i);
        }
        nz[wi] = m;
    }
}
#endif

// ---- parity dump -------------------------------------------------------------
#if LOB_IN_ENV
__global__ void dump_kernel(const DevParams* __restrict__ Pp, DevState S, int first, int n, lob_book_dump* out) {
    const DevParams& P = *Pp;  // parameters read through the scalar cache, never copied to scratch
    __shared__ TickLds tick_lds;
    stage_ticks(P, tick_lds);
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int b = first + t;
    EnvCtx c(P, S, b, &tick_lds);
    EnvR e;
    env_load(S, b, e);
    lob_book_dump d;
    memset(&d, 0, sizeof d);
    const BookMeta M = S.meta[b];
    for (int l = 0; l < P.D; l++) {
        d.ask_px[l] = rec_price(c, e.rec_cur, 0, l);  d.bid_px[l] = rec_price(c, e.rec_cur, 1, l);
        d.ask_last_px[l] = rec_price(c, e.rec_last, 0, l);  d.bid_last_px[l] = rec_price(c, e.rec_last, 1, l);
        d.ask_vol[l] = d.ask_px[l] != 0.0 ? book_volume(c, e.rec_cur, 0, d.ask_px[l]) : 0;
        d.bid_vol[l] = d.bid_px[l] != 0.0 ? book_volume(c, e.rec_cur, 1, d.bid_px[l]) : 0;
        d.ask_last_vol[l] = d.ask_last_px[l] != 0.0 ? book_volume(c, e.rec_last, 0, d.ask_last_px[l]) : 0;
        d.bid_last_vol[l] = d.bid_last_px[l] != 0.0 ? book_volume(c, e.rec_last, 1, d.bid_last_px[l]) : 0;
    }
    if (e.k > 0) {
        const Track t = c.track(e.k - 1);
        d.ask_total_volume = t.a_tv; d.bid_total_volume = t.b_tv;
        d.spread_mean = t.spread_mean; d.target_price = t.tp_val;
    }
    if (e.done == 2) {  // out of data: where the pre-pass stopped (incl. the rows an abandoned event still applied)
        d.ask_total_volume = S.prep[b].a_tv; d.bid_total_volume = S.prep[b].b_tv;
    }
    {   // last_total_volume_ = total before the last ApplyChanges row (book.cpp:71)
        // the last applied row is the current snapshot, except after an abandoned
        // (out-of-data) event that only stashed: then it is the stashed one
        const bool swapped = e.done == 2 && M.ex_cur < M.ex_first;
        i64 sa = 0, sb = 0;
        for (int l = 0; l < P.D; l++) {
            sa += swapped ? d.ask_last_vol[l] : d.ask_vol[l];
            sb += swapped ? d.bid_last_vol[l] : d.bid_vol[l];
        }
        d.ask_last_total_volume = d.ask_total_volume - sa;
        d.bid_last_total_volume = d.bid_total_volume - sb;
    }
    d.ask_n_transacted = e.a_ntr; d.bid_n_transacted = e.b_ntr;
    d.ask_has_order = e.a_on; d.bid_has_order = e.b_on;
    if (e.a_on) {
        d.ask_order_px = e.a_opx;
        i64 r = e.a_osz - e.a_oex; d.ask_order_rem = r > 0 ? r : 0;
        d.ask_q_head = e.a_oqh; d.ask_q_tail = e.a_oqt;
    }
    if (e.b_on) {
        d.bid_order_px = e.b_opx;
        i64 r = e.b_osz - e.b_oex; d.bid_order_rem = r > 0 ? r : 0;
        d.bid_q_head = e.b_oqh; d.bid_q_tail = e.b_oqt;
    }
    d.position = e.position;
    d.ask_quote = e.ask_quote; d.bid_quote = e.bid_quote;
    d.ask_level = e.ask_level; d.bid_level = e.bid_level;
    d.pnl_step = e.pnl_step; d.momentum_pnl_step = e.momentum_pnl_step;
    d.lo_vol_step = e.lo_vol_step; d.last_action = e.last_action;
    d.episode_reward = e.ep_reward; d.episode_pnl = e.ep_pnl; d.episode_bandh = e.ep_bandh;
    d.time_ms = e.time_ms;
    d.cursor = (e.k > 0 ? c.track(e.k - 1).rec_last : M.rec_cur0) + 1 + (e.done == 2 ? (int)M.ex_records : 0);
    d.terminal = e.done == 2 ? 2 : (is_open(P, e.time_ms) ? 0 : 1);
    d.total_ticks = e.total_ticks;
    d.market_buys = e.market_buys; d.market_sells = e.market_sells;
    d.ticks_with_ask = (i32)(uint32_t)e.tick_ab; d.ticks_with_bid = (i32)(uint32_t)(e.tick_ab >> 32); d.ticks_with_both = e.tick_both;
    d.ask_transactions = (i32)(uint32_t)e.ntr_snap; d.bid_transactions = (i32)(uint32_t)(e.ntr_snap >> 32);
    d.ticks_long = (i32)(uint32_t)e.tick_pos; d.ticks_short = (i32)(uint32_t)(e.tick_pos >> 32); d.ticks_with_position = d.ticks_long + d.ticks_short;
    int n_tr = 0;
    {
        const int ng = S.hdr[b].tr_n, head = S.hdr[b].tr_head;
        for (int k = 0; k < ng; k++) {
            const int slot = (head - k + P.trace_gens) & (P.trace_gens - 1);
            n_tr += __popc(S.tr_alive[(size_t)b * P.trace_gens + slot]);
        }
    }
    d.n_traces = n_tr;
    out[t] = d;
}
#endif

// Diagnostics of the fast path (lob_debug_fastpath; not on the step): how many weights the exact written-weights map shows,
// and the distribution of the live books' hit-list lengths.  out: [0] written weights, [1] live books, [2] live books without a
// list, [3] sum of the lengths, [4 + n] books whose list has n entries (n = LOB_FP_BINS - 1: that many or more).
#define LOB_FP_BINS 257
#if LOB_IN_MAIN
__global__ void fastpath_stats_kernel(DevState S, i64 M, int have_lists, i64* out) {
    __shared__ i32 hist[LOB_FP_BINS];
    __shared__ i32 s_none, s_live;
    __shared__ i64 s_sum, s_pop;
    for (int i = threadIdx.x; i < LOB_FP_BINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) { s_none = 0; s_live = 0; s_sum = 0; s_pop = 0; }
    __syncthreads();
    const i64 words = M / 32 + 1;
    i64 pop = 0;
    if (S.theta_nzx)
        for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (i64)gridDim.x * blockDim.x) pop += __popc(S.theta_nzx[w]);
    if (pop) atomicAdd((u64*)&s_pop, (u64)pop);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < S.B; b += gridDim.x * blockDim.x) {
        if (S.hdr[b].done) continue;
        atomicAdd(&s_live, 1);
        if (!have_lists) continue;
        const u64 n = S.hl_rec[(size_t)b * LOB_HL_REC];
        if (n == ~0ull) { atomicAdd(&s_none, 1); continue; }
        atomicAdd(&hist[n < LOB_FP_BINS - 1 ? (int)n : LOB_FP_BINS - 1], 1);
        atomicAdd((u64*)&s_sum, n);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LOB_FP_BINS; i += blockDim.x) if (hist[i]) atomicAdd((u64*)&out[4 + i], (u64)hist[i]);
    if (threadIdx.x == 0) {
        atomicAdd((u64*)&out[0], (u64)s_pop); atomicAdd((u64*)&out[1], (u64)s_live);
        atomicAdd((u64*)&out[2], (u64)s_none); atomicAdd((u64*)&out[3], (u64)s_sum);
    }
}
#endif

#endif
