// Device-resident state of the batched engine: struct-of-arrays over books.
//
// Layout (DESIGN.md "Data layout in HBM"):
//   * every per-book scalar is its own array  field[B]  -> lane b of the
//     lane-per-book environment kernel reads element b: fully coalesced;
//   * book levels are ping-pong  px[sel][side][level][B]  so that
//     Book::StashState (reference src/market/book.cpp:51-55) is a parity flip,
//     never a copy;
//   * rolling windows are rings  ring[slot][B];
//   * data consumed by the wave-per-book learner kernels (state variables,
//     trace lists) is book-major so that one wave reads one contiguous block.
#ifndef LOB_STATE_H
#define LOB_STATE_H

#include <stdint.h>

#include "../../include/lob_engine.h"

typedef long long i64;
typedef unsigned long long u64;
typedef int i32;
typedef double f64;
typedef float f32;

// A pointer READ FROM the device-resident DevState (kernels that take the state by pointer) is a generic pointer to the compiler:
// its loads and stores become flat_ instructions, which count as LDS traffic too and wait for it.  The round trip through the
// global address space tells InferAddressSpaces what the pointer is (a kernel ARGUMENT is promoted by itself).
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T* lob_g(T* p) {
    // (through an integer: a pointer-to-pointer round trip is folded away before the address spaces are inferred)
    return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p;
}
#else
template <class T> inline T* lob_g(T* p) { return p; }
#endif

// Pointer members of the device state.  On the device every use goes through lob_g (above): whether the structure arrived as a
// kernel argument or is read from its device-resident copy, the memory instructions are global_, never flat_.  One pointer wide,
// trivially copyable: the layout is that of the plain pointers it replaces.
template <class T> struct GP {
    T* p;
    __host__ __device__ __forceinline__ operator T*() const { return lob_g(p); }
    __host__ __device__ __forceinline__ T* operator->() const { return lob_g(p); }
    __host__ __device__ __forceinline__ GP& operator=(T* q) { p = q; return *this; }
};

// name, window-size parameter
#define LOB_ROLLING_MEANS(X) \
    X(f_midprice)            \
    X(f_volatility)          \
    X(f_ask_tx)              \
    X(f_bid_tx)              \
    X(spread_window)         \
    X(pnl_ups)               \
    X(pnl_downs)             \
    X(tp_mp)
#define LOB_ACCUMULATORS(X) \
    X(f_vwap_numer)         \
    X(f_vwap_denom)

// Per-book scalars of the AGENT-DEPENDENT part of the environment (orders,
// inventory, PnL, cursors).  Everything that depends only on the event stream
// (snapshots, cumulative volumes, windows, target price, five of the eight
// state variables, the step boundaries themselves) is precomputed once per
// episode into the market track below.
#define LOB_ENV_FIELDS(X)                                                                                     \
    X(i32, done)      /* 0 live, 1 isTerminal(), 2 out of data */                                              \
    X(i32, k)         /* number of completed NextState events = index of the next track entry */              \
    X(i32, time_ms)   /* Market::time_ */                                                                      \
    X(i32, rec_cur)   /* record holding the current depth snapshot (-1: none) */                              \
    X(i32, rec_last)  /* record holding the stashed snapshot (Book::StashState) (-1: none) */                  \
    X(i32, pf)        /* record up to which the trades have been handed over (= rec_first of event k - 1) */   \
    X(f64, mid)       /* midprice of the current snapshot */                                                   \
    X(f64, mid_prev)  /* midprice of the stashed snapshot */                                                   \
    X(i64, position)  /* RiskManager::position_ */                                                             \
    X(i32, last_action)                                                                                        \
    X(i32, lo_vol_step)                                                                                        \
    X(f64, pnl_step)                                                                                           \
    X(f64, momentum_pnl_step)                                                                                  \
    X(f64, ask_quote)                                                                                          \
    X(f64, bid_quote)                                                                                          \
    X(i32, ask_level)                                                                                          \
    X(i32, bid_level)                                                                                          \
    X(f64, ep_reward)                                                                                          \
    X(f64, ep_pnl)                                                                                             \
    X(f64, ep_bandh)                                                                                           \
    X(i32, total_ticks)                                                                                        \
    X(i32, market_buys)                                                                                        \
    X(i32, market_sells)                                                                                       \
    /* TickStatistics (Base::UpdateStats, base.cpp:412-442): plain 32-bit counters like the reference's ints, two per word --    \
     * with_ask | with_bid << 32 and long << 0 | short << 32 (with_position = long + short) -- and with_both in the 4 bytes the   \
     * three ints above leave before the next 8-byte field (a step consumes at least one event, n_events < 2^31) */              \
    X(i32, tick_both) X(i64, tick_ab) X(i64, tick_pos)                                                                         \
    /* TradeStatistics::ask_transactions | bid_transactions << 32: a_ntr / b_ntr as Base::UpdateStats copies them at decision     \
     * time (base.cpp:415-416), before the step's events */                                                                   \
    X(i64, ntr_snap)                                                                                                          \
    X(i64, events)    /* depth records consumed (incl. warm-up) */                                             \
    /* per side: n_transacted_ + the single live order (quirk Q13); otk = ToTicks(order price) */             \
    X(i32, a_ntr) X(i32, a_on) X(f64, a_opx) X(i64, a_osz) X(i64, a_oqh) X(i64, a_oqt) X(i64, a_oex) X(i64, a_oiq) X(i32, a_otk) \
    X(i32, b_ntr) X(i32, b_on) X(f64, b_opx) X(i64, b_osz) X(i64, b_oqh) X(i64, b_oqt) X(i64, b_oex) X(i64, b_oiq) X(i32, b_otk)

// One entry per NextState event of a book: the agent-independent outcome of
// Intraday::NextState (intraday.cpp:225-272) when started at record rec_first.  128 bytes; the first 64
// (TrackHead64) are all an event pass of env_kernel reads: which rows the event applies, time, touch, midprice
// and -- since they do not depend on the agent either -- the event's merged trade list (TimeAndSales::LoadUntil,
// at most two price levels: streams with more trade slots per record keep reading the records' own slots) and
// the best prices of the snapshot the event leaves behind (adverse selection); the rest feeds the quotes and the
// state extraction after the step.
struct __attribute__((aligned(16))) Track {
    i32 rec_first;  // record whose trade slots this event consumes (cursor at NextState entry)
    i32 rec_last;   // last depth record applied (the new current snapshot); rows rec_first..rec_last were applied
    i32 time_ms;
    i32 tick_ap0;   // ToTicks(best ask), ToTicks(best bid) of the new snapshot
    i32 tick_bp0;
    i32 info;       // bits 0-1: entries of the merged trade list held below; bit 2 (LOB_TRK_TRADES_OK): that is the whole list
    f64 mid;          // midprice of the new snapshot
    f32 tr_px[2];     // merged trades of the event, ascending price key (the order of the reference's std::map)
    f32 bap, bbp;     // best ask / bid price of the new snapshot
    i64 tr_vol[2];
    f64 tp_val;       // TargetPrice::get() after the update
    f64 spread_mean;  // spread_window.mean()
    i64 a_tv, b_tv;   // cumulative total_volume_ (quirk Q1)
    f32 mv[8];        // spd, mpm, imb, svl, vol, rsi, vwap (Intraday::getVariable), [7] unused
};
#define LOB_TRK_TRADES_OK 4
// The first 32 bytes of a Track entry: what the general event pass (next_state) reads of it.
struct __attribute__((aligned(16))) TrackHead {
    i32 rec_first, rec_last, time_ms, tick_ap0;
    i32 tick_bp0, info;
    f64 mid;
};
// The first 64: the fast pass (pass_fast, lob_env.h).
struct __attribute__((aligned(16))) TrackHead64 {
    i32 rec_first, rec_last, time_ms, tick_ap0;
    i32 tick_bp0, info;
    f64 mid;
    f32 tr_px[2];
    f32 bap, bbp;
    i64 tr_vol[2];
};
static_assert(sizeof(TrackHead) == 32 && sizeof(TrackHead64) == 64 && sizeof(Track) == 128, "TrackHead / TrackHead64 are prefixes of Track");
#define LOB_MV_SPD 0
#define LOB_MV_MPM 1
#define LOB_MV_IMB 2
#define LOB_MV_SVL 3
#define LOB_MV_VOL 4
#define LOB_MV_RSI 5
#define LOB_MV_VWAP 6

// Per-book summary of the market pre-pass.
struct BookMeta {
    i32 n_track;    // events whose Track entry exists so far (all of them once `complete`)
    i32 k_warm;     // events consumed by Intraday::Initialise (windows full)
    i32 init_ok;    // 0: ran out of data during Initialise
    i32 rec_cur0, rec_last0, time0;  // state after the "skip to market open" phase
    i32 ex_first;   // record whose trades the abandoned (out-of-data) event still matches
    i32 ex_cur, ex_last, ex_time;    // snapshot records / time after that abandoned event
    i64 ex_records; // depth records consumed by the abandoned event
    f64 mid0, mid_prev0;
    i32 complete;   // the pre-pass has reached the end of the stream (else it continues from `prep`: prepass_extend_kernel)
    i32 _pad;
};

// Where the (resumable) pre-pass of a book stands: the agent-independent registers of
// Intraday::NextState between two events.  The windows themselves are in DevState (rings in HBM).
struct PrepState {
    i32 cursor, time_ms, rec_cur, rec_last;
    f64 ap0, bp0, lap0, lbp0;   // best prices of the current / stashed snapshot (0 = undefined)
    i64 a_tv, b_tv;             // cumulative total_volume_ (quirk Q1)
    f64 ewma_up, ewma_down, tp_val;
    i64 records;
    i32 k;            // events produced so far
    i32 prev_first;   // record up to which the trades have been handed over
};

// Per-book learner header (Runner / Agent / Traces scalars), one 64-byte
// record per book: the wave-per-book kernels fetch it with a single scalar
// load (s_load_dwordx16) instead of a chain of dependent per-field loads.
struct __attribute__((aligned(64))) LHdr {
    i32 done;       // copy of the env's done flag (0 live, 1 isTerminal(), 2 out of data)
    i32 time_ms;    // copy of Market::time_ (isTerminal() test of Learner::_step)
    i32 slot_cur;   // which of the two rl::State objects is `state`
    i32 zero_mask;  // bit s: State object s still holds its constructor zeros
    i32 action;     // action chosen in the current step
    i32 stepped;    // act: an action is pending; env: performAction succeeded
    i32 tr_head;    // ring slot of the newest trace generation
    i32 tr_n;       // live generations
    u64 rng_ctr;    // draws consumed from this book's policy stream
    f64 reward;     // getReward() after performAction
    f64 td;         // last TD error
    f64 upd;        // alpha * delta to scatter
};

#define LOB_CBD_CAP 16384        /* dense slot ids: 128 KB of doubles in LDS */
#define LOB_ACD_MAX_BLOCKS 256   /* accumulate_dense_kernel: one block per CU */
#define LOB_ACD_MARK 0x7ff4000000000001ull /* a signalling NaN: no sum of terms is this bit pattern (arithmetic only produces quiet NaNs) */
#define LOB_ML_BLOCKS 256
#define LOB_ML_ROWS 8192
#define LOB_PERSIST_N 32
#define LOB_PROF_N 32
#define LOB_MK_REC 10         /* doubles per memo record: S0 of the nine actions + the theta version it was computed under */
#define LOB_MK_PROBES 16
#define LOB_HL_CAP 23         /* additions of a hit list that the kernels keep in registers / LDS rows (the mean is 8, the 99th percentile 17, once
                               * the written set has saturated at 160 k weights of 20 M) */
#define LOB_HL_ROW 24         /* u64 per LDS row of the wave-per-book kernels: the count + LOB_HL_CAP additions */
#define LOB_HL_MAX 35         /* additions per book a record can hold: the two-lanes-per-book learn kernel writes up to that many, and the
                               * act side replays the ones beyond LOB_HL_CAP in a pass of their own -- in a long run 8-11 books per step
                               * had 24 and more, and each of them cost a whole-wave evaluation in learn_q_rest_kernel AND in the env kernel
                               * (16-40 us of a 250-300 us step: NOTES.md "Round 5") */
#define LOB_HL_REC 36         /* u64 per book: [0] the count, then the additions: 288 bytes */
#define LOB_VD_STRIDE 72      /* u16 per book: 64 verdicts + epoch lo/hi + slot + valid, padded to 144 B */
/* theta's "ever written" map: one bit per LOB_NZ_GRAN = 8 consecutive weights (312 KB at M = 20M, so
 * it stays L2-resident under the streaming traffic; one bit per weight, 2.5 MB, did not).  A set
 * bit only means "fetch the weight"; an unwritten neighbour then reads as exactly +0.0. */
#define LOB_NZ_SHIFT 3
#define LOB_NZ_WORD(i) ((i) >> (LOB_NZ_SHIFT + 5))
#define LOB_NZ_BIT(i) (1u << (((i) >> LOB_NZ_SHIFT) & 31))
#define LOB_NZ_NWORDS(M) ((((size_t)(M) >> LOB_NZ_SHIFT) >> 5) + 1)
#define LOB_NZ_WORDS 256      /* per parity: [0] count of newly written weights, [128..255] their 4096-bit filter */
#define LOB_NZ_FILTER 128     /* filter words (bit = map bit index mod 4096) */
#define LOB_NZ_NEW_MAX 256    /* above this many new weights the filter is too dense to help: full look-ups */

struct RMPtrs {  // RollingMean<double>
    GP<f64> ring;   // [w][B]
    GP<i32> cnt;
    GP<i32> head;
    GP<f64> sum;
    GP<f64> mean;
    GP<f64> s;
    i32 w;
};
struct AccPtrs {  // Accumulator<double>
    GP<f64> ring;
    GP<i32> cnt;
    GP<i32> head;
    GP<f64> sum;
    i32 w;
};

struct DevState {
    i32 B;
    i32 D, T, W;       // depth, trade slots, record words
    i32 n_events;
    GP<const uint32_t> records;  // device layout: [B][n_events][Wd], or one replayed stream [n_total][Wd] when rec_phase is set
    GP<const i64> rec_phase;     // [B] first record of each book's n_events-long window (lob_load_events_shared), else null

#define X(t, n) GP<t> n;
    LOB_ENV_FIELDS(X)
#undef X
#define X(n) RMPtrs n;
    LOB_ROLLING_MEANS(X)
#undef X
#define X(n) AccPtrs n;
    LOB_ACCUMULATORS(X)
#undef X

    // Market track: one entry per NextState event, written by the pre-pass.  A stream of up to
    // `track_len` events has its whole track resident (index = k); a longer one (a recorded day of
    // tens of thousands of events) keeps a ring of the `track_len` (a power of two) latest entries per
    // book, refilled every few steps (prepass_extend_kernel): index = k & track_mask.
    GP<Track> track;      // [B][track_len]
    i32 track_len, track_mask;
    GP<BookMeta> meta;    // [B]
    GP<PrepState> prep;   // [B]
    GP<f64> ewma_up;      // [B] return_ups / return_downs EWMA means (persist across episodes)
    GP<f64> ewma_down;
    GP<f64> tp_val;       // [B] TargetPrice::val_ (persists)
    GP<f64> persist;      // [LOB_PERSIST_N][B] window sums at episode start (exact replay of quirk Q7)
    GP<i32> k_stop;       // [B] events consumed by the previous episode (for the replay)

    GP<LHdr> hdr;      // [B]
    GP<f32> vars;      // [B][3][16]  the two rl::State objects' state_vars + the latest getState()
    GP<f64> qs_last;   // [B][9] Q(last_state, .) of the current step
    GP<i32> tr_idx;    // [B][P.trace_gens][32]
    GP<uint32_t> tr_alive;  // [B][P.trace_gens]

    GP<f64> theta;       // [M] or [B][M]
    GP<f64> theta_b;        // DoubleAgent::theta_b (LOB_ALGO_DOUBLE_Q), same shape as theta, or null
    GP<uint32_t> theta_b_nz;
    GP<f64> qs_last_b;      // [B][9] Qb(last_state, .)
    GP<u64> mt_state;       // [B][312] std::mt19937_64 state of each book's Agent::gen (DoubleQLearn coin)
    GP<i32> mt_idx;         // [B]
    // Combined update (shared theta; DESIGN.md "generation combining"): many books hold the SAME trace
    // generation -- the live tiles one (group-0 state, action) left behind -- so their updates are
    // summed per distinct (generation identity, alive mask) first and applied to theta once.
    GP<i32> tr_sig;         // [B][trace_gens][4]: q0, q1, q2 (= (int)floor(32 x) of the three group-0 variables), action | zero << 8
    GP<i32> tr_cbslot;      // [B][trace_gens] slot of cb_key the generation's claim of THIS step ended on (-1: none), cb_claim_finish
    GP<i32> tr_mslot;       // [B][trace_gens] SARSA lane path (lob_fast.h trace_sarsa_kernel): memo slot of the generation's triple |
                         //   (episode epoch & 0x7fff) << 16, or -1: not known / slot's tiles not registered (DevParams::sarsa_lanes)
    GP<u64> cb_key;         // [cb_slots] 64-bit hash of (signature, mask), ~0 = empty
    GP<i32> cb_ident;       // [cb_slots][8]: q0, q1, q2, code, mask (0: free slot), the generation that claimed it (book * trace_gens + slot), -, -
    GP<f64> cb_acc;         // [cb_slots][2]: summed update for theta / theta_b
    GP<uint32_t> cb_touch;  // [cb_slots] bit 0: theta gets an update this step, bit 1: theta_b
    GP<i32> cb_list;        // [2 parities][cb_segs][cb_slots / cb_segs] the occupied slots, slot s in segment s & (cb_segs - 1): survivors of
                         //   the previous step (apply_kernel, one block per segment: no global counter), then this step's claims
    GP<i32> cb_count;       // [2][cb_segs]
    i32 cb_slots;        // power of two
    i32 cb_reps;         // copies of cb_acc ([cb_reps][cb_slots][2]): 1, or 8 = one per XCD (accumulate_kernel)
    i32 cb_segs;         // power of two <= cb_slots / 4: apply_kernel's grid
    i32 cb_par;          // parity of the current learner step (set by the host before the step's launches)
    // Dense ids of the occupied slots (accumulate_dense_kernel, lob_kernels.h): SARSA(lambda) -- and Q(lambda) once most actions are
    // greedy -- adds 1.6 M terms per step to ~10 k slots.  A slot that is claimed while `cb_dense_on` gets an id < LOB_CBD_CAP from
    // one of eight free lists (one per XCD: a single list's counter would queue every claim of the chip behind it); a block of
    // the kernel then sums its books' terms in a direct-indexed LDS array of LOB_CBD_CAP doubles -- no hash table, no
    // compare-and-swap -- and writes the array out as its row of `cb_part`; apply_kernel adds the rows up.  A slot without an id
    // (lists empty, claimed before the mode was on) takes the atomics on cb_acc as before.
    GP<i32> cb_dense;       // [cb_slots] id of the slot, -1: none
    GP<i32> cb_free;        // [8][LOB_CBD_CAP / 8] stacks of free ids: list x holds the ids = x mod 8, lowest on top at the start
    GP<i32> cb_free_n;      // [8][2]: entries on the stack, and the fewest it has ever held (ids >= 8 * (cap / 8 - that) were never out)
    GP<u64> tr_cbd;         // [B][trace_gens] slot << 32 | id (0xffffffff: the slot has none) as the kernel found them when it last
                         //   compared the generation's identity with the slot's; valid while tr_cbslot still names that slot, verified
    GP<f64> cb_part;        // [LOB_ACD_MAX_BLOCKS][LOB_CBD_CAP] the blocks' sums by id (LOB_ACD_MARK: no term this step)
    GP<f64> cb_red;         // [LOB_ACD_GROUPS = 8][LOB_CBD_CAP] ... added up per group of blocks (reduce_dense_kernel), for apply_kernel
    i32 cb_ids;          // ids in all (LOB_CBD_CAP; fewer with LOB_CBD_IDS, for the tests)
    i32 cb_dense_on;     // claims take ids (set by the host with the algorithm / epsilon: lob_engine.hip acc_blocked)
    // Verdict carry-over: learn(t) saves, per book, which group-1/2 tiles of s' hit a written
    // weight (9 bits per tiling); act(t+1) evaluates the same state and reuses them instead of 576
    // bitmap look-ups, OR-ed with a small filter of the bits the update in between newly set.
    GP<uint16_t> verdict;   // [B][LOB_VD_STRIDE] u16: [2 groups][32 tilings], then epoch (2 x u16), slot, valid
    GP<uint16_t> verdict_b; // [B][64]: the same for theta_b (double Q; shares the tag of `verdict`)
    GP<i32> nz_new;         // [2 targets: theta, theta_b][2 parities][LOB_NZ_WORDS]: count + filter of the map bits an update set for the first time
    GP<i32> nz_epoch;       // [1] bumped whenever theta / the bitmap change outside update_kernel
    GP<uint32_t> theta_nz;  // bitmap, bit i set once theta[i] has ever been written: clear bit => theta[i] == +0.0
    // Group-0 memo (shared theta; DESIGN.md "group-0 memo"): the first 32 of Q's 128 ordered terms --
    // the group-0 partial sum S0(a) = sum_j w0 * theta[tile(q0, q1, q2, a, j)] -- depend on the state
    // only through the three quantised group-0 variables (inventory, quote distances: a few hundred
    // distinct triples among 65 536 books), and the ordered sum STARTS with them.  So S0 is
    // evaluated once per distinct triple and theta version (memo_kernel) and every book continues
    // the sum from its triple's S0 with whatever group-1/2 weights are non-zero.
    GP<u64> mk_hash;        // [mk_slots] 64-bit hash of the triple, ~0 = empty (claimed by env_kernel)
    GP<i32> mk_ident;       // [mk_slots][4]: q0, q1, q2 (written by the claim winner, compared in full by the readers), and whether two of
                         //   the triple's 288 group-0 tiles coincide: 0 not known yet, 1 no, 2 yes (learnt by the trace kernel)
    GP<i32> mk_stamp;       // [mk_slots] step id of the last claim: first toucher of a step appends the slot to the list
    GP<i32> mk_list;        // [2 parities][mk_slots] slots in use this step
    GP<i32> mk_count;       // [2]
    GP<f64> mk_rec;         // [2: theta_t for learn, theta_{t+1} for the next act][mk_slots][LOB_MK_REC]: S0[9], theta version tag
    GP<f64> mk_rec_b;       // the same under theta_b (double Q on the fast path: both vectors share the triples, the tiles and the maps), or null
    GP<i32> mk_tiles;       // [mk_slots][9][32] the triple's 288 group-0 tile indices (action, tiling), written by memo_kernel the first time
                         //   the slot is on a step's list: the lane-per-book trace kernel copies a generation from here
    GP<i32> mk_tiles_ok;    // [mk_slots] bit 0: mk_tiles[slot] is filled; bit 1: ... and every tile is in the registry (ow_tab)
    GP<uint32_t> mk_marked; // [mk_slots] bit a: the 32 tiles of (triple, action a) are marked in the written-weights maps
    GP<i32> mk_marklist;    // [mk_slots] (slot * 16 + action) pairs whose tiles act_light_kernel wants marked: memo_kernel (which 0) does it,
    GP<i32> mk_markcount;   // [1]          a wave per pair, before the learn kernel looks; reset by memo_kernel (which 1)
    // Tile registry of the SARSA lane path: which weight indices are shared by two DIFFERENT group-0 tiles of the episode's
    // memo slots (a collision of the hash -- or of its 2048-entry table: coordinates 2048 apart, permuted terms).  Two tiles
    // are the same tile when tiling, action and the three coordinates & 2047 agree; every other pair on one index is
    // "ambiguous".  A generation loses a tile to a new state either because the two triples fall in the same tile of that tiling
    // (pure arithmetic on the quantised coordinates) or through an ambiguous index -- and only the latter needs the indices.
    GP<u64> ow_tab;         // [ow_slots] index << 32 | first registrant (slot * 288 + action * 32 + tiling), ~0 = empty
    i32 ow_slots;        // power of two
    GP<uint32_t> amb_bits;  // [M / 32 + 1] bit f: index f is ambiguous
    GP<i32> amb_new;        // [2 parities][amb_cap] indices that became ambiguous in this step's memo_kernel (which 0) ...
    GP<i32> amb_new_n;      // [2]   ... its launch which 1 marks them in mk_amb of every registered slot
    i32 amb_cap;
    GP<i32> amb_flag;       // [1] sticky until the next reset: amb_new overflowed (the lane path is off)
    GP<uint32_t> mk_amb;    // [mk_slots][9] bit j: tile (slot, action, tiling j) lies on an ambiguous index
    GP<i32> mk_all;         // [mk_slots] registered slots, in registration order
    GP<i32> mk_all_n;       // [1]
    GP<i32> mk_slot;        // [B] slot of the book's latest state (-1: none)
    GP<i32> mk_slot_last;   // [B] slot of the state before that (the learner's last_state in the next step)
    i32 mk_slots;        // power of two
    // Fast learner path (lob_fast.h): exact "ever written" map of the shared theta (one bit per weight) and
    // its coarse image (one bit per 2^cshift weights) that every CU keeps in LDS; work lists of the books
    // the fast kernels hand back to the general ones.
    GP<uint32_t> theta_nzx; // [M / 32 + 1]
    GP<uint32_t> theta_nzc; // [cwords4 * 4]
    // The exact map folded over the actions (learn_q_pair_kernel): the nine tiles of one tiling of group 1 / 2 are
    // (s + term[g][a]) mod M for ONE hash sum s, so bit s of group g's map = OR over a of theta_nzx[(s + term[g][a]) mod M]
    // answers "does any of this tiling's nine tiles lie on a written weight" with one look-up instead of nine (93 % of the
    // tilings: none, at 160 k written weights of 20 M).  Set with the exact bit, wherever that is set (nzd_mark).
    GP<uint32_t> theta_nzd;      // [2: tile group 1, 2][M / 32 + 1]
    GP<const uint32_t> nzd_terms; // [18]: term[1][0..8], term[2][0..8] (the engine's hash table + 2048 + 9)
    GP<i32> tr_list;        // [B] books the lane-per-book trace kernel leaves to the wave-per-book one
    GP<i32> tr_list_n;      // [2 parities]
    GP<i32> tr_list2;       // [B] Q(lambda): the entries of `tr_list` the lane-per-generation kernel (trace_lane_kernel) hands on to the wave-per-book one
    GP<i32> tr_list2_n;     // [2 parities]
    // The accumulation of a generation's update where its slot is resolved (learn_q_pair_kernel for the books whose step leaves one
    // new generation, trace_lane_kernel for the others): what they cannot finish goes on this list for accumulate_kernel --
    // entry = book (every generation of the book: its TD error was not known yet, or its trace step was handed on), or
    // book | 1 << 31 (only the book's generations without a slot: the direct, tile-by-tile path must wait until nobody reads theta).
    GP<i32> acc_list;       // [B]
    GP<i32> acc_list_n;     // [2 parities]
    GP<uint8_t> acc_pend;   // [B] bit 0: the learn kernel handed the book back (its TD error comes later): trace_lane_kernel must not add its update yet;
                         //     bit 1 (trace_rest_kernel's flow): trace_lane_kernel handed the book's trace step on
    // trace_rest_kernel's generations without a slot (book x trace_gens + ring slot), applied tile by tile by apply_kernel: other
    // waves of trace_rest_kernel read theta (learn_q_book) while it adds generations up
    GP<i32> dir_list;       // [B x trace_gens] (a generation is listed at most once per step)
    GP<i32> dir_list_n;     // [2 parities of the combine table's lists: apply_kernel empties the one it consumed a step ago]
    GP<i32> slow_list;      // [2 kinds: act, learn][B]
    GP<i32> slow_n;         // [2 parities][2 kinds]
    // Hit-list carry-over learn_q(t) -> act(t+1) (lob_fast.h act_light_kernel): the Q evaluation of the TD target and the
    // next step's action selection are over the SAME State, and between them only the weights change, not WHICH
    // group-1/2 tiles fall on a written weight -- the trace kernel marks a new generation's tiles in the maps when it
    // creates the generation, before learn_q looks.  So learn_q leaves, per book, the ordered list of additions
    // Agent::getQ makes beyond the memoised group-0 sum: entry = tile index | action << 32 | (weight w2 ? 1 : 0) << 36.
    // What a learn kernel READS that a trace kernel of the SAME step WRITES -- why SARSA(lambda)'s trace kernels run in front of its
    // learn kernels and may not run beside them (round 5 tried: not bit-clean, the reader "not found"; found in round 6):
    //   1. LHdr::td       trace_lane_kernel<SARSA> / trace_fast_kernel<SARSA, .> leave Q(s, a) there for the TD error (avoidable: qs_last);
    //   2. the written-weights maps theta_nzx / theta_nzd / theta_nzc (and mk_marked): the trace step MARKS the new generation's 32
    //      group-0 tiles (nzx_mark) -- the weights the update of this very step is going to write.  The learn kernel decides by
    //      those maps which group-1/2 tiles of s' go on the hit list below.  A tile of s' that hashes onto one of the new
    //      generation's weights is still exactly 0.0 when the learn kernel sums Q(s', .), so its TD error is right either way --
    //      but a list built BEFORE the mark lacks the entry, and the next step's replay (after the update has made the weight
    //      non-zero) drops that addition: the 2e-4 TD errors of three books in 16 384 that round 5 saw one step later.
    //      Watkins's flow, where the lane trace kernel does run after the learn kernel, marks through nzx_mark_late, which
    //      records the step in hl_dirty and so voids every list; SARSA(lambda)'s order makes the marks early instead.
    //   3. nothing else: tr_head / tr_n / tr_idx / tr_alive / tr_sig / tr_mslot / tr_cbslot are read by the update kernels only.
    GP<u64> hl_rec;         // [B][LOB_HL_REC]: [0] = number of entries, or ~0: no list (not evaluated by the fast learn kernel, or more
                         //   than it can record: LOB_HL_MAX, LOB_HL_CAP from the kernels with one lane or one wave per book); [1 + i] = entry i
    GP<i32> hl_dirty;       // [1] step id of the last update that set a map bit AFTER learn_q had looked (voids every list)
    // R-learning agents (RLearn / OnlineRLearn, src/rl/agent.cpp:357-412): the average reward rho of each agent -- one per weight
    // vector: [1] shared, [B] private --, the sum of a step's increments (folded in by rho_fold_kernel: every book reads rho_t),
    // and per book the bootstrap value the TD error used (maxQ(to_state) / Q(to_state, a')), which the rho update needs again
    GP<f64> rho;
    GP<f64> rho_inc;
    GP<i32> rho_cnt;     // books that contributed to rho_inc this step
    GP<f64> rl_t;        // [B]
    // model_log (Agent::HandleTransition, src/rl/agent.cpp:93-100: _agg_delta += |delta|; every 1000 updates one row _agg_delta /
    // 1000): per learner step the stepped books' |delta| are summed (td_stats_kernel: a partial per block, td_stats_fold_kernel:
    // the partials in order), added to the running aggregate, and once at least 1000 updates are in it a row aggregate / count is
    // written and both start again -- one book: exactly the reference's rows; a batch of 1000 books or more: a row per step, the
    // mean |delta| over the batch.  Off until lob_model_log_enable.
    GP<f64> ml_part;     // [LOB_ML_BLOCKS] a step's partial sums
    GP<i32> ml_npart;    // [LOB_ML_BLOCKS] ... and counts
    GP<f64> ml_agg;      // [1] the running aggregate
    GP<i64> ml_cnt;      // [2] updates in it; rows written since the last lob_model_log_read
    GP<f64> ml_rows;     // [LOB_ML_ROWS]
    GP<f64> theta_sync;  // [M] (multi-GPU) or null
    GP<f64> delta;       // [M] scratch for the all-reduce or null
    GP<i64> counters;    // [LOB_CNT_STRIPES][LOB_CNT_STRIDE] device counters, striped (cnt_add below; the host sums the stripes)
    GP<i64> prof;        // [B][LOB_PROF_N] clock64 per phase of the learner kernels (-DLOB_PROF builds only, tools/exp_prof.py), else null
    GP<i32> error_flag;  // [1] bits: reference-would-throw conditions
    const DevState* self;  // the device-resident copy of this very structure (kernels that take the state by pointer; lob_engine.hip push_state)
};

// Parameters copied to the device once (kernel argument, uniform).
struct DevParams {
    i32 D, T, W, V;
    i32 Wd;          // words per record in the device layout (lob_env.h drec_*)
    i32 vars[LOB_MAX_VARS];
    // tick table
    i32 n_bands;
    f64 band_lb[LOB_MAX_BANDS];
    f64 band_tick[LOB_MAX_BANDS];
    i64 band_cum[LOB_MAX_BANDS];
    f64 band_pp[LOB_MAX_BANDS];  // lobh::TickTable::pp / pt: what ToPrice / ToTicks have accumulated on reaching band i
    i32 band_pt[LOB_MAX_BANDS];
    i64 open_ms, close_ms;
    i32 order_size, reward_measure;
    i64 pos_lb, pos_ub;
    f32 damping_factor, pos_weight, trd_weight, pnl_weight;
    i32 target_price, quote_mode;
    f64 ewma_alpha;
    // learning
    i64 M;
    f64 w0, w1, w2;
    f64 gamma, alpha, epsilon;
    i32 policy;                       // LOB_POLICY_*
    f64 tau;                          // Boltzmann temperature
    f64 beta;                         // R-learning: step size of rho
    i32 r_learn;                      // 1: the agent is the average-reward variant of `algo` (RLearn of Q(lambda), OnlineRLearn of
                                      //   SARSA, DoubleRLearn of double Q): `algo` here is always one of the three base algorithms
    f32 trace_rate;                   // (float)(gamma*lambda)
    f32 trace_pow[LOB_TRACE_GENS + 1];  // eligibility by age, iterated float products
    i32 trace_kmax;                   // first age whose eligibility < tolerance
    i32 trace_gens;                   // ring size in generations: 32 or 64 (power of two >= trace_kmax)
    i32 algo, theta_private;
    i32 combine;         // shared theta: sum the updates per distinct trace generation first (0 with LOB_NO_COMBINE=1)
    i32 carry_verdicts;  // 0 with LOB_NO_CARRY=1 in the environment (A/B switch for the verdict carry-over)
    i32 sarsa_lanes;     // trace step with a lane per generation (trace_lane_kernel: SARSA(lambda), and the books of Q(lambda) that keep their traces) + the tile registry it needs
    i32 epi_epoch;       // episodes begun (lob_reset): tags the memo slots the generations refer to (tr_mslot)
    i32 memo;            // group-0 memo + fast learner kernels on (shared theta, SARSA / Q(lambda), one book group; 0 with LOB_NO_MEMO=1)
    i32 cshift, cwords4; // coarse map: bit = weight index >> cshift; size in 16-byte units
    i32 exp_learn_first; // (-DLOB_EXPERIMENTS builds, LOB_SARSA_LEARN_FIRST=1) SARSA(lambda)'s learn kernels IN FRONT OF its trace kernels: the order
                         //   the written-weights maps forbid (lob_state.h above hl_rec) -- kept so that a test can show that it fails
    u64 seed, book_id_offset;
};

// Kernels take the parameters and the state through their device-resident copies (lob_engine.hip P_dev / DevState::self), not as
// 3 KB of by-value arguments: the fields are loaded where they are used instead of all at the kernel's entry (scalar-register
// spills: learn_q_pair_kernel 145 -> 10, trace_rest_kernel 81 -> ..., NOTES.md "Round 6").
#define LOB_PS_ARGS const DevParams* __restrict__ Pp, const DevState* __restrict__ Sp
#define LOB_PS_REFS const DevParams& P = *Pp; const DevState& S = *Sp;

// Device counters (env-steps, events, path statistics): LOB_CNT_STRIPES copies, LOB_CNT_STRIDE words apart; a block adds to the copy
// blockIdx.x selects and the host sums the copies when it reads.  Round 6: the three end-of-wave atomics of env_step_kernel onto ONE
// cache line -- 1 024 waves finishing together, ~4.5 ns per atomic at the memory side -- held the kernel's completion back by 18 us of
// its 103 (0.103 -> 0.085 ms with the counters compiled out: what five rounds of phase clocks could not see, because no wave waits
// for a fire-and-forget atomic; the kernel's end does).
#define LOB_CNT_STRIPES 256
#define LOB_CNT_STRIDE 64   /* i64 per stripe: 512 bytes, 16 used */
#if defined(__HIP__)
__device__ __forceinline__ void cnt_add(const DevState& S, int idx, unsigned long long v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(+S.counters) + (size_t)(blockIdx.x & (LOB_CNT_STRIPES - 1)) * LOB_CNT_STRIDE + idx, v);
}
#else   // (the device headers compiled as host code by tests/host_env: one "block")
inline void cnt_add(const DevState& S, int idx, unsigned long long v) { S.counters.p[idx] += (i64)v; }
#endif

// Q(s, a) for SARSA's TD error: left in LHdr::td by the trace step -- or, in the experiment that runs the learn kernels first, from qs_last
#ifdef LOB_EXPERIMENTS
#define LOB_QSA(P, S, b, h, ALGO) (((ALGO) == LOB_ALGO_SARSA && (P).exp_learn_first) ? (S).qs_last[(size_t)(b) * LOB_N_ACTIONS + (h).action] : (h).td)
#define LOB_TD_KEEP(P) ((P).exp_learn_first != 0)
#else
#define LOB_QSA(P, S, b, h, ALGO) ((h).td)
#define LOB_TD_KEEP(P) false
#endif

#define LOB_ERR_BAD_ORDER_PRICE 1  /* Order ctor would throw (src/market/order.cpp:22-27) */
#define LOB_ERR_BAD_LEVEL 2        /* ApplyChanges would throw (src/market/book.cpp:74-77) */
#define LOB_ERR_UNDEF_PRICE 4
#define LOB_ERR_TRACK_UNDERRUN 16   /* a book consumed its market-track ring faster than it was refilled (LOB_TRACK_RING / LOB_TRACK_REFILL) */
#define LOB_ERR_TRADE_OVERFLOW 8   /* more distinct trade price keys in one event than max_trades slots */      /* Book::price() would throw (src/market/book.cpp:171-173) */

#endif
