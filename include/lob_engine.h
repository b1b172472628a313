/*
 * lob_engine.h — C ABI of the MI355X-native batched limit-order-book
 * environment + tile-coded TD(lambda) learner.
 *
 * This is the drop-in boundary for the ONE hot path of tspooner/rl_markets
 * (SURVEY.md §8): `market::Book` / `environment::Intraday` step, state/reward
 * extraction, CMAC tile coding and the linear-Q SARSA(lambda)/Q(lambda)
 * update, batched struct-of-arrays over thousands of independent books and
 * executed by hand-written HIP kernels for gfx950.  Plain pointers and sizes
 * only: no C++/torch types cross this boundary.  The reference has no FFI
 * layer of its own; each entry point below names the reference C++ interface
 * it stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returning `int` returns LOB_OK (0) or a negative
 *     LOB_E* code; `lob_last_error()` gives a human-readable message
 *     (the reference throws std::runtime_error / returns bool — see
 *     INTEGRATION.md for the mapping);
 *   - "host" pointers are caller-owned host memory, "dev" pointers are device
 *     (HBM) addresses valid on the engine's GPU;
 *   - one engine handle per GPU, not thread-safe per handle (same rule as the
 *     reference: one Environment + Runner per thread, src/main.cpp:45-58).
 *   - the library needs a gfx950 GPU: there is NO CPU fallback.  Without a
 *     device `lob_create` fails with LOB_ENODEV.
 */
#ifndef LOB_ENGINE_H
#define LOB_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOB_ABI_VERSION 6

#define LOB_N_ACTIONS 9   /* reference Intraday::DoAction table, src/environment/intraday.cpp:181-219 */
#define LOB_N_TILINGS 32  /* config/example.yaml:18 (compile-time in the kernels) */
#define LOB_MAX_DEPTH 10
#define LOB_MAX_TRADES 8
#define LOB_MAX_BANDS 20
#define LOB_MAX_VARS 13
#define LOB_MAX_WINDOW 256
#define LOB_TRACE_GENS 64 /* largest trace ring, in generations of 32 tiles: gamma*lambda up to ~0.93;
                             the engine allocates 32 when the decay allows (example.yaml: 25 generations) */

enum {
    LOB_OK = 0,
    LOB_EINVAL = -1,   /* bad argument / unsupported parameter */
    LOB_ENODEV = -2,   /* no usable gfx950 device */
    LOB_ENOMEM = -3,   /* hipMalloc failed */
    LOB_ESTATE = -4,   /* call sequence error (e.g. step before reset) */
    LOB_EDATA = -5,    /* malformed event stream (the reference throws, src/market/book.cpp:74-77) */
    LOB_EHIP = -6      /* HIP runtime error */
};

/* Variable ids: reference enum class Variable, include/environment/intraday.h:17-24 */
enum {
    LOB_VAR_POS = 0, LOB_VAR_SPD, LOB_VAR_MPM, LOB_VAR_IMB, LOB_VAR_SVL, LOB_VAR_VOL,
    LOB_VAR_RSI, LOB_VAR_VWAP, LOB_VAR_A_DIST, LOB_VAR_A_QUEUE, LOB_VAR_B_DIST,
    LOB_VAR_B_QUEUE, LOB_VAR_LAST_ACTION
};

/* Reward measures: reference enum class RewardMeasure, include/environment/base.h:24-35 */
enum {
    LOB_REWARD_NONE = 0, LOB_REWARD_PNL, LOB_REWARD_PNL_DAMPED, LOB_REWARD_SPREAD,
    LOB_REWARD_NORMED, LOB_REWARD_LOVOL, LOB_REWARD_MM_LINEAR, LOB_REWARD_MM_EXP,
    LOB_REWARD_MM_DIV
};

/* Target-price object actually instantiated by the reference factory
 * (src/environment/base.cpp:101-112, quirk Q5: "midprice" -> MicroPrice,
 * anything else -> MidPrice) and the quoting mode of the Intraday ctor
 * (src/environment/intraday.cpp:64-82). */
enum { LOB_TP_MIDPRICE = 0, LOB_TP_MICROPRICE = 1 };
enum { LOB_QUOTE_TARGET = 0, LOB_QUOTE_BOOK = 1 };

/* Learning algorithms: rl::SARSA (src/rl/agent.cpp:296-311),
 * rl::QLearn = Watkins Q(lambda) (src/rl/agent.cpp:268-292),
 * rl::DoubleQLearn (src/rl/agent.cpp:185-264,315-353; config/example.yaml's default): a second
 * weight vector theta_b (`which` = 1 in lob_theta_get/set for shared theta, book + n_books for
 * private), actions from (Qa+Qb)/2, a coin flip per step from the agent's own
 * std::mt19937_64 (seeded with seed + global book id) choosing which vector is updated. */
/* SARSA / QLearn / DoubleQLearn (src/rl/agent.cpp:268-353); RLearn / OnlineRLearn / DoubleRLearn (average-reward:
 * src/rl/agent.cpp:357-467, selected by learning.algorithm r_learn / online_r_learn / double_r_learn, src/main.cpp:179-186;
 * they run the general kernels). */
enum { LOB_ALGO_SARSA = 0, LOB_ALGO_QLAMBDA = 1, LOB_ALGO_DOUBLE_Q = 2, LOB_ALGO_R_LEARN = 3, LOB_ALGO_ONLINE_R_LEARN = 4,
       LOB_ALGO_DOUBLE_R_LEARN = 5 };

/* Weight sharing: one theta shared by all books of the engine (the batched
 * analogue of the reference's Hogwild threads, src/main.cpp:196-206), or one
 * private theta per book (B independent learners = B copies of the
 * single-book reference; used for exact parity tests). */
enum { LOB_THETA_SHARED = 0, LOB_THETA_PRIVATE = 1 };

/* Behaviour policy: rl::EpsilonGreedy (src/rl/policy.cpp:58-82; epsilon 0 = rl::Greedy, 1 = rl::Random's
 * uniform action) or rl::Boltzmann (policy.cpp:85-122): P(a) ~ exp(Q(a) / tau), one uniform draw.
 * The exponential is glibc 2.35's exp(double) restated on the device from the library's own table and constants
 * (rl_markets_amd/csrc/lob_exp_table.h; tools/check_exp.c: 0 differences from libm over 4e8 inputs), so the sampled
 * action -- an index -- is bit-exact like every other. */
enum { LOB_POLICY_EPS_GREEDY = 0, LOB_POLICY_BOLTZMANN = 1 };

/* Venue description: reference market::Market (include/market/market.h:13-52).
 * Bands ascending by lower bound, as std::map<double,double> pts_ iterates. */
typedef struct lob_market {
    int64_t open_ms;                   /* mo_ */
    int64_t close_ms;                  /* mc_ */
    int32_t n_bands;
    int32_t _pad;
    double band_lb[LOB_MAX_BANDS];     /* price lower bound of band i   */
    double band_tick[LOB_MAX_BANDS];   /* tick size inside band i       */
} lob_market;

/* Engine parameters = the subset of the reference YAML config the hot path
 * reads (config/example.yaml; consumers src/environment/base.cpp:14-115,
 * src/rl/agent.cpp:13-60, src/rl/policy.cpp:58-82). */
typedef struct lob_params {
    int32_t abi_version;        /* LOB_ABI_VERSION */
    int32_t depth;              /* book levels per side, 1..LOB_MAX_DEPTH (reference: 5, quirk Q18) */
    int32_t max_trades;         /* trade price levels per event, 1..LOB_MAX_TRADES */
    int32_t n_vars;             /* state variables V (example.yaml: 8) */
    int32_t vars[LOB_MAX_VARS]; /* LOB_VAR_* in config order */

    lob_market market;

    int32_t order_size;         /* market.order_size */
    int32_t reward_measure;     /* LOB_REWARD_* */
    int64_t pos_lb, pos_ub;     /* market.pos_lb / pos_ub */
    float damping_factor;       /* reward.damping_factor (float in the reference) */
    float pos_weight, trd_weight, pnl_weight;

    int32_t lb_mpm, lb_vlt, lb_svl, lb_vwap, lb_rsi; /* state.lookback.* (>=1 after max(.,1)) */
    int32_t lb_spread;          /* policy.spread_lookback */
    int32_t lb_pnl;             /* reward.pnl_lookback */
    int32_t lb_target;          /* market.target_price.lookback */
    int32_t target_price;       /* LOB_TP_* */
    int32_t quote_mode;         /* LOB_QUOTE_* */

    int64_t memory_size;        /* learning.memory_size M (theta length, < 2^31) */
    int32_t n_tilings;          /* must be LOB_N_TILINGS */
    int32_t n_actions;          /* must be LOB_N_ACTIONS */
    double group_weights[3];    /* learning.group_weights */
    double gamma, lambda;
    double alpha;               /* current step size (alpha schedule is host-side, lob_set_alpha) */
    double epsilon;             /* current epsilon (schedule host-side, lob_set_epsilon) */
    int32_t algo;               /* LOB_ALGO_* */
    int32_t theta_mode;         /* LOB_THETA_* */
    uint64_t seed;              /* counter-based policy RNG seed (DESIGN.md "RNG") */
    uint64_t book_id_offset;    /* global id of local book 0 (multi-GPU shards) */
    int32_t policy;             /* LOB_POLICY_* */
    int32_t random_init;        /* learning.random_init (src/rl/agent.cpp:37-39,190-192): 0 = every weight +0.0; 1 = every weight 2u - 1, u
                                 * drawn in index order from the agent's own std::mt19937_64 (debug.random_seed; theta, then theta_b
                                 * of the double agents) -- private theta: every book's agent draws its own vectors (seed + global
                                 * book id, as for the coin); shared theta: the one vector is global book 0's agent's, whatever
                                 * shard this engine holds, and that agent's generator moves on as in the reference */
    double tau;                 /* Boltzmann temperature (schedule host-side, lob_set_tau) */
    double beta;                /* learning.beta: step size of the average reward rho (R-learning agents) */
} lob_params;

/* Synthetic event-stream generator. */
typedef struct lob_gen_params {
    uint64_t seed;
    int32_t n_events;
    int32_t t0_ms;              /* timestamp of event 0 */
    int32_t dt_ms;              /* cadence */
    int32_t start_ticks;        /* best bid = start_ticks * tick (0.1-tick grid of LSE HSBA band [500,1000)) */
    int32_t min_ticks, max_ticks;
    int32_t move_prob_q16;      /* P(best bid moves +-1 tick) * 65536 */
    int32_t spread2_prob_q16;   /* P(spread == 2 ticks) * 65536 */
    int32_t trade_prob_q16;     /* P(a trade in the interval) * 65536 */
    int32_t trade2_prob_q16;    /* P(a second trade on the other side | first) * 65536 */
    int32_t touch_prob_q16;     /* P(trade at touch rather than 2nd level) * 65536 */
    int32_t vol_min, vol_max;   /* level volumes U{min..max} */
    int32_t trade_min, trade_max;
} lob_gen_params;

/* Parity dump of one book's full environment state (test / debugging aid;
 * mirrors the protected members of environment::Base, include/environment/base.h:39-95,
 * and market::Book, include/market/book.h:34-48). */
typedef struct lob_book_dump {
    double ask_px[LOB_MAX_DEPTH], bid_px[LOB_MAX_DEPTH];
    double ask_last_px[LOB_MAX_DEPTH], bid_last_px[LOB_MAX_DEPTH];
    int64_t ask_vol[LOB_MAX_DEPTH], bid_vol[LOB_MAX_DEPTH];
    int64_t ask_last_vol[LOB_MAX_DEPTH], bid_last_vol[LOB_MAX_DEPTH];
    int64_t ask_total_volume, bid_total_volume;
    int64_t ask_last_total_volume, bid_last_total_volume;
    int32_t ask_n_transacted, bid_n_transacted;
    /* the (at most one, quirk Q13) live order per side */
    int32_t ask_has_order, bid_has_order;
    double ask_order_px, bid_order_px;
    int64_t ask_order_rem, bid_order_rem;
    int64_t ask_q_head, bid_q_head, ask_q_tail, bid_q_tail;
    int64_t position;
    double ask_quote, bid_quote;
    int32_t ask_level, bid_level;
    double pnl_step, momentum_pnl_step;
    int32_t lo_vol_step;
    int32_t last_action;
    double episode_reward, episode_pnl, episode_bandh;
    double spread_mean, target_price;
    int64_t time_ms;
    int32_t cursor;            /* next unread event index */
    int32_t terminal;          /* 1 = isTerminal(), 2 = stream exhausted */
    int32_t total_ticks;       /* TickStatistics::total_ticks */
    int32_t n_traces;          /* live eligibility traces */
    /* TradeStatistics / TickStatistics of the episode (include/environment/statistics.h:19-50): what Base::ClearInventory
     * (base.cpp:339-349) and Base::UpdateStats (base.cpp:412-442) count; the reference never touches the placed / cancelled /
     * "no ..." counters */
    int32_t market_buys, market_sells;
    int32_t ticks_with_ask, ticks_with_bid, ticks_with_both;
    int32_t ticks_with_position, ticks_long, ticks_short;
    /* TradeStatistics::ask_transactions / bid_transactions: ask/bid_n_transacted AS OF THE LAST DECISION -- Base::UpdateStats
     * (base.cpp:415-416) copies them when performAction has placed its orders (base.cpp:278), before the step's events run;
     * what the step's own events fill shows here one decision later (after an episode's last step: never).  getTotalTransactions /
     * getOrderRatio / writeStats (base.cpp:451-473) read these. */
    int32_t ask_transactions, bid_transactions;
} lob_book_dump;

typedef struct lob_engine lob_engine;

/* ---- library / parameter helpers (host only, no GPU needed) -------------- */

int lob_abi_version(void);
const char* lob_last_error(void);

/* config/example.yaml defaults (D=5, T=2, 8 vars, SARSA, LSE HSBA venue). */
void lob_default_params(lob_params* p);

/* Venue tables: reference Market::make_market, src/market/market.cpp:39-59
 * (tables :142-314).  `ticker` = "SYMBOL.VENUE", e.g. "HSBA.L". */
int lob_market_preset(const char* ticker, lob_market* out);

/* Tick maths on the host (reference Market::ToTicks / ToPrice / tick_size,
 * src/market/market.cpp:78-138).  Exposed for the host adaptors and tests. */
int lob_to_ticks(const lob_market* m, double price, int32_t* ticks);
int lob_to_price(const lob_market* m, int32_t ticks, double* price);
int lob_tick_size(const lob_market* m, double price, double* tick);

/* ---- event streams -------------------------------------------------------
 * One record per (book, event), little-endian 32-bit words:
 *   [0] time_ms  [1] flags
 *   ask_px[D] f32, ask_vol[D] i32, bid_px[D] f32, bid_vol[D] i32,
 *   trade_px[T] f32 (ascending), trade_vol[T] i32 (0 = empty slot),
 *   zero padding to a multiple of 4 words.
 * An event = what one Intraday::NextState consumes (src/environment/intraday.cpp:225-272):
 * the trades aggregated per price since the previous depth snapshot
 * (data::TimeAndSalesRecord, include/data/records.h:30-37) followed by the
 * new depth snapshot (data::MarketDepthRecord, include/data/records.h:20-28).
 * Layout in memory: records[book][event] (each book's events contiguous).
 */
#define LOB_EVT_FLAG_SAME_TIME 1u /* more depth rows with this timestamp follow (quirk Q14) */
#define LOB_EVT_FLAG_TAS_DRY 2u   /* no event may START at this row: the time-and-sales stream has run dry (Streamer::LoadUntil
                                   * fails, src/data/streamer.cpp:61-85, so Intraday::NextState returns false before it
                                   * touches the books); set by lob_convert_csv from the last trade rows of the file */

int32_t lob_record_words(int32_t depth, int32_t max_trades);
void lob_default_gen_params(lob_gen_params* g);
/* Fill `out` (n_books * n_events * record_words * 4 bytes) on the host. */
int lob_gen_stream_host(const lob_gen_params* g, int32_t depth, int32_t max_trades,
                        uint64_t first_book_id, int32_t n_books, uint32_t* out);
/* Check a host stream against the engine preconditions (positive prices and
 * volumes, strictly monotone price keys per side, ascending trade prices). */
int lob_validate_stream(const uint32_t* records, int32_t depth, int32_t max_trades,
                        int32_t n_books, int32_t n_events);

/* ---- ingestion of recorded data -------------------------
 * lob_convert_csv: the reference's two CSV formats (include/data/basic.h:17-24,49-52:
 * 22-column 5-level market depth, 4-column time-and-sales; header row skipped,
 * `stof` prices, rows with a non-positive price dropped, src/data/basic.cpp:45-70,
 * 148-162) -> one record per depth row carrying the trades of its own interval
 * (time-and-sales rows with prev_depth_time < time <= depth_time, aggregated per
 * 1e-4 price key like TimeAndSalesRecord::transactions).  Depth is 5.
 * lob_convert_lobster: LOBSTER message + orderbook files (prices x 10000): one
 * record per distinct millisecond = last snapshot of that millisecond + the
 * executions (types 4, 5) of that millisecond aggregated per price; the first
 * `depth` of `levels_in_file` levels are kept, records with a missing level are
 * dropped.  Both allocate *out_records (n_events * lob_record_words * 4 bytes);
 * release with lob_free. */
int lob_convert_csv(const char* md_path, const char* tas_path, int32_t max_trades, uint32_t** out_records,
                    int32_t* n_events);
int lob_convert_lobster(const char* orderbook_path, const char* message_path, int32_t levels_in_file, int32_t depth,
                        int32_t max_trades, uint32_t** out_records, int32_t* n_events);
void lob_free(void* p);

/* ---- engine lifetime ----------------------------------------------------- */

/* Replaces constructing environment::Intraday<> + rl::Agent + serial::Learner
 * (src/main.cpp:45-58,140-189) for `n_books` books on GPU `device`. */
int lob_create(const lob_params* p, int32_t n_books, int32_t device, lob_engine** out);
void lob_destroy(lob_engine* e);

/* Replaces Intraday::LoadData (src/environment/intraday.cpp:141-150):
 * upload host records / synthesise the same records directly in HBM.
 * 2 <= n_events (an int32: the per-book TickStatistics counters are 32-bit like the reference's ints, one count per agent
 * step, and an agent step consumes at least one event). */
int lob_load_events(lob_engine* e, const uint32_t* host_records, int32_t n_events);
int lob_gen_events_device(lob_engine* e, const lob_gen_params* g);
/* The NEXT episode's per-book streams, handed over WHILE the current episode runs -- the reference loads a fresh day before every
 * episode (src/main.cpp:53-55: rs.sample() + env.LoadData per episode and thread): validation and the host-to-HBM copy run on a
 * host thread and a HIP stream of their own into a second record buffer (another n_books x n_events records of HBM) and the call
 * returns at once; the lob_reset that follows waits for the hand-over if it has to and makes the staged stream the current one.
 * `host_records` must stay valid until that lob_reset (or lob_stage_wait) returns; n_events must equal the loaded stream's.
 * lob_stage_wait: block until the hand-over is complete and return its status (LOB_EDATA etc. as lob_load_events would). */
int lob_stage_events(lob_engine* e, const uint32_t* host_records, int32_t n_events);
int lob_stage_wait(lob_engine* e);
/* One recorded stream replayed by every book (BASELINE config 5: a converted LOBSTER day):
 * `host_records` holds n_total records of ONE book; book b plays the n_events records
 * starting at record phase[b] (0 <= phase[b], phase[b] + n_events <= n_total), i.e. it
 * behaves exactly like a book loaded with that window through lob_load_events.  The
 * reference replays one file pair per episode and thread (Intraday::LoadData,
 * src/environment/intraday.cpp:141-150; file cycling in src/experiment/serial.cpp:38-50);
 * the phases stand in for its per-thread choice of episode file. */
int lob_load_events_shared(lob_engine* e, const uint32_t* host_records, int64_t n_total, const int64_t* phase,
                           int32_t n_events);

/* ---- environment interface (environment::Base, include/environment/base.h:117-151) */

/* Initialise() for every book: clear books/stats/windows, fast-forward to
 * market open, warm the windows, place the (1,1) quotes, extract the first
 * state and its tile features (Runner::RunEpisode prologue,
 * src/experiment/serial.cpp:18-26). */
int lob_reset(lob_engine* e);
/* performAction(action) for every live book; `actions` host int32[n_books]. */
int lob_step(lob_engine* e, const int32_t* host_actions);
/* getState(): host float[n_books][n_vars]. */
int lob_get_state(lob_engine* e, float* host_out);
/* getReward(): host double[n_books]. */
int lob_get_reward(lob_engine* e, double* host_out);
/* isTerminal() (1) / out of data (2) / live (0): host uint8[n_books]. */
int lob_get_terminal(lob_engine* e, uint8_t* host_out);
/* ClearInventory() for every book (Runner::RunEpisode epilogue, serial.cpp:31). */
int lob_clear_inventory(lob_engine* e);
int lob_get_book(lob_engine* e, int32_t book, lob_book_dump* out);
int lob_get_books(lob_engine* e, int32_t first, int32_t n, lob_book_dump* out);

/* ---- learner interface (rl::Agent, include/rl/agent.h:48-77) ------------- */

/* `n_steps` x Learner::_step (src/experiment/serial.cpp:53-70) for every live
 * book: action(s) -> performAction -> newState -> HandleTransition. */
int lob_td_step(lob_engine* e, int32_t n_steps);
/* One such step in two halves: lob_td_step_begin = swap, isTerminal, action, performAction, newState of every book;
 * lob_td_step_end = HandleTransition (traces, TD errors, update).  Between them no cached action-selection data is live, so
 * that is where a multi-GPU weight exchange goes (include/lob_comm.h lob_theta_allreduce): Q(from_state, .) of the step is
 * what it was when the action was chosen, Q(to_state, .) sees the exchanged weights.  Nothing else may come in between:
 * lob_td_step, lob_td_step_begin, lob_eval_step, lob_step, lob_clear_inventory, lob_handle_terminal and lob_theta_set return
 * LOB_ESTATE there; lob_reset abandons the half-done step with its episode.  begin + end without an exchange =
 * lob_td_step(e, 1), bit for bit. */
int lob_td_step_begin(lob_engine* e);
int lob_td_step_end(lob_engine* e);
/* 1 if this engine can split a step (always, except an experiments build run with two book groups, LOB_GROUPS=2): the
 * caller that cannot split runs the whole step and exchanges after it -- asked for explicitly instead of being inferred
 * from a LOB_ESTATE of lob_td_step_begin, which has other causes (no reset, a half step left open). */
int lob_td_split_supported(lob_engine* e);
/* Backtester::_step (serial.cpp:124-137): greedy action, no learning. */
int lob_eval_step(lob_engine* e, int32_t n_steps);
/* Agent::HandleTerminal (src/rl/agent.cpp:103-109): traces.decay(0). The
 * alpha / epsilon schedules are evaluated by the host adaptor. */
int lob_handle_terminal(lob_engine* e);

/* Replaces the `model_log` logger of Agent::HandleTransition (src/rl/agent.cpp:53-59,93-100: `_agg_delta += abs(delta)`, and
 * every 1000 updates one row `_agg_delta / 1000`).  After lob_model_log_enable(e, 1) every learner step adds the stepped books'
 * |delta| to a running aggregate on the device; once it holds 1000 updates or more, a row aggregate / count is written and
 * both start again.  One book: the reference's rows exactly (the count reaches 1000 one update at a time).  A batch: a row
 * per step once the batch has 1000 books, the mean |delta| of the step.  lob_model_log_read hands over the rows written since
 * the last read (at most `cap`; `n_lost`, if not NULL: rows that did not fit the device ring of 8192 or `cap`). */
int lob_model_log_enable(lob_engine* e, int32_t on);
int lob_model_log_read(lob_engine* e, double* rows, int32_t cap, int32_t* n_rows, int64_t* n_lost);
int lob_set_alpha(lob_engine* e, double alpha);
int lob_set_epsilon(lob_engine* e, double epsilon);
int lob_set_tau(lob_engine* e, double tau);

/* State::newState(vector<float>&) + getFeatures (src/rl/state.cpp:45-70):
 * n states of n_vars floats -> int32[n][9][96] tile indices. */
int lob_features(lob_engine* e, const float* host_vars, int32_t n, int32_t* host_out);
/* Agent::getQ for all actions (src/rl/agent.cpp:117-135): double[n][9]. */
int lob_q_values(lob_engine* e, const float* host_vars, int32_t n, double* host_out);

/* theta access: Agent::write_theta (src/rl/agent.cpp:176-181) + the missing
 * load path.  `which` = book for LOB_THETA_PRIVATE, 0 for shared. */
int lob_theta_get(lob_engine* e, int32_t which, double* host_out, int64_t count);
int lob_theta_set(lob_engine* e, int32_t which, const double* host_in, int64_t count);
/* Last actions / rewards / TD errors of the most recent lob_td_step. */
int lob_get_last_actions(lob_engine* e, int32_t* host_out);
int lob_get_last_td(lob_engine* e, double* host_out);
int lob_get_last_rewards(lob_engine* e, double* host_out);
/* 1 where the most recent step performed an env-step + TD update */
int lob_get_stepped(lob_engine* e, int32_t* host_out);
/* draws consumed so far from each book's policy RNG stream */
int lob_get_rng_counters(lob_engine* e, uint64_t* host_out);
/* state_vars of the rl::State produced by the last step: float[n_books][n_vars] */
int lob_get_learner_state(lob_engine* e, float* host_out);
/* Live traces of one book: indices and eligibilities (rl::Traces). */
int lob_get_traces(lob_engine* e, int32_t book, int32_t* idx, float* elig, int32_t cap, int32_t* n);

/* Counters: [0] env-steps performed, [1] market events consumed,
 * [2] live books, [3] td updates applied. */
int lob_get_counters(lob_engine* e, int64_t out[4]);
/* Which kernels served the books (diagnostics of the fast paths, cumulative since lob_create unless noted):
 * [0] books the SARSA lane trace kernel (trace_sarsa_kernel) handed back to the wave-per-book kernel,
 * [1] books whose action came from the hit-list replay (the light action selection), [2] memo slots registered this episode,
 * [3] weight indices found ambiguous this episode (tile registry), [4] 1 if the registry overflowed this episode,
 * [5] memo slots in use in the latest step, [6] books whose action the fused env kernel had to evaluate in full (no usable hit
 * list: act_book in-kernel), [7] books the lane learn kernels handed back to the wave-per-book evaluation (trace_rest_kernel /
 * learn_q_rest_kernel). */
int lob_get_path_stats(lob_engine* e, int64_t out[8]);

/* ---- multi-GPU weight exchange ---------------------------
 * The engine library never calls a collective itself: it exposes the dense
 * delta buffer, and include/lob_comm.h (liblob_comm.so, RCCL) all-reduces it
 * in place over xGMI on the engine's stream (lob_theta_allreduce) -- the batched
 * stand-in for the reference's shared Agent* of src/main.cpp:196-206.
 *   lob_delta_init  : theta_sync = theta (call once, after create; lob_theta_set keeps it in step)
 *   lob_delta_begin : dev_delta[i] = theta[i] - theta_sync[i]   (then waits for the stream;
 *                     lob_delta_begin_async only enqueues it)
 *   (caller: all-reduce SUM dev_delta over ranks)
 *   lob_delta_apply : theta = theta_sync + dev_delta ; theta_sync = theta
 * `count` = memory_size, or 2 x memory_size for LOB_ALGO_DOUBLE_Q (theta then theta_b); the average-reward agents
 * append two slots [rho - rho_sync, 1.0] (the sum's second slot = the number of ranks: rho moves by the mean change).
 */
int lob_delta_init(lob_engine* e);
int lob_delta_begin(lob_engine* e, double** dev_delta, int64_t* count);
int lob_delta_begin_async(lob_engine* e, double** dev_delta, int64_t* count);
int lob_delta_apply(lob_engine* e);
/* The same exchange without the dense vector (shared theta on the fast path, i.e. SARSA / Q(lambda); dense otherwise).  The
 * engine's exact written-weights map (one bit per weight) enumerates every weight a step of this rank has touched:
 *   lob_delta_sparse_maps  : the rank's map and a [world][words] buffer to ALL-GATHER the ranks' maps into (uint32 words)
 *   lob_delta_sparse_pack  : union of the gathered maps -> one compact vector layout common to all ranks;
 *                            dev_buf[p] = theta[f] - theta_sync[f] for the p-th weight f of the union.  Returns the element
 *                            count (identical on all ranks) after ONE stream synchronisation: the collective needs it.
 *   (caller: all-reduce SUM dev_buf[0 .. count) over ranks)
 *   lob_delta_sparse_apply : theta[f] = theta_sync[f] + dev_buf[p] ; theta_sync[f] = theta[f]
 * A few hundred thousand doubles instead of memory_size = 20 M of them. */
int lob_delta_sparse_supported(lob_engine* e);
int lob_delta_sparse_maps(lob_engine* e, int32_t world, uint32_t** dev_own, uint32_t** dev_gather, int64_t* words);
int lob_delta_sparse_pack(lob_engine* e, int32_t world, double** dev_buf, int64_t* count);
int lob_delta_sparse_apply(lob_engine* e);

/* Synchronise the engine's stream / expose it (hipStream_t as void*). */
int lob_sync(lob_engine* e);
void* lob_stream(lob_engine* e);
/* Average duration (ms) of the named kernel over the TIMED launches since the last
 * reset of the timers, measured with HIP events on the engine stream.
 * lob_kernel_timing(e, n): 0 = off, 1 = every launch, n > 1 = the launches of every n-th
 * step (two event records per launch cost ~9 % of a step when every launch is timed). */
int lob_kernel_time_ms(lob_engine* e, const char* kernel, double* avg_ms, int64_t* launches);
int lob_kernel_timing(lob_engine* e, int32_t enable);

#ifdef __cplusplus
}
#endif
#endif /* LOB_ENGINE_H */
