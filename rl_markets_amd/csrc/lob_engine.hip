// Host side of the C ABI (include/lob_engine.h): device memory management,
// kernel launches on one HIP stream, HIP-event kernel timing.  gfx950 only;
// there is no CPU execution path in this file.  (One of the library's four translation units, lob_launch.h: the environment,
// pre-pass and fast learner kernels are compiled in lob_tu_env.hip / lob_tu_prepass.hip / lob_tu_learn.hip.)
#define LOB_TU_SPLIT 1
#define LOB_TU_MAIN 1
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "lob_internal.h"
#include "lob_fast.h"
#include "lob_kernels.h"
#include "lob_launch.h"

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            lob_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
            return LOB_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

// parameters and state of the kernels that take them by pointer (lob_state.h LOB_PS_ARGS): their device-resident copies
#define LOB_PS(e) (const DevParams*)(e)->P_dev, (e)->S.self

#define LOB_HINT_RING 64
#define LOB_HINT_LAG 16
#define LOB_HINT_EVERY 8   /* the count goes to host memory in every 8th learner step only: the store costs the kernel that makes it 3.5 us */

struct KTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct lob_engine {
    int device = 0;
    int B = 0;
    lob_params params;
    DevParams P;
    DevParams* P_dev = nullptr;  // device copy for the lane-per-book kernels
    DevState S;
    // The device-resident copy of S, at one address for the engine's lifetime (S.self): kernels that take the state by POINTER read
    // the fields they need where they need them, through the scalar cache -- 1 728 bytes of by-value kernel arguments are loaded at
    // the kernel's entry, live across all of it and cost env_step_kernel 104 spilled scalar registers + 160 vector registers parked
    // in AGPRs (round 6).  Kept up by sync_state() before every launch sequence; the two words a step changes (cb_par,
    // cb_dense_on) are never read through the pointer: they travel as scalar arguments.
    DevState* S_dev = nullptr;
    DevState S_pushed;
    hipStream_t stream = nullptr;   // main stream (all API calls)
    hipStream_t stream2 = nullptr;  // second book group of the step pipeline
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_stagger = nullptr;
    hipEvent_t ev_rest_go = nullptr, ev_rest_done = nullptr;  // learn_q_rest_kernel on stream2 beside the trace kernels
    bool rest_side = true;      // (LOB_REST_SIDE=0: on the main stream, as before; A/B switch)
    bool rest_merge = true;     // the fused Q(lambda) / double Q flow: trace_rest_kernel (LOB_REST_MERGE=0: learn_q_rest_kernel + trace_fast_kernel<., 2> + accumulate_kernel over its list; A/B switch)
    bool dq_pair = true;        // double Q(lambda): learn_q_pair_kernel<DOUBLE_Q> (LOB_DQ_PAIR=0: a lane per book, learn_q_lane_kernel; A/B switch)
    int ts_lds = 0, ts_grid = 0; // (experiments: trace_lane_kernel<SARSA> with dynamic LDS / a persistent grid)
    bool no_hint = false;       // (experiment: learn_q_rest_kernel without its report to the host)
    bool exp_learn_first = false;  // (experiment LOB_SARSA_LEARN_FIRST=1: DevParams::exp_learn_first)
    long long flow[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // lob_debug_flow: learner steps by the shape of their update / action selection (see there)
    bool acc_fuse = true;       // Q(lambda), the pair kernel + the lane-per-generation trace kernel: updates added to their slots there, accumulate_kernel over a list (LOB_ACC_FUSE=0: over every book; A/B switch)
    bool acc_batches_set = false;
    int acc_batches = LOB_ACB_K;  // accumulate_block_kernel: batches of 1 024 books per block (LOB_ACC_BATCHES=1|2|4|8; A/B switch)
    bool acc_block = true;      // SARSA(lambda): sums per block first (LOB_ACC_BLOCK=0: accumulate_kernel; A/B switch)
    bool dense_ever = false;    // the dense sums have been on at some step: slots may hold dense ids (apply_kernel frees them with their slots)
    bool acc_dense = true;      // ... in a direct-indexed LDS array by the slots' dense ids, accumulate_dense_kernel (LOB_ACC_DENSE=0: accumulate_block_kernel's hash table; A/B switch)
    int env_step_lanes = 64;    // books per wave of env_step_kernel (LOB_ENV_STEP_LANES=32: two half-full waves per SIMD; experiment)
    int env16_max = 4096;       // largest batch that takes env_step16_kernel (16 lanes per book) instead of env_step_kernel (LOB_ENV16_MAX; 0: never)
    bool prepass_roles = false; // the pre-pass on two waves per 64 books (reset2_kernel / prepass_extend2_kernel; LOB_PREPASS_ROLES=1): measured slower, opt-in
    // learn_q_rest_kernel reports its list's length (the books the lane learn kernels handed back) through host-mapped memory:
    // a ring of LOB_HINT_RING words, one per learner step, each with an event recorded behind the kernel that writes it.  The
    // step launched now reads the word of LOB_HINT_LAG steps ago after waiting for THAT event (long past, unless the host is
    // more than LOB_HINT_LAG steps ahead of the device -- then it waits: the device still has that many steps queued), so which
    // path a step takes is a function of the run, not of the host's timing.  The first LOB_HINT_LAG steps after
    // lob_reset / lob_theta_set read 0.  A weight exchange (lob_delta_apply / lob_delta_sparse_apply) does NOT void the counts:
    // it runs every 64 steps, the written set only grows by the union of the ranks', and which books have no usable hit list
    // hardly changes -- voiding them would send a quarter of a dense-theta run's steps down the slower path.
    u64* rest_hint = nullptr;   // word = (step tag) << 32 | count: the host checks the tag, so it never reads a word the device has not written yet
    u64* rest_hint_dev = nullptr;
    hipEvent_t hint_ev[LOB_HINT_RING] = {};
    long long hint_step = 0;    // learner steps (fast path) since the hints were last void
    unsigned long long hint_serial = 0;  // launches of learn_q_rest_kernel since lob_create (the tag of a ring word)
    uint32_t hint_tags[LOB_HINT_RING] = {};
    int hint_now = 0;           // the hint this step goes by (run_steps)
    int force_general = -1;     // LOB_MOSTLY_GENERAL=0|1: the work-list act path never / always (tests); -1: by the hint
    int rest_recent = 0;        // steps left to keep the launch on the second stream after the last non-empty list seen
    int reg_apar = 0;   // which of the two lists of newly ambiguous indices this step's registry blocks fill (trace_lane_kernel's launch) and its scan blocks read (apply_kernel's)
    int n_groups = 1;
    int env_lanes = 0;     // books per env_kernel wave: 0 = by batch size, or 16 / 32 / 64; 256 = env_compact_kernel (LOB_ENV_LANES, read by lob_create)
    int reset_lanes = 64;  // books per reset_kernel wave (LOB_RESET_LANES)
    int td_parity = 0;
    int step_id = 0;            // stamps the memo claims of one step
    int list_par = 0;           // parity of the fast path's work lists
    int last_par = 0;           // `par` of the most recent step (its memo list is the current one)
    int n_cus = 256;            // compute units: the persistent learner kernels run one block on each
    int track_ring = 4096;      // longest resident market track, in events per book; longer streams keep a ring of this many (LOB_TRACK_RING)
    int track_refill = 64;      // steps between refills of the ring (LOB_TRACK_REFILL)
    bool chunked = false;       // the loaded stream is longer than the ring
    int steps_since_fill = 0;
    uint64_t theta_ver = 1;     // bumped whenever theta changes: memo records carry the version they were computed under
    bool half_open = false;     // between lob_td_step_begin and lob_td_step_end
    bool hits_ok = false;       // the previous call was a fast-path learner step and nothing has touched weights, maps or states since:
                                // the hit lists its learn kernel left are those of the States the next step acts on (act_light_kernel)
    bool light = true;          // use them (LOB_NO_LIGHT=1: always the full act kernel, for A/B runs)
    bool force_fuse_act = false;
    bool fuse_act = true;       // ... inside the env kernel (env_kernel<.., 1>; LOB_NO_FUSE_ACT=1: act_light_kernel as a launch of its own)
 bool t_light = true;        // trace_light_kernel in front of the wave-per-book trace kernel (Q(lambda); LOB_NO_TLIGHT=1: off)
    bool inline_general = true; // ... serving the books without a hit list itself (LOB_INLINE_GENERAL=0: through the work list, two more launches)
    int steps_on_lists = 0;     // fused steps since the hit lists were last void
    bool env_step = true;       // env_step_kernel for the fused action selection + step (LOB_ENV_STEP=0: env_kernel<64, 2, 1>; A/B switch)
    bool no_fuse = false;       // LOB_NO_FUSE=1: the light trace step as a kernel of its own, not inside the lane learner kernel (A/B switch)
    bool q_pair = true;         // ... two lanes per book (learn_q_pair_kernel; LOB_Q_PAIR=0: one)
    int q_lanes = -1;           // learn_q_lane_kernel (a lane per book) instead of learn_q_fast_kernel (a wave per book): -1 by batch size,
                                // 0 / 1 forced (LOB_Q_LANES)
    std::vector<void*> allocs;
    // sparse weight exchange (lob_delta_sparse_*): the ranks' gathered maps, their union, per-block counts / offsets, the packed deltas
    uint32_t* spx_gather = nullptr;
    int spx_world = 0;
    uint32_t* spx_union = nullptr;
    i32* spx_block_cnt = nullptr;
    i64* spx_block_off = nullptr;
    i64* spx_total = nullptr;
    f64* spx_buf = nullptr;
    int64_t spx_cap = 0, spx_count = 0;
    // The exchange without a host synchronisation: the packed vector's length is a host argument of the
    // collective, and the union's true size is only known on the device.  So the ranks exchange a FIXED count -- spx_fixed, the
    // same on every rank because it is a function of the union sizes of the exchanges before, which are the same on every rank --
    // with the tail behind the union zero-filled, and every exchange leaves its union's size in pinned host memory behind an event
    // that the NEXT exchange (64 steps later: long past) reads.  The union only grows, ever more slowly (the written set saturates):
    // an exchange goes without a synchronisation when the last size + twice the last increase still fits spx_fixed; the first two
    // exchanges, and any whose prediction does not fit, take the exact count with a synchronisation and set spx_fixed to at least
    // twice it.  A union that outgrows the count anyway (its growth more than doubled from one interval to the next) loses nothing:
    // the entries beyond it keep theta - theta_sync and travel with the next exchange (counted: lob_debug_exchange; the parity
    // tests of the exchange assert that it never happens in them).
    i64* spx_total_host = nullptr;   // pinned
    hipEvent_t spx_ev = nullptr;
    int64_t spx_fixed = 1 << 18, spx_last_total = -1, spx_prev_total = -1;
    bool spx_ev_pending = false;
    long long spx_nosync = 0, spx_synced = 0, spx_overflows = 0;
    uint32_t* rnd_dev = nullptr;
    uint32_t* records_dev = nullptr;
    // The NEXT episode's streams, handed over while the current episode runs (lob_stage_events): a second record buffer filled by
    // a host thread through a stream of its own -- validation, pinned staging, DMA, repack, none of it on the engine's stream --
    // and adopted by the lob_reset that follows.  The reference loads a fresh day before every episode (src/main.cpp:53-55).
    size_t records_bytes = 0, track_bytes = 0;   // sizes of records_dev / track_dev (set_records re-uses buffers of the right size)
    uint32_t* records_next = nullptr;
    hipStream_t stream_up = nullptr;
    std::thread stage_thread;
    bool stage_active = false;      // a hand-over has been started and not adopted / waited for yet
    int stage_rc = LOB_OK;
    std::string stage_err;
    i64* phase_dev = nullptr;  // replayed stream: first record of every book's window
    Track* track_dev = nullptr;
    i32* actions_dev = nullptr;
    i64* cnt_sum = nullptr;     // the striped device counters added up (read_counters)
    lob_book_dump* dump_dev = nullptr;
    int dump_cap = 0;
    bool have_events = false, was_reset = false;
    bool episode_open = false;  // a pre-pass ran and its window sums have not been rolled back to the stop point yet
    bool model_log = false;     // lob_model_log_enable: the step sums |delta| (td_stats_kernel)
    bool timing = false;
    int acc_shift = -1;     // accumulate_kernel lanes per book (log2); -1: chosen from the algorithm and epsilon
    int timing_period = 1;  // kernels of every n-th step are timed (two event records per launch are not free: 9 % at n = 1)
    std::map<std::string, KTimer> timers;
    std::vector<hipEvent_t> event_pool;
};

namespace {

// The 2048-entry table of the UNH CMAC hash (reference src/rl/tiles.cpp:133)
// is the low-byte stream of an unseeded glibc rand() (generator left commented
// out at tiles.cpp:141-149).  Regenerated from the glibc TYPE_3 additive
// feedback recurrence r[i] = r[i-3] + r[i-31]; tests compare it with the
// reference's own table.
void make_rndseq(uint32_t* t) {
    const int total = 344 + 4 * 2048;
    std::vector<int32_t> r(total);
    r[0] = 1;
    for (int i = 1; i < 31; i++) {
        int64_t v = (16807LL * r[i - 1]) % 2147483647LL;
        if (v < 0) v += 2147483647LL;
        r[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 34; i < total; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    for (int k = 0; k < 2048; k++) {
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) v = (v << 8) | ((((uint32_t)r[344 + 4 * k + i]) >> 1) & 0xff);
        t[k] = v;
    }
}

template <class T> int dev_alloc(lob_engine* e, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t err = hipMalloc(&q, bytes);
    if (err != hipSuccess) {
        lob_set_error(std::string("hipMalloc failed: ") + hipGetErrorString(err));
        return LOB_ENOMEM;
    }
    err = hipMemsetAsync(q, 0, bytes, e->stream);
    if (err != hipSuccess) { lob_set_error("hipMemset failed"); return LOB_EHIP; }
    e->allocs.push_back(q);
    *p = (T*)q;
    return LOB_OK;
}

template <class T> int dev_alloc(lob_engine* e, GP<T>* p, size_t count) { return dev_alloc(e, &p->p, count); }

// The device copy of DevState follows the host's (rare: a stream loaded, the exchange's buffers allocated, model_log switched on).
int sync_state(lob_engine* e) {
    DevState now = e->S;
    now.cb_par = 0;
    now.cb_dense_on = 0;
    now.self = e->S_dev;
    if (memcmp(&now, &e->S_pushed, sizeof(DevState)) == 0) return LOB_OK;
    if (hipMemcpyAsync(e->S_dev, &now, sizeof(DevState), hipMemcpyHostToDevice, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) {
        lob_set_error("state upload failed");
        return LOB_EHIP;
    }
    e->S_pushed = now;
    return LOB_OK;
}
int push_params(lob_engine* e) {
    if (hipMemcpyAsync(e->P_dev, &e->P, sizeof(DevParams), hipMemcpyHostToDevice, e->stream) != hipSuccess) { lob_set_error("param upload failed"); return LOB_EHIP; }
    return LOB_OK;
}

int grid_lanes(int B) { return (B + 255) / 256; }
int grid_waves(int B) { return (B + LOB_WAVES_PER_BLOCK - 1) / LOB_WAVES_PER_BLOCK; }

struct TimedLaunch {
    lob_engine* e;
    KTimer* t = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    // `always`: kernels that do not run every step (reset, weight exchange, ring refill) are timed whenever timing is on
    TimedLaunch(lob_engine* e_, const char* name, hipStream_t s = nullptr, bool always = false) : e(e_), st(s ? s : e_->stream) {
        if (!e->timing || (!always && e->timing_period > 1 && e->step_id % e->timing_period != 0)) return;
        t = &e->timers[name];
        auto get = [&]() {
            hipEvent_t ev;
            if (!e->event_pool.empty()) { ev = e->event_pool.back(); e->event_pool.pop_back(); }
            else hipEventCreateWithFlags(&ev, hipEventDisableSystemFence);
            return ev;
        };
        a = get();
        b = get();
        hipEventRecord(a, st);
    }
    ~TimedLaunch() {
        if (!t) return;
        hipEventRecord(b, st);
        t->pending.push_back({a, b});
    }
};

void drain_timers(lob_engine* e) {
    for (auto& kv : e->timers) {
        for (auto& pr : kv.second.pending) {
            hipEventSynchronize(pr.second);
            float ms = 0;
            hipEventElapsedTime(&ms, pr.first, pr.second);
            kv.second.total_ms += ms;
            kv.second.launches++;
            e->event_pool.push_back(pr.first);
            e->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

int check_device_errors(lob_engine* e) {
    i32 flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, e->S.error_flag, sizeof flag, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (flag) {
        std::string m = "device reported a condition on which the reference throws:";
        if (flag & LOB_ERR_BAD_ORDER_PRICE) m += " [order price <= 0]";
        if (flag & LOB_ERR_BAD_LEVEL) m += " [level price/volume <= 0]";
        if (flag & LOB_ERR_UNDEF_PRICE) m += " [undefined book price]";
        if (flag & LOB_ERR_TRADE_OVERFLOW) m += " [more trade price levels in one event than max_trades]";
        if (flag & LOB_ERR_TRACK_UNDERRUN) m += " [market-track ring underrun: raise LOB_TRACK_RING or lower LOB_TRACK_REFILL]";
        lob_set_error(m);
        return LOB_EDATA;
    }
    return LOB_OK;
}

}  // namespace

extern "C" {

static int theta_random_init(lob_engine* e);

int lob_create(const lob_params* p, int32_t n_books, int32_t device, lob_engine** out) {
    if (!p || !out || n_books < 1) { lob_set_error("lob_create: bad argument"); return LOB_EINVAL; }
    if (p->abi_version != LOB_ABI_VERSION) { lob_set_error("lob_create: ABI version mismatch"); return LOB_EINVAL; }
    if (p->depth < 1 || p->depth > LOB_MAX_DEPTH || p->max_trades < 1 || p->max_trades > LOB_MAX_TRADES) {
        lob_set_error("lob_create: depth/max_trades out of range"); return LOB_EINVAL;
    }
    if (p->n_tilings != LOB_N_TILINGS || p->n_actions != LOB_N_ACTIONS) {
        lob_set_error("lob_create: kernels are built for n_tilings=32, n_actions=9"); return LOB_EINVAL;
    }
    if (p->n_vars < 4 || p->n_vars > LOB_MAX_VARS) { lob_set_error("lob_create: n_vars must be in [4,13]"); return LOB_EINVAL; }
    if (p->memory_size < 1 || p->memory_size >= (1LL << 31)) { lob_set_error("lob_create: memory_size out of range"); return LOB_EINVAL; }
    if (p->market.n_bands < 1 || p->market.n_bands > LOB_MAX_BANDS) { lob_set_error("lob_create: bad tick table"); return LOB_EINVAL; }
    if (p->algo < LOB_ALGO_SARSA || p->algo > LOB_ALGO_DOUBLE_R_LEARN) { lob_set_error("lob_create: unknown algorithm"); return LOB_EINVAL; }
    if (p->policy != LOB_POLICY_EPS_GREEDY && p->policy != LOB_POLICY_BOLTZMANN) { lob_set_error("lob_create: unknown policy"); return LOB_EINVAL; }
    if (p->policy == LOB_POLICY_BOLTZMANN && !(p->tau > 0.0)) { lob_set_error("lob_create: Boltzmann temperature must be positive"); return LOB_EINVAL; }
    const int lbs[] = {p->lb_mpm, p->lb_vlt, p->lb_svl, p->lb_vwap, p->lb_rsi, p->lb_spread, p->lb_pnl, p->lb_target};
    for (int w : lbs)
        if (w < 1 || w > LOB_MAX_WINDOW) { lob_set_error("lob_create: lookbacks must be in [1,256]"); return LOB_EINVAL; }
    if (p->order_size < 1) { lob_set_error("lob_create: order_size"); return LOB_EINVAL; }

    // Traces::decay(float rate), quirk Q15: eligibility by age as iterated float products; checked before
    // any HIP resource exists
    float trace_pow[LOB_TRACE_GENS + 1];
    int trace_kmax = -1;
    const float trace_rate = (float)(p->gamma * p->lambda);
    trace_pow[0] = 1.0f;
    for (int k = 1; k <= LOB_TRACE_GENS; k++) {
        trace_pow[k] = trace_pow[k - 1] * trace_rate;
        if (trace_kmax < 0 && trace_pow[k] < 0.01f) trace_kmax = k;
    }
    if (trace_kmax < 0 || trace_kmax > LOB_TRACE_GENS) {
        lob_set_error("lob_create: gamma*lambda too close to 1 for the trace ring (LOB_TRACE_GENS = 64 generations, gamma*lambda <= ~0.93)");
        return LOB_EINVAL;
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
        lob_set_error("lob_create: no usable HIP device (the engine has no CPU fallback)");
        return LOB_ENODEV;
    }
    HIPCHK(hipSetDevice(device));
    lob_engine* e = new lob_engine();
    e->device = device;
    e->B = n_books;
    e->params = *p;
    // from here on every failure path releases what exists so far (lob_destroy copes with null members)
#define HIPCHK_E(expr)                                                                       \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            lob_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
            lob_destroy(e);                                                                  \
            return LOB_EHIP;                                                                 \
        }                                                                                    \
    } while (0)
    HIPCHK_E(hipStreamCreate(&e->stream));
    HIPCHK_E(hipStreamCreate(&e->stream2));
    HIPCHK_E(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_E(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_E(hipEventCreateWithFlags(&e->ev_stagger, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_E(hipEventCreateWithFlags(&e->ev_rest_go, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_E(hipEventCreateWithFlags(&e->ev_rest_done, hipEventDisableTiming | hipEventDisableSystemFence));
    // Runtime switches.  Honoured by every build: the A/B switches of first-class paths that the parity tests force on or off at
    // sizes that would not select them (LOB_TRACE_LANES, LOB_Q_LANES, LOB_Q_PAIR, LOB_DQ_PAIR, LOB_NO_TLIGHT, LOB_NO_MEMO, LOB_NO_LIGHT,
    // LOB_NO_FUSE, LOB_NO_COMBINE, LOB_NO_CARRY, LOB_MOSTLY_GENERAL, LOB_FUSE_ACT, LOB_ENV_LANES=16|64, LOB_ACC_LANES, LOB_ACC_FUSE,
    // LOB_ACC_DENSE, LOB_ACC_BATCHES, LOB_REST_MERGE), the table sizes the tests squeeze (LOB_CB_SLOTS, LOB_OW_SLOTS, LOB_AMB_CAP, LOB_CBD_IDS), the track
    // ring (LOB_TRACK_RING, LOB_TRACK_REFILL) and the exchange's form (LOB_DENSE_EXCHANGE, lob_comm.cpp).  The switches of variants
    // measured and lost, and of timing experiments, only by a -DLOB_EXPERIMENTS build (tools/exp_variants.sh; NOTES.md): `exps`.
    const bool exps = lobk_experiments() != 0;
    if (const char* g = getenv("LOB_REST_SIDE")) e->rest_side = !(exps && g[0] == '0');
    if (const char* g = getenv("LOB_ACC_BLOCK")) e->acc_block = !(exps && g[0] == '0');
    if (const char* g = getenv("LOB_DQ_PAIR")) e->dq_pair = !(g[0] == '0');
    if (const char* g = getenv("LOB_REST_MERGE")) e->rest_merge = !(g[0] == '0');
    if (exps) {
        if (const char* g = getenv("LOB_TS_LDS")) e->ts_lds = atoi(g);
        e->exp_learn_first = getenv("LOB_SARSA_LEARN_FIRST") != nullptr;
        if (const char* g = getenv("LOB_TS_GRID")) e->ts_grid = atoi(g);
        e->no_hint = getenv("LOB_NO_HINT") != nullptr;
    }
    if (const char* g = getenv("LOB_ACC_DENSE")) e->acc_dense = !(g[0] == '0');
    if (const char* g = getenv("LOB_ACC_FUSE")) e->acc_fuse = !(g[0] == '0');
    if (const char* g = getenv("LOB_ACC_BATCHES")) { const int v = atoi(g); if (v == 1 || v == 2 || v == 4 || v == 8) { e->acc_batches = v; e->acc_batches_set = true; } }
    if (const char* g = getenv("LOB_ENV_STEP_LANES")) { if (exps && atoi(g) == 32) e->env_step_lanes = 32; }
    if (const char* g = getenv("LOB_ENV16_MAX")) { const int v = atoi(g); if (v >= 0) e->env16_max = v; }
    if (const char* g = getenv("LOB_PREPASS_ROLES")) e->prepass_roles = exps && g[0] == '1';
    if (hipHostMalloc((void**)&e->rest_hint, LOB_HINT_RING * sizeof(u64), hipHostMallocMapped) == hipSuccess) {
        memset(e->rest_hint, 0, LOB_HINT_RING * sizeof(u64));
        if (hipHostGetDevicePointer((void**)&e->rest_hint_dev, e->rest_hint, 0) != hipSuccess) e->rest_hint_dev = nullptr;
    } else e->rest_hint = nullptr;   // (no hint: the launch stays on the main stream)
    if (!e->rest_hint_dev && e->rest_hint) { hipHostFree(e->rest_hint); e->rest_hint = nullptr; }
    // (these events DO fence to system scope: the host reads the word the kernel in front of them stored to host memory)
    for (int i = 0; i < LOB_HINT_RING && e->rest_hint; i++) HIPCHK_E(hipEventCreateWithFlags(&e->hint_ev[i], hipEventDisableTiming));
    if (const char* g = getenv("LOB_MOSTLY_GENERAL")) e->force_general = g[0] == '1' ? 1 : g[0] == '0' ? 0 : -1;
    // Optional (LOB_GROUPS=2): two book groups pipelined on two streams so that the latency-bound env
    // kernel of one group runs beside a gather kernel of the other.  It paid 7 % before the market
    // track made the env kernel cheap; now one group is as fast and gives clean per-kernel timings.
    e->n_groups = 1;
    if (const char* g = getenv("LOB_GROUPS")) { int v = atoi(g); if (exps && v >= 1 && v <= 2) e->n_groups = v; }
    // experiment / test switches for the books-per-wave choice of the lane-per-book kernels (32 and 256 = env_compact_kernel: experiments)
    if (const char* g = getenv("LOB_ENV_LANES")) { int v = atoi(g); if (v == 16 || v == 64 || (exps && (v == 32 || v == 256))) e->env_lanes = v; }
    if (const char* g = getenv("LOB_TRACK_RING")) { int v = atoi(g); if (v >= 256 && v <= (1 << 20) && (v & (v - 1)) == 0) e->track_ring = v; }
    if (const char* g = getenv("LOB_TRACK_REFILL")) { int v = atoi(g); if (v >= 1) e->track_refill = v; }
    if (const char* g = getenv("LOB_ACC_LANES")) { int v = atoi(g); if (v == 8 || v == 16 || v == 32 || v == 64) e->acc_shift = v == 8 ? 3 : v == 16 ? 4 : v == 32 ? 5 : 6; }
    if (const char* g = getenv("LOB_NO_LIGHT")) e->light = !(g[0] == '1');
    if (const char* g = getenv("LOB_NO_FUSE_ACT")) e->fuse_act = !(exps && g[0] == '1');
    if (const char* g = getenv("LOB_FUSE_ACT")) e->force_fuse_act = g[0] == '1';  // (also for small batches: the tests)
    if (const char* g = getenv("LOB_Q_LANES")) e->q_lanes = g[0] == '1' ? 1 : 0;
    if (const char* g = getenv("LOB_NO_TLIGHT")) e->t_light = !(g[0] == '1');
    if (const char* g = getenv("LOB_Q_PAIR")) e->q_pair = !(g[0] == '0');
    if (const char* g = getenv("LOB_NO_FUSE")) e->no_fuse = g[0] == '1';
    if (const char* g = getenv("LOB_ENV_STEP")) e->env_step = !(exps && g[0] == '0');
    if (const char* g = getenv("LOB_INLINE_GENERAL")) e->inline_general = !(exps && g[0] == '0');
    if (const char* g = getenv("LOB_RESET_LANES")) { int v = atoi(g); if (v == 64 || (exps && (v == 16 || v == 32))) e->reset_lanes = v; }

    // ---- DevParams ----
    DevParams& P = e->P;
    memset(&P, 0, sizeof P);
    P.D = p->depth; P.T = p->max_trades; P.W = lob_rec_words(p->depth, p->max_trades); P.Wd = drec_words(p->depth, p->max_trades); P.V = p->n_vars;
    for (int i = 0; i < LOB_MAX_VARS; i++) P.vars[i] = p->vars[i];
    lobh::TickTable tt;
    lobh::build_tick_table(p->market, tt);
    P.n_bands = tt.n;
    for (int i = 0; i < LOB_MAX_BANDS; i++) { P.band_lb[i] = tt.lb[i]; P.band_tick[i] = tt.tick[i]; P.band_cum[i] = tt.cum[i]; P.band_pt[i] = tt.pt[i]; P.band_pp[i] = tt.pp[i]; }
    P.open_ms = p->market.open_ms; P.close_ms = p->market.close_ms;
    P.order_size = p->order_size; P.reward_measure = p->reward_measure;
    P.pos_lb = p->pos_lb; P.pos_ub = p->pos_ub;
    P.damping_factor = p->damping_factor; P.pos_weight = p->pos_weight; P.trd_weight = p->trd_weight; P.pnl_weight = p->pnl_weight;
    P.target_price = p->target_price; P.quote_mode = p->quote_mode;
    P.ewma_alpha = 2.0 / ((double)(size_t)p->lb_rsi + 1.0);
    P.M = p->memory_size;
    P.w0 = p->group_weights[0]; P.w1 = p->group_weights[1]; P.w2 = p->group_weights[2];
    P.gamma = p->gamma; P.alpha = p->alpha; P.epsilon = p->epsilon;
    P.policy = p->policy; P.tau = p->tau; P.beta = p->beta;
    P.trace_rate = trace_rate;
    for (int k = 0; k <= LOB_TRACE_GENS; k++) P.trace_pow[k] = trace_pow[k];
    P.trace_kmax = trace_kmax;
    P.trace_gens = P.trace_kmax <= 32 ? 32 : LOB_TRACE_GENS;
    // the average-reward agents are their base algorithm + r_learn (RLearn: Q(lambda), OnlineRLearn: SARSA, DoubleRLearn: double Q)
    P.r_learn = p->algo >= LOB_ALGO_R_LEARN;
    P.algo = p->algo == LOB_ALGO_R_LEARN ? LOB_ALGO_QLAMBDA : p->algo == LOB_ALGO_ONLINE_R_LEARN ? LOB_ALGO_SARSA
             : p->algo == LOB_ALGO_DOUBLE_R_LEARN ? LOB_ALGO_DOUBLE_Q : p->algo;
    P.theta_private = p->theta_mode == LOB_THETA_PRIVATE;
    { const char* nc = getenv("LOB_NO_CARRY"); P.carry_verdicts = !P.theta_private && !(nc && nc[0] == '1'); }
    { const char* nc = getenv("LOB_NO_COMBINE"); P.combine = !P.theta_private && !(nc && nc[0] == '1'); }
    {   // group-0 memo: shared theta, one weight vector, one book group
        const char* nc = getenv("LOB_NO_MEMO");
        P.memo = !P.theta_private && !P.r_learn && e->n_groups == 1 && !(nc && nc[0] == '1');
        const bool dq = P.algo == LOB_ALGO_DOUBLE_Q;
        if (dq && P.memo) {
            // DoubleQLearn on the fast path: through the kernels that know two weight vectors -- env_step_kernel<., true> (two trade
            // slots), learn_q_lane_kernel<LOB_ALGO_DOUBLE_Q> with the fused Watkins trace step (a lane per book: big batches, or
            // LOB_Q_LANES=1), the lane trace kernel behind it.  Anything else: the general kernels, as before.
            hipDeviceProp_t prop;
            int n_cus = e->n_cus;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
            const bool q_lanes = e->q_lanes >= 0 ? e->q_lanes == 1 : (long long)n_books >= (long long)LOB_QL_BLOCK * n_cus / 2;
            const char* dqf = exps ? getenv("LOB_DQ_FAST") : nullptr;
            P.memo = q_lanes && e->t_light && !e->no_fuse && e->light && e->fuse_act && e->env_step && e->env_lanes == 0 && P.T <= 2 && P.combine &&
                     P.trace_gens == 32 && (n_books >= 1024 || e->force_fuse_act) && !(dqf && dqf[0] == '0');
        }
        // the memo path never reads the carry-over filter, and its hot counter serialises first writes
        if (P.memo) P.carry_verdicts = 0;
        // SARSA(lambda): the trace step with a lane per generation (trace_sarsa_kernel, lob_fast.h) + the tile registry it needs
        // ... for every book of SARSA(lambda); for Q(lambda) where the lane-per-book learn kernel takes the light trace steps and
        // lists the books that keep their traces (the same condition as in run_steps: big batches; LOB_TRACE_LANES=0 switches it off)
        const char* sl = exps ? getenv("LOB_SARSA_LANES") : nullptr;
        const char* tl = getenv("LOB_TRACE_LANES");
        bool lanes_ok = P.memo && P.combine && P.trace_gens == 32 && !(sl && sl[0] == '0') && !(tl && tl[0] == '0');
        if (lanes_ok && (P.algo == LOB_ALGO_QLAMBDA || dq)) {
            hipDeviceProp_t prop;
            int n_cus = e->n_cus;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
            const bool q_lanes = e->q_lanes >= 0 ? e->q_lanes == 1 : (long long)n_books >= (long long)LOB_QL_BLOCK * n_cus / 2;
            lanes_ok = e->t_light && q_lanes && !e->no_fuse;
        } else if (P.algo != LOB_ALGO_SARSA) lanes_ok = false;
        P.sarsa_lanes = lanes_ok;
        P.epi_epoch = 0;
    }
    P.seed = p->seed; P.book_id_offset = p->book_id_offset;
    P.exp_learn_first = e->exp_learn_first ? 1 : 0;

    // ---- DevState ----
    DevState& S = e->S;
    memset(&S, 0, sizeof S);
    const size_t B = (size_t)n_books;
    S.B = n_books; S.D = P.D; S.T = P.T; S.W = P.W;
    int rc = LOB_OK;
#define X(t, n) if (rc == LOB_OK) rc = dev_alloc(e, &S.n, B);
    LOB_ENV_FIELDS(X)
#undef X
    if (rc == LOB_OK) rc = dev_alloc(e, &S.hdr, B);
    auto alloc_rm = [&](RMPtrs& r, int w) {
        r.w = w;
        if (rc == LOB_OK) rc = dev_alloc(e, &r.ring, B * w);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.cnt, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.head, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.sum, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.mean, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.s, B);
    };
    auto alloc_acc = [&](AccPtrs& r, int w) {
        r.w = w;
        if (rc == LOB_OK) rc = dev_alloc(e, &r.ring, B * w);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.cnt, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.head, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &r.sum, B);
    };
    alloc_rm(S.f_midprice, p->lb_mpm);
    alloc_rm(S.f_volatility, p->lb_vlt);
    alloc_rm(S.f_ask_tx, p->lb_svl);
    alloc_rm(S.f_bid_tx, p->lb_svl);
    alloc_rm(S.spread_window, p->lb_spread);
    alloc_rm(S.pnl_ups, p->lb_pnl);
    alloc_rm(S.pnl_downs, p->lb_pnl);
    alloc_rm(S.tp_mp, p->lb_target);
    alloc_acc(S.f_vwap_numer, p->lb_vwap);
    alloc_acc(S.f_vwap_denom, p->lb_vwap);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.meta, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.prep, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.ewma_up, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.ewma_down, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.tp_val, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.persist, B * LOB_PERSIST_N);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.k_stop, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.vars, B * 3 * 16);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.qs_last, B * LOB_N_ACTIONS);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_idx, B * (size_t)P.trace_gens * 32);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_alive, B * (size_t)P.trace_gens);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.theta, (size_t)P.M * (P.theta_private ? B : 1));
    if (rc == LOB_OK) rc = dev_alloc(e, &S.theta_nz, LOB_NZ_NWORDS(P.M) * (P.theta_private ? B : 1));
    if (P.algo == LOB_ALGO_DOUBLE_Q) {
        if (rc == LOB_OK) rc = dev_alloc(e, &S.theta_b, (size_t)P.M * (P.theta_private ? B : 1));
        if (rc == LOB_OK) rc = dev_alloc(e, &S.theta_b_nz, LOB_NZ_NWORDS(P.M) * (P.theta_private ? B : 1));
        if (rc == LOB_OK) rc = dev_alloc(e, &S.qs_last_b, B * LOB_N_ACTIONS);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mt_state, B * LOB_MT_N);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mt_idx, B);
    }
    {
        const bool rl = P.r_learn != 0;
        const size_t nr = rl ? (P.theta_private ? B : 1) : 1;
        if (rc == LOB_OK) rc = dev_alloc(e, &S.rho, nr);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.rho_inc, nr);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.rho_cnt, nr);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.rl_t, rl ? B : 1);
    }
    if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_sig, B * (size_t)P.trace_gens * 4);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_cbslot, B * (size_t)P.trace_gens);
    {
        // (slots persist while in use: room for the generations of a few steps; LOB_CB_SLOTS overrides, for the tests)
        int slots = 2048;
        while ((size_t)slots < 8 * B && slots < (1 << 22)) slots <<= 1;
        if (const char* g = getenv("LOB_CB_SLOTS")) { int v = atoi(g); if (v >= 64 && v <= (1 << 24) && (v & (v - 1)) == 0) slots = v; }
        S.cb_slots = slots;
        if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_key, (size_t)slots);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_ident, (size_t)slots * 8);
        // SARSA(lambda) adds 1.6 M terms per step to the slots' sums: a copy of the sums per XCD (0.147 -> 0.100 ms; apply_kernel
        // 0.023 -> 0.03); Q(lambda)'s few additions are a latency chain that the copies do not shorten
        // ... and so do Watkins's Q(lambda) / double Q once the policy is mostly greedy (epsilon decays over the episodes: no cut,
        // 25 generations per book, and the greedy books crowd onto a few (triple, action) pairs: 0.44 ms per step with one copy at
        // epsilon = 0.01).  One copy per XCD for every algorithm on the fast path; apply_kernel adds them up.
        S.cb_reps = P.memo ? 8 : 1;
        if (const char* g = getenv("LOB_ACC_REPS")) { int v = atoi(g); if (exps && (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64)) S.cb_reps = v; }
        if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_acc, (size_t)S.cb_reps * slots * 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_touch, (size_t)slots);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_list, 2 * (size_t)slots);
        S.cb_segs = std::min(2048, slots / 4);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_count, 2 * (size_t)S.cb_segs);
        if (rc == LOB_OK && hipMemsetAsync(S.cb_key, 0xff, (size_t)slots * 8, e->stream) != hipSuccess) rc = LOB_EHIP;
        // dense ids of the slots (lob_state.h; accumulate_dense_kernel): for the algorithms whose update can come to sum per block
        // (SARSA(lambda) always, Q(lambda) once most actions are greedy), on the fast path, big batches
        const bool dense_ok = P.memo && P.combine && e->acc_dense && e->acc_block && (P.algo == LOB_ALGO_SARSA || P.algo == LOB_ALGO_QLAMBDA) &&
                              (long long)n_books >= 4 * LOB_ACB_BLOCK && P.trace_gens <= LOB_TRACE_GENS;
        if (dense_ok) {
            if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_dense, (size_t)slots);
            if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_free, (size_t)LOB_CBD_CAP);
            if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_free_n, 16);
            if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_cbd, B * (size_t)P.trace_gens);
            if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_part, (size_t)LOB_ACD_MAX_BLOCKS * LOB_CBD_CAP);
            if (rc == LOB_OK) rc = dev_alloc(e, &S.cb_red, (size_t)LOB_ACD_GROUPS * LOB_CBD_CAP);
            if (rc == LOB_OK && hipMemsetAsync(S.cb_dense, 0xff, (size_t)slots * 4, e->stream) != hipSuccess) rc = LOB_EHIP;
            if (rc == LOB_OK && hipMemsetAsync(S.tr_cbd, 0xff, B * (size_t)P.trace_gens * 8, e->stream) != hipSuccess) rc = LOB_EHIP;
            if (rc == LOB_OK) {
                // list x: the ids = x mod 8, the lowest on top; nothing handed out yet
                // (LOB_CBD_IDS=n: only n ids in all, for the tests -- most slots then go without one)
                std::vector<i32> fl(LOB_CBD_CAP), fn(16);
                const int per = LOB_CBD_CAP / 8;
                int have = per;
                if (const char* g = getenv("LOB_CBD_IDS")) { const int v = atoi(g); if (v >= 8 && v <= LOB_CBD_CAP) have = v / 8; }
                S.cb_ids = have * 8;
                for (int x = 0; x < 8; x++) {
                    for (int i = 0; i < have; i++) fl[(size_t)x * per + i] = (have - 1 - i) * 8 + x;
                    fn[2 * x] = have;
                    fn[2 * x + 1] = have;
                }
                if (hipMemcpyAsync(S.cb_free, fl.data(), fl.size() * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
                    hipMemcpyAsync(S.cb_free_n, fn.data(), fn.size() * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
                    hipStreamSynchronize(e->stream) != hipSuccess) rc = LOB_EHIP;
            }
            if (rc == LOB_OK && hipFuncSetAttribute((const void*)accumulate_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acd_lds_bytes()) != hipSuccess) {
                lob_set_error("hipFuncSetAttribute(accumulate_dense_kernel: dynamic LDS)");
                rc = LOB_EHIP;
            }
        }
    }
    {
        S.mk_slots = 1 << 16;
        const size_t ms = (size_t)S.mk_slots;
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_hash, ms);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_ident, ms * 4);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_stamp, ms);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_list, 2 * ms);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_count, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_rec, 2 * ms * LOB_MK_REC);
        if (P.memo && P.algo == LOB_ALGO_DOUBLE_Q && rc == LOB_OK) rc = dev_alloc(e, &S.mk_rec_b, 2 * ms * LOB_MK_REC);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_tiles, P.memo ? ms * LOB_N_ACTIONS * 32 : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_tiles_ok, ms);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_marked, ms);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_marklist, ms);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_markcount, 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_list, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_list_n, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_list2, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_list2_n, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.acc_list, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.acc_list_n, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.acc_pend, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.dir_list, P.combine ? B * (size_t)P.trace_gens : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.dir_list_n, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_slot, B);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_slot_last, B);
        if (rc == LOB_OK && hipMemsetAsync(S.mk_hash, 0xff, ms * 8, e->stream) != hipSuccess) rc = LOB_EHIP;
        if (rc == LOB_OK && hipMemsetAsync(S.mk_slot, 0xff, B * 4, e->stream) != hipSuccess) rc = LOB_EHIP;
        if (rc == LOB_OK && hipMemsetAsync(S.mk_slot_last, 0xff, B * 4, e->stream) != hipSuccess) rc = LOB_EHIP;
        // tile registry of the SARSA lane path (lob_state.h)
        const bool sl = P.sarsa_lanes != 0;
        S.ow_slots = sl ? 1 << 25 : 1;  // 65 536 memo slots x 288 tiles at most: load <= 0.56
        S.amb_cap = sl ? (int)std::min<long long>((long long)P.M, 1 << 18) : 1;
        // (LOB_OW_SLOTS / LOB_AMB_CAP: tiny values for the tests -- a registry without room leaves slots unregistered, a full list
        // of new ambiguous indices switches the lane kernel off until the next reset: every book then goes to the wave-per-book kernel)
        if (sl) if (const char* g = getenv("LOB_OW_SLOTS")) { int v = atoi(g); if (v >= 64 && v <= (1 << 26) && (v & (v - 1)) == 0) S.ow_slots = v; }
        if (sl) if (const char* g = getenv("LOB_AMB_CAP")) { int v = atoi(g); if (v >= 1 && v <= (1 << 18)) S.amb_cap = v; }
        if (rc == LOB_OK) rc = dev_alloc(e, &S.tr_mslot, sl ? B * (size_t)P.trace_gens : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.ow_tab, (size_t)S.ow_slots);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.amb_bits, sl ? (size_t)P.M / 32 + 1 : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.amb_new, 2 * (size_t)S.amb_cap);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.amb_new_n, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.amb_flag, 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_amb, sl ? ms * LOB_N_ACTIONS : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_all, sl ? ms : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.mk_all_n, 1);
        if (rc == LOB_OK && hipMemsetAsync(S.tr_mslot, 0xff, (sl ? B * (size_t)P.trace_gens : 1) * 4, e->stream) != hipSuccess) rc = LOB_EHIP;
        if (rc == LOB_OK && hipMemsetAsync(S.ow_tab, 0xff, (size_t)S.ow_slots * 8, e->stream) != hipSuccess) rc = LOB_EHIP;
    }
    {
        // coarse written-weights map of the fast path: the finest granularity whose image fits 80 KB of LDS
        int cs = 5;
        while ((((size_t)P.M >> cs) + 31) / 32 > 20480) cs++;
        P.cshift = cs;
        const size_t cwords = (((size_t)P.M >> cs) + 32) / 32;
        P.cwords4 = (int)((cwords + 3) / 4);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.theta_nzx, P.memo ? (size_t)P.M / 32 + 1 : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.theta_nzc, (size_t)P.cwords4 * 4);
        if (P.memo && rc == LOB_OK) rc = dev_alloc(e, &S.theta_nzd, 2 * ((size_t)P.M / 32 + 1));
        if (rc == LOB_OK) rc = dev_alloc(e, &S.slow_list, 2 * B);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.slow_n, 4);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.hl_rec, P.memo ? (size_t)LOB_HL_REC * B : 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &S.hl_dirty, 1);
        if (rc == LOB_OK && hipMemsetAsync(S.hl_rec, 0xff, (P.memo ? (size_t)LOB_HL_REC * B : 1) * 8, e->stream) != hipSuccess) rc = LOB_EHIP;
        if (rc == LOB_OK && hipMemsetAsync(S.hl_dirty, 0xff, 4, e->stream) != hipSuccess) rc = LOB_EHIP;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cus = prop.multiProcessorCount;
        if (P.memo) {
            const int q_lds = (int)fast_lds_bytes(P.cwords4, LOB_FAST_NB, false), tr_lds = (int)trace_lds_bytes();
            hipError_t er = hipFuncSetAttribute((const void*)act_fast_kernel<LOB_FAST_NB>, hipFuncAttributeMaxDynamicSharedMemorySize, q_lds);
            if (er == hipSuccess) er = lobk_learn_set_lds(q_lds, (int)qlane_lds_bytes(P.cwords4), (int)qpair_lds_bytes(P.cwords4));
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_fast_kernel<LOB_ALGO_SARSA, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_fast_kernel<LOB_ALGO_SARSA, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_fast_kernel<LOB_ALGO_QLAMBDA, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_fast_kernel<LOB_ALGO_QLAMBDA, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_fast_kernel<LOB_ALGO_QLAMBDA, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_rest_kernel<LOB_ALGO_QLAMBDA>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er == hipSuccess) er = hipFuncSetAttribute((const void*)trace_rest_kernel<LOB_ALGO_DOUBLE_Q>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds);
            if (er != hipSuccess && rc == LOB_OK) { lob_set_error(std::string("hipFuncSetAttribute(dynamic LDS): ") + hipGetErrorString(er)); rc = LOB_EHIP; }
        }
    }
    if (rc == LOB_OK) rc = dev_alloc(e, &S.verdict, B * LOB_VD_STRIDE);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.nz_new, 4 * LOB_NZ_WORDS);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.verdict_b, P.algo == LOB_ALGO_DOUBLE_Q ? B * 64 : 1);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.nz_epoch, 1);
    // (striped: lob_state.h cnt_add; [8]: generations apply_kernel applied from trace_rest_kernel's list, lob_debug_deferred)
    if (rc == LOB_OK) rc = dev_alloc(e, &S.counters, (size_t)LOB_CNT_STRIPES * LOB_CNT_STRIDE);
    if (rc == LOB_OK) rc = dev_alloc(e, &e->cnt_sum, 16);
    if (rc == LOB_OK) rc = dev_alloc(e, &S.error_flag, 1);
#ifdef LOB_PROF
    if (rc == LOB_OK) rc = dev_alloc(e, &S.prof, B * LOB_PROF_N);
#endif
    if (rc == LOB_OK) rc = dev_alloc(e, &e->rnd_dev, 2048 + 64);
    S.nzd_terms = e->rnd_dev ? e->rnd_dev + 2048 + LOB_N_ACTIONS : nullptr;  // term[1][.], term[2][.] (filled below)
    if (rc == LOB_OK) rc = dev_alloc(e, &e->actions_dev, B);
    if (rc == LOB_OK) rc = dev_alloc(e, &e->P_dev, 1);
    if (rc == LOB_OK) rc = dev_alloc(e, &e->S_dev, 1);
    if (rc == LOB_OK) { S.self = e->S_dev; memset(&e->S_pushed, 0xff, sizeof(DevState)); }
    if (rc == LOB_OK) rc = push_params(e);
    if (rc != LOB_OK) { lob_destroy(e); return rc; }
    // the two rl::State objects start with constructor zeros (src/rl/state.cpp:10-19)
    {
        std::vector<f64> neg1(B, -1.0);
        HIPCHK_E(hipMemcpyAsync(S.tp_val, neg1.data(), B * sizeof(f64), hipMemcpyHostToDevice, e->stream));
        HIPCHK_E(hipStreamSynchronize(e->stream));
        std::vector<LHdr> hdr(B);
        memset(hdr.data(), 0, B * sizeof(LHdr));
        for (size_t b = 0; b < B; b++) hdr[b].zero_mask = 3;
        HIPCHK_E(hipMemcpyAsync(S.hdr, hdr.data(), B * sizeof(LHdr), hipMemcpyHostToDevice, e->stream));
        // hash table followed by the 27 trailing-coordinate (action code) terms
        // rndseq[(code + 449*(nf+1)) & 2047], code = group*9 + action (state.cpp:56-63, tiles.cpp:46,152-163)
        uint32_t rnd[2048 + 64];
        memset(rnd, 0, sizeof rnd);
        make_rndseq(rnd);
        // (reduced mod M: hash_UNH's single final reduction distributes over the sum)
        uint32_t* terms = rnd + 2048;
        std::vector<uint32_t> raw(rnd, rnd + 2048);  // the action terms below index the unreduced table
        for (int g = 0; g < 3; g++) {
            const int nf = g == 0 ? 3 : (g == 1 ? P.V - 3 : P.V);
            for (int a = 0; a < LOB_N_ACTIONS; a++)
                terms[g * LOB_N_ACTIONS + a] = (uint32_t)((uint64_t)raw[((g * LOB_N_ACTIONS + a) + 449 * (nf + 1)) & 2047] % (uint64_t)P.M);
        }
        for (int k = 0; k < 2048; k++) rnd[k] = (uint32_t)((uint64_t)rnd[k] % (uint64_t)P.M);  // the device only ever sums the table mod M
        HIPCHK_E(hipMemcpyAsync(e->rnd_dev, rnd, sizeof rnd, hipMemcpyHostToDevice, e->stream));
        HIPCHK_E(hipStreamSynchronize(e->stream));
    }
    if (P.algo == LOB_ALGO_DOUBLE_Q) {
        hipLaunchKernelGGL(mt_init_kernel, dim3(grid_lanes(e->B)), dim3(256), 0, e->stream, e->P, e->S);
        HIPCHK_E(hipGetLastError());
        HIPCHK_E(hipStreamSynchronize(e->stream));
    }
#undef HIPCHK_E
    if (p->random_init) {
        const int rc_init = theta_random_init(e);
        if (rc_init != LOB_OK) { lob_destroy(e); return rc_init; }
    }
    *out = e;
    return LOB_OK;
}

void lob_destroy(lob_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stage_thread.joinable()) e->stage_thread.join();
    if (e->stream_up) { hipStreamSynchronize(e->stream_up); hipStreamDestroy(e->stream_up); }
    if (e->records_next) hipFree(e->records_next);
    if (e->stream2) hipStreamSynchronize(e->stream2);
    if (e->stream) hipStreamSynchronize(e->stream);
    drain_timers(e);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    if (e->ev_stagger) hipEventDestroy(e->ev_stagger);
    if (e->rest_hint) hipHostFree(e->rest_hint);
    for (int i = 0; i < LOB_HINT_RING; i++) if (e->hint_ev[i]) hipEventDestroy(e->hint_ev[i]);
    if (e->ev_rest_go) hipEventDestroy(e->ev_rest_go);
    if (e->ev_rest_done) hipEventDestroy(e->ev_rest_done);
    if (e->stream2) hipStreamDestroy(e->stream2);
    for (auto ev : e->event_pool) hipEventDestroy(ev);
    for (void* p : e->allocs) hipFree(p);
    if (e->records_dev) hipFree(e->records_dev);
    if (e->phase_dev) hipFree(e->phase_dev);
    if (e->track_dev) hipFree(e->track_dev);
    if (e->dump_dev) hipFree(e->dump_dev);
    if (e->spx_gather) hipFree(e->spx_gather);
    if (e->spx_ev) hipEventDestroy(e->spx_ev);
    if (e->spx_total_host) hipHostFree(e->spx_total_host);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

// learning.random_init (src/rl/agent.cpp:37-39, DoubleAgent: 190-192): generate(&theta[0], &theta[M], 2.0 * unif_dist(gen) - 1.0)
// with the agent's own std::mt19937_64 -- theta first, then theta_b, and the generator goes on to toss DoubleQLearn's coin from
// where the initialisation left it.  Done on the host (a sequential generator: 20 M draws take a tenth of a second, once per
// engine) with the same restatement of the generator the device uses for the coin (lob_learn.h), loaded through lob_theta_set
// (maps, memo records).  Private theta: every book's agent draws its own vectors.  Shared theta: global book 0's agent draws the
// one vector, on every shard the same; its generator's new state is installed where that book lives (book_id_offset == 0).
struct HostMt64 {
    u64 x[LOB_MT_N];
    int idx;
    explicit HostMt64(u64 seed) { mt64_seed(x, seed); idx = LOB_MT_N; }
    u64 draw() {
        if (idx >= LOB_MT_N) {  // the sequential in-place twist (the device does the same block in three phases, mt64_twist_wave)
            for (int i = 0; i < LOB_MT_N; i++) x[i] = mt64_mix(x[i], x[(i + 1) % LOB_MT_N], x[(i + LOB_MT_M) % LOB_MT_N]);
            idx = 0;
        }
        return mt64_temper(x[idx++]);
    }
};
static int theta_random_init(lob_engine* e) {
    const bool priv = e->P.theta_private != 0, dq = e->P.algo == LOB_ALGO_DOUBLE_Q;
    const int nt = priv ? e->B : 1;
    std::vector<f64> v((size_t)e->P.M);
    for (int t = 0; t < nt; t++) {
        HostMt64 g((u64)(uint32_t)(e->P.seed + (priv ? e->P.book_id_offset + (u64)t : 0ull)));
        for (int vec = 0; vec < (dq ? 2 : 1); vec++) {
            for (size_t i = 0; i < v.size(); i++) v[i] = 2.0 * mt64_canonical(g.draw()) - 1.0;
            int rc = lob_theta_set(e, vec * nt + t, v.data(), (int64_t)e->P.M);
            if (rc) return rc;
        }
        if (dq && (priv || e->P.book_id_offset == 0)) {  // Agent::gen of the book whose agent has just drawn: where the coin continues
            const i32 idx = g.idx;
            HIPCHK(hipMemcpyAsync(e->S.mt_state + (size_t)t * LOB_MT_N, g.x, sizeof g.x, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->S.mt_idx + t, &idx, sizeof idx, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
    }
    return LOB_OK;
}

static int finalize_episode(lob_engine* e) {
    if (!e->episode_open) return LOB_OK;
    lobk_finalize(e->stream, e->P.T <= 2, (const DevParams*)e->P_dev, e->S);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    e->episode_open = false;
    return LOB_OK;
}

static int stage_join(lob_engine* e);
// n_rows: records to allocate (B * n_events for per-book streams, the stream length for a replayed one)
static int set_records(lob_engine* e, int32_t n_events, size_t n_rows) {
    // (the TickStatistics counters of a book are 32-bit like the reference's ints, lob_state.h tick_ab / tick_pos / tick_both: one count
    // per agent step, at most one agent step per event, n_events < 2^31 -- a recorded day of millions of rows is fine)
    { int rc = finalize_episode(e); if (rc) return rc; }  // needs the old stream
    if (e->stage_active) { (void)stage_join(e); e->stage_active = false; }   // (a staged stream is void once another one is loaded)
    if (e->records_next) { hipFree(e->records_next); e->records_next = nullptr; }
    // the old stream is gone from here on, whatever happens below: no kernel may see a freed pointer
    e->have_events = false;
    e->was_reset = false;
    e->S.records = nullptr;
    e->S.track = nullptr;
    e->S.rec_phase = nullptr;
    e->S.n_events = 0;
    if (e->phase_dev) { hipFree(e->phase_dev); e->phase_dev = nullptr; }
    const size_t bytes = n_rows * e->P.Wd * 4 + 256;  // (tail pad: drec_levels reads whole 16-byte quads past a short level array)
    // market track: resident (one entry per event) up to track_ring events per book, a ring of that many beyond
    e->chunked = n_events > e->track_ring;
    e->S.track_len = e->chunked ? e->track_ring : n_events;
    e->S.track_mask = e->chunked ? e->track_ring - 1 : 0x7fffffff;
    const size_t track_bytes = (size_t)e->B * e->S.track_len * sizeof(Track);
    // (a stream of the size of the one before it -- a fresh day per episode -- moves into the buffers that are there: freeing and
    // allocating 31 + 18 GB cost more than the copy of a whole stream, round 6)
    hipError_t err = hipSuccess;
    if (!e->records_dev || e->records_bytes != bytes) {
        if (e->records_dev) { hipFree(e->records_dev); e->records_dev = nullptr; }
        err = hipMalloc((void**)&e->records_dev, bytes);
        if (err != hipSuccess) { e->records_dev = nullptr; e->records_bytes = 0; lob_set_error("hipMalloc(records) failed"); return LOB_ENOMEM; }
        e->records_bytes = bytes;
    }
    if (!e->track_dev || e->track_bytes != track_bytes) {
        if (e->track_dev) { hipFree(e->track_dev); e->track_dev = nullptr; }
        err = hipMalloc((void**)&e->track_dev, track_bytes);
        if (err != hipSuccess) {
            e->track_dev = nullptr; e->track_bytes = 0;
            hipFree(e->records_dev);
            e->records_dev = nullptr; e->records_bytes = 0;
            lob_set_error("hipMalloc(track) failed");
            return LOB_ENOMEM;
        }
        e->track_bytes = track_bytes;
    }
    e->S.track = e->track_dev;
    e->S.records = e->records_dev;
    e->S.n_events = n_events;
    e->have_events = true;
    e->was_reset = false;
    return LOB_OK;
}

// A few host threads that copy one piece into a pinned staging buffer together, started once per upload (round 5 started and joined
// up to seven per 64 MB piece: ~2 900 thread creations for 26.6 GB).  std::thread's constructor throws when the process may not
// start another thread; the pool then runs with the workers it got -- none at all is fine, the caller's thread copies alone.
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    char* dst = nullptr;
    const char* src = nullptr;
    size_t nb = 0;
    unsigned long long gen = 0;
    unsigned pending = 0;
    bool quit = false;
    unsigned parts() const { return (unsigned)th.size() + 1; }
    explicit CopyPool(unsigned want) {
        try {
            for (unsigned t = 1; t < want; t++) th.emplace_back([this, t] { worker(t); });
        } catch (const std::system_error&) {
        }
    }
    void worker(unsigned t) {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            char* d = dst; const char* s = src; const size_t n = nb; const unsigned np = parts();
            lk.unlock();
            if (t < np) { const size_t a = n * t / np, b = n * (t + 1) / np; memcpy(d + a, s + a, b - a); }
            lk.lock();
            if (--pending == 0) cv_done.notify_one();
        }
    }
    void copy(char* d, const char* s, size_t n) {
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = d; src = s; nb = n; pending = (unsigned)th.size(); gen++;
        }
        cv_go.notify_all();
        memcpy(d, s, n / parts());
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
};

// host ABI records -> HBM in the device layout.  Small uploads: one pageable copy into a temporary device buffer, one repack.
// Big ones (the 26.6 GB of 65 536 private streams): 64 MB pieces through two pinned staging buffers -- a few host threads copy
// piece k + 1 into its buffer while the DMA engine and the repack kernel are on piece k -- so the hand-over runs at what the
// slower of the host's memcpy and the link gives instead of the runtime's pageable path, and the temporary device copy is two
// pieces, not another whole stream.  Without pinned memory (a locked-memory limit; LOB_UPLOAD_PINNED=0 forces it, for the tests)
// the same pieces go from the caller's pageable memory directly: slower, and still two pieces of device memory, never a second
// whole stream.
static int upload_records(lob_engine* e, const uint32_t* host_records, size_t n_records, hipStream_t st = nullptr, uint32_t* dst = nullptr) {
    if (!st) st = e->stream;
    if (!dst) dst = e->records_dev;
    const size_t rec_bytes = (size_t)e->P.W * 4;
    const size_t bytes = n_records * rec_bytes;
    size_t piece_recs = std::max<size_t>(1, ((size_t)64 << 20) / rec_bytes);
    if (const char* g = getenv("LOB_UPLOAD_PIECE_RECS")) { const long v = atol(g); if (v >= 1) piece_recs = (size_t)v; }  // (tests: many small pieces)
    if (n_records <= 2 * piece_recs) {
        uint32_t* tmp = nullptr;
        if (hipMalloc((void**)&tmp, bytes) != hipSuccess) { lob_set_error("hipMalloc(upload buffer) failed"); return LOB_ENOMEM; }
        hipError_t err = hipMemcpyAsync(tmp, host_records, bytes, hipMemcpyHostToDevice, st);
        if (err == hipSuccess) {
            lobk_repack(st, (const uint32_t*)tmp, e->P.D, e->P.T, n_records, dst);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        hipFree(tmp);
        if (err != hipSuccess) { lob_set_error(std::string("record upload: ") + hipGetErrorString(err)); return LOB_EHIP; }
        return LOB_OK;
    }
    const size_t piece_bytes = piece_recs * rec_bytes;
    void* pinned[2] = {nullptr, nullptr};
    uint32_t* tmp[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    auto release = [&]() {
        for (int i = 0; i < 2; i++) {
            if (done[i]) hipEventDestroy(done[i]);
            if (tmp[i]) hipFree(tmp[i]);
            if (pinned[i]) hipHostFree(pinned[i]);
        }
    };
    hipError_t err = hipSuccess;
    for (int i = 0; i < 2 && err == hipSuccess; i++) {
        err = hipMalloc((void**)&tmp[i], piece_bytes);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    if (err != hipSuccess) { release(); lob_set_error("hipMalloc(upload pieces) failed"); return LOB_ENOMEM; }
    bool use_pinned = true;
    if (const char* g = getenv("LOB_UPLOAD_PINNED")) use_pinned = !(g[0] == '0');
    for (int i = 0; i < 2 && use_pinned; i++)
        if (hipHostMalloc(&pinned[i], piece_bytes, hipHostMallocDefault) != hipSuccess) {   // (no pinned memory to be had: the pageable pieces below)
            (void)hipGetLastError();
            pinned[i] = nullptr;
            use_pinned = false;
        }
    if (!use_pinned) for (int i = 0; i < 2; i++) if (pinned[i]) { hipHostFree(pinned[i]); pinned[i] = nullptr; }
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : nt > 8 ? 8 : nt;
    int rc = LOB_OK;
    try {
        CopyPool pool(use_pinned ? nt : 1);
        const size_t n_pieces = (n_records + piece_recs - 1) / piece_recs;
        for (size_t k = 0; k < n_pieces && err == hipSuccess; k++) {
            const int i = (int)(k & 1);
            const size_t r0 = k * piece_recs, nr = std::min(piece_recs, n_records - r0), nb = nr * rec_bytes;
            if (k >= 2) err = hipEventSynchronize(done[i]);  // (piece k - 2 has left this staging buffer and its device copy)
            if (err != hipSuccess) break;
            const char* src = reinterpret_cast<const char*>(host_records) + r0 * rec_bytes;
            if (use_pinned) {
                pool.copy(reinterpret_cast<char*>(pinned[i]), src, nb);
                err = hipMemcpyAsync(tmp[i], pinned[i], nb, hipMemcpyHostToDevice, st);
            } else {
                err = hipMemcpyAsync(tmp[i], src, nb, hipMemcpyHostToDevice, st);   // (pageable: the runtime stages it, synchronously)
            }
            if (err == hipSuccess) {
                lobk_repack(st, (const uint32_t*)tmp[i], e->P.D, e->P.T, nr, dst + r0 * (size_t)e->P.Wd);
                err = hipGetLastError();
            }
            if (err == hipSuccess) err = hipEventRecord(done[i], st);
        }
    } catch (const std::exception& ex) {   // (nothing may cross the C ABI)
        lob_set_error(std::string("record upload: ") + ex.what());
        rc = LOB_ENOMEM;
    }
    const hipError_t err2 = hipStreamSynchronize(st);
    if (err == hipSuccess) err = err2;
    release();
    if (rc != LOB_OK) return rc;
    if (err != hipSuccess) { lob_set_error(std::string("record upload: ") + hipGetErrorString(err)); return LOB_EHIP; }
    return LOB_OK;
}

int lob_load_events(lob_engine* e, const uint32_t* host_records, int32_t n_events) {
    if (!e || !host_records || n_events < 2) { lob_set_error("lob_load_events: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    int rc = lob_validate_stream(host_records, e->P.D, e->P.T, e->B, n_events);
    if (rc != LOB_OK) return rc;
    rc = set_records(e, n_events, (size_t)e->B * n_events);
    if (rc != LOB_OK) return rc;
    e->have_events = false;  // (a failed upload leaves no stream behind)
    rc = upload_records(e, host_records, (size_t)e->B * n_events);
    e->have_events = rc == LOB_OK;
    return rc;
}

static int stage_join(lob_engine* e) {
    if (e->stage_thread.joinable()) e->stage_thread.join();
    if (e->stage_rc != LOB_OK) lob_set_error(e->stage_err);
    return e->stage_rc;
}
int lob_stage_events(lob_engine* e, const uint32_t* host_records, int32_t n_events) {
    if (!e || !host_records) { lob_set_error("lob_stage_events: bad argument"); return LOB_EINVAL; }
    if (!e->have_events || e->S.rec_phase || n_events != e->S.n_events) {
        lob_set_error("lob_stage_events: needs a loaded per-book stream of the same length (lob_load_events first)");
        return LOB_ESTATE;
    }
    if (e->stage_active) { lob_set_error("lob_stage_events: the stream staged before has not been adopted (lob_reset) or waited for"); return LOB_ESTATE; }
    HIPCHK(hipSetDevice(e->device));
    const size_t n_rows = (size_t)e->B * n_events;
    if (!e->records_next) {
        if (hipMalloc((void**)&e->records_next, n_rows * e->P.Wd * 4 + 256) != hipSuccess) { e->records_next = nullptr; lob_set_error("hipMalloc(second record buffer) failed"); return LOB_ENOMEM; }
    }
    if (!e->stream_up) HIPCHK(hipStreamCreateWithFlags(&e->stream_up, hipStreamNonBlocking));
    e->stage_rc = LOB_OK;
    e->stage_err.clear();
    e->stage_active = true;
    try {
        e->stage_thread = std::thread([e, host_records, n_events, n_rows] {
            int rc = hipSetDevice(e->device) == hipSuccess ? LOB_OK : LOB_EHIP;
            if (rc == LOB_OK) rc = lob_validate_stream(host_records, e->P.D, e->P.T, e->B, n_events);
            if (rc == LOB_OK) rc = upload_records(e, host_records, n_rows, e->stream_up, e->records_next);
            e->stage_rc = rc;
            if (rc != LOB_OK) e->stage_err = lob_last_error();   // (the error text is the failing thread's: handed to whoever waits)
        });
    } catch (const std::system_error&) {   // no thread to be had: the hand-over happens here and now
        int rc = lob_validate_stream(host_records, e->P.D, e->P.T, e->B, n_events);
        if (rc == LOB_OK) rc = upload_records(e, host_records, n_rows, e->stream_up, e->records_next);
        e->stage_rc = rc;
        if (rc != LOB_OK) e->stage_err = lob_last_error();
    }
    return LOB_OK;
}
int lob_stage_wait(lob_engine* e) {
    if (!e) return LOB_EINVAL;
    if (!e->stage_active) { lob_set_error("lob_stage_wait: nothing staged"); return LOB_ESTATE; }
    return stage_join(e);
}
// (lob_reset) the staged stream becomes the current one: the buffers change places
static int stage_adopt(lob_engine* e) {
    if (!e->stage_active) return LOB_OK;
    const int rc = stage_join(e);
    e->stage_active = false;
    if (rc != LOB_OK) return rc;
    std::swap(e->records_dev, e->records_next);
    e->S.records = e->records_dev;
    return LOB_OK;
}

int lob_load_events_shared(lob_engine* e, const uint32_t* host_records, int64_t n_total, const int64_t* phase, int32_t n_events) {
    if (!e || !host_records || !phase || n_events < 2 || n_total < n_events || n_total > INT32_MAX) {
        lob_set_error("lob_load_events_shared: bad argument");
        return LOB_EINVAL;
    }
    for (int b = 0; b < e->B; b++)
        if (phase[b] < 0 || phase[b] + n_events > n_total) {
            lob_set_error("lob_load_events_shared: phase[" + std::to_string(b) + "] + n_events runs past the stream");
            return LOB_EINVAL;
        }
    HIPCHK(hipSetDevice(e->device));
    int rc = lob_validate_stream(host_records, e->P.D, e->P.T, 1, (int32_t)n_total);
    if (rc != LOB_OK) return rc;
    rc = set_records(e, n_events, (size_t)n_total);
    if (rc != LOB_OK) return rc;
    // until the phases are in place the buffer (n_total rows) must not pass for a per-book stream (B * n_events rows)
    e->have_events = false;
    hipError_t err = hipMalloc((void**)&e->phase_dev, (size_t)e->B * sizeof(i64));
    if (err != hipSuccess) { e->phase_dev = nullptr; lob_set_error("hipMalloc(phase) failed"); return LOB_ENOMEM; }
    rc = upload_records(e, host_records, (size_t)n_total);
    if (rc != LOB_OK) return rc;
    HIPCHK(hipMemcpyAsync(e->phase_dev, phase, (size_t)e->B * sizeof(i64), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->have_events = true;
    e->S.rec_phase = e->phase_dev;
    return LOB_OK;
}

int lob_gen_events_device(lob_engine* e, const lob_gen_params* g) {
    if (!e || !g || g->n_events < 2) { lob_set_error("lob_gen_events_device: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    int rc = set_records(e, g->n_events, (size_t)e->B * g->n_events);
    if (rc != LOB_OK) return rc;
    lobk_gen_events(e->stream, *g, e->P.D, e->P.T, e->P.book_id_offset, e->B, e->records_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}

// env_kernel with 64 books per wave, or 16 when the batch is too small to give every SIMD a wave
static void launch_env(lob_engine* e, hipStream_t st, const i32* actions, int count_updates, int b0, int nb, int par) {
    // (LOB_ENV_LANES: 16 | 64 forced; 32, and 256 = the event loop compacted across 256-book blocks, in experiment builds)
    const int force = e->env_lanes;
    const int lanes = force ? force : (e->B <= 16384 ? 16 : 64);
    lobk_env(st, lanes, e->P.T <= 2, (const DevParams*)e->P_dev, e->S, actions, count_updates, b0, nb, e->step_id, par);
}
// Long streams: let the pre-pass run on every `track_refill` steps (lob_kernels.h prepass_extend_kernel)
static void maybe_refill_track(lob_engine* e) {
    if (!e->chunked || ++e->steps_since_fill < e->track_refill) return;
    e->steps_since_fill = 0;
    TimedLaunch t(e, "prepass_extend_kernel", nullptr, true);
    lobk_prepass_extend(e->stream, e->P.T <= 2, e->prepass_roles, (const DevParams*)e->P_dev, e->S);
}
// S0 of every group-0 triple on this step's list (lob_kernels.h memo_kernel): `which` 0 = for learn_kernel
// (theta_t), 1 = for the next act_kernel (after the update)
static void launch_memo(lob_engine* e, int par, int which, int reset_lpar = -1) {
    TimedLaunch t(e, "memo_kernel");
    hipLaunchKernelGGL(memo_kernel, dim3(256), dim3(256), 0, e->stream, e->P, e->S, (const uint32_t*)e->rnd_dev, par, which, (u64)e->theta_ver, reset_lpar);
}
// The learner step's action selection inside the env kernel (env_kernel<64, TM, 1>), the general act kernel for the books it
// leaves on the work list, and their steps (env_kernel<64, TM, 2>).
static void launch_env_fused(lob_engine* e, hipStream_t st, int par, int lpar, u64 ver, const i32* act_list, const i32* act_n, int gl) {
    const DevParams* Pd = (const DevParams*)e->P_dev;
    const bool t2 = e->P.T <= 2;
    const int nb = e->B, sid = e->step_id;
    const EnvFuse F1{nullptr, nullptr, lpar, sid - 1, ver}, F2{act_list, act_n, lpar, sid - 1, ver};
    // Two trade slots per record: env_step_kernel (lob_envstep.h; LOB_ENV_STEP=0: env_kernel<64, 2, 1>).  It serves the books
    // without a usable hit list itself (act_book in-kernel) -- except in the first step on lists after they were void, when a late
    // map bit of the step before has voided them again and EVERY book takes the general path: that step goes through
    // the work list to the general act kernel and env_kernel<64, 2, 2> as before (either version is correct for any step).
    // ... and while MOST books have no usable list (a dense weight vector -- learning.random_init, a loaded checkpoint: every tile
    // lies on a written weight and no list fits a record): one wave per book in the general act kernel is far better than a
    // 64-lane wave going through its books one at a time.  The learn kernels hand the same books back: their list's length
    // of LOB_HINT_LAG steps ago (lob_engine::rest_hint) tells.
    const bool mostly_general = e->force_general >= 0 ? e->force_general == 1 : e->hint_now > e->B / 16;
    const bool half_waves = e->env_step_lanes == 32 && t2 && e->env_step && e->P.algo != LOB_ALGO_DOUBLE_Q;  // (experiment: LOB_ENV_STEP_LANES=32)
    const bool inline_general = t2 && e->env_step && e->inline_general && e->steps_on_lists >= 1 && !mostly_general && !half_waves;
    e->steps_on_lists++;
    e->flow[inline_general ? 4 : mostly_general ? 5 : 6]++;
    const bool dq = e->P.algo == LOB_ALGO_DOUBLE_Q;  // (then t2 && env_step: lob_create)
    {
        TimedLaunch t(e, "env_kernel", st);
        // (batches up to env16_max books: a book's levels across 16 lanes, four books per wave -- env_step16_kernel)
        const bool lanes16 = !dq && nb <= e->env16_max;
        if (t2 && e->env_step) lobk_env_step(st, inline_general, dq, half_waves, lanes16, Pd, e->S, nb, sid, par, F1, (const uint32_t*)e->rnd_dev);
        else lobk_env_mode(st, t2, 1, Pd, e->S, nb, sid, par, F1);
    }
    if (inline_general) return;
    {
        TimedLaunch t(e, "act_rest_kernel", st);
        if (dq) hipLaunchKernelGGL((act_kernel<LOB_ALGO_DOUBLE_Q, true>), dim3(gl), dim3(LOB_BLOCK), 0, st, LOB_PS(e), (const uint32_t*)e->rnd_dev, 0, 0, e->B, par, act_list, act_n);
        else hipLaunchKernelGGL((act_kernel<LOB_ALGO_SARSA, true>), dim3(gl), dim3(LOB_BLOCK), 0, st, LOB_PS(e), (const uint32_t*)e->rnd_dev, 0, 0, e->B, par, act_list, act_n);
    }
    {
        TimedLaunch t(e, "env_rest_kernel", st);
        lobk_env_mode(st, t2, 2, Pd, e->S, nb, sid, par, F2);
    }
}

int lob_reset(lob_engine* e) {
    if (!e) return LOB_EINVAL;
    if (!e->have_events) { lob_set_error("lob_reset: no event stream loaded"); return LOB_ESTATE; }
    HIPCHK(hipSetDevice(e->device));
    { int rc = finalize_episode(e); if (rc) return rc; }   // (the window sums of the episode that ends: on the stream it ran on)
    { int rc = stage_adopt(e); if (rc) return rc; }        // a stream handed over meanwhile (lob_stage_events) is this episode's
    { int rc = sync_state(e); if (rc) return rc; }
    // the memo table starts empty every episode (reset_kernel voids every book's slot)
    HIPCHK(hipMemsetAsync(e->S.mk_hash, 0xff, (size_t)e->S.mk_slots * 8, e->stream));
    HIPCHK(hipMemsetAsync(e->S.mk_tiles_ok, 0, (size_t)e->S.mk_slots * 4, e->stream));
    HIPCHK(hipMemsetAsync(e->S.mk_marked, 0, (size_t)e->S.mk_slots * 4, e->stream));
    // ... and so do its lists: a lob_theta_set / lob_delta_apply between this reset and the first step runs
    // memo_kernel over the `last_par` list, which must not hold the slots of the episode before (their hashes
    // are gone: re-stamping mk_tiles_ok for a stale triple would hand a later claimant of the slot the wrong tiles)
    HIPCHK(hipMemsetAsync(e->S.mk_count, 0, 2 * sizeof(i32), e->stream));
    HIPCHK(hipMemsetAsync(e->S.mk_markcount, 0, sizeof(i32), e->stream));
    // ... and the step's work lists (nothing is pending across a reset; a step abandoned after its first half -- below -- has
    // left its act list's count behind)
    HIPCHK(hipMemsetAsync(e->S.slow_n, 0, 4 * sizeof(i32), e->stream));
    HIPCHK(hipMemsetAsync(e->S.tr_list_n, 0, 2 * sizeof(i32), e->stream));
    HIPCHK(hipMemsetAsync(e->S.tr_list2_n, 0, 2 * sizeof(i32), e->stream));
    HIPCHK(hipMemsetAsync(e->S.acc_list_n, 0, 2 * sizeof(i32), e->stream));
    HIPCHK(hipMemsetAsync(e->S.acc_pend, 0, (size_t)e->B, e->stream));
    if (e->S.dir_list_n) HIPCHK(hipMemsetAsync(e->S.dir_list_n, 0, 2 * sizeof(i32), e->stream));
    if (e->half_open) {
        // a step abandoned between lob_td_step_begin and lob_td_step_end (a weight exchange that failed): its learner half never
        // ran, so the double-buffered lists are where the step found them -- the survivors of the combine table wait in the
        // abandoned step's parity, not the next one's.  The next step takes that parity again.
        e->td_parity ^= 1;
        e->list_par ^= 1;
    }
    if (e->P.sarsa_lanes) {
        // ... and the tile registry with it; the generations that still name a slot of the old table carry the old epoch
        HIPCHK(hipMemsetAsync(e->S.ow_tab, 0xff, (size_t)e->S.ow_slots * 8, e->stream));
        HIPCHK(hipMemsetAsync(e->S.amb_bits, 0, ((size_t)e->P.M / 32 + 1) * 4, e->stream));
        HIPCHK(hipMemsetAsync(e->S.amb_new_n, 0, 2 * sizeof(i32), e->stream));
        e->reg_apar = 0;
        HIPCHK(hipMemsetAsync(e->S.amb_flag, 0, sizeof(i32), e->stream));
        HIPCHK(hipMemsetAsync(e->S.mk_all_n, 0, sizeof(i32), e->stream));
        HIPCHK(hipMemsetAsync(e->S.tr_cbslot, 0xff, (size_t)e->B * e->P.trace_gens * 4, e->stream));  // (slots of books that stopped stepping may be gone)
        if (e->S.tr_cbd) HIPCHK(hipMemsetAsync(e->S.tr_cbd, 0xff, (size_t)e->B * e->P.trace_gens * 8, e->stream));
        hipLaunchKernelGGL(counters_zero_kernel, dim3(1), dim3(LOB_CNT_STRIPES), 0, e->stream, +e->S.counters, 7);
        e->P.epi_epoch++;
        { int rc = push_params(e); if (rc) return rc; }
    }
    {
        TimedLaunch t(e, "reset_kernel", nullptr, true);
        lobk_reset(e->stream, e->reset_lanes, e->P.T <= 2, e->prepass_roles, (const DevParams*)e->P_dev, e->S);
    }
    HIPCHK(hipGetLastError());
    e->was_reset = true;
    e->episode_open = true;
    e->steps_since_fill = 0;
    e->hits_ok = false;
    e->hint_step = 0;   // (the hand-back counts of the episode before say nothing about this one's first steps)
    e->rest_recent = 0;
    e->half_open = false;  // (a step begun before the reset -- e.g. an exchange that failed between the halves -- is abandoned with the episode)
    return check_device_errors(e);
}

static int need_reset(lob_engine* e, const char* who) {
    if (!e) return LOB_EINVAL;
    if (!e->was_reset) { lob_set_error(std::string(who) + ": call lob_reset first"); return LOB_ESTATE; }
    return LOB_OK;
}
// Between lob_td_step_begin and lob_td_step_end only the weight exchange may run (include/lob_engine.h): anything that moves
// the books, the traces or the weights there would change what the second half consumes.
static int not_mid_step(lob_engine* e, const char* who) {
    if (e && e->half_open) { lob_set_error(std::string(who) + ": a learner step is half done (lob_td_step_begin without lob_td_step_end)"); return LOB_ESTATE; }
    return LOB_OK;
}

int lob_step(lob_engine* e, const int32_t* host_actions) {
    int rc = need_reset(e, "lob_step");
    if (rc) return rc;
    if ((rc = not_mid_step(e, "lob_step"))) return rc;
    if (!host_actions) { lob_set_error("lob_step: actions == NULL"); return LOB_EINVAL; }
    for (int b = 0; b < e->B; b++)
        if (host_actions[b] < 0 || host_actions[b] >= LOB_N_ACTIONS) { lob_set_error("lob_step: action out of range"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->actions_dev, host_actions, (size_t)e->B * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->S.mk_count, 0, 2 * sizeof(i32), e->stream));  // claims of this step go on a fresh list (nobody evaluates it)
    e->step_id++;
    e->hits_ok = false;
    {
        TimedLaunch t(e, "env_kernel");
        launch_env(e, e->stream, (const i32*)e->actions_dev, 0, 0, e->B, 0);
    }
    maybe_refill_track(e);
    HIPCHK(hipGetLastError());
    return check_device_errors(e);
}

int lob_get_state(lob_engine* e, float* host_out) {
    int rc = need_reset(e, "lob_get_state");
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->device));
    f32* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)e->B * e->P.V * 4));
    lobk_get_state(e->stream, (const DevParams*)e->P_dev, e->S, d, (f64*)nullptr);
    hipError_t err = hipMemcpyAsync(host_out, d, (size_t)e->B * e->P.V * 4, hipMemcpyDeviceToHost, e->stream);
    hipStreamSynchronize(e->stream);
    hipFree(d);
    HIPCHK(err);
    return LOB_OK;
}

int lob_get_reward(lob_engine* e, double* host_out) {
    int rc = need_reset(e, "lob_get_reward");
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->device));
    f64* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)e->B * 8));
    lobk_get_state(e->stream, (const DevParams*)e->P_dev, e->S, (f32*)nullptr, d);
    hipError_t err = hipMemcpyAsync(host_out, d, (size_t)e->B * 8, hipMemcpyDeviceToHost, e->stream);
    hipStreamSynchronize(e->stream);
    hipFree(d);
    HIPCHK(err);
    return LOB_OK;
}

int lob_get_terminal(lob_engine* e, uint8_t* host_out) {
    int rc = need_reset(e, "lob_get_terminal");
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->device));
    std::vector<i32> done(e->B), tm(e->B);
    HIPCHK(hipMemcpyAsync(done.data(), e->S.done, (size_t)e->B * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(tm.data(), e->S.time_ms, (size_t)e->B * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int b = 0; b < e->B; b++) {
        bool open = ((i64)tm[b] > e->P.open_ms + 30 * 60000LL) && ((i64)tm[b] < e->P.close_ms - 30 * 60000LL);
        host_out[b] = done[b] == 2 ? 2 : (open ? 0 : 1);
    }
    return LOB_OK;
}

int lob_clear_inventory(lob_engine* e) {
    int rc = need_reset(e, "lob_clear_inventory");
    if (rc) return rc;
    if ((rc = not_mid_step(e, "lob_clear_inventory"))) return rc;
    HIPCHK(hipSetDevice(e->device));
    lobk_clear_inventory(e->stream, (const DevParams*)e->P_dev, e->S);
    e->hits_ok = false;
    HIPCHK(hipGetLastError());
    return check_device_errors(e);
}

int lob_get_books(lob_engine* e, int32_t first, int32_t n, lob_book_dump* out) {
    if (!e || !out || first < 0 || n < 1 || first + n > e->B) { lob_set_error("lob_get_books: bad range"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    if (e->dump_cap < n) {
        if (e->dump_dev) hipFree(e->dump_dev);
        HIPCHK(hipMalloc((void**)&e->dump_dev, (size_t)n * sizeof(lob_book_dump)));
        e->dump_cap = n;
    }
    lobk_dump(e->stream, (const DevParams*)e->P_dev, e->S, first, n, e->dump_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, e->dump_dev, (size_t)n * sizeof(lob_book_dump), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}
int lob_get_book(lob_engine* e, int32_t book, lob_book_dump* out) { return lob_get_books(e, book, 1, out); }

// One env-step of every book.  The books are split into n_groups contiguous
// groups, each running act -> env -> learn on its own stream; both groups only
// READ theta, so the synchronous-batch semantic is unchanged.  The update of all
// books runs on the main stream after both groups have joined.
// SARSA(lambda): every book keeps all its generations -- sums per slot inside 1 024-book blocks first (accumulate_block_kernel).
// The same for Q(lambda) once most actions are greedy (P(greedy) = 1 - eps + eps / 9 > 0.7: more than three live generations per book)
static bool acc_blocked(const lob_engine* e) {
    const bool many_gens = e->P.algo == LOB_ALGO_SARSA || (e->P.algo == LOB_ALGO_QLAMBDA && e->P.policy == LOB_POLICY_EPS_GREEDY && e->P.epsilon < 0.34);
    return e->acc_block && e->P.memo && many_gens && e->B >= 4 * LOB_ACB_BLOCK;
}
// accumulate_kernel's lanes per book (log2): one lane per trace generation.  Two books per wave measured
// best for both SARSA(lambda) (all trace_kmax generations alive) and Watkins's Q(lambda) at eps = 0.8 (1-2
// alive; 8 lanes per book: 0.035 ms, 16: 0.030, 32: 0.029, 64: 0.036).  LOB_ACC_LANES=8|16|32|64 overrides.
static int acc_lanes_shift(const lob_engine* e) {
    if (e->acc_shift >= 0) return e->acc_shift;
    if (e->P.algo == LOB_ALGO_SARSA && e->S.cb_reps > 1) return 6;  // (with a copy of the sums per XCD: 64 lanes 0.096 ms, 32 lanes 0.101)
    if (e->P.policy == LOB_POLICY_EPS_GREEDY && e->P.epsilon < 0.34) return 6;  // (mostly greedy: Q(lambda)'s books keep their generations like SARSA's)
    return 5;
}

// `half`: 0 = whole steps; 1 = the first half of ONE learner step (action selection + performAction), 2 = its second half
// (memo, traces, TD errors, update) -- lob_td_step_begin / lob_td_step_end: a weight exchange fits between the two, where
// no hit list is live (the step's action has consumed the previous step's, the learn kernel builds the next under
// whatever maps it then finds).
static int run_steps(lob_engine* e, int32_t n_steps, int mode, int half = 0) {
    const bool first = half != 2, second = half != 1;
    HIPCHK(hipSetDevice(e->device));
    { int rc = sync_state(e); if (rc) return rc; }
    const int G = e->B >= 1024 ? e->n_groups : 1;  // small batches: one group (an empty group would be an empty launch)
    const uint32_t* rnd = (const uint32_t*)e->rnd_dev;
    for (int s = 0; s < n_steps; s++) {
        // double-buffered list of newly written weights (verdict carry-over, lob_state.h)
        const int par = mode == 0 ? (first ? (e->td_parity ^= 1) : e->td_parity) : 0;
        if (first) e->step_id++;
        if (!e->hits_ok) e->steps_on_lists = 0;
        const u64 ver = (u64)e->theta_ver;
        if (G > 1) {
            HIPCHK(hipEventRecord(e->ev_fork, e->stream));
            HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
        }
        const bool fast = e->P.memo != 0;  // (implies one group)
        if (first) {
            // the learn kernels' hand-back count of LOB_HINT_LAG learner steps ago (0 until that many have run since the hints were void)
            e->hint_now = 0;
            if (fast && mode == 0 && e->rest_hint && e->hint_step >= LOB_HINT_LAG) {
                // (the latest reporting step that is at least LOB_HINT_LAG steps old)
                const int hs = (int)(((e->hint_step - LOB_HINT_LAG) / LOB_HINT_EVERY * LOB_HINT_EVERY) % LOB_HINT_RING);
                HIPCHK(hipEventSynchronize(e->hint_ev[hs]));
                // (the kernel is done; its store to host memory is a system-scope atomic and carries the launch's tag: wait for that very word)
                u64 w = *(volatile u64*)(e->rest_hint + hs);
                for (int spin = 0; (uint32_t)(w >> 32) != e->hint_tags[hs] && spin < (1 << 16); spin++) w = *(volatile u64*)(e->rest_hint + hs);
                // The count only chooses between two correct paths: a word that has still not arrived (a loaded host, a slow link)
                // is no reason to abort training -- this step goes by "no book handed back" 
                e->hint_now = (uint32_t)(w >> 32) == e->hint_tags[hs] ? (int)(uint32_t)w : 0;
            }
        }
        bool rest_pending = false;
        bool acc_fused = false;  // (this step: see the learn kernel's launch)
        bool rest_merged = false;  // ... and then trace_rest_kernel instead of trace_fast_kernel<., 2> + accumulate_kernel over its list
        bool registered = false;   // this step's trace_lane_kernel launch carries the tile registry's blocks (its apply_kernel launch then carries the scan's)
        bool rest_side_now = false;
        const bool dq = e->P.algo == LOB_ALGO_DOUBLE_Q;  // (fast && dq: every step without usable hit lists takes the general act kernel over the whole batch)
        const int lpar = first ? (e->list_par ^= 1) : e->list_par;
        e->last_par = par;
        if (mode == 0) e->S.cb_par = par;  // (DevState goes to the kernels by value: the launches below see it)
        {
            // slots claimed from here on take a dense id while the update sums per block.  The generations' records of their slots'
            // ids (tr_cbd) are only kept up by accumulate_dense_kernel: whenever the mode comes on (epsilon has fallen below 0.34,
            // lob_set_epsilon), every record is void -- the kernels that verified slots meanwhile did not look ids up
            const int on = e->S.cb_dense && acc_blocked(e) ? 1 : 0;
            if (on) e->dense_ever = true;
            if (on && !e->S.cb_dense_on) HIPCHK(hipMemsetAsync(e->S.tr_cbd, 0xff, (size_t)e->B * e->P.trace_gens * 8, e->stream));
            e->S.cb_dense_on = on;
        }
        for (int g = 0; g < G; g++) {
            hipStream_t st = g == 0 ? e->stream : e->stream2;
            const int b0 = (int)((long long)e->B * g / G), nb = (int)((long long)e->B * (g + 1) / G) - b0;
            const int gw = grid_waves(nb);
            // the persistent kernels: one 16-wave block per CU; the general kernels then serve the books handed back
            const int gf = std::min(e->n_cus, (nb + LOB_FAST_WAVES * LOB_FAST_NB - 1) / (LOB_FAST_WAVES * LOB_FAST_NB)), gl = 2 * e->n_cus;
            const i32* act_list = e->S.slow_list, *learn_list = e->S.slow_list + e->B;
            const i32* act_n = e->S.slow_n + lpar * 2, *learn_n = e->S.slow_n + lpar * 2 + 1;
            // stagger: group 1 starts acting when group 0 has finished acting, so that the
            // latency-bound env kernel of one group runs beside a gather kernel of the other
            if (first) {
            if (G > 1 && g == 1) HIPCHK(hipStreamWaitEvent(st, e->ev_stagger, 0));
            // (batches of 1 024 books and more, env_kernel<64>: the action selection rides in the env kernel -- 4 096 books: 25.0 -> 27.2 M
            // env-steps/s, 16 384: 65.5 -> 75.8 M)
            const bool fused_act = fast && mode == 0 && e->hits_ok && e->light && e->fuse_act && e->env_lanes == 0 && (e->B >= 1024 || e->force_fuse_act);
            if (fused_act) {
                launch_env_fused(e, st, par, lpar, ver, act_list, act_n, gl);
            } else if (fast && !dq) {
                {
                    TimedLaunch t(e, "act_kernel", st);
                    if (mode == 0 && e->hits_ok && e->light)
                        hipLaunchKernelGGL(act_light_kernel, dim3((nb + LOB_LIGHT_BLOCK - 1) / LOB_LIGHT_BLOCK), dim3(LOB_LIGHT_BLOCK), 0, st, LOB_PS(e), par, lpar, ver, e->step_id - 1);
                    else
                        hipLaunchKernelGGL(act_fast_kernel<LOB_FAST_NB>, dim3(gf), dim3(LOB_FAST_BLOCK), fast_lds_bytes(e->P.cwords4, LOB_FAST_NB, false), st, LOB_PS(e), rnd, mode, par, lpar, ver);
                }
                {
                    TimedLaunch t(e, "act_rest_kernel", st);
                    hipLaunchKernelGGL((act_kernel<LOB_ALGO_SARSA, true>), dim3(gl), dim3(LOB_BLOCK), 0, st, LOB_PS(e), rnd, mode, 0, e->B, par, act_list, act_n);
                }
            } else {
                TimedLaunch t(e, "act_kernel", st);
                if (e->P.algo == LOB_ALGO_DOUBLE_Q) hipLaunchKernelGGL((act_kernel<LOB_ALGO_DOUBLE_Q, false>), dim3(gw), dim3(LOB_BLOCK), 0, st, LOB_PS(e), rnd, mode, b0, nb, par, (const i32*)nullptr, (const i32*)nullptr);
                else hipLaunchKernelGGL((act_kernel<LOB_ALGO_SARSA, false>), dim3(gw), dim3(LOB_BLOCK), 0, st, LOB_PS(e), rnd, mode, b0, nb, par, (const i32*)nullptr, (const i32*)nullptr);
            }
            if (G > 1 && g == 0) HIPCHK(hipEventRecord(e->ev_stagger, st));
            if (!fused_act) {
                TimedLaunch t(e, "env_kernel", st);
                launch_env(e, st, (const i32*)nullptr, mode == 0 ? 1 : 0, b0, nb, par);
            }
            }  // first half
            if (!second) continue;
            if (fast) launch_memo(e, par, mode == 0 ? 0 : 1);  // learner: S0 under theta_t for learn_kernel; backtester: for the next act
            // (... and, which 0, the step's cb_par / cb_dense_on into the state's device-resident copy; without a memo launch:)
            else if (mode == 0 && e->P.combine) hipLaunchKernelGGL(step_words_kernel, dim3(1), dim3(1), 0, st, e->S_dev, e->S.cb_par, e->S.cb_dense_on);
            // (the tile registry's entries for the slots that were new in the step before -- the extra blocks of its trace_lane_kernel
            // and apply_kernel launches -- are read from here on: the dup flags by the light trace step, the rest by trace_lane_kernel)
            if (mode == 0 && fast) {
                const bool tl = (e->P.algo == LOB_ALGO_QLAMBDA || dq) && e->t_light;
                // Q(s', .) + TD error: a lane per book once the batch gives every CU a full block of them, else a wave per book
                const bool lanes = e->q_lanes >= 0 ? e->q_lanes == 1 : nb >= LOB_QL_BLOCK * e->n_cus / 2;
                // ... and then the lane kernel also takes the trace step of the books it can (those left go on a list for the
                // wave-per-book trace kernel, which runs AFTER it)
                const bool fuse = tl && lanes && !e->no_fuse;
                const int gt = std::min((4 * LOB_TRACE_OCC / LOB_TRACE_WAVES) * e->n_cus, (nb + LOB_TRACE_WAVES - 1) / LOB_TRACE_WAVES);
                const int sid = e->step_id;
                if (tl && !fuse) {
                    // a lane per book where the step leaves no older generation behind, the wave-per-book kernel for the rest
                    TimedLaunch t(e, "trace_light_kernel", st);
                    hipLaunchKernelGGL(trace_light_kernel, dim3((nb + LOB_LIGHT_BLOCK - 1) / LOB_LIGHT_BLOCK), dim3(LOB_LIGHT_BLOCK), 0, st, LOB_PS(e), lpar);
                }
                auto launch_traces = [&]() {
                    if (!fuse) {
                        TimedLaunch t(e, "trace_kernel", st);
                        if (tl) hipLaunchKernelGGL((trace_fast_kernel<LOB_ALGO_QLAMBDA, 1>), dim3(gt), dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), st, LOB_PS(e), rnd, par, lpar, sid);
                        else if (e->P.algo == LOB_ALGO_QLAMBDA) hipLaunchKernelGGL((trace_fast_kernel<LOB_ALGO_QLAMBDA, 0>), dim3(gt), dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), st, LOB_PS(e), rnd, par, lpar, sid);
                        else if (e->P.sarsa_lanes) {
                            // a lane per generation; the wave-per-book kernel for the books it leaves on the list
                            // (LOB_TS_GRID / LOB_TS_LDS, experiments: fewer, persistent blocks / dynamic LDS to throttle the occupancy -- halving
                            // it costs 28 %, a persistent grid changes nothing: NOTES.md "Round 5")
                            const int ts_full = (nb + LOB_TS_BLOCK / 32 - 1) / (LOB_TS_BLOCK / 32);
                            // (its first LOB_REG_BLOCKS blocks: the tile registry for the memo slots that are new on this step's list)
                            registered = true;
                            hipLaunchKernelGGL(trace_lane_kernel<LOB_ALGO_SARSA>, dim3((e->ts_grid > 0 ? std::min(e->ts_grid, ts_full) : ts_full) + LOB_REG_BLOCKS), dim3(LOB_TS_BLOCK), e->ts_lds, st,
                                               LOB_PS(e), lpar, sid, 0, rnd, par, e->reg_apar, LOB_REG_BLOCKS);
                            hipLaunchKernelGGL((trace_fast_kernel<LOB_ALGO_SARSA, 1>), dim3(gt), dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), st, LOB_PS(e), rnd, par, lpar, sid);
                        }
                        else hipLaunchKernelGGL((trace_fast_kernel<LOB_ALGO_SARSA, 0>), dim3(gt), dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), st, LOB_PS(e), rnd, par, lpar, sid);
                    }
                };
                const bool learn_first = lobk_experiments() && e->P.exp_learn_first && e->P.algo == LOB_ALGO_SARSA;
                // (SARSA(lambda)'s trace kernels run IN FRONT OF its learn kernels: they mark the new generation's tiles in the
                // written-weights maps the learn kernels build the hit lists by -- lob_state.h above hl_rec.  The other order exists in
                // -DLOB_EXPERIMENTS builds only (LOB_SARSA_LEARN_FIRST=1), for the test that shows it to fail.)
                if (!learn_first) launch_traces();
                {
                    TimedLaunch t(e, "learn_kernel", st);
                    if (lanes) {
                        const bool pair = e->q_pair && e->P.M < (1ll << 27) && (!dq || e->dq_pair);  // (its LDS rows hold tile indices in 27 bits)
                        // the updates added to their slots by this kernel and trace_lane_kernel (lob_state.h acc_list): Q(lambda) while its
                        // books keep few generations (else accumulate_block_kernel's sums per block win)
                        acc_fused = fuse && e->acc_fuse && e->P.combine && e->P.sarsa_lanes && (e->P.algo == LOB_ALGO_QLAMBDA || dq) && !acc_blocked(e) && G == 1;
                        // (not while the learn kernel hands most books back -- a dense theta: every one of them would go on the list
                        // through one counter; the list's length of a few steps ago, as launch_env_fused reads it)
                        if (e->hint_now > std::max(1024, e->B / 16)) acc_fused = false;
                        rest_merged = acc_fused && e->rest_merge;
                        const int gq = pair ? std::min(LOB_QP_OCC * e->n_cus, (nb + LOB_QP_BOOKS - 1) / LOB_QP_BOOKS) : std::min(e->n_cus, (nb + LOB_QL_BLOCK - 1) / LOB_QL_BLOCK);
                        const size_t lds = pair ? qpair_lds_bytes(e->P.cwords4) : qlane_lds_bytes(e->P.cwords4);
                        lobk_learn_q(st, pair, e->P.algo, e->P.V == 8, fuse, gq, lds, (const DevParams*)e->P_dev, e->S, rnd, lpar, ver, sid, acc_fused ? (rest_merged ? 2 : 1) : 0);
                    } else lobk_learn_q_fast(st, e->P.algo, gf, fast_lds_bytes(e->P.cwords4, LOB_FAST_NB, false), (const DevParams*)e->P_dev, e->S, rnd, lpar, ver);
                }
                if (!rest_merged) {   // (rest_merged: trace_rest_kernel serves the books handed back too, behind the lane trace kernel)
                    // The books the lane kernels hand back (a list that is empty in most steps, a handful of books in the others).
                    // When trace kernels follow on the main stream (the fused Q(lambda) / double Q flow), this launch runs beside
                    // them on the second stream: nothing they do depends on it -- the update kernels wait for both.
                    // (worth it only while books ARE handed back -- an empty launch is cheaper than the two cross-stream waits: the
                    // kernel reports its list's length through a host-mapped word, read here a few steps late)
                    if (e->hint_now > 0) e->rest_recent = 64;
                    else if (e->rest_recent > 0) e->rest_recent--;
                    const bool side = fuse && e->rest_side && e->rest_recent > 0;
                    hipStream_t rs = side ? e->stream2 : st;
                    if (side) {
                        HIPCHK(hipEventRecord(e->ev_rest_go, st));
                        HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_rest_go, 0));
                    }
                    {
                        TimedLaunch t(e, "learn_rest_kernel", rs);
                        const int hs = (int)(e->hint_step % LOB_HINT_RING);
                        const bool no_hint = e->no_hint;  // (experiment: the hint's store to host memory costs the kernel 3.5 us and the step nothing)
                        const bool reports = e->rest_hint && !no_hint && e->hint_step % LOB_HINT_EVERY == 0;
                        u64* hint_dev = reports ? e->rest_hint_dev + hs : nullptr;
                        const uint32_t hint_tag = (uint32_t)(++e->hint_serial);
                        if (dq) hipLaunchKernelGGL(learn_q_rest_kernel<LOB_ALGO_DOUBLE_Q>, dim3(gl), dim3(LOB_BLOCK), 0, rs, LOB_PS(e), rnd, learn_list, learn_n, hint_dev, hint_tag);
                        else if (e->P.algo == LOB_ALGO_QLAMBDA) hipLaunchKernelGGL(learn_q_rest_kernel<LOB_ALGO_QLAMBDA>, dim3(gl), dim3(LOB_BLOCK), 0, rs, LOB_PS(e), rnd, learn_list, learn_n, hint_dev, hint_tag);
                        else hipLaunchKernelGGL(learn_q_rest_kernel<LOB_ALGO_SARSA>, dim3(gl), dim3(LOB_BLOCK), 0, rs, LOB_PS(e), rnd, learn_list, learn_n, hint_dev, hint_tag);
                        if (reports) { HIPCHK(hipEventRecord(e->hint_ev[hs], rs)); e->hint_tags[hs] = hint_tag; }
                        if (e->rest_hint && !no_hint) e->hint_step++;
                    }
                    if (side) {
                        HIPCHK(hipEventRecord(e->ev_rest_done, e->stream2));
                        rest_pending = true;
                        rest_side_now = true;
                    }
                }
                if (learn_first) launch_traces();
                if (fuse) {
                    TimedLaunch t(e, "trace_kernel", st);
                    // the listed books (their traces survive the step): a lane per generation, then the wave-per-book kernel for
                    // those the lane kernel hands on
                    if (e->P.sarsa_lanes) {
                        // (its first LOB_REG_BLOCKS blocks: the tile registry for the memo slots that are new on this step's list)
                        registered = true;
                        hipLaunchKernelGGL(trace_lane_kernel<LOB_ALGO_QLAMBDA>, dim3(std::min(4 * e->n_cus, (nb + LOB_TS_BLOCK / 32 - 1) / (LOB_TS_BLOCK / 32)) + LOB_REG_BLOCKS), dim3(LOB_TS_BLOCK), 0, st,
                                           LOB_PS(e), lpar, sid, acc_fused ? (rest_merged ? 2 : 1) : 0, rnd, par, e->reg_apar, LOB_REG_BLOCKS);
                    }
                    // (rest_merged: what the lane kernel hands on is served by trace_rest_kernel, in the place of accumulate_kernel below)
                    if (!rest_merged) hipLaunchKernelGGL((trace_fast_kernel<LOB_ALGO_QLAMBDA, 2>), dim3(gt), dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), st, LOB_PS(e), rnd, par, lpar, sid);
                }
            } else if (mode == 0) {
                TimedLaunch t(e, "learn_kernel", st);
                if (e->P.algo == LOB_ALGO_DOUBLE_Q) hipLaunchKernelGGL((learn_kernel<LOB_ALGO_DOUBLE_Q, false>), dim3(gw), dim3(LOB_BLOCK), 0, st, LOB_PS(e), rnd, b0, nb, par, (const i32*)nullptr, (const i32*)nullptr);
                else if (e->P.algo == LOB_ALGO_QLAMBDA) hipLaunchKernelGGL((learn_kernel<LOB_ALGO_QLAMBDA, false>), dim3(gw), dim3(LOB_BLOCK), 0, st, LOB_PS(e), rnd, b0, nb, par, (const i32*)nullptr, (const i32*)nullptr);
                else hipLaunchKernelGGL((learn_kernel<LOB_ALGO_SARSA, false>), dim3(gw), dim3(LOB_BLOCK), 0, st, LOB_PS(e), rnd, b0, nb, par, (const i32*)nullptr, (const i32*)nullptr);
            }
        }
        if (G > 1) {
            HIPCHK(hipEventRecord(e->ev_join, e->stream2));
            HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
        }
        if (!second) { e->half_open = true; continue; }
        e->half_open = false;
        if (rest_pending) { HIPCHK(hipStreamWaitEvent(e->stream, e->ev_rest_done, 0)); rest_pending = false; }
        // model_log: the step's |delta| sums, launched where every stepped book's TD error is final -- behind trace_rest_kernel in
        // the merged flow (a book the learn kernel handed back still carries Q(s, a) in LHdr::td until that kernel's learn_q_book
        // has run: ADVICE r5), in front of the update kernels everywhere else
        auto td_stats = [&]() {
            if (!(mode == 0 && e->model_log)) return;
            TimedLaunch t(e, "td_stats_kernel");
            const int nblk = std::min(LOB_ML_BLOCKS, (e->B + 255) / 256), per = (e->B + nblk - 1) / nblk;
            hipLaunchKernelGGL(td_stats_kernel, dim3(nblk), dim3(256), 0, e->stream, e->S, per);
            hipLaunchKernelGGL(td_stats_fold_kernel, dim3(1), dim3(64), 0, e->stream, e->S, nblk);
        };
        if (!rest_merged) td_stats();
        if (mode == 0 && e->P.combine) {
            int dense_blocks = 0;
            {
                TimedLaunch t(e, "accumulate_kernel");
                // SARSA(lambda): every book keeps all its generations -- sums per slot inside 1 024-book blocks first.  The same for
                // Q(lambda) once most actions are greedy (P(greedy) = 1 - eps + eps / 9 > 0.7: more than three live generations per book)
                e->flow[acc_blocked(e) ? 2 : acc_fused ? 0 : 3]++;
                if (acc_blocked(e) && e->S.cb_dense) e->flow[7]++;
                if (rest_side_now) e->flow[1]++;
                if (acc_blocked(e) && e->S.cb_dense) {
                    // sums by the slots' dense ids in a direct-indexed LDS array, one block per CU (accumulate_dense_kernel)
                    dense_blocks = std::min(std::min(e->n_cus, LOB_ACD_MAX_BLOCKS), (e->B + 127) / 128);   // (a block takes 128 books per pass)
                    const int bpb = ((e->B + dense_blocks - 1) / dense_blocks + 31) / 32 * 32;
                    dense_blocks = (e->B + bpb - 1) / bpb;
                    hipLaunchKernelGGL(accumulate_dense_kernel, dim3(dense_blocks), dim3(LOB_ACD_BLOCK), acd_lds_bytes(), e->stream, LOB_PS(e), par, e->step_id, bpb);
                    hipLaunchKernelGGL(reduce_dense_kernel, dim3((e->S.cb_ids + 255) / 256, LOB_ACD_GROUPS), dim3(256), 0, e->stream, e->S, dense_blocks);
                } else if (acc_blocked(e)) {
                    // (batches per block: SARSA(lambda) 80 us with one, 89 with two or four -- its blocks are bound by their LDS insertions, not by what
                    // they send to memory; mostly-greedy Q(lambda) 58 -> 50 us with four)
                    const int nbat = e->acc_batches_set ? e->acc_batches : (e->P.algo == LOB_ALGO_SARSA ? 1 : std::max(1, std::min(e->acc_batches, e->B / (16 * LOB_ACB_BLOCK))));
                    hipLaunchKernelGGL(accumulate_block_kernel, dim3((e->B + LOB_ACB_BLOCK * nbat - 1) / (LOB_ACB_BLOCK * nbat), e->P.trace_kmax), dim3(LOB_ACB_BLOCK), 0, e->stream, LOB_PS(e), par, e->step_id, nbat);
                } else if (rest_merged) {
                    // the books the learn kernel handed back (TD error + their sums), the books the lane trace kernel handed on (trace
                    // step + their sums) and what the fused accumulation left, in one launch; it reports the hand-back count to the
                    // host as learn_q_rest_kernel does in the other flows
                    const int hs = (int)(e->hint_step % LOB_HINT_RING);
                    const bool reports = e->rest_hint && !e->no_hint && e->hint_step % LOB_HINT_EVERY == 0;
                    u64* hint_dev = reports ? e->rest_hint_dev + hs : nullptr;
                    const uint32_t hint_tag = (uint32_t)(++e->hint_serial);
                    const dim3 rg(std::min(e->n_cus, (std::min(e->B, 4096) + LOB_TRACE_WAVES - 1) / LOB_TRACE_WAVES));
                    if (e->P.algo == LOB_ALGO_DOUBLE_Q)
                        hipLaunchKernelGGL(trace_rest_kernel<LOB_ALGO_DOUBLE_Q>, rg, dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), e->stream, LOB_PS(e), rnd, par, lpar, e->step_id, acc_lanes_shift(e), hint_dev, hint_tag);
                    else
                        hipLaunchKernelGGL(trace_rest_kernel<LOB_ALGO_QLAMBDA>, rg, dim3(LOB_TRACE_BLOCK), trace_lds_bytes(), e->stream, LOB_PS(e), rnd, par, lpar, e->step_id, acc_lanes_shift(e), hint_dev, hint_tag);
                    if (reports) { HIPCHK(hipEventRecord(e->hint_ev[hs], e->stream)); e->hint_tags[hs] = hint_tag; }
                    if (e->rest_hint && !e->no_hint) e->hint_step++;
                } else if (acc_fused) {
                    // what the fused accumulation left: a few hundred books (the grid's waves stride over the list)
                    const int sh = acc_lanes_shift(e);
                    const int waves = (std::min(e->B, 4096) + (64 >> sh) - 1) / (64 >> sh);
                    hipLaunchKernelGGL(accumulate_kernel, dim3((waves + LOB_WAVES_PER_BLOCK - 1) / LOB_WAVES_PER_BLOCK), dim3(LOB_BLOCK), 0, e->stream, LOB_PS(e), par, sh, e->step_id,
                                       (const i32*)e->S.acc_list, (const i32*)&e->S.acc_list_n[lpar]);
                } else {
                    const int sh = acc_lanes_shift(e);
                    const int waves = (e->B + (64 >> sh) - 1) / (64 >> sh);
                    hipLaunchKernelGGL(accumulate_kernel, dim3((waves + LOB_WAVES_PER_BLOCK - 1) / LOB_WAVES_PER_BLOCK), dim3(LOB_BLOCK), 0, e->stream, LOB_PS(e), par, sh, e->step_id,
                                       (const i32*)nullptr, (const i32*)nullptr);
                }
            }
            if (rest_merged) td_stats();
            {
                TimedLaunch t(e, "apply_kernel");
                const int blocks = e->S.cb_segs;
                // (dense_blocks -1: no slot has ever been given a dense id -- apply_kernel does not look any up)
                // (+ the indices the step's registry blocks -- trace_lane_kernel's launch -- found ambiguous, marked in every registered slot)
                hipLaunchKernelGGL(apply_kernel, dim3(blocks + (registered ? LOB_SCAN_BLOCKS : 0)), dim3(256), 0, e->stream, LOB_PS(e), rnd, par, e->step_id,
                                   e->dense_ever ? dense_blocks : -1, e->reg_apar);
                if (registered) e->reg_apar ^= 1;
            }
        } else if (mode == 0) {
            TimedLaunch t(e, "update_kernel");
            hipLaunchKernelGGL(update_kernel, dim3(grid_waves(e->B)), dim3(LOB_BLOCK), 0, e->stream, LOB_PS(e), par, e->step_id);
        }
        if (mode == 0 && e->P.r_learn) {
            // R-learning: the average reward rho, after updateQ (rho_kernel, lob_kernels.h)
            TimedLaunch t(e, "rho_kernel");
            const int nr = e->P.theta_private ? e->B : 1;
            hipLaunchKernelGGL(rho_kernel, dim3(grid_waves(e->B)), dim3(LOB_BLOCK), 0, e->stream, e->P, e->S, rnd);
            hipLaunchKernelGGL(rho_fold_kernel, dim3((nr + 255) / 256), dim3(256), 0, e->stream, e->S, nr);
        }
        if (mode == 0) {
            e->theta_ver++;                          // theta_{t+1}
            if (e->P.memo) launch_memo(e, par, 1, lpar);   // the same triples again, for the next action selection (+ its list resets)
        }
        e->hits_ok = mode == 0 && fast;              // learn_q_fast_kernel has left the hit lists of the States the next step acts on
        maybe_refill_track(e);
    }
    HIPCHK(hipGetLastError());
    return LOB_OK;
}

int lob_td_step(lob_engine* e, int32_t n_steps) {
    int rc = need_reset(e, "lob_td_step");
    if (rc) return rc;
    if (n_steps < 0) return LOB_EINVAL;
    if (e->half_open) { lob_set_error("lob_td_step: a step is half done (lob_td_step_begin without lob_td_step_end)"); return LOB_ESTATE; }
    return run_steps(e, n_steps, 0);
}
int lob_td_step_begin(lob_engine* e) {
    int rc = need_reset(e, "lob_td_step_begin");
    if (rc) return rc;
    if (e->half_open) { lob_set_error("lob_td_step_begin: the previous step has not been ended"); return LOB_ESTATE; }
    if (e->B >= 1024 && e->n_groups > 1) { lob_set_error("lob_td_step_begin: not with two book groups (LOB_GROUPS=2)"); return LOB_ESTATE; }
    return run_steps(e, 1, 0, 1);
}
int lob_td_split_supported(lob_engine* e) { return e && !(e->B >= 1024 && e->n_groups > 1) ? 1 : 0; }
int lob_td_step_end(lob_engine* e) {
    int rc = need_reset(e, "lob_td_step_end");
    if (rc) return rc;
    if (!e->half_open) { lob_set_error("lob_td_step_end: no step begun"); return LOB_ESTATE; }
    return run_steps(e, 1, 0, 2);
}
int lob_eval_step(lob_engine* e, int32_t n_steps) {
    int rc = need_reset(e, "lob_eval_step");
    if (rc) return rc;
    if (n_steps < 0) return LOB_EINVAL;
    if (e->half_open) { lob_set_error("lob_eval_step: a learner step is half done"); return LOB_ESTATE; }
    return run_steps(e, n_steps, 1);
}

int lob_model_log_enable(lob_engine* e, int32_t on) {
    if (!e) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    if (on && !e->S.ml_part) {
        int rc = dev_alloc(e, &e->S.ml_part, (size_t)LOB_ML_BLOCKS);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->S.ml_npart, (size_t)LOB_ML_BLOCKS);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->S.ml_agg, 1);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->S.ml_cnt, 2);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->S.ml_rows, (size_t)LOB_ML_ROWS);
        if (rc != LOB_OK) return rc;
    }
    e->model_log = on != 0;
    return LOB_OK;
}
int lob_model_log_read(lob_engine* e, double* rows, int32_t cap, int32_t* n_rows, int64_t* n_lost) {
    if (!e || !rows || cap < 0 || !n_rows) return LOB_EINVAL;
    *n_rows = 0;
    if (n_lost) *n_lost = 0;
    if (!e->S.ml_part) return LOB_OK;
    HIPCHK(hipSetDevice(e->device));
    i64 cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, e->S.ml_cnt, sizeof cnt, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const i64 kept = std::min<i64>(cnt[1], LOB_ML_ROWS), n = std::min<i64>(kept, cap);
    if (n > 0) HIPCHK(hipMemcpyAsync(rows, e->S.ml_rows, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
    // the rows are handed over once: the counter starts again (the running aggregate stays)
    HIPCHK(hipMemsetAsync(e->S.ml_cnt + 1, 0, sizeof(i64), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *n_rows = (int32_t)n;
    if (n_lost) *n_lost = cnt[1] - n;   // rows beyond the ring (LOB_ML_ROWS between two reads) or beyond `cap`
    return LOB_OK;
}

int lob_handle_terminal(lob_engine* e) {
    if (!e) return LOB_EINVAL;
    { int rc = not_mid_step(e, "lob_handle_terminal"); if (rc) return rc; }
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(clear_traces_kernel, dim3(grid_lanes(e->B)), dim3(256), 0, e->stream, e->S);
    HIPCHK(hipGetLastError());
    return LOB_OK;
}
int lob_set_alpha(lob_engine* e, double alpha) { if (!e) return LOB_EINVAL; e->P.alpha = alpha; e->params.alpha = alpha; hipSetDevice(e->device); return push_params(e); }
int lob_set_tau(lob_engine* e, double tau) { if (!e || !(tau > 0.0)) return LOB_EINVAL; e->P.tau = tau; e->params.tau = tau; hipSetDevice(e->device); return push_params(e); }
int lob_set_epsilon(lob_engine* e, double eps) { if (!e) return LOB_EINVAL; e->P.epsilon = eps; e->params.epsilon = eps; hipSetDevice(e->device); return push_params(e); }

static int features_impl(lob_engine* e, const float* host_vars, int32_t n, int32_t* out_idx, double* out_q) {
    if (!e || !host_vars || n < 1) { lob_set_error("lob_features: bad argument"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    f32* dv = nullptr; i32* di = nullptr; f64* dq = nullptr;
    HIPCHK(hipMalloc((void**)&dv, (size_t)n * e->P.V * 4));
    if (out_idx) HIPCHK(hipMalloc((void**)&di, (size_t)n * LOB_N_ACTIONS * 96 * 4));
    if (out_q) HIPCHK(hipMalloc((void**)&dq, (size_t)n * LOB_N_ACTIONS * 8));
    HIPCHK(hipMemcpyAsync(dv, host_vars, (size_t)n * e->P.V * 4, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(features_kernel, dim3(grid_waves(n)), dim3(LOB_BLOCK), 0, e->stream, e->P, (const f64*)e->S.theta,
                       (const uint32_t*)e->S.theta_nz, (const uint32_t*)e->rnd_dev, (const f32*)dv, n, di, dq);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess && out_idx) err = hipMemcpyAsync(out_idx, di, (size_t)n * LOB_N_ACTIONS * 96 * 4, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess && out_q) err = hipMemcpyAsync(out_q, dq, (size_t)n * LOB_N_ACTIONS * 8, hipMemcpyDeviceToHost, e->stream);
    hipStreamSynchronize(e->stream);
    hipFree(dv);
    if (di) hipFree(di);
    if (dq) hipFree(dq);
    HIPCHK(err);
    return LOB_OK;
}
int lob_features(lob_engine* e, const float* host_vars, int32_t n, int32_t* host_out) { return features_impl(e, host_vars, n, host_out, nullptr); }
int lob_q_values(lob_engine* e, const float* host_vars, int32_t n, double* host_out) { return features_impl(e, host_vars, n, nullptr, host_out); }

// `which`: shared theta: 0 = theta, 1 = theta_b (double Q); private theta: book, or n_books + book for theta_b
static int theta_slot(lob_engine* e, int32_t which, f64** th, uint32_t** nz) {
    const int n = e->P.theta_private ? e->B : 1;
    const bool dq = e->P.algo == LOB_ALGO_DOUBLE_Q;
    if (which < 0 || which >= (dq ? 2 * n : n)) { lob_set_error("lob_theta_*: `which` out of range"); return LOB_EINVAL; }
    const bool second = which >= n;
    const int idx = second ? which - n : which;
    *th = (second ? e->S.theta_b : e->S.theta) + (size_t)idx * e->P.M;
    *nz = (second ? e->S.theta_b_nz : e->S.theta_nz) + (size_t)idx * LOB_NZ_NWORDS(e->P.M);
    return LOB_OK;
}
int lob_theta_get(lob_engine* e, int32_t which, double* host_out, int64_t count) {
    if (!e || !host_out || count < 0 || count > e->P.M) return LOB_EINVAL;
    f64* th; uint32_t* nz;
    int rc = theta_slot(e, which, &th, &nz);
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(host_out, th, (size_t)count * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}
int lob_theta_set(lob_engine* e, int32_t which, const double* host_in, int64_t count) {
    if (!e || !host_in || count < 0 || count > e->P.M) return LOB_EINVAL;
    f64* th; uint32_t* nz;
    int rc = theta_slot(e, which, &th, &nz);
    if (rc) return rc;
    if ((rc = not_mid_step(e, "lob_theta_set"))) return rc;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(th, host_in, (size_t)count * 8, hipMemcpyHostToDevice, e->stream));
    if (e->S.theta_sync) {
        // multi-GPU exchange already initialised: the loaded weights are the new common base, not a local
        // delta (every rank loads the same checkpoint); otherwise the next all-reduce would add
        // world_size x (loaded - sync)
        f64* sync = e->S.theta_sync + (th == e->S.theta_b ? (size_t)e->P.M : 0);
        HIPCHK(hipMemcpyAsync(sync, th, (size_t)count * 8, hipMemcpyDeviceToDevice, e->stream));
    }
    e->theta_ver++;  // memo records computed under the old weights are void
    e->hits_ok = false;
    e->hint_step = 0;  // (... and so is what the learn kernels handed back under them)
    e->rest_recent = 0;
    hipLaunchKernelGGL(rebuild_nz_kernel, dim3(1024), dim3(256), 0, e->stream, (const f64*)th, nz, e->S.nz_epoch, e->P.M);
    if (e->P.memo && (th == e->S.theta || th == e->S.theta_b)) {  // (double Q: one pair of maps for both vectors)
        // the maps keep the bits they have (monotone: the tiles of live trace generations stay marked, whatever the
        // loaded value of their weights -- a set bit only means "fetch the weight") and gain those of the loaded non-zeros
        hipLaunchKernelGGL(rebuild_nzx_kernel, dim3(2048), dim3(256), 0, e->stream, (const f64*)th, e->S.theta_nzx, e->S.theta_nzc, e->P.cshift, e->P.M);
        hipLaunchKernelGGL(rebuild_nzd_kernel, dim3(2048), dim3(256), 0, e->stream, (const uint32_t*)e->S.theta_nzx, e->S.theta_nzd, e->S.nzd_terms, e->P.M);
        launch_memo(e, e->last_par, 1);  // the current triples under the loaded weights: the next act stays on the fast path
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}

static int fetch_hdr(lob_engine* e, std::vector<LHdr>& h) {
    h.resize(e->B);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(h.data(), e->S.hdr, (size_t)e->B * sizeof(LHdr), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}
#define HDR_GETTER(name, T, field)                                   \
    int name(lob_engine* e, T* host_out) {                           \
        if (!e || !host_out) return LOB_EINVAL;                      \
        std::vector<LHdr> h;                                         \
        int rc = fetch_hdr(e, h);                                    \
        if (rc) return rc;                                           \
        for (int b = 0; b < e->B; b++) host_out[b] = (T)h[b].field;  \
        return LOB_OK;                                               \
    }
HDR_GETTER(lob_get_last_actions, int32_t, action)
HDR_GETTER(lob_get_last_td, double, td)
HDR_GETTER(lob_get_last_rewards, double, reward)
HDR_GETTER(lob_get_stepped, int32_t, stepped)
HDR_GETTER(lob_get_rng_counters, uint64_t, rng_ctr)

/* current `state` variables of the learner (the rl::State the last step produced): float[B][n_vars] */
int lob_get_learner_state(lob_engine* e, float* host_out) {
    if (!e || !host_out) return LOB_EINVAL;
    std::vector<LHdr> h;
    int rc = fetch_hdr(e, h);
    if (rc) return rc;
    std::vector<f32> v((size_t)e->B * 48);
    HIPCHK(hipMemcpyAsync(v.data(), e->S.vars, v.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int b = 0; b < e->B; b++)
        for (int i = 0; i < e->P.V; i++) host_out[(size_t)b * e->P.V + i] = v[((size_t)b * 3 + h[b].slot_cur) * 16 + i];
    return LOB_OK;
}

int lob_get_traces(lob_engine* e, int32_t book, int32_t* idx, float* elig, int32_t cap, int32_t* n) {
    if (!e || book < 0 || book >= e->B || !n) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    const int G = e->P.trace_gens;
    std::vector<i32> ti((size_t)G * 32);
    std::vector<uint32_t> al(G);
    i32 head = 0, ng = 0;
    HIPCHK(hipMemcpyAsync(ti.data(), e->S.tr_idx + (size_t)book * G * 32, ti.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(al.data(), e->S.tr_alive + (size_t)book * G, al.size() * 4, hipMemcpyDeviceToHost, e->stream));
    LHdr hb;
    HIPCHK(hipMemcpyAsync(&hb, e->S.hdr + book, sizeof hb, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    head = hb.tr_head;
    ng = hb.tr_n;
    int k = 0;
    for (int age = 0; age < ng; age++) {
        int slot = (head - age + G) & (G - 1);
        for (int j = 0; j < 32; j++)
            if ((al[slot] >> j) & 1u) {
                if (k < cap) {
                    if (idx) idx[k] = ti[slot * 32 + j];
                    if (elig) elig[k] = e->P.trace_pow[age];
                }
                k++;
            }
    }
    *n = k;
    return LOB_OK;
}

// the striped device counters, summed on the device, into c[16] (asynchronous: the caller synchronises the stream)
static int read_counters(lob_engine* e, i64* c) {
    hipLaunchKernelGGL(counters_fold_kernel, dim3(1), dim3(LOB_CNT_STRIPES), 0, e->stream, (const i64*)e->S.counters, e->cnt_sum, (const i32*)e->S.done, e->B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c, e->cnt_sum, 16 * sizeof(i64), hipMemcpyDeviceToHost, e->stream));
    return LOB_OK;
}

int lob_get_counters(lob_engine* e, int64_t out[4]) {
    if (!e || !out) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    i64 c[16];
    { int rc = read_counters(e, c); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    out[0] = c[0]; out[1] = c[1]; out[2] = c[15]; out[3] = c[3];   // ([15]: the live books, counted by the fold kernel)
    return LOB_OK;
}

int lob_get_path_stats(lob_engine* e, int64_t out[8]) {
    if (!e || !out) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    i64 c[16];
    i32 n_all = 0, flag = 0, mk_n[2] = {0, 0};
    { int rc = read_counters(e, c); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(mk_n, e->S.mk_count, sizeof mk_n, hipMemcpyDeviceToHost, e->stream));
    if (e->P.sarsa_lanes) {
        HIPCHK(hipMemcpyAsync(&n_all, e->S.mk_all_n, 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&flag, e->S.amb_flag, 4, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[0] = c[6]; out[1] = c[5]; out[2] = n_all; out[3] = c[7]; out[4] = flag; out[5] = mk_n[e->last_par & 1];
    out[6] = c[2]; out[7] = c[4];
    return LOB_OK;
}

// Double Q carries two weight vectors: sync / delta buffers hold [theta | theta_b] back to back and
// one all-reduce of 2M doubles exchanges both.
static int delta_vectors(const lob_engine* e) { return e->P.algo == LOB_ALGO_DOUBLE_Q ? 2 : 1; }
// R-learning: the shared average reward rho travels behind the weights as [rho - rho_sync, 1.0] (rho_delta_*_kernel)
static int delta_extra(const lob_engine* e) { return e->P.r_learn ? 2 : 0; }

int lob_delta_init(lob_engine* e) {
    if (!e) return LOB_EINVAL;
    if (e->P.theta_private) { lob_set_error("lob_delta_*: shared theta only"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    const size_t M = (size_t)e->P.M;
    const int nv = delta_vectors(e);
    if (!e->S.theta_sync) {
        int rc = dev_alloc(e, &e->S.theta_sync, M * nv + delta_extra(e));
        if (rc == LOB_OK) rc = dev_alloc(e, &e->S.delta, M * nv + delta_extra(e));
        if (rc != LOB_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(e->S.theta_sync, e->S.theta, M * 8, hipMemcpyDeviceToDevice, e->stream));
    if (nv == 2) HIPCHK(hipMemcpyAsync(e->S.theta_sync + M, e->S.theta_b, M * 8, hipMemcpyDeviceToDevice, e->stream));
    if (delta_extra(e)) HIPCHK(hipMemcpyAsync(e->S.theta_sync + M * nv, e->S.rho, 8, hipMemcpyDeviceToDevice, e->stream));
    return LOB_OK;
}
int lob_delta_begin_async(lob_engine* e, double** dev_delta, int64_t* count) {
    if (!e || !dev_delta || !count) return LOB_EINVAL;
    if (!e->S.theta_sync) { lob_set_error("lob_delta_begin: call lob_delta_init first"); return LOB_ESTATE; }
    HIPCHK(hipSetDevice(e->device));
    const size_t M = (size_t)e->P.M;
    const int nv = delta_vectors(e);
    for (int v = 0; v < nv; v++) {
        TimedLaunch t(e, "delta_begin_kernel", nullptr, true);
        hipLaunchKernelGGL(delta_begin_kernel, dim3(2048), dim3(256), 0, e->stream, (const f64*)(v ? e->S.theta_b : e->S.theta),
                           (const f64*)(e->S.theta_sync + v * M), e->S.delta + v * M, e->P.M);
    }
    if (delta_extra(e))
        hipLaunchKernelGGL(rho_delta_begin_kernel, dim3(1), dim3(1), 0, e->stream, (const f64*)e->S.rho, (const f64*)(e->S.theta_sync + M * nv), e->S.delta + M * nv);
    HIPCHK(hipGetLastError());
    *dev_delta = e->S.delta;
    *count = (int64_t)(M * nv + delta_extra(e));
    return LOB_OK;
}
int lob_delta_begin(lob_engine* e, double** dev_delta, int64_t* count) {
    int rc = lob_delta_begin_async(e, dev_delta, count);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}
int lob_delta_apply(lob_engine* e) {
    if (!e) return LOB_EINVAL;
    if (!e->S.theta_sync) { lob_set_error("lob_delta_apply: call lob_delta_init first"); return LOB_ESTATE; }
    HIPCHK(hipSetDevice(e->device));
    const size_t M = (size_t)e->P.M;
    const int nv = delta_vectors(e);
    e->theta_ver++;  // memo records computed under the pre-exchange weights are void
    // between lob_td_step_begin and lob_td_step_end no hit list is live and the step's own memo launches follow: nothing to void
    const bool mid_step = e->half_open;
    if (!mid_step) e->hits_ok = false;
    for (int v = 0; v < nv; v++) {
        TimedLaunch t(e, "delta_apply_kernel", nullptr, true);
        hipLaunchKernelGGL(delta_apply_kernel, dim3(2048), dim3(256), 0, e->stream, v ? e->S.theta_b : e->S.theta, e->S.theta_sync + v * M,
                           (const f64*)(e->S.delta + v * M), v ? e->S.theta_b_nz : e->S.theta_nz, e->S.nz_epoch, e->P.M,
                           e->P.memo ? e->S.theta_nzx : (uint32_t*)nullptr, e->S.theta_nzc, e->P.cshift, e->S.theta_nzd, e->S.nzd_terms);
    }
    if (delta_extra(e))
        hipLaunchKernelGGL(rho_delta_apply_kernel, dim3(1), dim3(1), 0, e->stream, e->S.rho, e->S.theta_sync + M * nv, (const f64*)(e->S.delta + M * nv));
    if (e->P.memo && !mid_step) launch_memo(e, e->last_par, 1);  // the current triples under the exchanged weights
    HIPCHK(hipGetLastError());
    return LOB_OK;
}

// ---- sparse exchange (include/lob_engine.h) ------------------------------------------------------------------------------
static int64_t spx_words(const lob_engine* e) { return (int64_t)((size_t)e->P.M / 32 + 1); }
int lob_delta_sparse_supported(lob_engine* e) { return e && e->P.memo && e->P.algo != LOB_ALGO_DOUBLE_Q && e->S.theta_sync && e->S.theta_nzx ? 1 : 0; }
int lob_delta_sparse_maps(lob_engine* e, int32_t world, uint32_t** dev_own, uint32_t** dev_gather, int64_t* words) {
    if (!e || world < 1 || !dev_own || !dev_gather || !words) return LOB_EINVAL;
    if (!lob_delta_sparse_supported(e)) { lob_set_error("lob_delta_sparse_*: needs the shared-theta fast path and lob_delta_init"); return LOB_ESTATE; }
    HIPCHK(hipSetDevice(e->device));
    const int64_t W = spx_words(e);
    if (e->spx_world < world) {
        if (e->spx_gather) { hipFree(e->spx_gather); e->spx_gather = nullptr; }
        HIPCHK(hipMalloc((void**)&e->spx_gather, (size_t)world * W * 4));
        e->spx_world = world;
    }
    if (!e->spx_union) {
        // the pinned word the union's size is handed over in, and its event.  A rank without them would ask for the exact count at
        // every exchange while the others use the fixed one -- collectives of different lengths: it fails HERE instead, where a
        // failing rank makes all of them fail together (see below)
        if (!e->spx_total_host && hipHostMalloc((void**)&e->spx_total_host, sizeof(i64), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); e->spx_total_host = nullptr;
            lob_set_error("lob_delta_sparse_maps: no pinned memory for the exchange's size hand-over");
            return LOB_ENOMEM;
        }
        if (!e->spx_ev && hipEventCreateWithFlags(&e->spx_ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError(); hipHostFree(e->spx_total_host); e->spx_total_host = nullptr;
            lob_set_error("lob_delta_sparse_maps: no event for the exchange's size hand-over");
            return LOB_ENOMEM;
        }
        const int nb = (int)((W + LOB_SPX_BLOCK - 1) / LOB_SPX_BLOCK);
        int rc = dev_alloc(e, &e->spx_union, (size_t)W);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->spx_block_cnt, (size_t)nb);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->spx_block_off, (size_t)nb);
        if (rc == LOB_OK) rc = dev_alloc(e, &e->spx_total, 1);
        if (rc != LOB_OK) return rc;
        // (the first fixed count: a 64th of the table, 4 096 .. 262 144 entries; LOB_SPX_COUNT: the tests' small one)
        e->spx_fixed = std::max<int64_t>(4096, std::min<int64_t>(1 << 18, e->P.M / 64));
        if (const char* g = getenv("LOB_SPX_COUNT")) { const long long v = atoll(g); if (v >= 64) e->spx_fixed = v; }
        e->spx_fixed = std::min<int64_t>(e->spx_fixed, e->P.M);
    }
    // every buffer of the exchange exists before its first collective starts (lob_theta_allreduce calls this once BEFORE the ranks
    // agree on the exchange's form, and a rank that failed here makes all of them fail together): a rank that failed to allocate
    // between two collectives would leave the others waiting in the second.  The packed deltas go into the dense exchange's
    // scratch vector (lob_delta_init: M doubles, idle while the exchange is sparse): the union has at most M entries, and
    // nothing of that size is allocated for it.
    e->spx_buf = e->S.delta;
    e->spx_cap = e->P.M;
    *dev_own = e->S.theta_nzx;
    *dev_gather = e->spx_gather;
    *words = W;
    return LOB_OK;
}
int lob_delta_sparse_pack(lob_engine* e, int32_t world, double** dev_buf, int64_t* count) {
    if (!e || !dev_buf || !count || world < 1 || world > e->spx_world || !e->spx_buf) {
        lob_set_error("lob_delta_sparse_pack: bad argument (call lob_delta_sparse_maps with this world size first)");
        return LOB_EINVAL;
    }
    HIPCHK(hipSetDevice(e->device));
    const int64_t W = spx_words(e);
    const int nb = (int)((W + LOB_SPX_BLOCK - 1) / LOB_SPX_BLOCK);
    {
        TimedLaunch t(e, "delta_begin_kernel", nullptr, true);
        hipLaunchKernelGGL(sparse_union_kernel, dim3(nb), dim3(LOB_SPX_BLOCK), 0, e->stream, (const uint32_t*)e->spx_gather, (int)world, (i64)W, e->spx_union, e->spx_block_cnt);
        hipLaunchKernelGGL(sparse_scan_kernel, dim3(1), dim3(1024), 0, e->stream, (const i32*)e->spx_block_cnt, nb, e->spx_block_off, e->spx_total);
    }
    // the union's size of the exchange BEFORE this one (its copy to pinned memory has had a whole exchange interval to land)
    if (e->spx_ev_pending) {
        HIPCHK(hipEventSynchronize(e->spx_ev));
        if (e->spx_total_host[0] != e->spx_last_total) { e->spx_prev_total = e->spx_last_total; e->spx_last_total = e->spx_total_host[0]; }
        e->spx_ev_pending = false;
        if (e->spx_last_total > e->spx_count) e->spx_overflows++;   // (that exchange left entries for this one)
    }
    const bool no_sync = e->spx_total_host && e->spx_last_total >= 0 && e->spx_prev_total >= 0 &&
                         e->spx_last_total + 2 * std::max<int64_t>(e->spx_last_total - e->spx_prev_total, 0) <= e->spx_fixed;
    i64 count_now;
    if (no_sync) {
        count_now = e->spx_fixed;
        e->spx_nosync++;
    } else {
        // the exact count, with the one synchronisation the exchange used to need every time
        i64 total = 0;
        HIPCHK(hipMemcpyAsync(&total, e->spx_total, sizeof total, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (total > e->spx_cap) { lob_set_error("lob_delta_sparse_pack: more union entries than weights"); return LOB_ESTATE; }  // (cannot happen: cap = M)
        while (e->spx_fixed < 2 * total && e->spx_fixed < e->spx_cap) e->spx_fixed = std::min<int64_t>(e->spx_fixed * 2, e->spx_cap);
        if (total != e->spx_last_total) { e->spx_prev_total = e->spx_last_total; e->spx_last_total = total; }
        count_now = total;
        e->spx_synced++;
    }
    e->spx_count = count_now;
    {
        TimedLaunch t(e, "delta_begin_kernel", nullptr, true);
        hipLaunchKernelGGL(sparse_pack_kernel, dim3(nb), dim3(LOB_SPX_BLOCK), 0, e->stream, (const uint32_t*)e->spx_union, (i64)W, (const i64*)e->spx_block_off,
                           (const f64*)e->S.theta, (const f64*)e->S.theta_sync, e->spx_buf, (i64)count_now);
        if (no_sync) hipLaunchKernelGGL(sparse_tail_kernel, dim3(64), dim3(256), 0, e->stream, e->spx_buf, (const i64*)e->spx_total, (i64)count_now);
    }
    if (e->spx_total_host) {   // this exchange's union size, for the next one
        HIPCHK(hipMemcpyAsync(e->spx_total_host, e->spx_total, sizeof(i64), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipEventRecord(e->spx_ev, e->stream));
        e->spx_ev_pending = true;
    }
    HIPCHK(hipGetLastError());
    *dev_buf = e->spx_buf;
    *count = count_now;
    return LOB_OK;
}
int lob_delta_sparse_apply(lob_engine* e) {
    if (!e || !e->spx_union) { lob_set_error("lob_delta_sparse_apply: no packed exchange to apply"); return LOB_EINVAL; }
    HIPCHK(hipSetDevice(e->device));
    const int64_t W = spx_words(e);
    const int nb = (int)((W + LOB_SPX_BLOCK - 1) / LOB_SPX_BLOCK);
    e->theta_ver++;  // memo records computed under the pre-exchange weights are void
    const bool mid_step = e->half_open;  // (see lob_delta_apply)
    if (!mid_step) e->hits_ok = false;
    if (e->spx_count > 0) {
        TimedLaunch t(e, "delta_apply_kernel", nullptr, true);
        hipLaunchKernelGGL(sparse_apply_kernel, dim3(nb), dim3(LOB_SPX_BLOCK), 0, e->stream, (const uint32_t*)e->spx_union, (i64)W, (const i64*)e->spx_block_off,
                           e->S.theta, e->S.theta_sync, (const f64*)e->spx_buf, e->S.theta_nz, e->S.nz_epoch, e->S.theta_nzx, e->S.theta_nzc, e->P.cshift,
                           e->S.theta_nzd, e->S.nzd_terms, (i64)e->P.M, (i64)e->spx_count);
    }
    if (e->P.memo && !mid_step) launch_memo(e, e->last_par, 1);  // the current triples under the exchanged weights
    HIPCHK(hipGetLastError());
    return LOB_OK;
}

int lob_sync(lob_engine* e) {
    if (!e) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return check_device_errors(e);
}
void* lob_stream(lob_engine* e) { return e ? (void*)e->stream : nullptr; }

int lob_kernel_timing(lob_engine* e, int32_t enable) {
    if (!e) return LOB_EINVAL;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    drain_timers(e);
    e->timers.clear();
    e->timing = enable != 0;
    e->timing_period = enable > 1 ? enable : 1;
    return LOB_OK;
}
int lob_kernel_time_ms(lob_engine* e, const char* kernel, double* avg_ms, int64_t* launches) {
    if (!e || !kernel || !avg_ms || !launches) return LOB_EINVAL;
    hipSetDevice(e->device);
    drain_timers(e);
    auto it = e->timers.find(kernel);
    if (it == e->timers.end() || it->second.launches == 0) { *avg_ms = 0.0; *launches = 0; return LOB_OK; }
    *avg_ms = it->second.total_ms / (double)it->second.launches;
    *launches = it->second.launches;
    return LOB_OK;
}

}  // extern "C"

// Diagnostics (not part of include/lob_engine.h): 1 in a -DLOB_EXPERIMENTS build -- the kernel variants measured and lost
// (env_compact_kernel, the two-wave pre-pass, 32-lane env_step_kernel, two book groups, ...) exist and their switches work.
extern "C" int lob_experiments_enabled(void) { return lobk_experiments(); }

// Diagnostics (not part of include/lob_engine.h): phase clocks of a -DLOB_PROF build, summed over books
// (tools/exp_prof.py); LOB_ESTATE on a regular build.
extern "C" int lob_debug_prof(lob_engine* e, int64_t out[LOB_PROF_N]) {
    if (!e || !out) return LOB_EINVAL;
    if (!e->S.prof) { lob_set_error("lob_debug_prof: not a -DLOB_PROF build"); return LOB_ESTATE; }
    HIPCHK(hipSetDevice(e->device));
    std::vector<i64> h((size_t)e->B * LOB_PROF_N);
    HIPCHK(hipMemcpyAsync(h.data(), e->S.prof, h.size() * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < LOB_PROF_N; i++) out[i] = 0;
    for (size_t b = 0; b < (size_t)e->B; b++)
        for (int i = 0; i < LOB_PROF_N; i++) out[i] += h[b * LOB_PROF_N + i];
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): the state the fast path's shortcuts depend on -- [0] weights the exact
// written-weights map shows, [1] live books, [2] live books without a hit list, [3] sum of the live books' list lengths,
// [4 + n] live books whose list has n entries (the last bin, n = 256: that many or more).  bench.py's sustained leg prints it.
extern "C" int lob_debug_fastpath(lob_engine* e, int64_t* out, int32_t n_out) {
    if (!e || !out || n_out < 4 + LOB_FP_BINS) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    i64* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)(4 + LOB_FP_BINS) * 8));
    hipError_t err = hipMemsetAsync(d, 0, (size_t)(4 + LOB_FP_BINS) * 8, e->stream);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(fastpath_stats_kernel, dim3(512), dim3(256), 0, e->stream, e->S, (i64)e->P.M, e->P.memo ? 1 : 0, d);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(out, d, (size_t)(4 + LOB_FP_BINS) * 8, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    hipFree(d);
    HIPCHK(err);
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): learner steps (combined update) since lob_create by the shape of their update
// -- [0] updates added to their slots by the learn / trace kernels, accumulate_kernel over the list they left (Q(lambda)),
// [1] steps whose learn_q_rest_kernel ran beside the trace kernels on the second stream, [2] accumulate_block_kernel,
// [3] accumulate_kernel over every book; steps with the action selection inside the env kernel (launch_env_fused): [4] books without a
// usable hit list served in-kernel (act_book), [5] through the work list to the wave-per-book act kernel because the learn
// kernels' hand-back count said most books have none (a dense theta), [6] through the work list for another reason (the first
// step on lists, LOB_INLINE_GENERAL=0); [7] of the steps counted under [2], those whose sums went through the dense ids
// (accumulate_dense_kernel).  The tests use it to know which path they have compared with the oracle.
extern "C" int lob_debug_flow(lob_engine* e, int64_t out[8]) {
    if (!e || !out) return LOB_EINVAL;
    for (int i = 0; i < 8; i++) out[i] = e->flow[i];
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): books whose action came from act_light_kernel so far, and the step id of
// the last update that voided the hit lists (-1: never).
extern "C" int lob_debug_light(lob_engine* e, int64_t out[2]) {
    if (!e || !out) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    i64 c[16];
    i32 d = 0;
    { int rc = read_counters(e, c); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(&d, e->S.hl_dirty, sizeof d, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    out[0] = c[5];
    out[1] = d;
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): sparse exchanges without / with a host synchronisation, exchanges whose union outgrew
// the fixed count (their surplus travelled one exchange later), the fixed count, the last union size known to the host.
extern "C" int lob_debug_exchange(lob_engine* e, int64_t out[5]) {
    if (!e || !out) return LOB_EINVAL;
    out[0] = e->spx_nosync; out[1] = e->spx_synced; out[2] = e->spx_overflows; out[3] = e->spx_fixed; out[4] = e->spx_last_total;
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): generations without a combine slot that trace_rest_kernel left to apply_kernel so far.
extern "C" int lob_debug_deferred(lob_engine* e, int64_t out[1]) {
    if (!e || !out) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    i64 c[16];
    { int rc = read_counters(e, c); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    out[0] = c[8];
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): rho of the R-learning agents (`n` = 1 shared, n_books private).
extern "C" int lob_debug_rho(lob_engine* e, double* out, int32_t n) {
    if (!e || !out || n < 1 || !e->S.rho) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(out, e->S.rho, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return LOB_OK;
}

// Diagnostics (not part of include/lob_engine.h): how many weights the last two updates wrote for
// the first time -- the quantity that decides whether act_kernel can reuse learn_kernel's verdicts.
extern "C" int lob_debug_new_weights(lob_engine* e, int32_t out[2]) {
    if (!e || !out) return LOB_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    i32 h[2 * LOB_NZ_WORDS];
    HIPCHK(hipMemcpyAsync(h, e->S.nz_new, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    out[0] = h[0]; out[1] = h[LOB_NZ_WORDS];
    return LOB_OK;
}
