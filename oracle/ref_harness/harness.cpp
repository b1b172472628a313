// TEST INFRASTRUCTURE ONLY — never linked into or called by the product.
//
// Harness around the UNMODIFIED reference sources (compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/).  It drives the
// reference's own classes through their public interfaces:
//   environment::Intraday<>   (include/environment/intraday.h:27-76)
//   rl::State / rl::SARSA / rl::QLearn (include/rl/state.h, include/rl/agent.h)
//   experiment::serial::Learner (include/experiment/serial.h:38-52)
// and dumps per-step trajectories / known-answer vectors that pin the CPU
// restatement in oracle/lob_oracle.cpp and, through it, the HIP engine.
//
// The only substitutions are the two implementation-defined RNG sources
// (SURVEY.md §7 step 1, §8c): the policy's libstdc++ mt19937_64 draws are
// replaced by a Policy subclass drawing from the counter-based generator
// `lob_rng` (rl::Agent takes the policy by unique_ptr, include/rl/agent.h:48),
// and libc rand() (tie-breaks, src/rl/agent.cpp:160, src/rl/policy.cpp:49) is
// interposed below with the same generator.
//
// Modes
//   episode  : replay one book of a binary event stream (lob_engine.h record
//              layout, depth 5) through Intraday + agent, mirroring
//              Learner::_step (src/experiment/serial.cpp:53-70); writes a
//              trajectory file.
//   learner  : same inputs through the reference's own Learner::RunEpisode
//              (src/experiment/serial.cpp:72-93); prints steps and seconds
//              (CPU baseline) and optionally dumps theta.
//   dropin   : (ref_dropin binary, -DLOB_DROPIN, links liblob_engine.so) the reference's UNMODIFIED
//              Learner::RunEpisode + rl::Agent driving environment::GpuIntraday
//              (rl_markets_amd/host/ref_binding/gpu_intraday.h), a subclass of the reference's own
//              environment::Base backed by one book of the GPU engine; writes the same trajectory file
//              as `episode` (minus the post-reset record).
//   tiles    : rl::State::newState(vector<float>&) known-answer vectors.
//   ticks    : Market::ToTicks / ToPrice known-answer vectors.
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

// DoubleAgent::theta_b is private (include/rl/agent.h:82) and the probes below read protected
// members: widen access for the reference headers only.  Access specifiers do not change the
// object layout the reference objects were compiled with.
// (standard / shim headers first so that the widening below touches the reference only)
#include <algorithm>
#include <array>
#include <deque>
#include <exception>
#include <functional>
#include <iomanip>
#include <iterator>
#include <list>
#include <random>
#include <set>
#include <sstream>
#include <stdexcept>
#include <tuple>
#include <utility>
#include <spdlog/spdlog.h>
#include <yaml-cpp/yaml.h>
#define private public
#define protected public
#include "environment/intraday.h"
#include "experiment/serial.h"
#include "market/market.h"
#include "rl/agent.h"
#include "rl/policy.h"
#include "rl/state.h"
#include "rl/tiles.h"
#include "utilities/config.h"
#undef private
#undef protected

#include "../../rl_markets_amd/csrc/lob_stream.h"  // record layout + lob_rng (inputs only)
#include "../lob_oracle.h"                          // oracle_step_rec layout only
#ifdef LOB_DROPIN
#include "../../rl_markets_amd/host/ref_binding/gpu_intraday.h"  // the thing under test in `dropin` mode
#include "../../rl_markets_amd/host/ref_binding/gpu_learner.h"   // ... and in `dropin_learner` mode
#endif

// ---------------------------------------------------------------------------
// RNG injection
static uint64_t g_seed = 1994, g_stream = 0, g_ctr = 0;
static uint64_t next_raw() { return lob_rng(g_seed, g_stream, g_ctr++); }

extern "C" int rand(void) { return (int)(next_raw() >> 33); }

class ReplayPolicy : public rl::Greedy {
public:
    double eps;
    ReplayPolicy(unsigned n_actions, double eps) : rl::Greedy(n_actions, 1), eps(eps) {}
    unsigned int Sample(std::vector<double>& qs) override {
        // rl::EpsilonGreedy::Sample (src/rl/policy.cpp:69-75) with injected draws
        double u = (double)(next_raw() >> 11) * (1.0 / 9007199254740992.0);
        if (u < eps) return (unsigned)(((next_raw() >> 32) * (uint64_t)N_ACTIONS) >> 32);
        return rl::Greedy::Sample(qs);
    }
    double descr() override { return eps; }
};

// rl::Boltzmann (src/rl/policy.cpp:85-122) with its one uniform draw injected: Sample restated only as far
// as the draw forces (the probabilities and the cumulative scan are the reference's own arithmetic).
class ReplayBoltzmann : public rl::Boltzmann {
public:
    ReplayBoltzmann(unsigned n_actions, double tau) : rl::Boltzmann(n_actions, tau, tau, 1, 1) {}
    unsigned int Sample(std::vector<double>& qs) override {
        double z = 0.0;
        for (int a = 0; a < N_ACTIONS; a++) {
            probabilities[a] = std::exp(qs[a] / tau);
            z += probabilities[a];
        }
        double acc = 0.0;
        double r = (double)(next_raw() >> 11) * (1.0 / 9007199254740992.0);
        for (int a = 0; a < N_ACTIONS; a++) {
            acc += probabilities[a] / z;
            if (r < acc) return a;
        }
        return N_ACTIONS - 1;
    }
};
// ---------------------------------------------------------------------------
// Expose protected state of the reference classes (no behaviour change).
class ProbeEnv : public environment::Intraday<> {
public:
    explicit ProbeEnv(Config& c) : environment::Intraday<>(c) {}

    void fill(lob_book_dump& d) {
        memset(&d, 0, sizeof(d));
        for (int l = 0; l < 5; l++) {
            try { d.ask_px[l] = ask_book_.price(l); d.ask_vol[l] = ask_book_.volume(d.ask_px[l]); } catch (...) {}
            try { d.bid_px[l] = bid_book_.price(l); d.bid_vol[l] = bid_book_.volume(d.bid_px[l]); } catch (...) {}
            try { d.ask_last_px[l] = ask_book_.last_price(l); d.ask_last_vol[l] = ask_book_.last_volume(d.ask_last_px[l]); } catch (...) {}
            try { d.bid_last_px[l] = bid_book_.last_price(l); d.bid_last_vol[l] = bid_book_.last_volume(d.bid_last_px[l]); } catch (...) {}
        }
        d.ask_total_volume = ask_book_.total_volume();
        d.bid_total_volume = bid_book_.total_volume();
        d.ask_last_total_volume = ask_book_.last_total_volume();
        d.bid_last_total_volume = bid_book_.last_total_volume();
        d.ask_n_transacted = ask_book_.n_transacted();
        d.bid_n_transacted = bid_book_.n_transacted();
        d.ask_has_order = ask_book_.order_count();
        d.bid_has_order = bid_book_.order_count();
        if (d.ask_has_order) {
            double p = ask_book_.best_open_order_price();
            d.ask_order_px = p;
            d.ask_order_rem = ask_book_.order_remaining_volume(p);
            d.ask_q_head = ask_book_.queue_ahead(p);
            d.ask_q_tail = ask_book_.queue_behind(p);
        }
        if (d.bid_has_order) {
            double p = bid_book_.best_open_order_price();
            d.bid_order_px = p;
            d.bid_order_rem = bid_book_.order_remaining_volume(p);
            d.bid_q_head = bid_book_.queue_ahead(p);
            d.bid_q_tail = bid_book_.queue_behind(p);
        }
        d.position = risk_manager_.exposure();
        d.ask_quote = ask_quote;
        d.bid_quote = bid_quote;
        d.ask_level = ask_level;
        d.bid_level = bid_level;
        d.pnl_step = pnl_step;
        d.momentum_pnl_step = momentum_pnl_step;
        d.lo_vol_step = lo_vol_step;
        d.last_action = last_action;
        d.episode_reward = episode_stats.reward;
        d.episode_pnl = episode_stats.pnl;
        d.episode_bandh = episode_stats.bandh;
        d.spread_mean = spread_window.mean();
        d.target_price = target_price_->get();
        d.time_ms = market ? market->time() : 0;
        d.cursor = -1;
        d.terminal = isTerminal() ? 1 : 0;
        d.total_ticks = tick_stats.total_ticks;
        d.n_traces = -1;
        d.market_buys = trade_stats.market_buys; d.market_sells = trade_stats.market_sells;
        d.ticks_with_ask = tick_stats.ticks_with_ask; d.ticks_with_bid = tick_stats.ticks_with_bid; d.ticks_with_both = tick_stats.ticks_with_both;
        d.ticks_with_position = tick_stats.ticks_with_position; d.ticks_long = tick_stats.ticks_long; d.ticks_short = tick_stats.ticks_short;
        d.ask_transactions = trade_stats.ask_transactions; d.bid_transactions = trade_stats.bid_transactions;
    }
};

template <class A> class ProbeAgent : public A {
public:
    double last_delta = 0.0;
    ProbeAgent(std::unique_ptr<rl::Policy> p, Config& c) : A(std::move(p), c) {}
    double UpdateWeights(rl::State& f, int a, double r, rl::State& t) override {
        last_delta = A::UpdateWeights(f, a, r, t);
        return last_delta;
    }
    double* theta_ptr() { return this->theta; }
    rl::Traces& traces_ref() { return this->traces; }
    long mem() { return this->MEMORY_SIZE; }
};

// ---------------------------------------------------------------------------
static std::string ms_to_str(long t) {
    char buf[32];
    long ms = t % 1000; t /= 1000;
    long s = t % 60; t /= 60;
    long m = t % 60; t /= 60;
    snprintf(buf, sizeof buf, "%02ld:%02ld:%02ld.%03ld", t, m, s, ms);
    return buf;
}

struct Args {
    std::map<std::string, std::string> kv;
    std::string get(const std::string& k, const std::string& d = "") const {
        auto it = kv.find(k);
        return it == kv.end() ? d : it->second;
    }
    long geti(const std::string& k, long d) const { return kv.count(k) ? atol(kv.at(k).c_str()) : d; }
    double getd(const std::string& k, double d) const { return kv.count(k) ? atof(kv.at(k).c_str()) : d; }
};

static Args parse(int argc, char** argv, int from) {
    Args a;
    for (int i = from; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k.rfind("--", 0) == 0) k = k.substr(2);
        a.kv[k] = argv[i + 1];
    }
    return a;
}

static std::unique_ptr<rl::Policy> make_policy(const Args& a) {
    if (a.get("policy", "epsilon_greedy") == "boltzmann") return std::unique_ptr<rl::Policy>(new ReplayBoltzmann(9, a.getd("tau", 1.0)));
    return std::unique_ptr<rl::Policy>(new ReplayPolicy(9, a.getd("eps", 0.8)));
}

// Write the reference's two CSV formats (include/data/basic.h:17-24,49-52)
// for one book of a depth-5 binary stream.
static void write_csvs(const std::vector<uint32_t>& rec, int D, int T, int n_events,
                       const std::string& md_path, const std::string& tas_path) {
    if (D != 5) { fprintf(stderr, "reference records are hard-wired to 5 levels (quirk Q18)\n"); exit(2); }
    const int W = lob_rec_words(D, T);
    FILE* md = fopen(md_path.c_str(), "w");
    FILE* ts = fopen(tas_path.c_str(), "w");
    if (!md || !ts) { perror("csv"); exit(2); }
    fprintf(md, "date,time,ap1,ap2,ap3,ap4,ap5,av1,av2,av3,av4,av5,bp1,bp2,bp3,bp4,bp5,bv1,bv2,bv3,bv4,bv5\n");
    fprintf(ts, "date,time,price,size\n");
    const int date = 20200102;
    long last_t = 0;
    for (int e = 0; e < n_events; e++) {
        const uint32_t* r = &rec[(size_t)e * W];
        long t = (long)(int32_t)r[LOB_REC_TIME];
        last_t = t;
        for (int i = 0; i < T; i++) {
            int32_t v = (int32_t)r[lob_rec_trade_vol(D, T) + i];
            if (v > 0)
                fprintf(ts, "%d,%s,%.9g,%d\n", date, ms_to_str(t - 1).c_str(),
                        (double)lob_bits_f32(r[lob_rec_trade_px(D, T) + i]), v);
        }
        fprintf(md, "%d,%s", date, ms_to_str(t).c_str());
        for (int l = 0; l < D; l++) fprintf(md, ",%.9g", (double)lob_bits_f32(r[lob_rec_ask_px(D, T) + l]));
        for (int l = 0; l < D; l++) fprintf(md, ",%d", (int32_t)r[lob_rec_ask_vol(D, T) + l]);
        for (int l = 0; l < D; l++) fprintf(md, ",%.9g", (double)lob_bits_f32(r[lob_rec_bid_px(D, T) + l]));
        for (int l = 0; l < D; l++) fprintf(md, ",%d", (int32_t)r[lob_rec_bid_vol(D, T) + l]);
        fprintf(md, "\n");
    }
    // Two sentinel trade groups after the last depth row so that the T&S
    // streamer never reports end-of-data before the depth stream does
    // (Streamer::LoadUntil, src/data/streamer.cpp:57-80; TimeAndSales::_LoadNext,
    // src/data/basic.cpp:164-181).
    fprintf(ts, "%d,%s,1.0,1\n", date, ms_to_str(last_t + 3600000).c_str());
    fprintf(ts, "%d,%s,1.0,1\n", date, ms_to_str(last_t + 3600001).c_str());
    fclose(md);
    fclose(ts);
}

static std::string make_yaml(const Args& a, const std::string& path) {
    std::ofstream f(path);
    f << "debug:\n    inspect_books: false\n    random_seed: " << a.geti("agent_seed", 1) << "\n";
    f << "learning:\n";
    f << "    memory_size: " << a.geti("mem", 20000000) << "\n";
    f << "    n_tilings: 32\n    n_actions: 9\n";
    f << "    algorithm: " << a.get("algo", "sarsa") << "\n";
    f << "    group_weights: [" << a.get("w0", "0.65") << ", " << a.get("w1", "0.25") << ", " << a.get("w2", "0.10") << "]\n";
    f << "    gamma: " << a.get("gamma", "0.975") << "\n    lambda: " << a.get("lambda", "0.85") << "\n";
    f << "    omega: 1.0\n    alpha_start: " << a.get("alpha", "0.001") << "\n    alpha_floor: " << a.get("alpha", "0.001") << "\n";
    f << "    beta: " << a.get("beta", "0.005") << "\n";  // RLearn / OnlineRLearn (src/rl/agent.cpp:357-412)
    if (a.geti("random_init", 0)) f << "    random_init: true\n";  // Agent::Agent, src/rl/agent.cpp:37-39
    f << "policy:\n    type: epsilon_greedy\n    eps_init: 0.8\n    eps_floor: 0.0001\n    eps_T: 800\n";
    f << "    spread_lookback: " << a.geti("lb_spread", 45) << "\n";
    f << "reward:\n    measure: " << a.get("reward", "pnl_damped") << "\n";
    f << "    damping_factor: " << a.get("damping", "0.15") << "\n    pnl_lookback: " << a.geti("lb_pnl", 0) << "\n";
    f << "    pos_weight: " << a.get("pos_weight", "0.0") << "\n    pnl_weight: " << a.get("pnl_weight", "1.0") << "\n";
    f << "state:\n    variables: [" << a.get("vars", "\"pos\", \"a_dist\", \"b_dist\", \"mpm\", \"spd\", \"vol\", \"imb\", \"svl\"") << "]\n";
    f << "    lookback:\n        mpm: " << a.geti("lb_mpm", 15) << "\n        vlt: " << a.geti("lb_vlt", 60)
      << "\n        svl: " << a.geti("lb_svl", 60) << "\n        rsi: " << a.geti("lb_rsi", 0)
      << "\n        vwap: " << a.geti("lb_vwap", 0) << "\n";
    f << "market:\n    transaction_fee: 0.0\n    target_price:\n        type: " << a.get("tp", "midprice")
      << "\n        lookback: " << a.geti("lb_target", 1) << "\n";
    f << "    latency:\n        type: fixed\n        floor: 0.0\n";
    f << "    pos_ub: " << a.geti("pos_ub", 50) << "\n    pos_lb: " << a.geti("pos_lb", -50) << "\n";
    f << "    order_size: " << a.geti("order_size", 10) << "\n";
    f << "logging:\n    log_learning: true\n    log_backtest: " << (a.geti("backtest", 0) ? "true" : "false") << "\n    max_size: 1000000\n";
    f << "output_dir: /tmp/\n";
    f.close();
    return path;
}

static std::vector<uint32_t> load_book(const std::string& path, int D, int T, int n_events, long book) {
    const int W = lob_rec_words(D, T);
    std::vector<uint32_t> rec((size_t)n_events * W);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { perror(path.c_str()); exit(2); }
    fseek(f, (long)((size_t)book * n_events * W * 4), SEEK_SET);
    if (fread(rec.data(), 4, rec.size(), f) != rec.size()) { fprintf(stderr, "short stream read\n"); exit(2); }
    fclose(f);
    return rec;
}

// Trajectory record written per step: oracle_step_rec (oracle/lob_oracle.h).
typedef oracle_step_rec StepRec;

// sparse dump of DoubleAgent::theta_b (no-op for single-vector agents)
static void dump_theta_b_impl(rl::DoubleAgent* ag, const std::string& path) {
    double* tb = ag->theta_b;
    long M = ag->MEMORY_SIZE;
    FILE* f = fopen(path.c_str(), "wb");
    int64_t n = 0;
    for (long i = 0; i < M; i++) if (tb[i] != 0.0) n++;
    fwrite(&n, 8, 1, f);
    for (long i = 0; i < M; i++)
        if (tb[i] != 0.0) { int64_t idx = i; fwrite(&idx, 8, 1, f); fwrite(&tb[i], 8, 1, f); }
    fclose(f);
}
static void dump_theta_b_impl(void*, const std::string&) {}
template <class A> static void dump_theta_b(A& ag, const std::string& path) {
    typedef typename std::conditional<std::is_base_of<rl::DoubleAgent, A>::value, rl::DoubleAgent*, void*>::type P;
    dump_theta_b_impl((P)&ag, path);
}

template <class AGENT>
static int run_episode(const Args& a, Config& c, ProbeEnv& env, const std::string& out_path) {
    long max_steps = a.geti("steps", 1L << 40);
    ProbeAgent<AGENT> agent(make_policy(a), c);
    if (a.kv.count("theta_in")) {
        FILE* f = fopen(a.get("theta_in").c_str(), "rb");
        if (!f || fread(agent.theta_ptr(), 8, agent.mem(), f) != (size_t)agent.mem()) { fprintf(stderr, "theta_in\n"); return 2; }
        fclose(f);
    }
    rl::State s1(c), s2(c);
    rl::State *state = &s1, *last_state = &s2;

    FILE* out = fopen(out_path.c_str(), "wb");
    if (!out) { perror("out"); return 2; }

    StepRec r;
    auto dump = [&](int action, double reward, double td, rl::State* st) {
        memset(&r, 0, sizeof r);
        r.action = action;
        r.reward = reward;
        r.td = td;
        auto& v = st->toVector();
        r.n_vars = (int)v.size();
        for (size_t i = 0; i < v.size() && i < LOB_MAX_VARS; i++) r.vars[i] = v[i];
        r.rng_ctr = g_ctr;
        env.fill(r.book);
        r.book.n_traces = agent.traces_ref().n_nonzero_traces;
        fwrite(&r, sizeof r, 1, out);
    };
    long steps = 0;
    int end_reason = 0;
    std::string ends;
    const int episodes = (int)a.geti("episodes", 1);
    for (int ep = 0; ep < episodes; ep++) {
        // src/main.cpp:55: every episode re-opens its data files
        if (ep > 0) env.LoadData(a.get("ticker", "HSBA.L"), a.get("md"), a.get("tas"));
        // Runner::RunEpisode prologue (src/experiment/serial.cpp:18-26)
        if (!env.Initialise()) { fprintf(stderr, "Initialise failed\n"); return 3; }
        last_state->newState(env);
        dump(-1, 0.0, 0.0, last_state);  // state after reset
        long ep_steps = 0;
        end_reason = 0;
        // Learner::_step (src/experiment/serial.cpp:53-70)
        while (ep_steps < max_steps) {
            std::swap(state, last_state);
            if (env.isTerminal()) { end_reason = 1; break; }
            int action = agent.action(*last_state);
            if (!env.performAction(action)) { end_reason = 2; break; }
            state->newState(env);
            double reward = env.getReward();
            agent.HandleTransition(*last_state, action, reward, *state);
            steps++;
            ep_steps++;
            dump(action, reward, agent.last_delta, state);
        }
        if (a.geti("clear_inventory", 0) || episodes > 1) {
            env.ClearInventory();  // Runner::RunEpisode epilogue (serial.cpp:31)
            dump(-2, 0.0, 0.0, state);
        }
        if (episodes > 1) agent.HandleTerminal(ep);  // Learner::RunEpisode (serial.cpp:79)
        ends += (ep ? "," : "") + std::to_string(end_reason);
    }
    fclose(out);

    if (a.kv.count("theta_out")) {
        // sparse dump: int64 count, then (int64 index, double value) pairs
        FILE* f = fopen(a.get("theta_out").c_str(), "wb");
        int64_t n = 0;
        for (long i = 0; i < agent.mem(); i++) if (agent.theta_ptr()[i] != 0.0) n++;
        fwrite(&n, 8, 1, f);
        for (long i = 0; i < agent.mem(); i++)
            if (agent.theta_ptr()[i] != 0.0) {
                int64_t idx = i;
                fwrite(&idx, 8, 1, f);
                fwrite(&agent.theta_ptr()[i], 8, 1, f);
            }
        fclose(f);
    }
    if (a.kv.count("theta_b_out")) {
        dump_theta_b(agent, a.get("theta_b_out"));
    }
    if (a.geti("backtest", 0)) {
        // the testing phase of src/main.cpp:216-226: GoGreedy, re-open the data, the reference's own Backtester
        // (Runner::RunEpisode + Backtester::_step, serial.cpp:18-34,124-137) with log_backtest on; the rows
        // Intraday::LogProfit hands to its "profit_log" logger go to --profit_out (12 doubles per row)
        agent.GoGreedy();
        env.LoadData(a.get("ticker", "HSBA.L"), a.get("md"), a.get("tas"));
        experiment::serial::Backtester bt(c, env);
        bool ok = bt.RunEpisode(&agent);
        auto lg = spdlog::get("profit_log");
        if (!lg) { fprintf(stderr, "no profit_log logger\n"); return 4; }
        FILE* f = fopen(a.get("profit_out").c_str(), "wb");
        if (!f) { perror("profit_out"); return 2; }
        int64_t n = 0;
        for (auto& row : lg->rows) if (row.size() == 12) n++;   // (the header line has no numeric argument)
        fwrite(&n, 8, 1, f);
        for (auto& row : lg->rows) if (row.size() == 12) fwrite(row.data(), 8, 12, f);
        lob_book_dump d;
        env.fill(d);
        fwrite(&d, sizeof d, 1, f);   // the state Runner::RunEpisode leaves behind (after ClearInventory)
        fclose(f);
        printf("{\"backtest_ok\": %d, \"backtest_rows\": %lld}\n", ok ? 1 : 0, (long long)n);
    }
    if (a.kv.count("traces_out")) {
        FILE* f = fopen(a.get("traces_out").c_str(), "wb");
        auto& tr = agent.traces_ref();
        int32_t n = tr.n_nonzero_traces;
        fwrite(&n, 4, 1, f);
        for (int i = 0; i < n; i++) {
            int32_t idx = tr.nonzero_traces[i];
            float e = tr.get(idx);
            fwrite(&idx, 4, 1, f);
            fwrite(&e, 4, 1, f);
        }
        fclose(f);
    }
    // what Agent::HandleTransition handed to its "model_log" logger (agent.cpp:93-100: one row per 1000 updates)
    std::string ml;
    if (auto lg = spdlog::get("model_log"))
        for (auto& row : lg->rows)
            if (row.size() == 1) { char buf[64]; snprintf(buf, sizeof buf, "%s%.17g", ml.empty() ? "" : ", ", row[0]); ml += buf; }
    printf("{\"steps\": %ld, \"end\": %d, \"ends\": [%s], \"rng_ctr\": %llu, \"sizeof_steprec\": %zu, \"model_log\": [%s]}\n", steps, end_reason,
           ends.c_str(), (unsigned long long)g_ctr, sizeof(StepRec), ml.c_str());
    return 0;
}

template <class AGENT>
static int run_learner(const Args& a, Config& c, ProbeEnv& env) {
    double eps = a.getd("eps", 0.8);
    int episodes = (int)a.geti("episodes", 1);
    ProbeAgent<AGENT> agent(std::unique_ptr<rl::Policy>(new ReplayPolicy(9, eps)), c);
    experiment::serial::Learner learner(c, env);
    double best = 1e30, total = 0;
    long steps = 0;
    for (int ep = 0; ep < episodes; ep++) {
        g_ctr = 0;
        // re-open the CSVs for each episode
        env.LoadData(a.get("ticker", "HSBA.L"), a.get("md"), a.get("tas"));
        auto t0 = std::chrono::steady_clock::now();
        bool ok = learner.RunEpisode(&agent);
        auto t1 = std::chrono::steady_clock::now();
        double sec = std::chrono::duration<double>(t1 - t0).count();
        lob_book_dump d;
        env.fill(d);
        steps = d.total_ticks;
        total += sec;
        if (sec < best) best = sec;
        (void)ok;
    }
    if (a.kv.count("theta_out")) {
        FILE* f = fopen(a.get("theta_out").c_str(), "wb");
        int64_t n = 0;
        for (long i = 0; i < agent.mem(); i++) if (agent.theta_ptr()[i] != 0.0) n++;
        fwrite(&n, 8, 1, f);
        for (long i = 0; i < agent.mem(); i++)
            if (agent.theta_ptr()[i] != 0.0) {
                int64_t idx = i;
                fwrite(&idx, 8, 1, f);
                fwrite(&agent.theta_ptr()[i], 8, 1, f);
            }
        fclose(f);
    }
    printf("{\"steps_per_episode\": %ld, \"episodes\": %d, \"best_sec\": %.6f, \"mean_sec\": %.6f, \"steps_per_sec\": %.1f}\n",
           steps, episodes, best, total / episodes, steps / best);
    return 0;
}

#ifdef LOB_DROPIN
// The agent records a trajectory row from inside the reference's own HandleTransition (UpdateWeights is
// its last RNG-drawing stage): the Learner driving it is the reference's, untouched.
template <class A> class RecordingAgent : public ProbeAgent<A> {
public:
    environment::GpuIntraday* env = nullptr;
    FILE* out = nullptr;
    long steps = 0;
    RecordingAgent(std::unique_ptr<rl::Policy> p, Config& c) : ProbeAgent<A>(std::move(p), c) {}
    double UpdateWeights(rl::State& f, int a, double r, rl::State& t) override {
        const double d = ProbeAgent<A>::UpdateWeights(f, a, r, t);
        StepRec rec;
        memset(&rec, 0, sizeof rec);
        rec.action = a;
        rec.reward = r;
        rec.td = d;
        auto& v = t.toVector();
        rec.n_vars = (int)v.size();
        for (size_t i = 0; i < v.size() && i < LOB_MAX_VARS; i++) rec.vars[i] = v[i];
        rec.rng_ctr = g_ctr;
        rec.book = env->book();
        rec.book.n_traces = this->traces_ref().n_nonzero_traces;
        fwrite(&rec, sizeof rec, 1, out);
        steps++;
        return d;
    }
};
template <class AGENT>
static int run_dropin(const Args& a, Config& c, environment::GpuIntraday& env, const std::string& out_path) {
    RecordingAgent<AGENT> agent(make_policy(a), c);
    agent.env = &env;
    agent.out = fopen(out_path.c_str(), "wb");
    if (!agent.out) { perror("out"); return 2; }
    experiment::serial::Learner learner(c, env);           // the reference's runner, unmodified
    g_ctr = 0;  // (bringing up the HIP runtime goes through the interposed rand() too)
    const bool ok = learner.RunEpisode(&agent);            // Initialise -> _step ... -> ClearInventory -> HandleTerminal
    fclose(agent.out);
    if (a.kv.count("theta_out")) {
        FILE* f = fopen(a.get("theta_out").c_str(), "wb");
        int64_t n = 0;
        for (long i = 0; i < agent.mem(); i++) if (agent.theta_ptr()[i] != 0.0) n++;
        fwrite(&n, 8, 1, f);
        for (long i = 0; i < agent.mem(); i++)
            if (agent.theta_ptr()[i] != 0.0) { int64_t idx = i; fwrite(&idx, 8, 1, f); fwrite(&agent.theta_ptr()[i], 8, 1, f); }
        fclose(f);
    }
    printf("{\"steps\": %ld, \"ok\": %d, \"rng_ctr\": %llu, \"sizeof_steprec\": %zu, \"episode_reward\": %.17g, \"episode_pnl\": %.17g}\n",
           agent.steps, ok ? 1 : 0, (unsigned long long)g_ctr, sizeof(StepRec), env.getEpisodeReward(), env.getEpisodePnL());
    return 0;
}
// `dropin_learner`: the call sequence of src/main.cpp's train() (:47-70) -- runner.RunEpisode(agent) through a Runner& and an
// rl::Agent*, then the episode getters of environment::Base -- over experiment::serial::GpuLearner / rl::GpuAgent /
// environment::GpuIntraday with `books` books in one engine and the weights in HBM.
static int run_dropin_learner(const Args& a, Config& c, environment::GpuIntraday& env) {
    std::unique_ptr<rl::Agent> owner(new rl::GpuAgent(make_policy(a), c, env));
    rl::Agent* agent = owner.get();
    experiment::serial::GpuLearner gl(c, env, (int)a.geti("steps_per_call", 8));
    experiment::serial::Runner& runner = gl;
    environment::Base& base = env;
    const int episodes = (int)a.geti("episodes", 1);
    for (int ep = 0; ep < episodes; ep++) {
        if (!runner.RunEpisode(agent)) { fprintf(stderr, "RunEpisode failed\n"); return 3; }
        printf("{\"episode\": %d, \"reward\": %.17g, \"pnl\": %.17g, \"rho\": %.17g, \"nTr\": %d, \"order_ratio\": %.9g, \"steps\": %lu, \"descr\": %.17g}\n",
               ep + 1, base.getEpisodeReward(), base.getEpisodePnL(), base.getMeanEpisodeReward(), base.getTotalTransactions(), (double)base.getOrderRatio(),
               gl.step_counter(), agent->policy->descr());
    }
    if (auto tl = spdlog::get("training_log")) {   // (--log_learning 1) what GpuLearner handed to the training_log logger: the numeric columns
        for (auto& row : tl->rows) {
            if (row.empty()) continue;                 // the header line
            printf("{\"training_log\": [");
            for (size_t i = 0; i < row.size(); i++) printf("%s%.17g", i ? ", " : "", row[i]);
            printf("]}\n");
        }
    }
    if (auto ml = spdlog::get("model_log")) {      // ... and to the model_log logger (Agent::HandleTransition's rows, agent.cpp:93-100)
        printf("{\"model_log\": [");
        bool first = true;
        for (auto& row : ml->rows)
            if (row.size() == 1) { printf("%s%.17g", first ? "" : ", ", row[0]); first = false; }
        printf("]}\n");
    }
    if (a.kv.count("stats_out")) base.writeStats(a.get("stats_out"));
    if (a.kv.count("theta_out")) {
        const std::string path = a.get("theta_out");
        agent->write_theta(path + ".raw");   // Agent::write_theta, unmodified (non-virtual): the host copy GpuLearner keeps current
        const double* th = static_cast<rl::GpuAgent*>(agent)->theta_host();
        const long M = env.params().memory_size;
        FILE* f = fopen(path.c_str(), "wb");
        int64_t n = 0;
        for (long i = 0; i < M; i++) if (th[i] != 0.0) n++;
        fwrite(&n, 8, 1, f);
        for (long i = 0; i < M; i++)
            if (th[i] != 0.0) { int64_t idx = i; fwrite(&idx, 8, 1, f); fwrite(&th[i], 8, 1, f); }
        fclose(f);
    }
    return 0;
}
#endif

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_harness <episode|learner|tiles|ticks> --key value ...\n"); return 1; }
    std::string mode = argv[1];
    Args a = parse(argc, argv, 2);
    g_seed = (uint64_t)a.geti("seed", 1994);
    g_stream = (uint64_t)a.geti("rng_stream", a.geti("book", 0));
    g_ctr = 0;

    if (mode == "tiles") {
        // in: float32[n][V]; out: int32[n][9][96]  (rl::State, src/rl/state.cpp:45-65)
        long M = a.geti("mem", 20000000);
        int V = (int)a.geti("nvars", 8);
        std::ifstream in(a.get("in"), std::ios::binary);
        std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        size_t n = buf.size() / (4 * V);
        const float* fv = (const float*)buf.data();
        FILE* out = fopen(a.get("out").c_str(), "wb");
        rl::State st(M, 9, 32);
        for (size_t i = 0; i < n; i++) {
            std::vector<float> v(fv + i * V, fv + (i + 1) * V);
            st.newState(v);
            for (int act = 0; act < 9; act++) {
                auto& f = st.getFeatures(act);
                fwrite(f.data(), 4, 96, out);
            }
        }
        fclose(out);
        return 0;
    }
    if (mode == "rndseq") {
        // The 2048-entry table of hash_UNH (src/rl/tiles.cpp:133), read back through the
        // reference function itself: with one coordinate k, increment 0 and m > 2^32 the
        // hash returns rndseq[k & 2047] (tiles.cpp:152-166).
        FILE* out = fopen(a.get("out").c_str(), "wb");
        for (int k = 0; k < 2048; k++) {
            int c = k;
            uint32_t v = (uint32_t)hash_UNH(&c, 1, 1L << 40, 0);
            fwrite(&v, 4, 1, out);
        }
        fclose(out);
        return 0;
    }
    if (mode == "ticks") {
        // in: float64[n] prices; out: int32[n] ToTicks, float64[n] ToPrice(ToTicks), float64[n] tick_size
        std::string ticker = a.get("ticker", "HSBA.L");
        std::string sym = ticker.substr(0, ticker.find('.')), ven = ticker.substr(ticker.find('.') + 1);
        market::Market* m = market::Market::make_market(sym, ven);
        std::ifstream in(a.get("in"), std::ios::binary);
        std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        size_t n = buf.size() / 8;
        const double* pv = (const double*)buf.data();
        FILE* out = fopen(a.get("out").c_str(), "wb");
        for (size_t i = 0; i < n; i++) {
            int32_t t = m->ToTicks(pv[i]);
            double back = m->ToPrice(t), ts = m->tick_size(pv[i]);
            fwrite(&t, 4, 1, out);
            int32_t pad = 0;
            fwrite(&pad, 4, 1, out);
            fwrite(&back, 8, 1, out);
            fwrite(&ts, 8, 1, out);
        }
        fclose(out);
        printf("{\"open_ms\": %ld, \"close_ms\": %ld}\n", m->open_time(), m->close_time());
        return 0;
    }

    // ---- episode / learner -------------------------------------------------
    int D = (int)a.geti("depth", 5), T = (int)a.geti("trades", 2);
    int n_events = (int)a.geti("events", 0);
    long book = a.geti("book", 0);
    std::string tmp = a.get("tmp", "/tmp/ref_harness_" + std::to_string((long)getpid()));
    std::string md = tmp + "_md.csv", tas = tmp + "_tas.csv", yaml = tmp + ".yaml";
    if (a.kv.count("stream")) {
        auto rec = load_book(a.get("stream"), D, T, n_events, book);
        write_csvs(rec, D, T, n_events, md, tas);
    } else {
        md = a.get("md");
        tas = a.get("tas");
    }
    a.kv["md"] = md;
    a.kv["tas"] = tas;
    make_yaml(a, yaml);
    Config c(yaml);
#ifdef LOB_DROPIN
    if (mode == "dropin_learner") {
        environment::GpuIntraday genv(c, 0, (int)a.geti("books", 1), true);
        genv.set_rng(g_seed, g_stream);
        genv.LoadData(a.get("ticker", "HSBA.L"), md, tas);
        const int rc = run_dropin_learner(a, c, genv);
        if (!a.geti("keep", 0) && a.kv.count("stream")) { remove(md.c_str()); remove(tas.c_str()); }
        remove(yaml.c_str());
        return rc;
    }
    if (mode == "dropin") {
        environment::GpuIntraday genv(c);
        genv.LoadData(a.get("ticker", "HSBA.L"), md, tas);   // the call site of src/main.cpp:55 with the class swapped
        const std::string algo = a.get("algo", "sarsa");
        int rc = 2;
        if (algo == "sarsa") rc = run_dropin<rl::SARSA>(a, c, genv, a.get("out", tmp + ".traj"));
        else if (algo == "q_learn") rc = run_dropin<rl::QLearn>(a, c, genv, a.get("out", tmp + ".traj"));
        else if (algo == "double_q_learn") rc = run_dropin<rl::DoubleQLearn>(a, c, genv, a.get("out", tmp + ".traj"));
        else if (algo == "r_learn") rc = run_dropin<rl::RLearn>(a, c, genv, a.get("out", tmp + ".traj"));
        else if (algo == "online_r_learn") rc = run_dropin<rl::OnlineRLearn>(a, c, genv, a.get("out", tmp + ".traj"));
        else if (algo == "double_r_learn") rc = run_dropin<rl::DoubleRLearn>(a, c, genv, a.get("out", tmp + ".traj"));
        if (!a.geti("keep", 0) && a.kv.count("stream")) { remove(md.c_str()); remove(tas.c_str()); }
        remove(yaml.c_str());
        return rc;
    }
#endif
    ProbeEnv env(c);
    env.LoadData(a.get("ticker", "HSBA.L"), md, tas);
    std::string algo = a.get("algo", "sarsa");
    int rc;
    if (mode == "episode") {
        if (algo == "sarsa") rc = run_episode<rl::SARSA>(a, c, env, a.get("out", tmp + ".traj"));
        else if (algo == "q_learn") rc = run_episode<rl::QLearn>(a, c, env, a.get("out", tmp + ".traj"));
        else if (algo == "double_q_learn") rc = run_episode<rl::DoubleQLearn>(a, c, env, a.get("out", tmp + ".traj"));
        else if (algo == "r_learn") rc = run_episode<rl::RLearn>(a, c, env, a.get("out", tmp + ".traj"));
        else if (algo == "online_r_learn") rc = run_episode<rl::OnlineRLearn>(a, c, env, a.get("out", tmp + ".traj"));
        else if (algo == "double_r_learn") rc = run_episode<rl::DoubleRLearn>(a, c, env, a.get("out", tmp + ".traj"));
        else { fprintf(stderr, "unknown algo\n"); rc = 2; }
    } else if (mode == "learner") {
        if (algo == "sarsa") rc = run_learner<rl::SARSA>(a, c, env);
        else if (algo == "q_learn") rc = run_learner<rl::QLearn>(a, c, env);
        else if (algo == "double_q_learn") rc = run_learner<rl::DoubleQLearn>(a, c, env);
        else if (algo == "r_learn") rc = run_learner<rl::RLearn>(a, c, env);
        else if (algo == "online_r_learn") rc = run_learner<rl::OnlineRLearn>(a, c, env);
        else if (algo == "double_r_learn") rc = run_learner<rl::DoubleRLearn>(a, c, env);
        else { fprintf(stderr, "unknown algo\n"); rc = 2; }
    } else {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        rc = 1;
    }
    if (!a.geti("keep", 0) && a.kv.count("stream")) {
        remove(md.c_str());
        remove(tas.c_str());
    }
    remove(yaml.c_str());
    return rc;
}
