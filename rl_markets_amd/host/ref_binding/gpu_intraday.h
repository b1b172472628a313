// environment::GpuIntraday -- the reference-side binding of the engine: a subclass of the reference's OWN
// environment::Base (/root/reference/include/environment/base.h:37-151), compiled against the
// reference's headers, so that its unmodified experiment::serial::Learner / Backtester
// (src/experiment/serial.cpp:18-34,53-70,124-137) and rl::Agent drive one book of the GPU engine through
// exactly the call sites they use for environment::Intraday<>:
//
//     Config c(path);
//     environment::GpuIntraday env(c);                        // was: environment::Intraday<> env(c);
//     env.LoadData("HSBA.L", md_csv, tas_csv);                // same signature (intraday.h:67)
//     experiment::serial::Learner learner(c, env);            // unchanged
//     learner.RunEpisode(agent);                              // unchanged
//
// This is the single-book (B = 1) plumbing of BASELINE config 1; the batched path goes through
// include/lob_engine.h directly (rl_markets_amd/host/lob_host.hpp).  Not part of liblob_engine.so and
// not built on the GPU box: it needs the reference checkout (the test driver that runs the reference's
// Learner over it is the `dropin` mode of the reference harness in the test tree).
//
// What the virtual interface carries: Initialise(), performAction(), getState(), isTerminal(),
// getEpisodeId().  Base::getReward(), getEpisodeReward(), getEpisodePnL() and ClearInventory() are NOT
// virtual in the reference; they read Base's protected members, so after every step this class
// mirrors the engine's values (pnl_step, momentum_pnl_step, lo_vol_step, position, episode totals) into
// them: getReward() then evaluates the reference's own formula on the engine's numbers.  That covers
// the rewards built from those members (none, pnl, pnl_damped, lovol, mm_linear, mm_exp, mm_div); `spread` and
// `normed` read Base's rolling windows, which live on the GPU here: the constructor rejects them unless
// the one-line change of INTEGRATION.md (`virtual` on Base::getReward) is applied.  Base::ClearInventory()
// (non-virtual too; Runner::RunEpisode's epilogue calls it through a Base&) runs the REFERENCE's market
// order on Base's own books: so the current snapshot and the position are mirrored into them as well,
// the reference code computes the same fill from the same levels, and the engine is brought in step
// (lob_clear_inventory) before its next call.  (Base's books only see one snapshot per step, so their
// cumulative total_volume_ -- the abort test of WalkTheBook, quirk Q1 -- is smaller than the engine's;
// both exceed any reachable inventory by orders of magnitude.)
#ifndef LOB_REF_BINDING_GPU_INTRADAY_H
#define LOB_REF_BINDING_GPU_INTRADAY_H

#include <array>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "environment/base.h"

#include "../../../include/lob_engine.h"

namespace environment {

class GpuIntraday : public Base {
    lob_engine* engine_ = nullptr;
    lob_params params_;
    int device_;
    bool have_data_ = false;
    int init_date_ = 0;
    lob_book_dump last_;  // the engine's view (of book 0) after the most recent call
    int n_books_ = 1;
    bool device_learning_ = false;

    static void check(int rc, const char* what) {
        if (rc != LOB_OK) throw std::runtime_error(std::string(what) + ": " + lob_last_error());
    }
    // the engine's numbers into the members Base's non-virtual getters read
    void mirror() {
        check(lob_get_book(engine_, 0, &last_), "GpuIntraday");
        last_action = last_.last_action;
        lo_vol_step = last_.lo_vol_step;
        pnl_step = last_.pnl_step;
        momentum_pnl_step = last_.momentum_pnl_step;
        ask_quote = last_.ask_quote;
        bid_quote = last_.bid_quote;
        risk_manager_.Update((long)last_.position - risk_manager_.exposure());
        // the current snapshot into Base's own books (what Base::ClearInventory walks)
        std::array<double, 5> ap, bp;
        std::array<long, 5> av, bv;
        bool whole = true;
        for (int l = 0; l < 5; l++) {
            ap[l] = last_.ask_px[l]; av[l] = (long)last_.ask_vol[l];
            bp[l] = last_.bid_px[l]; bv[l] = (long)last_.bid_vol[l];
            whole = whole && ap[l] > 0.0 && bp[l] > 0.0 && av[l] > 0 && bv[l] > 0;
        }
        if (whole) {
            ask_book_.StashState(); ask_book_.ApplyChanges(ap, av);
            bid_book_.StashState(); bid_book_.ApplyChanges(bp, bv);
        }
        // Base::getReward()'s `spread` measure divides by spread_window.mean(): a one-slot window holding the engine's mean
        // (RollingMean is not assignable -- a const member --: rebuilt in place, empty, then mean = 0 + (v - 0) / 1 = v exactly)
        spread_window.~RollingMean<double>();
        new (&spread_window) RollingMean<double>(1);
        spread_window.push(last_.spread_mean);
        episode_stats.reward = last_.episode_reward;
        episode_stats.pnl = last_.episode_pnl;
        episode_stats.bandh = last_.episode_bandh;
        tick_stats.total_ticks = last_.total_ticks;
        // TradeStatistics / TickStatistics (statistics.h:19-50): Base::getTotalTransactions / getOrderRatio / writeStats read these
        trade_stats.ask_transactions = last_.ask_transactions; trade_stats.bid_transactions = last_.bid_transactions;  // (the snapshot of the last UpdateStats)
        trade_stats.market_buys = last_.market_buys; trade_stats.market_sells = last_.market_sells;
        tick_stats.ticks_with_ask = last_.ticks_with_ask; tick_stats.ticks_with_bid = last_.ticks_with_bid;
        tick_stats.ticks_with_both = last_.ticks_with_both; tick_stats.ticks_with_position = last_.ticks_with_position;
        tick_stats.ticks_long = last_.ticks_long; tick_stats.ticks_short = last_.ticks_short;
    }

    // Base::ClearInventory ran on the mirror (the reference's runner calls it through a Base&): same market
    // order on the engine's side
    void sync_inventory() {
        if (engine_ && have_data_ && risk_manager_.exposure() != (long)last_.position) {
            const long after = risk_manager_.exposure();
            check(lob_clear_inventory(engine_), "ClearInventory");
            check(lob_get_book(engine_, 0, &last_), "GpuIntraday");
            if ((long)last_.position != after) throw std::runtime_error("GpuIntraday: the engine's inventory disagrees with Base's after ClearInventory");
        }
    }

protected:
    // the engine steps whole actions: Base's per-event hooks are never reached
    void DoAction(int) override {}
    bool NextState() override { return false; }
    void LogProfit(int, double, double) override {}
    void LogTrade(char, char, double, long, double) override {}

public:
    // The configuration reads of Base / Intraday (src/environment/base.cpp:14-115,
    // src/environment/intraday.cpp:37-82) folded into lob_params; Base(c) itself still runs (its own
    // books and windows stay empty).
    // `n_books` > 1 / `device_learning`: the batched use -- B books in one engine, the learner's weights, traces and TD
    // updates in HBM (experiment::serial::GpuLearner + rl::GpuAgent, gpu_learner.h); the virtual interface then shows book 0.
    explicit GpuIntraday(Config& c, int device = 0, int n_books = 1, bool device_learning = false)
        : Base(c), device_(device), n_books_(n_books < 1 ? 1 : n_books), device_learning_(device_learning) {
        memset(&last_, 0, sizeof last_);
        lob_default_params(&params_);
        static const std::map<std::string, int> v2i = {
            {"pos", LOB_VAR_POS}, {"spd", LOB_VAR_SPD}, {"mpm", LOB_VAR_MPM}, {"imb", LOB_VAR_IMB},
            {"svl", LOB_VAR_SVL}, {"vol", LOB_VAR_VOL}, {"rsi", LOB_VAR_RSI}, {"vwap", LOB_VAR_VWAP},
            {"a_dist", LOB_VAR_A_DIST}, {"a_queue", LOB_VAR_A_QUEUE}, {"b_dist", LOB_VAR_B_DIST},
            {"b_queue", LOB_VAR_B_QUEUE}, {"last_action", LOB_VAR_LAST_ACTION}};
        auto v = c["state"]["variables"].as<std::list<std::string>>();
        if (v.size() > LOB_MAX_VARS) throw std::runtime_error("too many state variables");
        params_.n_vars = 0;
        for (const auto& name : v) params_.vars[params_.n_vars++] = v2i.at(name);  // out_of_range like intraday.cpp:50-58
        params_.depth = 5;  // data::MarketDepthRecord (include/data/records.h:20-28)
        params_.max_trades = c["engine"]["max_trades"].as<int>(4);
        params_.order_size = c["market"]["order_size"].as<int>(1);
        params_.pos_lb = c["market"]["pos_lb"].as<long>();
        params_.pos_ub = c["market"]["pos_ub"].as<long>();
        static const std::map<std::string, int> r2i = {
            {"none", LOB_REWARD_NONE}, {"pnl", LOB_REWARD_PNL}, {"pnl_damped", LOB_REWARD_PNL_DAMPED},
            {"lovol", LOB_REWARD_LOVOL}, {"mm_linear", LOB_REWARD_MM_LINEAR}, {"mm_exp", LOB_REWARD_MM_EXP}, {"mm_div", LOB_REWARD_MM_DIV},
            {"spread", LOB_REWARD_SPREAD}};  // (spread: Base's spread_window is given the engine's mean after every step, mirror())
        const std::string rm = c["reward"]["measure"].as<std::string>("pnl");
        if (!r2i.count(rm) && !device_learning_)
            throw std::runtime_error("GpuIntraday: reward measure " + rm + " reads Base's windows through the non-virtual getReward(); "
                                     "make Base::getReward virtual (INTEGRATION.md) or use the C ABI's lob_get_reward");
        if (r2i.count(rm)) params_.reward_measure = r2i.at(rm);
        else if (rm == "normed") params_.reward_measure = LOB_REWARD_NORMED;  // (device learning: the engine's own getReward)
        else throw std::runtime_error("Unknown reward measure: " + rm);
        params_.pos_weight = c["reward"]["pos_weight"].as<float>(0.0);
        params_.trd_weight = c["reward"]["trd_weight"].as<float>(0.0);
        params_.pnl_weight = c["reward"]["pnl_weight"].as<float>(1.0);
        params_.damping_factor = c["reward"]["damping_factor"].as<float>(1.0);
        auto lb = [&](int x) { return x > 1 ? x : 1; };
        params_.lb_vwap = lb(c["state"]["lookback"]["vwap"].as<int>(0));
        params_.lb_mpm = lb(c["state"]["lookback"]["mpm"].as<int>(0));
        params_.lb_vlt = lb(c["state"]["lookback"]["vlt"].as<int>(0));
        params_.lb_svl = lb(c["state"]["lookback"]["svl"].as<int>(0));
        params_.lb_rsi = lb(c["state"]["lookback"]["rsi"].as<int>(0));
        params_.lb_spread = lb(c["policy"]["spread_lookback"].as<int>(10));
        params_.lb_pnl = lb(c["reward"]["pnl_lookback"].as<int>(0));
        const std::string tp = c["market"]["target_price"]["type"].as<std::string>("midprice");
        params_.target_price = (tp != "midprice") ? LOB_TP_MIDPRICE : LOB_TP_MICROPRICE;  // the factory's inverted tests, base.cpp:101-112
        params_.quote_mode = (tp == "book") ? LOB_QUOTE_BOOK : LOB_QUOTE_TARGET;          // intraday.cpp:64
        params_.lb_target = c["market"]["target_price"]["lookback"].as<int>(1);
        params_.memory_size = 1;  // the weights stay with the reference's rl::Agent on the host
        params_.theta_mode = LOB_THETA_PRIVATE;
        if (device_learning_) {
            // the reads of rl::Agent / the agents' constructors (src/rl/agent.cpp:21-64, src/main.cpp:140-188)
            params_.memory_size = c["learning"]["memory_size"].as<long>();
            params_.n_tilings = c["learning"]["n_tilings"].as<int>();
            params_.n_actions = c["learning"]["n_actions"].as<int>();
            params_.theta_mode = LOB_THETA_SHARED;
            if (c["learning"]["group_weights"]) {
                auto gw = c["learning"]["group_weights"].as<std::vector<double>>();
                params_.group_weights[0] = gw.at(0);
                params_.group_weights[1] = gw.at(1);
                params_.group_weights[2] = gw.size() > 2 ? gw[2] : 1.0 - (gw[0] + gw[1]);
            } else params_.group_weights[0] = params_.group_weights[1] = params_.group_weights[2] = 1.0 / 3;
            params_.gamma = c["learning"]["gamma"].as<double>();
            params_.lambda = c["learning"]["lambda"].as<double>();
            params_.alpha = c["learning"]["alpha_start"].as<double>(0.2);
            const std::string pt = c["policy"]["type"].as<std::string>("");
            if (pt == "greedy") params_.epsilon = 0.0;
            else if (pt == "epsilon_greedy") params_.epsilon = (double)c["policy"]["eps_init"].as<float>(0.0);
            else if (pt == "random") params_.epsilon = 1.0;
            else if (pt == "boltzmann") { params_.policy = LOB_POLICY_BOLTZMANN; params_.tau = (double)c["policy"]["tau_init"].as<float>(); }
            else throw std::runtime_error("Please specify a valid policy!");
            const std::string algo = c["learning"]["algorithm"].as<std::string>("sarsa");
            if (algo == "sarsa") params_.algo = LOB_ALGO_SARSA;
            else if (algo == "q_learn") params_.algo = LOB_ALGO_QLAMBDA;
            else if (algo == "double_q_learn") params_.algo = LOB_ALGO_DOUBLE_Q;
            else if (algo == "r_learn") params_.algo = LOB_ALGO_R_LEARN;
            else if (algo == "online_r_learn") params_.algo = LOB_ALGO_ONLINE_R_LEARN;
            else if (algo == "double_r_learn") params_.algo = LOB_ALGO_DOUBLE_R_LEARN;
            else throw std::invalid_argument("Unknown learning algorithm: " + algo);
            if (params_.algo >= LOB_ALGO_R_LEARN) params_.beta = c["learning"]["beta"].as<double>();
            params_.seed = (uint64_t)c["debug"]["random_seed"].as<unsigned>(1994);
        }
    }
    ~GpuIntraday() { lob_destroy(engine_); }
    GpuIntraday(const GpuIntraday&) = delete;
    GpuIntraday& operator=(const GpuIntraday&) = delete;

    // Intraday::LoadData (intraday.cpp:141-150): the CSV pair -> event records -> HBM
    void LoadData(std::string ticker, std::string md_path, std::string tas_path) {
        check(lob_market_preset(ticker.c_str(), &params_.market), "Market::make_market");
        if (engine_ == nullptr) check(lob_create(&params_, n_books_, device_, &engine_), "GpuIntraday");
        uint32_t* rec = nullptr;
        int32_t n = 0;
        check(lob_convert_csv(md_path.c_str(), tas_path.c_str(), params_.max_trades, &rec, &n), "LoadData");
        int rc;
        if (n_books_ == 1) rc = lob_load_events(engine_, rec, n);
        else {  // every book replays the loaded day (one copy in HBM)
            std::vector<int64_t> phase((size_t)n_books_, 0);
            rc = lob_load_events_shared(engine_, rec, n, phase.data(), n);
        }
        lob_free(rec);
        check(rc, "LoadData");
        have_data_ = true;
    }
    // B synthetic streams generated in HBM (the bench's workload) instead of a CSV pair
    void LoadSynthetic(std::string ticker, const lob_gen_params& g) {
        check(lob_market_preset(ticker.c_str(), &params_.market), "Market::make_market");
        if (engine_ == nullptr) check(lob_create(&params_, n_books_, device_, &engine_), "GpuIntraday");
        check(lob_gen_events_device(engine_, &g), "LoadData");
        have_data_ = true;
    }
    // the policy RNG streams of the books: draw k of book b = lob_rng(seed, first_stream + b, k) (include/lob_engine.h)
    void set_rng(uint64_t seed, uint64_t first_stream) { params_.seed = seed; params_.book_id_offset = first_stream; }
    int n_books() const { return n_books_; }
    bool device_learning() const { return device_learning_; }
    const lob_params& params() const { return params_; }
    void refresh() { mirror(); }   // book 0's numbers into Base's members (GpuLearner calls it at the end of an episode)

    bool Initialise() override {  // Intraday::Initialise (intraday.cpp:103-138); false: ran out of data before the windows filled
        if (!have_data_) return false;
        sync_inventory();
        Base::Initialise();
        check(lob_reset(engine_), "Initialise");
        mirror();
        init_date_++;
        return last_.terminal != 2;
    }
    bool performAction(int action) override {  // false: the depth stream is exhausted (base.cpp:254-337)
        const int32_t a = action;
        sync_inventory();
        check(lob_step(engine_, &a), "performAction");
        mirror();
        return last_.terminal != 2;
    }
    void getState(std::vector<float>& out) override {  // APPENDS the state variables (intraday.cpp:411-416)
        float v[LOB_MAX_VARS];
        check(lob_get_state(engine_, v), "getState");
        out.insert(out.end(), v, v + params_.n_vars);
    }
    bool isTerminal() override { return last_.terminal != 0; }
    std::string getEpisodeId() override { return std::to_string(init_date_); }
    void ClearInventory() {  // hides Base::ClearInventory (non-virtual, base.cpp:339-349) for callers that hold a GpuIntraday
        sync_inventory();
        check(lob_clear_inventory(engine_), "ClearInventory");
        mirror();
    }
    const lob_book_dump& book() const { return last_; }
    lob_engine* handle() { return engine_; }
};

}  // namespace environment
#endif
