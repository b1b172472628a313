// Host-side C++ mirror of the reference's interfaces for the hot path, on top
// of the C ABI (include/lob_engine.h).  Same names, argument meaning and error
// behaviour as the reference classes so that its call sites read unchanged:
//
//   lob::Config            <- Config (include/utilities/config.h:7-12): the YAML keys the
//                             path reads (config/example.yaml), parsed by a small subset
//                             reader (block maps, flow sequences, scalars, comments)
//   lob::BatchedIntraday   <- environment::Base / Intraday<> (include/environment/base.h:117-151,
//                             include/environment/intraday.h:62-75): Initialise(),
//                             performAction(), getState(), getReward(), isTerminal(),
//                             ClearInventory(), getEpisodeReward()/getEpisodePnL()...
//                             for ALL books at once, with a per-book view
//   lob::Agent             <- rl::Agent (include/rl/agent.h:48-77): HandleTerminal(episode)
//                             (alpha / epsilon schedules, src/rl/agent.cpp:103-109,
//                             src/rl/policy.cpp:79-82), GoGreedy(), write_theta()
//   lob::Learner / lob::Backtester
//                          <- experiment::serial::Learner / Backtester
//                             (include/experiment/serial.h:38-58): RunEpisode(), _step()
//
// Errors: the reference throws std::runtime_error / std::invalid_argument and
// returns false for "out of data"; so do these classes (every non-OK ABI status
// becomes a std::runtime_error carrying lob_last_error()).
#ifndef LOB_HOST_HPP
#define LOB_HOST_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lob_comm.h"
#include "../../include/lob_engine.h"

namespace lob {

inline void check(int rc, const char* what) {
    if (rc != LOB_OK) throw std::runtime_error(std::string(what) + ": " + lob_last_error());
}

// ---------------------------------------------------------------------------
// YAML-subset configuration (same key paths as the reference's Config).
class Config {
    std::map<std::string, std::string> kv_;                // "a.b.c" -> scalar
    std::map<std::string, std::vector<std::string>> seq_;  // "a.b" -> flow sequence

    static std::string strip(const std::string& s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static std::string unq(const std::string& s) {
        if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) return s.substr(1, s.size() - 2);
        return s;
    }

public:
    Config() {}
    explicit Config(const std::string& path) {
        std::ifstream f(path);
        if (!f.is_open()) throw std::runtime_error("[Config] cannot open " + path);
        parse(f);
    }
    static Config FromString(const std::string& text) {
        Config c;
        std::istringstream is(text);
        c.parse(is);
        return c;
    }
    void parse(std::istream& in) {
        std::vector<std::pair<int, std::string>> stack;
        std::string line;
        while (std::getline(in, line)) {
            size_t h = line.find(" #");
            if (h != std::string::npos) line = line.substr(0, h);
            if (!line.empty() && line[0] == '#') continue;
            if (strip(line).empty()) continue;
            int indent = 0;
            while ((size_t)indent < line.size() && line[indent] == ' ') indent++;
            std::string body = strip(line);
            size_t colon = body.find(':');
            if (colon == std::string::npos) throw std::runtime_error("[Config] unsupported line: " + line);
            std::string key = unq(strip(body.substr(0, colon))), val = strip(body.substr(colon + 1));
            while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
            std::string path;
            for (auto& s : stack) path += s.second + ".";
            path += key;
            if (val.empty()) stack.push_back({indent, key});
            else if (val.front() == '[' && val.back() == ']') {
                std::vector<std::string> items;
                std::string cur;
                for (char ch : val.substr(1, val.size() - 2)) {
                    if (ch == ',') { if (!strip(cur).empty()) items.push_back(unq(strip(cur))); cur.clear(); }
                    else cur.push_back(ch);
                }
                if (!strip(cur).empty()) items.push_back(unq(strip(cur)));
                seq_[path] = items;
            } else kv_[path] = unq(val);
        }
    }
    bool has(const std::string& k) const { return kv_.count(k) || seq_.count(k); }
    void set(const std::string& k, const std::string& v) { kv_[k] = v; }
    std::string str(const std::string& k) const {
        auto it = kv_.find(k);
        if (it == kv_.end()) throw std::runtime_error("[Config] missing key " + k);
        return it->second;
    }
    std::string str(const std::string& k, const std::string& d) const { return kv_.count(k) ? kv_.at(k) : d; }
    double num(const std::string& k) const { return std::stod(str(k)); }
    double num(const std::string& k, double d) const { return kv_.count(k) ? std::stod(kv_.at(k)) : d; }
    long integer(const std::string& k) const { return std::stol(str(k)); }
    long integer(const std::string& k, long d) const { return kv_.count(k) ? std::stol(kv_.at(k)) : d; }
    // YAML 1.1 booleans as yaml-cpp's as<bool>() takes them (y / yes / true / on, n / no / false / off, any case)
    bool boolean(const std::string& k, bool d) const {
        if (!kv_.count(k)) return d;
        std::string v = kv_.at(k);
        for (auto& ch : v) ch = (char)tolower((unsigned char)ch);
        if (v == "y" || v == "yes" || v == "true" || v == "on") return true;
        if (v == "n" || v == "no" || v == "false" || v == "off") return false;
        throw std::runtime_error("[Config] not a boolean: " + k + ": " + kv_.at(k));
    }
    const std::vector<std::string>& list(const std::string& k) const {
        auto it = seq_.find(k);
        if (it == seq_.end()) throw std::runtime_error("[Config] missing sequence " + k);
        return it->second;
    }

    // The parameter reads of Base/Intraday/Agent/Policy constructors
    // (src/environment/base.cpp:14-115, src/environment/intraday.cpp:37-82,
    //  src/rl/agent.cpp:13-60, src/main.cpp:140-189) folded into lob_params.
    lob_params to_params(const std::string& ticker, int depth = 5, int max_trades = 2) const {
        lob_params p;
        lob_default_params(&p);
        p.depth = depth;
        p.max_trades = max_trades;
        check(lob_market_preset(ticker.c_str(), &p.market), "Market::make_market");
        static const std::map<std::string, int> v2i = {
            {"pos", LOB_VAR_POS}, {"spd", LOB_VAR_SPD}, {"mpm", LOB_VAR_MPM}, {"imb", LOB_VAR_IMB},
            {"svl", LOB_VAR_SVL}, {"vol", LOB_VAR_VOL}, {"rsi", LOB_VAR_RSI}, {"vwap", LOB_VAR_VWAP},
            {"a_dist", LOB_VAR_A_DIST}, {"a_queue", LOB_VAR_A_QUEUE}, {"b_dist", LOB_VAR_B_DIST},
            {"b_queue", LOB_VAR_B_QUEUE}, {"last_action", LOB_VAR_LAST_ACTION}};
        const auto& vars = list("state.variables");
        if (vars.size() > LOB_MAX_VARS) throw std::runtime_error("too many state variables");
        p.n_vars = (int)vars.size();
        for (size_t i = 0; i < vars.size(); i++) {
            auto it = v2i.find(vars[i]);
            if (it == v2i.end()) throw std::out_of_range("Unknown state variable: " + vars[i]);  // intraday.cpp:50-58
            p.vars[i] = it->second;
        }
        p.order_size = (int)integer("market.order_size", 1);
        p.pos_lb = integer("market.pos_lb");
        p.pos_ub = integer("market.pos_ub");
        static const std::map<std::string, int> r2i = {
            {"none", LOB_REWARD_NONE}, {"pnl", LOB_REWARD_PNL}, {"pnl_damped", LOB_REWARD_PNL_DAMPED},
            {"spread", LOB_REWARD_SPREAD}, {"normed", LOB_REWARD_NORMED}, {"lovol", LOB_REWARD_LOVOL},
            {"mm_linear", LOB_REWARD_MM_LINEAR}, {"mm_exp", LOB_REWARD_MM_EXP}, {"mm_div", LOB_REWARD_MM_DIV}};
        std::string rm = str("reward.measure", "pnl");
        if (!r2i.count(rm)) throw std::runtime_error("Unknown reward measure: " + rm);  // base.cpp:75
        p.reward_measure = r2i.at(rm);
        p.pos_weight = (float)num("reward.pos_weight", 0.0);
        p.trd_weight = (float)num("reward.trd_weight", 0.0);
        p.pnl_weight = (float)num("reward.pnl_weight", 1.0);
        p.damping_factor = (float)num("reward.damping_factor", 1.0);
        auto lb = [&](const char* k, long d) { return (int)std::max(integer(k, d), 1L); };
        p.lb_vwap = lb("state.lookback.vwap", 0);
        p.lb_mpm = lb("state.lookback.mpm", 0);
        p.lb_vlt = lb("state.lookback.vlt", 0);
        p.lb_svl = lb("state.lookback.svl", 0);
        p.lb_rsi = lb("state.lookback.rsi", 0);
        p.lb_spread = lb("policy.spread_lookback", 10);
        p.lb_pnl = lb("reward.pnl_lookback", 0);
        // Latency::make (src/market/latency.cpp:9-22) accepts fixed / normal / lognormal and throws on anything
        // else; the sample only feeds a variable nothing reads (base.cpp:258), so every accepted type is the same here
        std::string lt = str("market.latency.type", "fixed");
        if (lt != "fixed" && lt != "normal" && lt != "lognormal") throw std::invalid_argument("Unknown latency type: " + lt);
        // quirk Q5 (base.cpp:101-112): "midprice" builds MicroPrice, anything else MidPrice
        std::string tp = str("market.target_price.type", "midprice");
        p.target_price = (tp != "midprice") ? LOB_TP_MIDPRICE : LOB_TP_MICROPRICE;
        p.quote_mode = (tp == "book") ? LOB_QUOTE_BOOK : LOB_QUOTE_TARGET;  // intraday.cpp:64
        p.lb_target = (int)integer("market.target_price.lookback", 1);
        p.memory_size = integer("learning.memory_size");
        p.n_tilings = (int)integer("learning.n_tilings");
        p.n_actions = (int)integer("learning.n_actions");
        if (has("learning.group_weights")) {  // agent.cpp:42-50
            const auto& gw = list("learning.group_weights");
            p.group_weights[0] = std::stod(gw.at(0));
            p.group_weights[1] = std::stod(gw.at(1));
            p.group_weights[2] = gw.size() > 2 ? std::stod(gw[2]) : 1.0 - (p.group_weights[0] + p.group_weights[1]);
        } else p.group_weights[0] = p.group_weights[1] = p.group_weights[2] = 1.0 / 3;
        p.gamma = num("learning.gamma");
        p.lambda = num("learning.lambda");
        p.alpha = num("learning.alpha_start", 0.2);
        // policy factory of src/main.cpp:140-165.  eps_init / eps_floor are read as float there
        // (policy.cpp:58-67): 0.8 is 0.800000011920929 in every comparison and in the schedule.
        std::string pt = str("policy.type", "");
        if (pt == "greedy") p.epsilon = 0.0;
        else if (pt == "epsilon_greedy") p.epsilon = (double)(float)num("policy.eps_init", 0.0);
        else if (pt == "random") p.epsilon = 1.0;   // RandomPolicy::Sample = the uniform branch of EpsilonGreedy, always taken
        else if (pt == "boltzmann") { p.policy = LOB_POLICY_BOLTZMANN; p.tau = (double)(float)num("policy.tau_init"); }  // float in main.cpp:156-157
        else throw std::runtime_error("Please specify a valid policy!");  // main.cpp:164-165
        std::string algo = str("learning.algorithm", "sarsa");
        if (algo == "sarsa") p.algo = LOB_ALGO_SARSA;
        else if (algo == "q_learn") p.algo = LOB_ALGO_QLAMBDA;
        else if (algo == "double_q_learn") p.algo = LOB_ALGO_DOUBLE_Q;
        else if (algo == "r_learn") p.algo = LOB_ALGO_R_LEARN;                // src/main.cpp:179-183
        else if (algo == "online_r_learn") p.algo = LOB_ALGO_ONLINE_R_LEARN;
        else if (algo == "double_r_learn") p.algo = LOB_ALGO_DOUBLE_R_LEARN;
        else throw std::invalid_argument("Unknown learning algorithm: " + algo);  // as src/main.cpp:187-188
        if (p.algo >= LOB_ALGO_R_LEARN) p.beta = num("learning.beta");  // (required, as c["learning"]["beta"].as<double>())
        p.seed = (uint64_t)integer("debug.random_seed", 1994);
        p.random_init = boolean("learning.random_init", false) ? 1 : 0;  // agent.cpp:37-39: theta (and theta_b) = 2u - 1 from the agent's generator
        return p;
    }
};

// ---------------------------------------------------------------------------
// environment::Base-shaped handle over all books of one engine.
class BatchedIntraday {
    lob_engine* e_ = nullptr;
    lob_params p_;
    int B_;
    std::vector<float> state_cache_;
    std::vector<double> reward_cache_;
    std::vector<uint8_t> term_cache_;
    bool cache_ok_ = false;

    void refresh() {
        if (cache_ok_) return;
        state_cache_.resize((size_t)B_ * p_.n_vars);
        reward_cache_.resize(B_);
        term_cache_.resize(B_);
        check(lob_get_state(e_, state_cache_.data()), "getState");
        check(lob_get_reward(e_, reward_cache_.data()), "getReward");
        check(lob_get_terminal(e_, term_cache_.data()), "isTerminal");
        cache_ok_ = true;
    }

public:
    BatchedIntraday(const lob_params& p, int n_books, int device = 0) : p_(p), B_(n_books) {
        check(lob_create(&p, n_books, device, &e_), "Intraday::Intraday");
    }
    ~BatchedIntraday() { lob_destroy(e_); }
    BatchedIntraday(const BatchedIntraday&) = delete;
    BatchedIntraday& operator=(const BatchedIntraday&) = delete;

    lob_engine* handle() { return e_; }
    const lob_params& params() const { return p_; }
    int n_books() const { return B_; }
    void invalidate() { cache_ok_ = false; }

    // LoadData(symbol, md_path, tas_path) (intraday.cpp:141-150): records instead of CSV paths
    void LoadData(const uint32_t* records, int n_events) { check(lob_load_events(e_, records, n_events), "LoadData"); invalidate(); }
    // one recorded day replayed by every book, book b from record phase[b] on (BASELINE config 5)
    void LoadReplay(const uint32_t* records, int64_t n_total, const std::vector<int64_t>& phase, int n_events) {
        if ((int)phase.size() != B_) throw std::invalid_argument("LoadReplay: one phase per book");
        check(lob_load_events_shared(e_, records, n_total, phase.data(), n_events), "LoadData");
        invalidate();
    }
    void LoadSynthetic(const lob_gen_params& g) { check(lob_gen_events_device(e_, &g), "LoadData"); invalidate(); }
    // the NEXT episode's day, handed over while this episode runs (src/main.cpp:53-55: rs.sample() + env.LoadData before every
    // episode): returns at once, the Initialise() that follows adopts the stream; `records` must stay valid until then
    void StageData(const uint32_t* records, int n_events) { check(lob_stage_events(e_, records, n_events), "StageData"); }
    void StageWait() { check(lob_stage_wait(e_), "StageData"); }

    bool Initialise() {  // false = no data for at least one book (base.h:122)
        check(lob_reset(e_), "Initialise");
        invalidate();
        refresh();
        for (int b = 0; b < B_; b++) if (term_cache_[b] == 2) return false;
        return true;
    }
    // performAction for every book; returns false when any book ran out of data (base.h:132)
    bool performAction(const std::vector<int32_t>& actions) {
        if ((int)actions.size() != B_) throw std::invalid_argument("performAction: one action per book");
        check(lob_step(e_, actions.data()), "performAction");
        invalidate();
        refresh();
        for (int b = 0; b < B_; b++) if (term_cache_[b] == 2) return false;
        return true;
    }
    bool performAction(int action) { return performAction(std::vector<int32_t>(B_, action)); }
    void getState(std::vector<float>& out, int book = 0) {  // APPENDS n_vars floats (base.h:125)
        refresh();
        out.insert(out.end(), state_cache_.begin() + (size_t)book * p_.n_vars, state_cache_.begin() + (size_t)(book + 1) * p_.n_vars);
    }
    double getReward(int book = 0) { refresh(); return reward_cache_[book]; }
    double getPotential() { return 0.0; }  // base.cpp:239-242
    bool isTerminal(int book = 0) { refresh(); return term_cache_[book] != 0; }
    void ClearInventory() { check(lob_clear_inventory(e_), "ClearInventory"); invalidate(); }
    lob_book_dump book(int b) { lob_book_dump d; check(lob_get_book(e_, b, &d), "book"); return d; }
    double getEpisodeReward(int b = 0) { return book(b).episode_reward; }
    double getEpisodePnL(int b = 0) { return book(b).episode_pnl; }
    double getMeanEpisodeReward(int b = 0) { lob_book_dump d = book(b); return d.episode_reward / d.total_ticks; }
    // Base::getTotalTransactions / getOrderRatio (base.cpp:463-473; "nTr" and the denominator of "Ppt" in src/main.cpp:234-236)
    int getTotalTransactions(int b = 0) {
        lob_book_dump d = book(b);
        return d.ask_transactions + d.bid_transactions + d.market_buys + d.market_sells;
    }
    float getOrderRatio(int b = 0) {
        lob_book_dump d = book(b);
        return float(d.ask_transactions + d.bid_transactions) / (d.market_buys + d.market_sells);
    }
    // Base::writeStats (base.cpp:451-456): experiment, tick and trade statistics are each written to the SAME path, truncating
    // it (statistics.cpp:12,36,70) -- what survives is TradeStatistics::write, eight lines (quirk Q17).  The reference never
    // counts placed / cancelled orders: those four stay 0 there too.
    void writeStats(const std::string& path, int b = 0) {
        lob_book_dump d = book(b);
        std::ofstream ofs(path, std::ofstream::out);
        ofs << "asks_placed," << 0 << std::endl;
        ofs << "bids_placed," << 0 << std::endl;
        ofs << "asks_cancelled," << 0 << std::endl;
        ofs << "bids_cancelled," << 0 << std::endl;
        ofs << "ask_transactions," << d.ask_transactions << std::endl;
        ofs << "bid_transactions," << d.bid_transactions << std::endl;
        ofs << "market_sells," << d.market_sells << std::endl;
        ofs << "market_buys," << d.market_buys << std::endl;
    }
    // TickStatistics::write's six occupancy figures (statistics.cpp:70-92), for callers that want what writeStats overwrites
    std::string tickStatsText(int b = 0) {
        lob_book_dump d = book(b);
        std::ostringstream os;
        const int tt = d.total_ticks;
        os << "ask_occupancy," << 100 * float(d.ticks_with_ask) / tt << "%\n" << "bid_occupancy," << 100 * float(d.ticks_with_bid) / tt << "%\n"
           << "both_occupancy," << 100 * float(d.ticks_with_both) / tt << "%\n" << "pos_occupancy," << 100 * float(d.ticks_with_position) / tt << "%\n"
           << "short_occupancy," << 100 * float(d.ticks_short) / tt << "%\n" << "long_occupancy," << 100 * float(d.ticks_long) / tt << "%\n";
        return os.str();
    }
    std::string getEpisodeId() { return "synthetic"; }
};

// ---------------------------------------------------------------------------
// rl::Agent-shaped object: theta lives on the GPU inside the engine.
class Agent {
    BatchedIntraday& env_;
    double alpha_start_, alpha_floor_, omega_;
    double eps_init_, eps_floor_, eps_T_;
    bool eps_schedule_;  // only EpsilonGreedy::HandleTerminal moves epsilon (policy.cpp:19,79-82)
    bool tau_schedule_;  // Boltzmann::HandleTerminal (policy.cpp:119-122)
    double tau_init_ = 1.0, tau_floor_ = 1.0, tau_T_ = 1.0;
    bool greedy_ = false;

public:
    Agent(BatchedIntraday& env, const Config& c)
        : env_(env), alpha_start_(c.num("learning.alpha_start", 0.2)), alpha_floor_(c.num("learning.alpha_floor", 0.001)),
          omega_(c.num("learning.omega", 1.0)), eps_init_((double)(float)c.num("policy.eps_init", 0.0)),
          eps_floor_((double)(float)c.num("policy.eps_floor", 0.0)), eps_T_(c.num("policy.eps_T", 1.0)),  // float in main.cpp:149-150
          eps_schedule_(c.str("policy.type", "") == "epsilon_greedy"), tau_schedule_(c.str("policy.type", "") == "boltzmann") {
        if (tau_schedule_) {
            tau_init_ = (double)(float)c.num("policy.tau_init");
            tau_floor_ = (double)(float)c.num("policy.tau_floor");
            tau_T_ = c.num("policy.tau_T", 1.0);
        }
    }
    void GoGreedy() { greedy_ = true; }
    bool greedy() const { return greedy_; }
    // Agent::HandleTerminal (agent.cpp:103-109) + EpsilonGreedy::HandleTerminal (policy.cpp:79-82)
    void HandleTerminal(int episode) {
        check(lob_handle_terminal(env_.handle()), "HandleTerminal");
        double alpha = std::max(alpha_floor_, alpha_start_ * std::pow(omega_, (double)episode));
        check(lob_set_alpha(env_.handle(), alpha), "HandleTerminal");
        if (eps_schedule_) {
            double eps = eps_init_ * std::pow(eps_floor_ / eps_init_, (double)episode / eps_T_);
            check(lob_set_epsilon(env_.handle(), eps), "HandleTerminal");
            epsilon_ = eps;
        }
        if (tau_schedule_) {
            double tau = tau_init_ * std::pow(tau_floor_ / tau_init_, (double)episode / tau_T_);
            check(lob_set_tau(env_.handle(), tau), "HandleTerminal");
            epsilon_ = tau;  // Policy::descr() of the training log
        }
    }
    double epsilon_ = -1.0;
    // Agent::write_theta (agent.cpp:176-181): raw double[MEMORY_SIZE]
    void write_theta(const std::string& filename) {
        std::vector<double> th((size_t)env_.params().memory_size);
        check(lob_theta_get(env_.handle(), 0, th.data(), (int64_t)th.size()), "write_theta");
        std::ofstream f(filename.c_str(), std::ios::binary);
        f.write((const char*)th.data(), (std::streamsize)(th.size() * sizeof(double)));
    }
    void load_theta(const std::string& filename) {
        std::vector<double> th((size_t)env_.params().memory_size);
        std::ifstream f(filename.c_str(), std::ios::binary);
        if (!f.read((char*)th.data(), (std::streamsize)(th.size() * sizeof(double)))) throw std::runtime_error("load_theta: short file");
        check(lob_theta_set(env_.handle(), 0, th.data(), (int64_t)th.size()), "load_theta");
    }
};

// ---------------------------------------------------------------------------
// experiment::serial::Runner family (serial.cpp:18-34, 53-70, 124-137).
class Runner {
protected:
    BatchedIntraday& environment;
    virtual bool _step(Agent* m) = 0;  // true when every book is terminal
    long n_live() {
        int64_t c[4];
        check(lob_get_counters(environment.handle(), c), "counters");
        return (long)c[2];
    }

public:
    explicit Runner(BatchedIntraday& env) : environment(env) {}
    virtual ~Runner() {}
    virtual bool RunEpisode(Agent* m) {
        if (!environment.Initialise()) return false;
        bool is_terminal;
        do { is_terminal = _step(m); } while (!is_terminal);
        environment.ClearInventory();
        return true;
    }
};

class Learner : public Runner {
    int steps_per_call_;
    unsigned long _step_counter = 0;
    int _episode_counter = 0;
    lob_comm* comm_ = nullptr;  // multi-GPU: the shared Agent* of main.cpp:196-206 becomes a periodic all-reduce
    int sync_every_ = 64;
    unsigned long since_sync_ = 0;

protected:
    bool _step(Agent*) override {
        int n = steps_per_call_;
        if (comm_ && since_sync_ + n > (unsigned long)sync_every_) n = (int)(sync_every_ - since_sync_);
        const bool sync_now = comm_ && since_sync_ + n >= (unsigned long)sync_every_;
        if (!sync_now) {
            check(lob_td_step(environment.handle(), n), "Learner::_step");
        } else {
            // the sync step carries the exchange between its two halves: no cached action-selection data is live there
            // (include/lob_engine.h lob_td_step_begin)
            if (n > 1) check(lob_td_step(environment.handle(), n - 1), "Learner::_step");
            if (!lob_td_split_supported(environment.handle())) {
                // no half steps with this engine configuration (an experiments build with two book groups): the whole step, then
                // the exchange -- correct too, the exchange then voids the cached action-selection data of one step.  Asked for
                // explicitly: a LOB_ESTATE from lob_td_step_begin has other causes and is reported as the error it is.
                check(lob_td_step(environment.handle(), 1), "Learner::_step");
                check(lob_theta_allreduce(environment.handle(), comm_), "Learner::_step (weight exchange)");
            } else {
                check(lob_td_step_begin(environment.handle()), "Learner::_step");
                check(lob_theta_allreduce(environment.handle(), comm_), "Learner::_step (weight exchange)");
                check(lob_td_step_end(environment.handle()), "Learner::_step");
            }
        }
        environment.invalidate();
        _step_counter += n;
        if (!comm_) return n_live() == 0;
        since_sync_ += n;
        if (!sync_now) return false;
        since_sync_ = 0;
        // every rank keeps stepping (a no-op for finished books) until NO rank has a live book:
        // the exchange is a collective, all ranks must leave the episode at the same sync point
        double live = (double)n_live();
        check(lob_comm_reduce_host_f64(comm_, &live, 1, LOB_COMM_SUM), "Learner::_step (live books)");
        return live == 0.0;
    }

public:
    // steps_per_call: how many env-steps of every book are enqueued per host round trip
    Learner(BatchedIntraday& env, int steps_per_call = 1) : Runner(env), steps_per_call_(steps_per_call) {}
    // one process per GPU: exchange delta-theta over `comm` every `sync_every` steps
    void set_comm(lob_comm* comm, int sync_every = 64) {
        comm_ = comm;
        sync_every_ = sync_every < 1 ? 1 : sync_every;
        since_sync_ = 0;
        if (comm_) check(lob_delta_init(environment.handle()), "Learner::set_comm");
    }
    unsigned long step_counter() const { return _step_counter; }
    bool RunEpisode(Agent* m) override {  // serial.cpp:72-93
        _step_counter = 0;
        if (Runner::RunEpisode(m)) {
            m->HandleTerminal(_episode_counter++);
            return true;
        }
        return false;
    }
};

class Backtester : public Runner {
    std::ofstream profit_log_;
    double last_bandh_ = 0.0;
    int date_ = 0;

protected:
    bool _step(Agent*) override {
        check(lob_eval_step(environment.handle(), 1), "Backtester::_step");
        environment.invalidate();
        if (profit_log_.is_open()) LogProfit();
        return n_live() == 0;
    }
    // Intraday::LogProfit (src/environment/intraday.cpp:438-451), book 0, one row per performed step
    void LogProfit() {
        int32_t stepped = 0;
        {
            std::vector<int32_t> st(environment.n_books());
            check(lob_get_stepped(environment.handle(), st.data()), "LogProfit");
            stepped = st[0];
        }
        if (!stepped) return;
        lob_book_dump d = environment.book(0);
        char buf[512];
        snprintf(buf, sizeof buf, "%d,%lld,%d,%lld,%.10g,%.10g,%.10g,%.10g,%d,%d,%.10g,%.10g\n", date_, (long long)d.time_ms,
                 d.last_action, (long long)d.position, (d.ask_px[0] + d.bid_px[0]) / 2.0, d.ask_px[0] - d.bid_px[0], d.ask_quote,
                 d.bid_quote, d.ask_level, d.bid_level, d.pnl_step, d.episode_bandh - last_bandh_);
        profit_log_ << buf;
        last_bandh_ = d.episode_bandh;
    }

public:
    explicit Backtester(BatchedIntraday& env) : Runner(env) {}
    // Backtester ctor + Base::start_logging (serial.cpp:97-122): profit_log.csv with the reference's header
    void start_logging(const std::string& path, int date = 0) {
        profit_log_.open(path.c_str());
        if (!profit_log_.is_open()) throw std::runtime_error("Loggers not registered!");  // base.cpp:402-403
        profit_log_ << "episode,step,action,position,midprice,spread,quoted_ask,quoted_bid,ask_level,bid_level,pnl_step,bandh_step\n";
        date_ = date;
    }
    void stop_logging() { profit_log_.close(); }
    bool RunEpisode(Agent* m) override {
        last_bandh_ = 0.0;
        return Runner::RunEpisode(m);
    }
};

}  // namespace lob
#endif
