// rl::GpuAgent and experiment::serial::GpuLearner -- the learner half of the reference-side binding.
//
// The reference's training loop (src/main.cpp:47-70) holds its runner through `experiment::serial::Learner`, a subclass of
// `Runner` whose RunEpisode / _step are virtual (include/experiment/serial.h:14-33), and its agent through `rl::Agent*`, whose
// action() / UpdateWeights() / UpdateTraces() are virtual (include/rl/agent.h:44-65).  Compiled against those headers:
//
//     environment::GpuIntraday env(c, device, n_books, true);    // was: environment::Intraday<> env(c);
//     env.LoadData(symbol, md_csv, tas_csv);                     // unchanged call site (every book replays the day)
//     experiment::serial::GpuLearner learner(c, env);            // was: experiment::serial::Learner learner(c, env);
//     rl::Agent* agent = new rl::GpuAgent(std::move(policy), c, env);   // was: new rl::QLearn(std::move(policy), c)
//     while (learner.RunEpisode(agent)) { ... env.getEpisodeReward() ... }   // unchanged
//     agent->write_theta(path);                                  // unchanged (the host copy is refreshed after every episode)
//
// One RunEpisode = Runner::RunEpisode's sequence (Initialise -> _step until terminal -> ClearInventory, serial.cpp:18-34) for
// ALL books of the engine at once, then Learner::RunEpisode's epilogue (HandleTerminal, serial.cpp:72-93): a `_step` is
// `steps_per_call` batched Learner::_step's (lob_td_step: action, performAction, newState, HandleTransition of every live book),
// terminal when no book is live.  The schedules stay with the reference's own objects: Agent::HandleTerminal (non-virtual)
// computes alpha and lets the Policy compute epsilon / tau; GpuLearner then hands both to the engine (lob_set_alpha /
// lob_set_epsilon / lob_set_tau).  The agent's weights live in HBM; action() -- what the reference's unmodified Backtester
// calls -- evaluates Q on the device (lob_q_values) and lets the reference's own Policy object sample.
#ifndef LOB_REF_BINDING_GPU_LEARNER_H
#define LOB_REF_BINDING_GPU_LEARNER_H

#include <atomic>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "experiment/serial.h"
#include "rl/agent.h"

#include "gpu_intraday.h"

namespace rl {

class GpuAgent : public Agent {
    environment::GpuIntraday& env_;

    static void check(int rc, const char* what) {
        if (rc != LOB_OK) throw std::runtime_error(std::string(what) + ": " + lob_last_error());
    }

protected:
    void UpdateTraces(State&, int) override {}  // traces live in HBM (lob_td_step)

public:
    GpuAgent(std::unique_ptr<Policy> policy, Config& c, environment::GpuIntraday& env) : Agent(std::move(policy), c), env_(env) {
        if (!env.device_learning()) throw std::invalid_argument("rl::GpuAgent needs an environment::GpuIntraday built for device learning");
        if (MEMORY_SIZE != env.params().memory_size) throw std::invalid_argument("rl::GpuAgent: memory_size differs from the engine's");
    }
    // Agent::action (agent.cpp:67-74): the nine Q values from the weights in HBM, the sample from the reference's Policy
    unsigned int action(State& s) override {
        std::vector<float>& v = s.toVector();
        std::vector<double> qs(N_ACTIONS, 0.0);
        check(lob_q_values(env_.handle(), v.data(), 1, qs.data()), "GpuAgent::action");
        return policy->Sample(qs);
    }
    // HandleTransition of a single transition has no meaning here: the engine learns inside lob_td_step (GpuLearner)
    double UpdateWeights(State&, int, double, State&) override {
        throw std::logic_error("rl::GpuAgent learns on the device: drive it with experiment::serial::GpuLearner");
    }
    // what Agent::HandleTerminal (non-virtual) left in the protected members
    double alpha_now() const { return alpha; }
    // the host copy Agent::write_theta (non-virtual) writes <-> the weights in HBM
    void pull_theta() { check(lob_theta_get(env_.handle(), 0, theta, MEMORY_SIZE), "GpuAgent::pull_theta"); }
    void push_theta() { check(lob_theta_set(env_.handle(), 0, theta, MEMORY_SIZE), "GpuAgent::push_theta"); }
    const double* theta_host() const { return theta; }
};

}  // namespace rl

namespace experiment {
namespace serial {

class GpuLearner : public Runner {
    environment::GpuIntraday& genv_;
    int steps_per_call_;
    unsigned long _step_counter = 0;
    int _episode_counter = 0;
    bool model_log_on_ = false;
    std::vector<double> ml_rows_ = std::vector<double>(8192);

    static void check(int rc, const char* what) {
        if (rc != LOB_OK) throw std::runtime_error(std::string(what) + ": " + lob_last_error());
    }

protected:
    // `steps_per_call` x Learner::_step of every book; true when no book is live any more
    bool _step(rl::Agent*) override {
        check(lob_td_step(genv_.handle(), steps_per_call_), "GpuLearner::_step");
        _step_counter += (unsigned long)steps_per_call_;
        // Agent::HandleTransition's `model_log` rows (agent.cpp:93-100: mean |delta| per 1000 updates): the engine aggregates on
        // the device; what it has written since the last call goes to the logger the Agent's constructor registered
        if (model_log_on_) {
            if (auto log = spdlog::get("model_log")) {
                int32_t n = 0;
                check(lob_model_log_read(genv_.handle(), ml_rows_.data(), (int32_t)ml_rows_.size(), &n, nullptr), "GpuLearner::_step (model_log)");
                for (int32_t i = 0; i < n; i++) log->info(ml_rows_[(size_t)i]);
            }
        }
        int64_t cnt[4];
        check(lob_get_counters(genv_.handle(), cnt), "GpuLearner::_step");
        return cnt[2] == 0;
    }

    // alpha as Agent::HandleTerminal left it; epsilon / tau as the Policy object reports them (Policy::descr(): eps of
    // EpsilonGreedy, tau of Boltzmann, 0 for Greedy -- epsilon 0 IS greedy; Random draws uniformly: epsilon 1)
    void push_schedules(rl::Agent* m, rl::GpuAgent* ga) {
        check(lob_set_alpha(genv_.handle(), ga->alpha_now()), "GpuLearner");
        if (genv_.params().policy == LOB_POLICY_BOLTZMANN) check(lob_set_tau(genv_.handle(), m->policy->descr()), "GpuLearner");
        else check(lob_set_epsilon(genv_.handle(), dynamic_cast<rl::Random*>(m->policy.get()) ? 1.0 : m->policy->descr()), "GpuLearner");
    }

public:
    GpuLearner(Config& c, environment::GpuIntraday& env, int steps_per_call = 8) : Runner(c, env), genv_(env), steps_per_call_(steps_per_call < 1 ? 1 : steps_per_call) {
        if (!env.device_learning()) throw std::invalid_argument("GpuLearner needs an environment::GpuIntraday built for device learning");
        // Learner::Learner (src/experiment/serial.cpp:38-51): the per-episode training log, same logger name, same header
        if (c["logging"] and c["logging"]["log_learning"].as<bool>()) {
            try {
                auto log = spdlog::rotating_logger_mt("training_log", c["output_dir"].as<std::string>() + "training_log.csv",
                                                      c["logging"]["max_size"].as<size_t>(), 1);
                log->info("episode,episode_id,reward,pnl,n_steps,epsilon");
            } catch (spdlog::spdlog_ex& e) {}
        }
    }
    unsigned long step_counter() const { return _step_counter; }

    bool RunEpisode(rl::Agent* m) override {  // Learner::RunEpisode (serial.cpp:72-93) over Runner::RunEpisode (serial.cpp:18-34)
        rl::GpuAgent* ga = dynamic_cast<rl::GpuAgent*>(m);
        if (!ga) throw std::invalid_argument("GpuLearner::RunEpisode: the agent must be an rl::GpuAgent");
        _step_counter = 0;
        if (!model_log_on_ && spdlog::get("model_log")) {   // (registered by Agent::Agent when logging.log_learning is on, agent.cpp:53-59)
            check(lob_model_log_enable(genv_.handle(), 1), "GpuLearner (model_log)");
            model_log_on_ = true;
        }
        push_schedules(m, ga);                          // the agent's and the policy's objects are the source of truth
        environment.resetStats();
        if (!environment.Initialise()) return false;   // lob_reset of every book
        bool is_terminal;
        do { is_terminal = _step(m); } while (!is_terminal);
        check(lob_clear_inventory(genv_.handle()), "ClearInventory");
        check(lob_handle_terminal(genv_.handle()), "HandleTerminal");   // traces.decay(0.0) of every book
        m->HandleTerminal(_episode_counter++);                          // the reference's schedules: alpha, and the policy's epsilon / tau
        push_schedules(m, ga);
        genv_.refresh();     // book 0's episode totals / statistics into Base's members (getEpisodeReward() ... getTotalTransactions())
        // the row Learner::RunEpisode writes (serial.cpp:81-88), book 0's episode; no row without the logger (the reference
        // dereferences a null pointer there when logging.log_learning is off)
        if (auto log = spdlog::get("training_log"))
            log->info("{},{},{},{},{},{}", _episode_counter, environment.getEpisodeId(), environment.getEpisodeReward(), environment.getEpisodePnL(),
                      _step_counter, m->policy->descr());
        ga->pull_theta();    // write_theta() is not virtual: keep the host copy current
        return true;
    }
};

}  // namespace serial
}  // namespace experiment
#endif
