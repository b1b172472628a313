// Thin driver with the reference's factory choices (src/main.cpp:82-245) for the
// hot path: read a YAML config, build the batched environment + learner, train
// n episodes on synthetic streams, evaluate greedily, print the per-episode
// rows of the reference's training_log (serial.cpp:81-88) for book 0.
//
//   lob_run -c config/engine.yaml [-n books] [-e episodes (default: training.n_episodes)] [-a sarsa|q_learn|double_q_learn|r_learn|online_r_learn|double_r_learn] [--events N] [--depth D]
//           [--theta out.bin] [--profit-log profit_log.csv]
//           [--gpus N [--sync-every K]]   one process per GPU (forked here), -n books EACH, book ids rank * n ..,
//            delta-theta all-reduced over RCCL/xGMI every K steps (include/lob_comm.h): the stand-in for the
//            reference's N training threads on one shared Agent (src/main.cpp:196-206)
//           [--md depth.csv --tas trades.csv | --lobster orderbook.csv message.csv LEVELS]   (a recorded day, replayed
//            by every book from evenly spread starting records; default: synthetic streams)
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lob_host.hpp"

static int run(int argc, char** argv, int rank, int world, const std::string& rdzv);

int main(int argc, char** argv) {
    int gpus = 1;
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], "--gpus")) gpus = atoi(argv[i + 1]);
    // rendezvous token of the RCCL communicator: a file in a private (0700) directory of this run
    char rdir[] = "/tmp/lob_run_XXXXXX";
    if (!mkdtemp(rdir)) { perror("mkdtemp"); return 2; }
    char rdzv[128];
    snprintf(rdzv, sizeof rdzv, "%s/rdzv", rdir);
    // LOB_FORCE_DIST=1: the whole exchange path with a one-rank communicator (a 1-GPU box can run it)
    const char* fd = getenv("LOB_FORCE_DIST");
    if (gpus <= 1) {
        const int rc1 = run(argc, argv, 0, 1, (fd && fd[0] == '1') ? rdzv : "");
        unlink(rdzv);
        rmdir(rdir);
        return rc1;
    }
    // one process per GPU, forked before anything touches the HIP runtime
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; r++) {
        pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 2; }
        if (pid == 0) _exit(run(argc, argv, r, gpus, rdzv));
        kids.push_back(pid);
    }
    // wait for whichever rank ends first: when one fails (no device, no data, an exception) the others would sit in
    // ncclCommInitRank or the next all-reduce for ever -- stop exactly the processes forked above
    int rc = 0;
    size_t left = kids.size();
    while (left > 0) {
        int st = 0;
        const pid_t k = waitpid(-1, &st, 0);
        if (k < 0) break;
        bool ours = false;
        for (pid_t& q : kids) if (q == k) { q = -1; ours = true; }
        if (!ours) continue;
        left--;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 2;
        if (code && !rc) {
            rc = code;
            fprintf(stderr, "[lob_run] a rank exited with %d: stopping the other ranks\n", code);
            for (pid_t q : kids) if (q > 0) kill(q, SIGTERM);
        }
    }
    unlink(rdzv);
    rmdir(rdir);
    return rc;
}

static int run(int argc, char** argv, int rank, int world, const std::string& rdzv) {
    std::string cfg_path, algo, theta_out, profit_log, stats_out, md, tas, lob_ob, lob_msg;
    int lob_levels = 0;
    int books = 1, episodes = -1, events = 2112, depth = 5, sync_every = 64;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return std::string(argv[++i]); };
        if (a == "-c" || a == "--config") cfg_path = next();
        else if (a == "-n") books = atoi(next().c_str());
        else if (a == "-e") episodes = atoi(next().c_str());
        else if (a == "-a" || a == "--algorithm") algo = next();
        else if (a == "--events") events = atoi(next().c_str());
        else if (a == "--depth") depth = atoi(next().c_str());
        else if (a == "--theta") theta_out = next();
        else if (a == "--profit-log") profit_log = next();
        else if (a == "--stats-out") stats_out = next();   // env.writeStats(output_dir + "test_stats.csv"), src/main.cpp:242
        else if (a == "--md") md = next();
        else if (a == "--tas") tas = next();
        else if (a == "--lobster") { lob_ob = next(); lob_msg = next(); lob_levels = atoi(next().c_str()); }
        else if (a == "--gpus") next();
        else if (a == "--sync-every") sync_every = atoi(next().c_str());
        else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 1; }
    }
    try {
        if (cfg_path.empty()) throw std::runtime_error("A configuration file must be provided (-c)");  // main.cpp:300-304
        lob::Config c(cfg_path);
        if (!algo.empty()) c.set("learning.algorithm", algo);  // CLI override, main.cpp:342-347
        std::string ticker = c.has("data.symbols") ? c.list("data.symbols").at(0) : "HSBA.L";
        lob_params p = c.to_params(ticker, depth, 2);
        // training.n_episodes (src/main.cpp:91 reads it into n_train_episodes -- a required key there too --, train() runs until that
        // many episodes are done, main.cpp:53-77); -e overrides it (the reference has no such flag: its tests edit the yaml)
        if (episodes < 0) episodes = (int)c.integer("training.n_episodes");
        p.book_id_offset = (uint64_t)rank * (uint64_t)books;  // global book ids: streams and RNG draws do not depend on the sharding
        lob::BatchedIntraday env(p, books, rank);
        lob_comm* comm = nullptr;
        if (!rdzv.empty()) lob::check(lob_comm_create_file(rdzv.c_str(), rank, world, rank, 300, &comm), "lob_comm_create_file");
        if (!md.empty() || !lob_ob.empty()) {
            // the reference's data files (Intraday::LoadData reads the CSV pair, intraday.cpp:141-150)
            uint32_t* rec = nullptr;
            int32_t n = 0;
            if (!md.empty()) lob::check(lob_convert_csv(md.c_str(), tas.c_str(), p.max_trades, &rec, &n), "LoadData");
            else lob::check(lob_convert_lobster(lob_ob.c_str(), lob_msg.c_str(), lob_levels, depth, p.max_trades, &rec, &n), "LoadData");
            const int window = events < n ? events : n;
            std::vector<int64_t> phase(books);
            for (int b = 0; b < books; b++) phase[b] = books > 1 ? (int64_t)(n - window) * b / (books - 1) : 0;
            env.LoadReplay(rec, n, phase, window);
            lob_free(rec);
        } else {
            lob_gen_params g;
            lob_default_gen_params(&g);
            g.n_events = events;
            g.seed = p.seed;
            env.LoadSynthetic(g);
        }
        lob::Agent agent(env, c);
        lob::Learner learner(env, 8);
        if (comm) learner.set_comm(comm, sync_every);
        if (rank == 0) printf("episode,episode_id,reward,pnl,n_steps,epsilon\n");
        for (int ep = 0; ep < episodes; ep++) {
            auto t0 = std::chrono::steady_clock::now();
            if (!learner.RunEpisode(&agent)) { fprintf(stderr, "[!] no data\n"); return 2; }
            double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            int64_t cnt[4];
            lob::check(lob_get_counters(env.handle(), cnt), "counters");
            if (rank == 0) printf("%d,%s,%.10g,%.10g,%d,%.6g\n", ep + 1, env.getEpisodeId().c_str(), env.getEpisodeReward(0), env.getEpisodePnL(0),
                   env.book(0).total_ticks, agent.epsilon_);
            fprintf(stderr, "[rank %d/%d] episode %d: %lld env-steps over %d books in %.3f s\n", rank, world, ep + 1, (long long)cnt[0], books, sec);
        }
        if (!theta_out.empty() && rank == 0) agent.write_theta(theta_out);  // replicas agree after the last exchange
        if (comm) { lob_comm_barrier(comm); lob_comm_destroy(comm); comm = nullptr; }
        if (!profit_log.empty() && rank == 0) {
            // src/main.cpp:217-239: GoGreedy() then one Backtester episode with profit logging (book 0)
            agent.GoGreedy();
            lob::Backtester bt(env);
            bt.start_logging(profit_log, 20200102);
            if (!bt.RunEpisode(&agent)) { fprintf(stderr, "[!] no data\n"); return 2; }
            bt.stop_logging();
            printf("backtest,%.10g,%.10g,%d\n", env.getEpisodeReward(0), env.getEpisodePnL(0), env.book(0).total_ticks);
        }
        if (!stats_out.empty() && rank == 0) {
            // src/main.cpp:234-242: nTr / Ppt on the console, then writeStats (book 0; quirk Q17: the trade statistics survive)
            printf("stats,%d,%.10g\n", env.getTotalTransactions(0), env.getEpisodePnL(0) / env.getTotalTransactions(0));
            env.writeStats(stats_out, 0);
        }
    } catch (std::exception& e) {
        fprintf(stderr, "Unhandled Exception: %s\n", e.what());  // main.cpp:364-368
        return 2;
    }
    return 0;
}
