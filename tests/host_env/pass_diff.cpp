// TEST INFRASTRUCTURE ONLY.  Differential test, on the CPU, of the two event passes of env_kernel:
//   general:  step_event<2> -> next_state<2>            (rl_markets_amd/csrc/lob_env.h)
//   fast:     pass_fast (select form, trades / touch from the track entry)
// compiled from the engine's own device header through tests/host_env/shim.  Random order / inventory states are
// thrown at random rows and trade lists (prices on the tick grid around the touch so that keys collide, volumes
// that grow and shrink, queue positions incl. the negative q_tail of quirk Q2, zero totals -> the x86 cvttsd2si
// corner); after every pass the complete EnvR and StepAgg of both must be bit-identical.
//   g++ -std=c++17 -O1 -ffp-contract=off -Itests/host_env/shim -o pass_diff tests/host_env/pass_diff.cpp && ./pass_diff [cases]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "../../rl_markets_amd/csrc/lob_state.h"
#include "../../rl_markets_amd/csrc/lob_stream.h"
struct TickLds {
    int n;
    f64 lb[LOB_MAX_BANDS];
    f64 tick[LOB_MAX_BANDS];
    i64 cum[LOB_MAX_BANDS];
    f64 pp[LOB_MAX_BANDS];
    i32 pt[LOB_MAX_BANDS];
};
void lob_set_error(const std::string&) {}
#include "../../rl_markets_amd/csrc/lob_env.h"

static std::mt19937_64 rng(12345);
static int ri(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); }
static double ru() { return (double)(rng() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 200000;
    long n_pass = 0, n_fill = 0, n_cancel = 0, n_adverse = 0, n_multi = 0;
    for (long cs = 0; cs < cases; cs++) {
        DevParams P;
        memset(&P, 0, sizeof P);
        P.D = ri(1, 10); P.T = ri(1, 2); P.Wd = drec_words(P.D, P.T); P.W = lob_rec_words(P.D, P.T);
        P.order_size = ri(1, 3) == 1 ? 1 : ri(1, 100);
        P.pos_ub = ri(1, 5) * P.order_size; P.pos_lb = -ri(1, 5) * P.order_size;
        const int rewards[] = {LOB_REWARD_PNL, LOB_REWARD_PNL_DAMPED, LOB_REWARD_LOVOL, LOB_REWARD_MM_LINEAR, LOB_REWARD_MM_DIV, LOB_REWARD_MM_EXP, LOB_REWARD_SPREAD};
        P.reward_measure = rewards[ri(0, 6)];
        P.damping_factor = (float)ru(); P.pos_weight = (float)(2 * ru()); P.pnl_weight = (float)(2 * ru());
        P.open_ms = 8 * 3600000LL; P.close_ms = 16 * 3600000LL + 1800000LL;
        // a little stream of rows on a 0.1 / 0.0001 / 0.5 grid
        const int n_rows = 12;
        const double tick = ri(0, 2) == 0 ? 0.1 : (ri(0, 1) ? 0.0001 : 0.5);
        const int base = ri(200, 90000);
        std::vector<uint32_t> rows((size_t)n_rows * P.Wd, 0);
        int bid_tick = base;
        for (int r = 0; r < n_rows; r++) {
            uint32_t rec[64];
            memset(rec, 0, sizeof rec);
            bid_tick += ri(-1, 1);
            const int spread = ri(1, 2);
            rec[LOB_REC_TIME] = (uint32_t)(9 * 3600000 + r * 500 + (ri(0, 9) == 0 ? 0 : 0));
            for (int l = 0; l < P.D; l++) {
                rec[lob_rec_ask_px(P.D, P.T) + l] = lob_f32_bits((float)((bid_tick + spread + l) * tick));
                rec[lob_rec_bid_px(P.D, P.T) + l] = lob_f32_bits((float)((bid_tick - l) * tick));
                rec[lob_rec_ask_vol(P.D, P.T) + l] = (uint32_t)ri(1, ri(0, 3) ? 30 : 5000);
                rec[lob_rec_bid_vol(P.D, P.T) + l] = (uint32_t)ri(1, ri(0, 3) ? 30 : 5000);
            }
            for (int i = 0; i < P.T; i++)
                if (ri(0, 1)) {
                    rec[lob_rec_trade_px(P.D, P.T) + i] = lob_f32_bits((float)((bid_tick + ri(-1, spread + 1)) * tick));
                    rec[lob_rec_trade_vol(P.D, P.T) + i] = (uint32_t)ri(1, ri(0, 2) ? 40 : 3000);
                }
            drec_from_abi(rec, P.D, P.T, rows.data() + (size_t)r * P.Wd);
        }
        DevState S;
        memset(&S, 0, sizeof S);
        S.B = 1; S.D = P.D; S.T = P.T; S.W = P.W; S.n_events = n_rows;
        S.records = rows.data();
        std::vector<Track> track(16);
        memset(track.data(), 0, track.size() * sizeof(Track));
        for (auto& t : track) t.spread_mean = 0.05 + ru();
        S.track = track.data(); S.track_len = 16; S.track_mask = 0x7fffffff;
        BookMeta M;
        memset(&M, 0, sizeof M);
        M.n_track = 1000; M.complete = 1;
        S.meta = &M;
        int err = 0;
        S.error_flag = &err;
        TickLds tl;
        memset(&tl, 0, sizeof tl);
        EnvCtx c(P, S, 0, &tl);
        // the agent's state
        EnvR e;
        memset(&e, 0, sizeof e);
        e.rec_cur = ri(0, 3); e.rec_last = e.rec_cur - 1; e.pf = e.rec_cur - ri(0, 1); e.k = ri(1, 5);
        e.time_ms = 9 * 3600000;
        const uint32_t* rc = c.row(e.rec_cur);
        const double ap0 = (double)__uint_as_float(rc[drec_ask_px(P.D, P.T)]), bp0 = (double)__uint_as_float(rc[drec_bid_px(P.D, P.T)]);
        e.mid = (ap0 + bp0) / 2.0; e.mid_prev = e.mid + ri(-1, 1) * tick;
        e.position = ri(-6, 6) * (i64)P.order_size;
        e.a_on = ri(0, 4) != 0; e.b_on = ri(0, 4) != 0;
        // order prices: mostly the f64 image of a level's float price (so that keys match), sometimes off the grid / far away
        auto px = [&](double touch, int dir) {
            const int lvl = ri(-1, 4);
            double p = (double)(float)(touch + dir * lvl * tick);
            if (ri(0, 9) == 0) p = touch + dir * lvl * tick;  // the exact decimal, one float off
            if (ri(0, 30) == 0) p += 0.00004;
            return p;
        };
        e.a_opx = px(ap0, +1); e.b_opx = px(bp0, -1);
        e.a_osz = e.b_osz = P.order_size;
        auto q = [&](i64& qh, i64& qt, i64& ex) {
            qh = ri(0, 3) ? ri(0, 60) : ri(0, 5000);
            qt = ri(0, 2) == 0 ? 0 : (ri(0, 1) ? ri(-80, 80) : -qh);  // -qh: total 0 -> division by zero -> cvttsd2si corner
            ex = ri(0, 3) == 0 ? ri(0, P.order_size) : 0;
        };
        q(e.a_oqh, e.a_oqt, e.a_oex); q(e.b_oqh, e.b_oqt, e.b_oex);
        e.a_oiq = e.a_oqh; e.b_oiq = e.b_oqh;
        e.momentum_pnl_step = ru() - 0.5; e.ep_pnl = 100 * ru(); e.lo_vol_step = ri(0, 5);
        StepAgg g;
        memset(&g, 0, sizeof g);
        g.n_track = M.n_track; g.complete = 1; g.cv_valid = 1;
        g.cv_a = ri(0, 4) ? (i64)ri(0, 60) : 0; g.cv_b = ri(0, 4) ? (i64)ri(0, 5000) : 0;
        g.r = ru(); g.pnl = ru(); g.mpm = 0.0;
        EnvR e1 = e, e2 = e;
        StepAgg g1 = g, g2 = g;
        const FastKeys K{key4(e.a_opx), key4(e.b_opx)};
        RowFull L;
        int first = e.rec_cur + 1;
        row_full_load(c, first, L);
        for (int pass = 0; pass < 3 && first < n_rows - 1; pass++) {
            // the event: rows first..last (mostly one), its trades = slots of rows pf+1..first merged (what the pre-pass stores)
            const int last = (ri(0, 5) == 0 && first + 1 < n_rows - 1) ? first + 1 : first;
            n_multi += last > first;
            f64 tp[2]; i64 tv[2];
            load_trades<2>(c, e1.pf + 1, first, tp, tv);
            TrackHead64 t;
            memset(&t, 0, sizeof t);
            t.rec_first = first; t.rec_last = last;
            const uint32_t* rl = c.row(last);
            t.time_ms = (i32)rl[LOB_REC_TIME];
            t.bap = __uint_as_float(rl[drec_ask_px(P.D, P.T)]); t.bbp = __uint_as_float(rl[drec_bid_px(P.D, P.T)]);
            t.mid = ((double)t.bap + (double)t.bbp) / 2.0;
            int ntr = 0;
            for (int i = 0; i < 2; i++) ntr += (i < P.T && tv[i] > 0);
            t.info = ntr | LOB_TRK_TRADES_OK;
            t.tr_px[0] = (f32)tp[0]; t.tr_px[1] = (f32)tp[1]; t.tr_vol[0] = tv[0]; t.tr_vol[1] = tv[1];
            TrackHead t32;
            t32.rec_first = first; t32.rec_last = last; t32.time_ms = t.time_ms; t32.tick_ap0 = 0; t32.tick_bp0 = 0; t32.info = t.info; t32.mid = t.mid;
            const EnvR before = e1;
            const int s1 = step_event<2>(c, e1, g1, t32);
            if (t.rec_first != e2.rec_cur + 1) row_full_load(c, t.rec_first, L);
            const int s2 = pass_fast(c, e2, g2, t, L, K);
            n_pass++;
            n_fill += (e1.a_oex != before.a_oex) || (e1.b_oex != before.b_oex);
            n_cancel += (e1.a_oqh != before.a_oqh) || (e1.b_oqt != before.b_oqt);
            n_adverse += (before.a_on && !e1.a_on && e1.a_oex < e1.a_osz) || (before.b_on && !e1.b_on && e1.b_oex < e1.b_osz);
            if (s1 != s2 || memcmp(&e1, &e2, sizeof e1) != 0 || memcmp(&g1, &g2, sizeof g1) != 0) {
                printf("MISMATCH case %ld pass %d: status %d / %d\n", cs, pass, s1, s2);
#define X(t, n) if (memcmp(&e1.n, &e2.n, sizeof e1.n) != 0) printf("  field %s: general %.17g fast %.17g\n", #n, (double)e1.n, (double)e2.n);
                LOB_ENV_FIELDS(X)
#undef X
                printf("  g: r %.17g/%.17g pnl %.17g/%.17g mpm %.17g/%.17g cv %lld,%lld / %lld,%lld\n", g1.r, g2.r, g1.pnl, g2.pnl, g1.mpm, g2.mpm,
                       g1.cv_a, g1.cv_b, g2.cv_a, g2.cv_b);
                return 1;
            }
            if (s1 != 0) { /* keep going anyway: more passes from the state reached */ }
            first = last + 1;
            row_full_load(c, first < n_rows - 1 ? first : n_rows - 1, L);
        }
    }
    printf("pass_diff OK: %ld passes identical (%ld with fills, %ld with queue changes, %ld adverse, %ld multi-row)\n", n_pass, n_fill, n_cancel, n_adverse, n_multi);
    return 0;
}
