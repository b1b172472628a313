// Host-only part of the C ABI (no GPU needed): parameter defaults, venue tick
// tables, host tick maths, synthetic stream generation and stream validation.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "lob_stream.h"
#include "lob_internal.h"

static thread_local std::string g_err;
void lob_set_error(const std::string& s) { g_err = s; }

// ---------------------------------------------------------------------------
// Venue tables.  Data restated from the reference venue factory
// (src/market/market.cpp:39-59, tables :142-314): per venue the session
// open/close (ms of day) and the price -> tick-size bands.
namespace {
struct Band { double lb, tick; };
const Band kEuronext[] = {{0., 0.001}, {10., 0.005}, {50., 0.01}, {100., 0.05}};
const Band kNordic[] = {{0., 0.0001}, {0.5, 0.0005}, {1., 0.001}, {2., 0.002}, {5., 0.005}, {10., 0.01},
                        {50., 0.05}, {100., 0.1}, {500., 0.5}, {1000., 1.}, {5000., 5.}, {10000., 10.},
                        {20000., 20.}, {40000., 40.}, {50000., 50.}, {80000., 80.}, {100000., 100.}};
const Band kLseA[] = {{0., 0.0001}, {1., 0.0005}, {5., 0.001}, {10., 0.005}, {50., 0.01},
                      {100., 0.05}, {500., 0.1}, {1000., 0.5}, {5000., 1.}, {10000., 5.}};
const Band kLseB[] = {{0., 0.0001}, {0.5, 0.0005}, {1., 0.001}, {5., 0.005}, {10., 0.01}, {50., 0.05},
                      {100., 0.1}, {500., 0.5}, {1000., 1.}, {5000., 5.}, {10000., 10.}};
const Band kMilan[] = {{0., 0.0001}, {0.25, 0.0005}, {1., 0.001}, {2., 0.0025}, {5., 0.005}, {50., 0.01}};
const Band kSwiss[] = {{0., 0.0001}, {0.5, 0.0005}, {1., 0.001}, {5., 0.005}, {10., 0.01}, {50., 0.05},
                       {100., 0.1}, {500., 0.5}, {1000., 1.}, {5000., 5.}, {10000., 10.}};
const Band kVienna[] = {{0., 0.001}, {10., 0.005}, {50., 0.01}, {100., 0.5}};

long hm(long h, long m) { return h * 3600000L + m * 60000L; }

template <size_t N> void fill(lob_market* m, long open, long close, const Band (&b)[N]) {
    memset(m, 0, sizeof *m);
    m->open_ms = open;
    m->close_ms = close;
    m->n_bands = (int)N;
    for (size_t i = 0; i < N; i++) {
        m->band_lb[i] = b[i].lb;
        m->band_tick[i] = b[i].tick;
    }
}
bool in(const std::string& s, std::initializer_list<const char*> l) {
    for (auto x : l)
        if (s == x) return true;
    return false;
}
}  // namespace

extern "C" {

int lob_abi_version(void) { return LOB_ABI_VERSION; }
const char* lob_last_error(void) { return g_err.c_str(); }


int lob_market_preset(const char* ticker, lob_market* out) {
    if (!ticker || !out) return LOB_EINVAL;
    std::string t(ticker);
    size_t dot = t.find('.');
    if (dot == std::string::npos) { lob_set_error("ticker must be SYMBOL.VENUE"); return LOB_EINVAL; }
    std::string sym = t.substr(0, dot), ven = t.substr(dot + 1);
    for (auto& c : ven) c = (char)toupper(c);
    if (ven == "AS" || ven == "BR") fill(out, hm(9, 0), hm(17, 40), kEuronext);
    else if (ven == "PA") fill(out, hm(9, 0), hm(17, 30), kEuronext);
    else if (ven == "CO") fill(out, hm(9, 0), hm(17, 0), kNordic);
    else if (ven == "HE") fill(out, hm(10, 0), hm(16, 30), kNordic);
    else if (ven == "ST") fill(out, hm(9, 0), hm(17, 30), kNordic);
    else if (ven == "OL") fill(out, hm(9, 0), hm(16, 30), kNordic);
    else if (ven == "DE" || ven == "MC") fill(out, hm(9, 0), hm(17, 30), kEuronext);
    else if (ven == "I") fill(out, hm(8, 0), hm(16, 16) + 40, kEuronext);  // sic: add_minutes(16, 40) in the reference
    else if (ven == "MI") fill(out, hm(9, 0), hm(17, 25), kMilan);
    else if (ven == "S" || ven == "VX") fill(out, hm(9, 0), hm(17, 30), kSwiss);
    else if (ven == "VI") fill(out, hm(9, 0), hm(17, 30), kVienna);
    else if (ven == "L") {
        if (in(sym, {"AAL", "BATS", "GSK", "VOD", "HSBA"})) fill(out, hm(8, 0), hm(16, 30), kLseA);
        else if (in(sym, {"BAES", "UU", "LGEN", "LSE", "NXT"})) fill(out, hm(8, 0), hm(16, 30), kLseB);
        else { lob_set_error("unknown LSE symbol " + sym); return LOB_EINVAL; }
    } else {
        lob_set_error("unknown venue " + ven);
        return LOB_EINVAL;
    }
    return LOB_OK;
}

// ---------------------------------------------------------------------------
// Host tick maths: same arithmetic as the device functions in lob_device.h
// (reference Market::ToTicks/ToPrice/tick_size, src/market/market.cpp:78-138).
int lob_tick_size(const lob_market* m, double price, double* tick) {
    if (!m || !tick || m->n_bands < 1) return LOB_EINVAL;
    if (!(price >= m->band_lb[0])) { lob_set_error("invalid price for tick conversion"); return LOB_EINVAL; }
    *tick = lobh::tick_size(*m, price);
    return LOB_OK;
}
int lob_to_ticks(const lob_market* m, double price, int32_t* ticks) {
    if (!m || !ticks || m->n_bands < 1) return LOB_EINVAL;
    if (price < m->band_lb[0]) { lob_set_error("invalid price for tick conversion"); return LOB_EINVAL; }
    *ticks = lobh::to_ticks(*m, price);
    return LOB_OK;
}
int lob_to_price(const lob_market* m, int32_t ticks, double* price) {
    if (!m || !price || m->n_bands < 1) return LOB_EINVAL;
    if (ticks < 0) { lob_set_error("invalid tick count"); return LOB_EINVAL; }
    *price = lobh::to_price(*m, ticks);
    return LOB_OK;
}

// ---------------------------------------------------------------------------
void lob_default_params(lob_params* p) {
    memset(p, 0, sizeof *p);
    p->abi_version = LOB_ABI_VERSION;
    p->depth = 5;
    p->max_trades = 2;
    // state.variables of config/example.yaml:54
    const int v[8] = {LOB_VAR_POS, LOB_VAR_A_DIST, LOB_VAR_B_DIST, LOB_VAR_MPM,
                      LOB_VAR_SPD, LOB_VAR_VOL, LOB_VAR_IMB, LOB_VAR_SVL};
    p->n_vars = 8;
    for (int i = 0; i < 8; i++) p->vars[i] = v[i];
    lob_market_preset("HSBA.L", &p->market);
    p->order_size = 10;
    p->reward_measure = LOB_REWARD_PNL_DAMPED;
    p->pos_lb = -50;
    p->pos_ub = 50;
    p->damping_factor = 0.15f;
    p->pos_weight = 0.0f;
    p->trd_weight = 0.0f;
    p->pnl_weight = 1.0f;
    p->lb_mpm = 15;
    p->lb_vlt = 60;
    p->lb_svl = 60;
    p->lb_vwap = 1;   // max(0, 1)
    p->lb_rsi = 1;
    p->lb_spread = 45;
    p->lb_pnl = 1;
    p->lb_target = 1;
    p->target_price = LOB_TP_MICROPRICE;  // "midprice" -> MicroPrice, quirk Q5
    p->quote_mode = LOB_QUOTE_TARGET;
    p->memory_size = 20000000;
    p->n_tilings = LOB_N_TILINGS;
    p->n_actions = LOB_N_ACTIONS;
    p->group_weights[0] = 0.65;
    p->group_weights[1] = 0.25;
    p->group_weights[2] = 0.10;
    p->gamma = 0.975;
    p->lambda = 0.85;
    p->alpha = 0.001;
    p->epsilon = 0.8;
    p->algo = LOB_ALGO_SARSA;
    p->theta_mode = LOB_THETA_SHARED;
    p->seed = 1994;
    p->book_id_offset = 0;
}

void lob_default_gen_params(lob_gen_params* g) {
    memset(g, 0, sizeof *g);
    g->seed = 1994;
    g->n_events = 2112;            // 64 warm-up + 2048
    g->t0_ms = 8 * 3600000 + 30 * 60000 + 500;  // first snapshot after open + 30 min
    g->dt_ms = 500;
    g->start_ticks = 7000;         // 700.0 on the 0.1 grid
    g->min_ticks = 5500;
    g->max_ticks = 9500;
    g->move_prob_q16 = (int)(0.35 * 65536);
    g->spread2_prob_q16 = (int)(0.25 * 65536);
    g->trade_prob_q16 = (int)(0.50 * 65536);
    g->trade2_prob_q16 = (int)(0.15 * 65536);
    g->touch_prob_q16 = (int)(0.80 * 65536);
    g->vol_min = 100;
    g->vol_max = 5000;
    g->trade_min = 50;
    g->trade_max = 3000;
}

int32_t lob_record_words(int32_t depth, int32_t max_trades) { return lob_rec_words(depth, max_trades); }

int lob_gen_stream_host(const lob_gen_params* g, int32_t D, int32_t T, uint64_t first_book_id,
                        int32_t n_books, uint32_t* out) {
    if (!g || !out || D < 1 || D > LOB_MAX_DEPTH || T < 1 || T > LOB_MAX_TRADES || n_books < 0) {
        lob_set_error("lob_gen_stream_host: bad argument");
        return LOB_EINVAL;
    }
    const int W = lob_rec_words(D, T);
    for (int b = 0; b < n_books; b++) {
        lob_gen_state s;
        lob_gen_init(*g, s);
        for (int e = 0; e < g->n_events; e++)
            lob_gen_event(*g, D, T, first_book_id + (uint64_t)b, e, s,
                          out + ((size_t)b * g->n_events + e) * W);
    }
    return LOB_OK;
}

int lob_validate_stream(const uint32_t* rec, int32_t D, int32_t T, int32_t n_books, int32_t n_events) {
    if (!rec || D < 1 || D > LOB_MAX_DEPTH || T < 1 || T > LOB_MAX_TRADES) return LOB_EINVAL;
    const int W = lob_rec_words(D, T);
    char buf[160];
    for (int b = 0; b < n_books; b++) {
        int32_t last_t = -1;
        for (int e = 0; e < n_events; e++) {
            const uint32_t* r = rec + ((size_t)b * n_events + e) * W;
            int32_t t = (int32_t)r[LOB_REC_TIME];
            if (t < last_t) { snprintf(buf, sizeof buf, "book %d event %d: time goes backwards", b, e); lob_set_error(buf); return LOB_EDATA; }
            last_t = t;
            double pa = 0, pb = 1e300;
            for (int l = 0; l < D; l++) {
                double ap = lob_bits_f32(r[lob_rec_ask_px(D, T) + l]), bp = lob_bits_f32(r[lob_rec_bid_px(D, T) + l]);
                int32_t av = (int32_t)r[lob_rec_ask_vol(D, T) + l], bv = (int32_t)r[lob_rec_bid_vol(D, T) + l];
                // reference throws on <= 0 (src/market/book.cpp:74-77)
                if (!(ap > 0.0) || !(bp > 0.0) || av <= 0 || bv <= 0) {
                    snprintf(buf, sizeof buf, "book %d event %d level %d: non-positive price/volume", b, e, l);
                    lob_set_error(buf);
                    return LOB_EDATA;
                }
                // engine precondition: strictly monotone 1e-4 price keys (no duplicate level keys)
                if (!(rint(ap * 10000.0) > rint(pa * 10000.0)) || !(rint(bp * 10000.0) < rint(pb * 10000.0))) {
                    snprintf(buf, sizeof buf, "book %d event %d level %d: price keys not strictly best->worst", b, e, l);
                    lob_set_error(buf);
                    return LOB_EDATA;
                }
                pa = ap;
                pb = bp;
            }
            double pt = 0;
            bool ended = false;
            for (int i = 0; i < T; i++) {
                int32_t v = (int32_t)r[lob_rec_trade_vol(D, T) + i];
                double p = lob_bits_f32(r[lob_rec_trade_px(D, T) + i]);
                if (v == 0) { ended = true; continue; }
                if (ended || v < 0 || !(p > 0.0) || !(rint(p * 10000.0) > rint(pt * 10000.0))) {
                    snprintf(buf, sizeof buf, "book %d event %d trade %d: trades must be packed, positive, ascending keys", b, e, i);
                    lob_set_error(buf);
                    return LOB_EDATA;
                }
                pt = p;
            }
        }
    }
    return LOB_OK;
}

}  // extern "C"
