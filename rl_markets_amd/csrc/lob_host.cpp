// Host-only part of the C ABI (no GPU needed): parameter defaults, venue tick
// tables, host tick maths, synthetic stream generation and stream validation.
#include <system_error>
#include <thread>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <fstream>
#include <map>
#include <string>
#include <vector>

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
    p->policy = LOB_POLICY_EPS_GREEDY;
    p->tau = 1.0;
    p->beta = 0.005;
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

// One book's stream against the preconditions; the first offence's message into `buf` (LOB_EDATA), else LOB_OK.
static int validate_book(const uint32_t* rec, int32_t D, int32_t T, int b, int32_t n_events, char* buf, size_t cap) {
    const int W = lob_rec_words(D, T);
    int32_t last_t = -1;
    for (int e = 0; e < n_events; e++) {
        const uint32_t* r = rec + ((size_t)b * n_events + e) * W;
        int32_t t = (int32_t)r[LOB_REC_TIME];
        if (t < last_t) { snprintf(buf, cap, "book %d event %d: time goes backwards", b, e); return LOB_EDATA; }
        last_t = t;
        double pa = 0, pb = 1e300;
        for (int l = 0; l < D; l++) {
            double ap = lob_bits_f32(r[lob_rec_ask_px(D, T) + l]), bp = lob_bits_f32(r[lob_rec_bid_px(D, T) + l]);
            int32_t av = (int32_t)r[lob_rec_ask_vol(D, T) + l], bv = (int32_t)r[lob_rec_bid_vol(D, T) + l];
            // reference throws on <= 0 (src/market/book.cpp:74-77)
            if (!(ap > 0.0) || !(bp > 0.0) || av <= 0 || bv <= 0) {
                snprintf(buf, cap, "book %d event %d level %d: non-positive price/volume", b, e, l);
                return LOB_EDATA;
            }
            // engine precondition: strictly monotone 1e-4 price keys (no duplicate level keys)
            if (!(rint(ap * 10000.0) > rint(pa * 10000.0)) || !(rint(bp * 10000.0) < rint(pb * 10000.0))) {
                snprintf(buf, cap, "book %d event %d level %d: price keys not strictly best->worst", b, e, l);
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
                snprintf(buf, cap, "book %d event %d trade %d: trades must be packed, positive, ascending keys", b, e, i);
                return LOB_EDATA;
            }
            pt = p;
        }
    }
    return LOB_OK;
}

int lob_validate_stream(const uint32_t* rec, int32_t D, int32_t T, int32_t n_books, int32_t n_events) {
    if (!rec || D < 1 || D > LOB_MAX_DEPTH || T < 1 || T > LOB_MAX_TRADES) return LOB_EINVAL;
    // Books are independent: big uploads (65 536 books x 2 112 events = 138 M records, ~50 ns each on one core: 7 s) are checked by
    // up to 32 threads over contiguous book ranges.  Each stops at its first offence; the one reported is that of the lowest
    // book, which is what the serial scan would have found.
    const size_t total = (size_t)(n_books > 0 ? n_books : 0) * (size_t)(n_events > 0 ? n_events : 0);
    unsigned nt = 1;
    if (n_books >= 64 && total >= ((size_t)1 << 18)) {
        nt = std::thread::hardware_concurrency();
        if (nt < 1) nt = 1;
        if (nt > 32) nt = 32;
    }
    struct Slot { int rc = LOB_OK; int book = -1; char msg[160]; };
    std::vector<Slot> slots(nt);
    auto work = [&](unsigned k) {
        const int b0 = (int)((long long)n_books * k / nt), b1 = (int)((long long)n_books * (k + 1) / nt);
        for (int b = b0; b < b1; b++) {
            const int rc = validate_book(rec, D, T, b, n_events, slots[k].msg, sizeof slots[k].msg);
            if (rc != LOB_OK) { slots[k].rc = rc; slots[k].book = b; return; }
        }
    };
    if (nt == 1) work(0);
    else {
        // (std::thread's constructor throws std::system_error when the process may not start another thread -- a container's pids
        // limit, RLIMIT_NPROC: no exception may cross the C ABI, so the ranges whose thread could not be started are scanned here)
        std::vector<std::thread> th;
        unsigned started = 0;
        try {
            for (; started < nt; started++) th.emplace_back(work, started);
        } catch (const std::system_error&) {
        }
        for (unsigned k = started; k < nt; k++) work(k);
        for (auto& t : th) t.join();
    }
    for (unsigned k = 0; k < nt; k++)   // (ranges ascend with k: the first slot with an offence holds the lowest book)
        if (slots[k].rc != LOB_OK) { lob_set_error(slots[k].msg); return slots[k].rc; }
    return LOB_OK;
}


}  // extern "C"

// ---------------------------------------------------------------------------
// Ingestion
namespace {
void split_csv(const std::string& line, std::vector<std::string>& cols) {  // utilities/csv.cpp:30-45
    cols.clear();
    size_t pos = 0;
    while (true) {
        size_t next = line.find(',', pos);
        if (next == std::string::npos) { cols.push_back(line.substr(pos)); break; }
        cols.push_back(line.substr(pos, next - pos));
        pos = next + 1;
    }
}
void split_csv_append(const std::string& line, std::vector<std::string>& cols) {  // CSV::next: parseRow onto the caller's vector
    std::vector<std::string> one;
    split_csv(line, one);
    cols.insert(cols.end(), one.begin(), one.end());
}
bool parse_time(const std::string& s, long& out) {  // utilities/time.h:28-39 "HH:MM:SS.mmm"
    if (s.size() < 12) return false;
    out = atol(s.substr(0, 2).c_str()) * 3600000L + atol(s.substr(3, 2).c_str()) * 60000L +
          atol(s.substr(6, 2).c_str()) * 1000L + atol(s.substr(9, 3).c_str());
    return true;
}
struct Snap {
    long time;
    float ap[LOB_MAX_DEPTH], bp[LOB_MAX_DEPTH];
    int32_t av[LOB_MAX_DEPTH], bv[LOB_MAX_DEPTH];
};
struct Trade { long time; float price; long size; };

int emit_records(const std::vector<Snap>& snaps, const std::vector<Trade>& trades, int D, int T, uint32_t** out, int32_t* n, long t_dry = LONG_MAX) {
    const int W = lob_rec_words(D, T);
    const size_t N = snaps.size();
    if (N < 2) { lob_set_error("convert: fewer than 2 usable depth rows"); return LOB_EDATA; }
    uint32_t* rec = (uint32_t*)calloc(N * W, 4);
    if (!rec) return LOB_ENOMEM;
    size_t ti = 0;
    char buf[160];
    for (size_t r = 0; r < N; r++) {
        uint32_t* w = rec + r * W;
        const Snap& s = snaps[r];
        if (r > 0 && s.time < snaps[r - 1].time) { free(rec); lob_set_error("convert: depth rows go backwards in time"); return LOB_EDATA; }
        w[LOB_REC_TIME] = (uint32_t)(int32_t)s.time;
        w[LOB_REC_FLAGS] = ((r + 1 < N && snaps[r + 1].time == s.time) ? LOB_EVT_FLAG_SAME_TIME : 0) | (s.time >= t_dry ? LOB_EVT_FLAG_TAS_DRY : 0);
        // levels best -> worst (the reference sorts them itself, book.cpp:86)
        std::vector<std::pair<float, int32_t>> a, b;
        for (int l = 0; l < D; l++) { a.push_back({s.ap[l], s.av[l]}); b.push_back({s.bp[l], s.bv[l]}); }
        std::stable_sort(a.begin(), a.end(), [](const std::pair<float, int32_t>& x, const std::pair<float, int32_t>& y) { return rint((double)x.first * 10000) < rint((double)y.first * 10000); });
        std::stable_sort(b.begin(), b.end(), [](const std::pair<float, int32_t>& x, const std::pair<float, int32_t>& y) { return rint((double)x.first * 10000) > rint((double)y.first * 10000); });
        for (int l = 0; l < D; l++) {
            w[lob_rec_ask_px(D, T) + l] = lob_f32_bits(a[l].first);
            w[lob_rec_ask_vol(D, T) + l] = (uint32_t)a[l].second;
            w[lob_rec_bid_px(D, T) + l] = lob_f32_bits(b[l].first);
            w[lob_rec_bid_vol(D, T) + l] = (uint32_t)b[l].second;
        }
        // trades of (previous depth time, this depth time], per 1e-4 key, first-seen price kept (std::map semantics)
        std::map<long long, std::pair<float, long>> agg;
        while (ti < trades.size() && trades[ti].time <= s.time) {
            const Trade& t = trades[ti++];
            long long key = (long long)rint((double)t.price * 10000);
            auto it = agg.find(key);
            if (it == agg.end()) agg[key] = {t.price, t.size};
            else it->second.second += t.size;
            if (it != agg.end() && it->second.second > INT32_MAX) { free(rec); lob_set_error("convert: aggregated trade volume outside int32"); return LOB_EDATA; }
        }
        if ((int)agg.size() > T) {
            free(rec);
            snprintf(buf, sizeof buf, "convert: %zu trade price levels before depth row %zu, max_trades is %d", agg.size(), r, T);
            lob_set_error(buf);
            return LOB_EDATA;
        }
        int i = 0;
        for (auto& kv : agg) {
            w[lob_rec_trade_px(D, T) + i] = lob_f32_bits(kv.second.first);
            w[lob_rec_trade_vol(D, T) + i] = (uint32_t)kv.second.second;
            i++;
        }
    }
    int rc = lob_validate_stream(rec, D, T, 1, (int32_t)N);
    if (rc != LOB_OK) { free(rec); return rc; }
    *out = rec;
    *n = (int32_t)N;
    return LOB_OK;
}
}  // namespace

extern "C" {

void lob_free(void* p) { free(p); }

int lob_convert_csv(const char* md_path, const char* tas_path, int32_t T, uint32_t** out, int32_t* n) {
    if (!md_path || !tas_path || !out || !n || T < 1 || T > LOB_MAX_TRADES) { lob_set_error("lob_convert_csv: bad argument"); return LOB_EINVAL; }
    std::ifstream md(md_path), ts(tas_path);
    if (!md.is_open() || !ts.is_open()) { lob_set_error("lob_convert_csv: cannot open input (the reference exits, utilities/csv.cpp:16-19)"); return LOB_EINVAL; }
    std::string line;
    std::vector<std::string> c;
    std::vector<Snap> snaps;
    int date0 = 0;
    bool cut_md = false, cut_tas = false;
    std::getline(md, line);  // header
    // MarketDepth::_LoadRow (basic.cpp:31-43) keeps APPENDING the columns of the lines it reads to its row until the
    // row has exactly 22 of them (utilities/csv.cpp:31-53 never clears the vector): a line with fewer columns is glued
    // to the next one, and once the count has passed 22 no row is ever complete again -- the reference's day ends
    // there.  Same reader, same outcome: the rows after such a line are not part of the stream.
    c.clear();
    while (std::getline(md, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        split_csv_append(line, c);
        if (c.size() < 22) continue;
        if (c.size() > 22) { cut_md = true; break; }
        Snap s;
        int date = atoi(c[0].c_str());
        if (!date0) date0 = date;
        if (date != date0) { lob_set_error("lob_convert_csv: more than one date in the depth file (one episode = one day)"); return LOB_EDATA; }
        if (!parse_time(c[1], s.time)) { lob_set_error("lob_convert_csv: bad time"); return LOB_EDATA; }
        bool ok = true;
        for (int i = 0; i < 5 && ok; i++) {
            s.ap[i] = strtof(c[2 + i].c_str(), nullptr);   // stof: float32 prices (quirk Q8)
            s.bp[i] = strtof(c[12 + i].c_str(), nullptr);
            if (s.ap[i] <= 0.0f || s.bp[i] <= 0.0f) { ok = false; break; }  // row dropped (basic.cpp:54-58)
            const long av = atol(c[7 + i].c_str()), bv = atol(c[17 + i].c_str());
            if (av > INT32_MAX || bv > INT32_MAX || av < INT32_MIN || bv < INT32_MIN) {
                lob_set_error("lob_convert_csv: level volume outside int32 (records carry 32-bit volumes)");
                return LOB_EDATA;
            }
            s.av[i] = (int32_t)av;
            s.bv[i] = (int32_t)bv;
        }
        if (ok) snaps.push_back(s);
        c.clear();   // _ParseRow clears the row whether it keeps the record or not
    }
    std::vector<Trade> trades;
    std::vector<long> tas_times;   // the time of every complete row, kept or not: the streamer groups rows by it
    std::getline(ts, line);
    c.clear();
    while (std::getline(ts, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        split_csv_append(line, c);   // TimeAndSales::_LoadRow (basic.cpp:138-150): as above, with 4 columns
        if (c.size() < 4) continue;
        if (c.size() > 4) { cut_tas = true; break; }
        Trade t;
        if (!parse_time(c[1], t.time)) { lob_set_error("lob_convert_csv: bad time"); return LOB_EDATA; }
        t.price = strtof(c[2].c_str(), nullptr);
        t.size = atol(c[3].c_str());
        if (t.size > INT32_MAX) { lob_set_error("lob_convert_csv: trade size outside int32 (records carry 32-bit volumes)"); return LOB_EDATA; }
        if (t.price > 0.0f && t.size > 0) trades.push_back(t);  // basic.cpp:156-157
        tas_times.push_back(t.time);
        c.clear();
    }
    // The time-and-sales streamer runs dry before its file does: Streamer::LoadUntil(t) (streamer.cpp:61-85) merges
    // the row groups (runs of rows, each row no later than the group's first: TimeAndSales::_LoadNext, basic.cpp:164-181)
    // that are due at t into one record and then needs the NEXT group loaded behind it -- and _LoadNext reports failure
    // when the group it loads ends the file.  So a NextState aimed at a depth row at or after the time of the last
    // group but one fails before it touches the books (Intraday::NextState, intraday.cpp:225-232) and the reference's
    // episode is over.  Those rows are flagged LOB_EVT_FLAG_TAS_DRY: no event starts at them (they are still there
    // for an event that began earlier and runs through invalid states).  No trade rows at all: no event ever starts.
    std::vector<long> group_first;
    for (size_t i = 0; i < tas_times.size();) {
        const long target = tas_times[i];
        group_first.push_back(target);
        for (i++; i < tas_times.size() && tas_times[i] <= target; i++) {}
    }
    const long t_dry = group_first.size() >= 2 ? group_first[group_first.size() - 2] : group_first.size() == 1 ? group_first[0] : LONG_MIN;
    size_t n_dry = 0;
    for (const Snap& sn : snaps) n_dry += sn.time >= t_dry;
    // Trades are consumed in FILE order, as TimeAndSales::_LoadNext does (basic.cpp:164-181): no sort.
    int rc = emit_records(snaps, trades, 5, T, out, n, t_dry);
    if (rc == LOB_OK && (cut_md || cut_tas || n_dry)) {
        char buf[400];
        snprintf(buf, sizeof buf, "lob_convert_csv: %s%s%zu depth row(s) at the end flagged LOB_EVT_FLAG_TAS_DRY: the time-and-sales stream "
                 "runs dry before them (the reference's day ends at the same row)", cut_md ? "depth file cut at a row without 22 columns; " : "",
                 cut_tas ? "time-and-sales file cut at a row without 4 columns; " : "", n_dry);
        lob_set_error(buf);  // informational: the call succeeded, lob_last_error() carries the note
    }
    return rc;
}

int lob_convert_lobster(const char* ob_path, const char* msg_path, int32_t L, int32_t D, int32_t T, uint32_t** out, int32_t* n) {
    if (!ob_path || !msg_path || !out || !n || L < 1 || D < 1 || D > L || D > LOB_MAX_DEPTH || T < 1 || T > LOB_MAX_TRADES) {
        lob_set_error("lob_convert_lobster: bad argument");
        return LOB_EINVAL;
    }
    std::ifstream ob(ob_path), msg(msg_path);
    if (!ob.is_open() || !msg.is_open()) { lob_set_error("lob_convert_lobster: cannot open input"); return LOB_EINVAL; }
    std::string lo, lm;
    std::vector<std::string> co, cm;
    std::vector<Snap> snaps;
    std::vector<Trade> trades;
    while (std::getline(ob, lo) && std::getline(msg, lm)) {
        if (!lo.empty() && lo.back() == '\r') lo.pop_back();
        if (!lm.empty() && lm.back() == '\r') lm.pop_back();
        if (lo.empty() || lm.empty()) continue;
        split_csv(lo, co);
        split_csv(lm, cm);
        if ((int)co.size() < 4 * L || cm.size() < 6) { lob_set_error("lob_convert_lobster: short row"); return LOB_EDATA; }
        const double tsec = atof(cm[0].c_str());
        const long tms = (long)floor(tsec * 1000.0 + 1e-6);
        const int type = atoi(cm[1].c_str());
        if (type == 4 || type == 5) {
            Trade t;
            t.time = tms;
            t.price = (float)((double)atoll(cm[4].c_str()) / 10000.0);
            t.size = atol(cm[3].c_str());
            if (t.price > 0.0f && t.size > 0) trades.push_back(t);
        }
        Snap s;
        s.time = tms;
        bool ok = true;
        for (int l = 0; l < D; l++) {
            long long apx = atoll(co[4 * l + 0].c_str()), bpx = atoll(co[4 * l + 2].c_str());
            long asz = atol(co[4 * l + 1].c_str()), bsz = atol(co[4 * l + 3].c_str());
            if (apx <= 0 || bpx <= 0 || apx >= 9999999999LL || asz <= 0 || bsz <= 0) { ok = false; break; }  // LOBSTER dummy levels
            s.ap[l] = (float)((double)apx / 10000.0);
            s.bp[l] = (float)((double)bpx / 10000.0);
            s.av[l] = (int32_t)asz;
            s.bv[l] = (int32_t)bsz;
        }
        if (!ok) continue;
        if (!snaps.empty() && snaps.back().time == tms) snaps.back() = s;  // keep the last snapshot of a millisecond
        else snaps.push_back(s);
    }
    return emit_records(snaps, trades, D, T, out, n);
}

}  // extern "C"
