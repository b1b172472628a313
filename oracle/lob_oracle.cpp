// TEST INFRASTRUCTURE ONLY — see lob_oracle.h.  CPU restatement of the
// reference hot path (scalar, sequential, std::map-based like the reference
// itself so that each block can be read against the file:line it follows).
// Nothing here is shared with the HIP engine under rl_markets_amd/csrc/.
#include "lob_oracle.h"

#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <cfloat>
#include <deque>
#include <map>
#include <memory>
#include <random>
#include <stdexcept>
#include <tuple>
#include <unordered_map>
#include <thread>
#include <vector>

#include "../rl_markets_amd/csrc/lob_stream.h"  // record layout + lob_rng only (inputs, RNG)

// Debug statistics for kernel design (tools/env_pass_stats.py): what an event pass (NextState) did to the agent's two
// orders.  Bits of g_pass; counted per pass and OR-ed per step in the env loop below.  Not part of any comparison.
enum { PS_TOUCH = 1, PS_FILL = 2, PS_EXECUTED = 4, PS_CANCEL = 8, PS_LEVEL_GONE = 16, PS_NO_LAST = 32, PS_VOL_BEHIND = 64,
       PS_ADVERSE = 128, PS_ERASED = 256 };
static thread_local unsigned g_pass = 0;
static thread_local long long g_pass_stats[16] = {0};  // (of the calling thread: meaningful with ORACLE_THREADS unset)

namespace {

// ---------------------------------------------------------------------------
// include/utilities/comparison.h:4-34 — price keys at 1e-4.
struct Less {
    bool operator()(double a, double b) const { return rint(a * 10000) < rint(b * 10000); }
};
struct Greater {
    bool operator()(double a, double b) const { return rint(b * 10000) < rint(a * 10000); }
};
inline bool approx_equal(double a, double b) { return rint(a * 10000) == rint(b * 10000); }

// x86 cvttsd2si semantics for the double -> long conversions the reference
// performs implicitly in Order::doCancellation (quirk Q2): out-of-range / NaN
// give LONG_MIN ("integer indefinite").  Written out so that the GPU engine
// can be held to the same value instead of relying on UB.
inline long cvt_long(double d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return (long)0x8000000000000000ull;
    return (long)d;
}
inline long wrap_add(long a, long b) { return (long)((unsigned long)a + (unsigned long)b); }

typedef std::map<double, long, Less> TradeMap;  // data::TimeAndSalesRecord::transactions

// ---------------------------------------------------------------------------
// market::Order — include/market/order.h:9-48, src/market/order.cpp:12-141
struct Order {
    double price;
    long size;
    long q_head, q_tail;
    long total_executed = 0;
    long initial_queue;
    Order(double p, long s, long qh) : price(p), size(s), q_head(qh), q_tail(0), initial_queue(qh) {
        if (p <= 0) throw std::runtime_error("Order price must be non-zero and positive.");
        if (s <= 0) throw std::runtime_error("Order size must be non-zero and positive.");
        if (qh < 0) throw std::runtime_error("Order queue must be positive.");
    }
    long remaining() const { return std::max(size - total_executed, 0L); }
    bool isExecuted() const { return total_executed >= size; }
    long doTransaction(long volume) {  // order.cpp:54-82
        if (volume < 0) throw std::runtime_error("Transaction volume must be positive.");
        long remaining_volume = volume - q_head;
        if (remaining_volume > 0) {
            q_head = 0;
            if (remaining() <= remaining_volume) {
                total_executed = size;
                remaining_volume -= size;
            } else {
                total_executed += remaining_volume;
                remaining_volume = 0;
            }
        } else {
            q_head -= volume;
        }
        return std::max(remaining_volume, 0L);
    }
    void doCancellation(long volume) {  // order.cpp:84-107
        if (volume < 0) throw std::runtime_error("Cancellation volume must be positive.");
        if (q_tail == 0) {
            q_head -= volume;
        } else {
            double total = (double)(q_head + q_tail);
            // `q_head -= ceil(...)`: long op= double -> computed in double, converted back
            q_head = cvt_long((double)q_head - ceil((double)(volume * q_head) / total));
            q_tail = cvt_long((double)q_tail - floor((double)(volume * q_tail) / total));
        }
        if (q_head < 0) {
            q_tail = wrap_add(q_tail, q_head);
            q_head = 0;
        }
        if (q_tail < 0) q_tail = 0;
    }
    void addVolumeBehind(long v) { q_tail += v; }
    void clearQueues() { q_head = 0; q_tail = 0; }
    float getQueueProgress() const { return q_head / std::max(1.0f, (float)initial_queue); }
};

// ---------------------------------------------------------------------------
// market::Book<C,DEPTH> — include/market/book.h:22-110, src/market/book.cpp:17-378
template <class C> struct Book {
    typedef std::map<double, std::unique_ptr<Order>, C> OrderMap;
    int D;
    C cmp;
    OrderMap open_orders;
    std::vector<double> prices, last_prices;
    std::map<double, long, C> levels, last_levels;
    long total_volume_ = 0, last_total_volume_ = 0;
    int n_transacted_ = 0;
    double observed_transaction_value_ = 0.0;
    long observed_transaction_volume_ = 0;

    explicit Book(int depth) : D(depth), prices(depth, 0.0), last_prices(depth, 0.0) {}

    void Reset() {  // book.cpp:144-160
        n_transacted_ = 0;
        observed_transaction_value_ = 0.0;
        observed_transaction_volume_ = 0;
        total_volume_ = 0;
        last_total_volume_ = 0;
        std::fill(prices.begin(), prices.end(), 0.0);
        std::fill(last_prices.begin(), last_prices.end(), 0.0);
        levels.clear();
        last_levels.clear();
        open_orders.clear();
    }
    void StashState() {  // book.cpp:51-55
        prices.swap(last_prices);
        levels.swap(last_levels);
    }
    bool HasStash() const { return !approx_equal(last_prices[0], 0.0); }  // book.cpp:57-61
    double price(int level) const {  // book.cpp:167-177
        if (level < 0) level = D + level;
        if (level >= D || level < 0 || prices[level] == 0.0) throw std::runtime_error("undefined price");
        return prices[level];
    }
    double last_price(int level) const {  // book.cpp:179-189
        if (level < 0) level = D + level;
        if (level >= D || level < 0 || last_prices[level] == 0.0) throw std::runtime_error("undefined last price");
        return last_prices[level];
    }
    long volume(double p) const {
        auto it = levels.find(p);
        return it == levels.end() ? 0L : it->second;
    }
    long last_volume(double p) const {
        auto it = last_levels.find(p);
        return it == last_levels.end() ? 0L : it->second;
    }
    int price_level(double target) const {  // book.cpp:191-198
        for (int l = 0; l < D; l++)
            if (approx_equal(price(l), target)) return l;
        return -1;
    }
    void ApplyChanges(const double* new_prices, const long* new_volumes, const TradeMap& transactions) {
        // book.cpp:64-99
        levels.clear();
        last_total_volume_ = total_volume_;  // quirk Q1: cumulative, never reset
        for (int l = 0; l < D; l++) {
            if (new_prices[l] <= 0.0) throw std::runtime_error("Prices must be non-zero positive");
            else if (new_volumes[l] <= 0) throw std::runtime_error("Volumes must be non-zero positive");
            else {
                prices[l] = new_prices[l];
                levels[new_prices[l]] = new_volumes[l];
                total_volume_ += new_volumes[l];
            }
        }
        std::sort(prices.begin(), prices.end(), cmp);
        auto it = open_orders.begin();
        while (it != open_orders.end()) {
            auto it_t = transactions.find(it->first);
            long tv = (it_t == transactions.end()) ? 0L : it_t->second;
            it = UpdateOrder(it, tv);
        }
    }
    typename OrderMap::iterator UpdateOrder(typename OrderMap::iterator it, long transaction_volume) {
        // book.cpp:102-141
        double p = it->first;
        Order& o = *it->second;
        if (o.isExecuted()) { g_pass |= PS_ERASED; return open_orders.erase(it); }
        long lv = last_volume(p);
        if (lv == 0) { g_pass |= PS_NO_LAST; ++it; return it; }
        long v = volume(p);
        if (v == 0) { g_pass |= PS_LEVEL_GONE; o.clearQueues(); ++it; return it; }
        long vol_diff = lv - v;
        if (vol_diff >= 0) {
            long cancelled = vol_diff - transaction_volume;
            if (cancelled > 0) { g_pass |= PS_CANCEL; o.doCancellation(cancelled); }
        } else {
            g_pass |= PS_VOL_BEHIND;
            o.addVolumeBehind(vol_diff);  // quirk Q2: negative
        }
        ++it;
        return it;
    }
    bool PlaceOrder(double p, long size) {  // book.cpp:250-261
        if (open_orders.find(p) != open_orders.end()) return false;
        open_orders.emplace(p, std::unique_ptr<Order>(new Order(p, size, volume(p))));
        return true;
    }
    void CancelWorst() { open_orders.erase(--open_orders.rbegin().base()); }  // book.cpp:289-293
    void CancelAllOrders() { open_orders.clear(); }
    int order_count() const { return (int)open_orders.size(); }
    double best_open_order_price() const { return open_orders.begin()->first; }
    float queue_progress() const {  // book.cpp:346-352 (returns long in the reference: truncation)
        auto it = open_orders.begin();
        return it != open_orders.end() ? (float)(long)it->second->getQueueProgress() : -1.0f;
    }
};

typedef std::tuple<long, double, double> Fill;

struct AskBook : Book<Less> {
    explicit AskBook(int d) : Book<Less>(d) {}
    Fill ApplyTransactions(const TradeMap& transactions, double reference_price) {  // book.cpp:383-427
        auto o_it = open_orders.begin();
        observed_transaction_value_ = 0.0;
        observed_transaction_volume_ = 0;
        long volume = 0;
        double proxy = 0.0, value = 0.0;
        for (auto it = transactions.begin(); it != transactions.end(); ++it) {
            if (it->first < reference_price) continue;
            long vol = it->second;
            observed_transaction_value_ += it->first * vol;
            observed_transaction_volume_ += vol;
            while (o_it != open_orders.end() && o_it->first <= it->first) {
                long rem0 = o_it->second->remaining();
                vol = o_it->second->doTransaction(vol);
                long exec = rem0 - o_it->second->remaining();
                g_pass |= PS_TOUCH | (exec ? PS_FILL : 0);
                volume -= exec;
                proxy += (o_it->first - reference_price) * exec;
                value += o_it->first * exec;
                if (o_it->second->isExecuted()) {
                    g_pass |= PS_EXECUTED;
                    o_it = open_orders.erase(o_it);
                    n_transacted_++;
                }
                if (vol <= 0) break;
            }
        }
        return std::make_tuple(volume, proxy, value);
    }
    Fill WalkTheBook(double reference_price, long size) {  // book.cpp:431-456
        long abs_size = labs(size);
        if (abs_size > total_volume_) return std::make_tuple(0L, 0.0, 0.0);
        long executed = 0;
        double proxy = 0.0, value = 0.0;
        for (auto const& kv : levels) {
            long l_ex = std::min(kv.second, abs_size - executed);
            executed += l_ex;
            proxy -= l_ex * fabs(kv.first - reference_price);
            value -= l_ex * kv.first;
            if (executed >= abs_size) { n_transacted_++; break; }
        }
        return std::make_tuple(executed, proxy, value);
    }
};

struct BidBook : Book<Greater> {
    explicit BidBook(int d) : Book<Greater>(d) {}
    Fill ApplyTransactions(const TradeMap& transactions, double reference_price) {  // book.cpp:468-510
        // open_orders is keyed descending; rbegin() = lowest bid.  With the
        // one-order-per-side limit (quirk Q13) there is at most one entry.
        observed_transaction_value_ = 0.0;
        observed_transaction_volume_ = 0;
        long volume = 0;
        double proxy = 0.0, value = 0.0;
        for (auto it = transactions.rbegin(); it != transactions.rend(); ++it) {
            if (it->first > reference_price) continue;
            long vol = it->second;
            observed_transaction_value_ += it->first * vol;
            observed_transaction_volume_ += vol;
            while (!open_orders.empty() && open_orders.rbegin()->first >= it->first) {
                auto o_it = --open_orders.end();
                long rem0 = o_it->second->remaining();
                vol = o_it->second->doTransaction(vol);
                long exec = rem0 - o_it->second->remaining();
                g_pass |= PS_TOUCH | (exec ? PS_FILL : 0);
                volume += exec;
                proxy += (reference_price - o_it->first) * exec;
                value -= o_it->first * exec;
                bool erased = false;
                if (o_it->second->isExecuted()) {
                    g_pass |= PS_EXECUTED;
                    open_orders.erase(o_it);
                    n_transacted_++;
                    erased = true;
                }
                if (vol <= 0) break;
                if (!erased) break;  // single order not executed yet vol>0 cannot happen; guard against spinning
            }
        }
        return std::make_tuple(volume, proxy, value);
    }
    Fill WalkTheBook(double reference_price, long size) {  // book.cpp:514-539
        long abs_size = labs(size);
        if (abs_size > total_volume_) return std::make_tuple(0L, 0.0, 0.0);
        long executed = 0;
        double proxy = 0.0, value = 0.0;
        for (auto const& kv : levels) {
            long l_ex = std::min(kv.second, abs_size - executed);
            executed += l_ex;
            proxy -= l_ex * fabs(kv.first - reference_price);
            value += l_ex * kv.first;
            if (executed >= abs_size) { n_transacted_++; break; }
        }
        return std::make_tuple(-executed, proxy, value);
    }
};

// include/market/measures.h:9-75
inline double spread(AskBook& a, BidBook& b) { return a.price(0) - b.price(0); }
inline double midprice(AskBook& a, BidBook& b) { return (a.price(0) + b.price(0)) / 2.0f; }
inline double last_midprice(AskBook& a, BidBook& b) { return (a.last_price(0) + b.last_price(0)) / 2.0f; }
inline double midprice_move(AskBook& a, BidBook& b) { return midprice(a, b) - last_midprice(a, b); }
inline double microprice(AskBook& a, BidBook& b) {
    double ap = a.price(0), bp = b.price(0);
    long av = a.total_volume_, bv = b.total_volume_;
    double div = (double)(av + bv);
    double mpm_a = av * bp, mpm_b = ap * bv;
    return (mpm_a + mpm_b) / div;
}

// BookUtils — src/market/book.cpp:551-625
Fill HandleAdverseSelection(AskBook& ask, BidBook& bid) {
    const double bap = ask.price(0), bbp = bid.price(0), rp = last_midprice(ask, bid);
    long volume = 0;
    double proxy = 0.0, value = 0.0;
    auto it_a = ask.open_orders.begin();
    while (it_a != ask.open_orders.end() && it_a->first <= bbp) {
        long rem = it_a->second->remaining();
        volume -= rem;
        proxy += rem * (it_a->first - rp);
        value += rem * it_a->first;
        it_a->second->doTransaction(rem);
        it_a = ask.open_orders.erase(it_a);
        ask.n_transacted_++;
    }
    auto it_b = bid.open_orders.begin();
    while (it_b != bid.open_orders.end() && it_b->first >= bap) {
        long rem = it_b->second->remaining();
        volume += rem;
        proxy += rem * (rp - it_b->first);
        value -= rem * it_b->first;
        it_b->second->doTransaction(rem);
        it_b = bid.open_orders.erase(it_b);
        bid.n_transacted_++;
    }
    return std::make_tuple(volume, proxy, value);
}
Fill MarketOrder(long size, AskBook& ask, BidBook& bid) {
    double mip = midprice(ask, bid);
    if (size == 0L) return std::make_tuple(0L, 0.0, 0.0);
    else if (size > 0) return ask.WalkTheBook(mip, size);
    else return bid.WalkTheBook(mip, size);
}
bool IsValidState(AskBook& ask, BidBook& bid) {
    double mp = midprice(ask, bid);
    if (ask.HasStash() && bid.HasStash())
        return (spread(ask, bid) >= 0.0) && (mp > 0.0) && (fabs(midprice_move(ask, bid)) < mp);
    return true;
}

// ---------------------------------------------------------------------------
// market::Market tick maths — src/market/market.cpp:11-37,67-138
struct Market {
    std::map<double, double> pts_;
    std::map<int, double> tts_;
    long mo_, mc_;
    long time_ = 0;
    explicit Market(const lob_market& m) : mo_(m.open_ms), mc_(m.close_ms) {
        for (int i = 0; i < m.n_bands; i++) pts_[m.band_lb[i]] = m.band_tick[i];
        tts_[0] = pts_.begin()->second;
        long acc_ticks = 0;
        for (auto it = std::next(pts_.begin()); it != pts_.end(); it++) {
            auto pit = std::prev(it);
            acc_ticks = (long)((double)acc_ticks + (it->first - pit->first) / pit->second);
            tts_[(int)acc_ticks] = it->second;
        }
    }
    bool IsOpen() const { return (time_ > mo_ + 30 * 60000L) && (time_ < mc_ - 30 * 60000L); }
    double tick_size(double price) const {
        auto it = pts_.upper_bound(price);
        if (it == pts_.begin()) throw std::invalid_argument("invalid price");
        return std::prev(it)->second;
    }
    int ToTicks(double price) const {
        int ticks = 0;
        auto it = pts_.begin();
        if (price < it->first) throw std::invalid_argument("invalid price");
        while (it != pts_.end() && price + tick_size(it->first) / 2.0 > it->first) {
            double ub;
            auto nx = std::next(it);
            if (nx == pts_.end() || price < nx->first) ub = price + tick_size(price) / 2.0;
            else ub = nx->first;
            ticks = (int)((double)ticks + (ub - it->first) / it->second);
            it++;
        }
        return ticks;
    }
    double ToPrice(int ticks) const {
        double price = 0;
        auto it = tts_.begin();
        if (ticks < it->first) throw std::invalid_argument("invalid ticks");
        while (it != tts_.end() && ticks > it->first) {
            double ub;
            auto nx = std::next(it);
            if (nx == tts_.end() || ticks < nx->first) ub = ticks;
            else ub = nx->first;
            price += (ub - it->first) * it->second;
            it++;
        }
        return price;
    }
};

// ---------------------------------------------------------------------------
// utilities/accumulators — src/utilities/accumulators.cpp:11-175
struct Accumulator {
    size_t window_size;
    std::deque<double> window;
    double _sum = 0.0;
    explicit Accumulator(size_t w) : window_size(w) {}
    void push(double val) {
        _sum += val;
        window.push_front(val);
        if (window.size() > window_size) {
            _sum -= window.back();
            window.pop_back();
        }
    }
    double sum() const { return _sum; }
    double front() const { return window.front(); }
    double back() const { return window.back(); }
    void clear() { window.clear(); }  // quirk Q7: sums persist
    bool full() const { return window.size() == window_size; }
};
struct RollingMean : Accumulator {
    double _mean = 0.0, _s = 0.0;
    explicit RollingMean(size_t w) : Accumulator(w) {}
    void push(double val) {
        _sum += val;
        window.push_front(val);
        double n = (double)window.size();
        double old_mean = _mean;
        _mean += (val - _mean) / n;
        _s += (val - _mean) * (val - old_mean);
        if (window.size() > window_size) {
            double old = window.back();
            window.pop_back();
            _sum -= old;
            double n2 = (double)window.size();
            double old_mean2 = _mean;
            _mean -= (old - _mean) / n2;
            _s -= (old - _mean) * (old - old_mean2);
        }
    }
    double mean() const { return _mean; }
    double var() const { return _s / (double)(window.size() - 1); }
    double std() const {
        double v = var();
        return v > 0 ? sqrt(v) : 0.0;
    }
};
struct EWMA {
    double _alpha, _mean = 0.0;
    explicit EWMA(size_t w) : _alpha(2.0 / (w + 1.0)) {}
    void push(double val) { _mean = (_alpha * val) + ((1 - _alpha) * _mean); }
    double mean() const { return _mean; }
};

// ---------------------------------------------------------------------------
// rl/tiles — src/rl/tiles.cpp:31-75,130-169.  The 2048-entry table of the
// reference (tiles.cpp:133) is the byte stream `rand() & 0xff` of an unseeded
// glibc rand() (see the commented-out generator at tiles.cpp:141-149); it is
// regenerated here from the published glibc TYPE_3 additive-feedback
// algorithm (r[i] = r[i-3] + r[i-31]) and checked against the reference's
// table in tests/golden/.
struct RndSeq {
    uint32_t t[2048];
    RndSeq() {
        int32_t r[34 + 310 + 4 * 2048];
        r[0] = 1;
        for (int i = 1; i < 31; i++) {
            int64_t v = (16807LL * r[i - 1]) % 2147483647LL;
            if (v < 0) v += 2147483647LL;
            r[i] = (int32_t)v;
        }
        for (int i = 31; i < 34; i++) r[i] = r[i - 31];
        const int total = 344 + 4 * 2048;
        for (int i = 34; i < total; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
        for (int k = 0; k < 2048; k++) {
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) {
                uint32_t o = ((uint32_t)r[344 + 4 * k + i]) >> 1;
                v = (v << 8) | (o & 0xff);
            }
            t[k] = v;
        }
    }
};
const RndSeq& rndseq() {
    static RndSeq s;
    return s;
}
int hash_UNH(const int* ints, int num_ints, long m, int increment) {
    long sum = 0;
    for (int i = 0; i < num_ints; i++) {
        long index = ints[i];
        index += (increment * i);
        index = index & 2047;
        while (index < 0) index += 2048;
        sum += (long)rndseq().t[(int)index];
    }
    long index = (int)(sum % m);
    while (index < 0) index += m;
    return (int)index;
}
void tiles(int* the_tiles, int num_tilings, int memory_size, const float* floats, int num_floats, int h1) {
    int qstate[20], base[20], coordinates[42];
    int num_coordinates = num_floats + 1 + 1;
    coordinates[num_floats + 1] = h1;
    for (int i = 0; i < num_floats; i++) {
        // float multiply, double floor; the conversion is x86 `cvttsd2si`: NaN / out of range -> INT_MIN
        // (the reference does feed NaN through here: vwap over a window without trades is 0/0)
        double fd = floor(floats[i] * num_tilings);
        qstate[i] = (fd >= -2147483648.0 && fd < 2147483648.0) ? (int)fd : INT_MIN;
        base[i] = 0;
    }
    for (int j = 0; j < num_tilings; j++) {
        int i;
        for (i = 0; i < num_floats; i++) {
            // tiles.cpp:61-64.  With qstate = INT_MIN the subtractions overflow; the compiled reference
            // wraps (two's complement) and takes a signed remainder -- pinned by
            // tests/golden/kat_nonfinite.npz -- so that is spelt out here instead of left to the optimiser.
            const uint32_t q = (uint32_t)qstate[i], bs = (uint32_t)base[i], nt = (uint32_t)num_tilings;
            if (qstate[i] >= base[i]) coordinates[i] = (int)(q - (uint32_t)((int)(q - bs) % num_tilings));
            else coordinates[i] = (int)(q + 1u + (uint32_t)((int)(bs - q - 1u) % num_tilings) - nt);
            base[i] += 1 + (2 * i);
        }
        coordinates[i] = j;
        the_tiles[j] = hash_UNH(coordinates, num_coordinates, memory_size, 449);
    }
}
// rl::State::populateFeatures — src/rl/state.cpp:53-65
void populate_features(long M, int NT, int NA, const std::vector<float>& v, std::vector<std::vector<int>>& f) {
    for (int a = 0; a < NA; a++) {
        tiles(&f[a][0], NT, (int)M, &v[0], 3, a);
        tiles(&f[a][NT], NT, (int)M, &v[3], (int)v.size() - 3, NA + a);
        tiles(&f[a][2 * NT], NT, (int)M, &v[0], (int)v.size(), 2 * NA + a);
    }
}

// ---------------------------------------------------------------------------
// rl::Traces — src/rl/traces.cpp:8-102, sparse (per-book) instead of two dense
// MEMORY_SIZE arrays; list order and swap-with-last removal are kept.
struct Traces {
    float tolerance = 0.01f;
    std::vector<int> nonzero;
    std::unordered_map<int, std::pair<float, int>> el;  // feature -> (eligibility, loc)
    float get(int f) const {
        auto it = el.find(f);
        return it == el.end() ? 0.0f : it->second.first;
    }
    void clearExisting(int f, int loc) {
        el.erase(f);
        int last = nonzero.back();
        nonzero.pop_back();
        if (loc < (int)nonzero.size()) {
            nonzero[loc] = last;
            el[last].second = loc;
        }
    }
    void decay(float rate) {
        for (int loc = (int)nonzero.size() - 1; loc >= 0; loc--) {
            int f = nonzero[loc];
            auto& e = el[f];
            e.first *= rate;
            if (e.first < tolerance) clearExisting(f, loc);
        }
    }
    void set(int f, float value) {
        auto it = el.find(f);
        if (it != el.end() && it->second.first >= tolerance) it->second.first = value;
        else {
            el[f] = std::make_pair(value, (int)nonzero.size());
            nonzero.push_back(f);
        }
    }
    void clear(int f) {
        auto it = el.find(f);
        if (it != el.end() && it->second.first != 0.0f) clearExisting(f, it->second.second);
    }
    void update(const std::vector<std::vector<int>>& feats, int action, int NA, int NT) {
        for (int a = 0; a < NA; a++) {
            if (a != action) for (int t = 0; t < NT; t++) clear(feats[a][t]);
            else for (int t = 0; t < NT; t++) set(feats[a][t], 1.0f);
        }
    }
};

// ---------------------------------------------------------------------------
// environment::Base + Intraday<> — src/environment/base.cpp, intraday.cpp
struct Env {
    lob_params P;
    int D, T, W;
    const uint32_t* rec;  // this book's events
    int n_events;
    int cursor = 0;       // next depth row to load (market_depth.record_next)
    int trades_from = 0;  // first row whose trade slots have not been handed over yet (T&S stream position)
    bool exhausted = false;

    AskBook ask;
    BidBook bid;
    Market market;
    long position = 0;  // RiskManager::position_
    int last_action = 0, lo_vol_step = 0;
    double pnl_step = 0.0, momentum_pnl_step = 0.0;
    double ask_quote = 0.0, bid_quote = 0.0;
    int ask_level = 0, bid_level = 0;
    Accumulator f_vwap_numer, f_vwap_denom;
    RollingMean f_midprice, f_volatility, f_ask_tx, f_bid_tx, spread_window, pnl_ups, pnl_downs, tp_mp;
    EWMA return_ups, return_downs;
    double tp_val = -1.0;
    double ep_reward = 0, ep_pnl = 0, ep_bandh = 0;
    int total_ticks = 0, market_buys = 0, market_sells = 0;
    int ticks_with_ask = 0, ticks_with_bid = 0, ticks_with_both = 0, ticks_with_position = 0, ticks_long = 0, ticks_short = 0;  // TickStatistics
    int ask_transactions = 0, bid_transactions = 0;  // TradeStatistics: n_transacted of the two books as of the last UpdateStats (base.cpp:415-416)
    int64_t events_consumed = 0;

    Env(const lob_params& p, const uint32_t* r, int ne)
        : P(p), D(p.depth), T(p.max_trades), W(lob_rec_words(p.depth, p.max_trades)), rec(r), n_events(ne),
          ask(p.depth), bid(p.depth), market(p.market),
          f_vwap_numer(p.lb_vwap), f_vwap_denom(p.lb_vwap), f_midprice(p.lb_mpm), f_volatility(p.lb_vlt),
          f_ask_tx(p.lb_svl), f_bid_tx(p.lb_svl), spread_window(p.lb_spread), pnl_ups(p.lb_pnl),
          pnl_downs(p.lb_pnl), tp_mp(p.lb_target), return_ups(p.lb_rsi), return_downs(p.lb_rsi) {}

    // RiskManager — src/environment/risk_manager.cpp:26-113 (ORDER_LIMIT == 1)
    void CheckOrders() {
        if (position >= P.pos_ub) bid.CancelAllOrders();
        else if (position <= P.pos_lb) ask.CancelAllOrders();
    }
    template <class B> void rmPlace(B& book, double price, long size) {
        int oc = book.order_count();
        if (oc < 1) book.PlaceOrder(price, size);
        else { book.CancelWorst(); book.PlaceOrder(price, size); }
    }

    bool isTerminal() const { return !market.IsOpen(); }  // single-date streams

    double getReward() {  // base.cpp:166-237
        double r = 0.0;
        int abs_pos = (int)labs(position);
        switch (P.reward_measure) {
            case LOB_REWARD_NONE: break;
            case LOB_REWARD_PNL: r = pnl_step; break;
            case LOB_REWARD_PNL_DAMPED: r = pnl_step - P.damping_factor * std::max(0.0, momentum_pnl_step); break;
            case LOB_REWARD_SPREAD: r = pnl_step / spread_window.mean(); break;
            case LOB_REWARD_LOVOL: r = lo_vol_step; break;
            case LOB_REWARD_MM_LINEAR: r = -P.pos_weight * abs_pos; r += P.pnl_weight * pnl_step; break;
            case LOB_REWARD_MM_EXP:
                r = -pow(1.0 - (double)expf(P.pos_weight * abs_pos), 2);  // std::exp(float) in the reference
                r += P.pnl_weight * pnl_step;
                break;
            case LOB_REWARD_MM_DIV:
                if (pnl_step > 0) r = pnl_step / std::max(1.0, (double)abs_pos);
                else r = pnl_step;
                break;
            case LOB_REWARD_NORMED: {
                if (!(pnl_ups.full() && pnl_downs.full())) r = 0.0;
                else {
                    double u = pnl_ups.mean(), d = pnl_downs.mean(), su = pnl_ups.std(), sd = pnl_downs.std();
                    double numer = (u * sd - d * su), denom = (su + sd);
                    if (std::isnan(numer) || std::isinf(numer)) numer = 0.0;
                    if (std::isnan(denom) || std::isinf(denom)) denom = 0.0;
                    r = (fabs(denom) < 1e-5) ? numer : (numer / denom);
                }
                break;
            }
        }
        return r * 100;
    }

    const uint32_t* row(int i) const { return rec + (size_t)i * W; }

    // Intraday::UpdateBookProfiles — intraday.cpp:275-313 over the record stream.
    // market_depth.LoadNext() fails when there is no row AFTER the one being
    // made current (Streamer::LoadNext, src/data/streamer.cpp:42-49), so the
    // last record of a stream is never applied.
    bool UpdateBookProfiles(const TradeMap& transactions) {
        ask.StashState();
        bid.StashState();
        while (true) {
            if (cursor + 1 >= n_events) { exhausted = true; return false; }
            const uint32_t* r = row(cursor);
            cursor++;
            events_consumed++;
            market.time_ = (long)(int32_t)r[LOB_REC_TIME];
            double ap[LOB_MAX_DEPTH], bp[LOB_MAX_DEPTH];
            long av[LOB_MAX_DEPTH], bv[LOB_MAX_DEPTH];
            for (int l = 0; l < D; l++) {
                ap[l] = (double)lob_bits_f32(r[lob_rec_ask_px(D, T) + l]);
                bp[l] = (double)lob_bits_f32(r[lob_rec_bid_px(D, T) + l]);
                av[l] = (int32_t)r[lob_rec_ask_vol(D, T) + l];
                bv[l] = (int32_t)r[lob_rec_bid_vol(D, T) + l];
            }
            ask.ApplyChanges(ap, av, transactions);
            bid.ApplyChanges(bp, bv, transactions);
            // WillTimeChange(): the next row has a different timestamp
            if ((long)(int32_t)row(cursor)[LOB_REC_TIME] == market.time_) continue;
            try {
                if (IsValidState(ask, bid)) break;
            } catch (std::runtime_error&) {
                continue;
            }
        }
        return true;
    }

    // Intraday::NextState — intraday.cpp:225-272
    bool NextState() {
        if (cursor >= n_events) { exhausted = true; return false; }
        // time_and_sales.LoadUntil fails (streamer.cpp:61-85): the converter marked the rows from which on it does
        if (row(cursor)[LOB_REC_FLAGS] & LOB_EVT_FLAG_TAS_DRY) { exhausted = true; return false; }
        // trades carried by the record about to be applied (= LoadUntil(next depth time))
        // time_and_sales.LoadUntil(next depth row's time): every trade up to that row which
        // has not been handed over yet (rows carry the trades of their own interval)
        g_pass = 0;
        TradeMap tx;
        for (int rr = trades_from; rr <= cursor; rr++) {
            const uint32_t* r = row(rr);
            for (int i = 0; i < T; i++) {
                int32_t v = (int32_t)r[lob_rec_trade_vol(D, T) + i];
                double p = (double)lob_bits_f32(r[lob_rec_trade_px(D, T) + i]);
                if (p > 0.0 && v > 0) tx[p] += v;
            }
        }
        trades_from = cursor + 1;
        double mp = midprice(ask, bid);
        Fill au = ask.ApplyTransactions(tx, mp), bu = bid.ApplyTransactions(tx, mp);
        if (!UpdateBookProfiles(tx)) return false;
        Fill adv = HandleAdverseSelection(ask, bid);
        if (std::get<0>(adv) != 0) g_pass |= PS_ADVERSE;
        pnl_step += std::get<1>(au) + std::get<1>(bu) + std::get<1>(adv);
        lo_vol_step += (int)(std::get<0>(bu) - std::get<0>(au) + labs(std::get<0>(adv)));
        ep_pnl += std::get<2>(au) + std::get<2>(bu) + std::get<2>(adv);
        position += std::get<0>(bu) + std::get<0>(au) + std::get<0>(adv);
        CheckOrders();
        long mpt = market.ToTicks(midprice(ask, bid));
        double mpm = midprice_move(ask, bid), sp = spread(ask, bid);
        f_midprice.push((double)mpt);
        f_volatility.push((double)mpt);
        f_vwap_numer.push(ask.observed_transaction_value_ + bid.observed_transaction_value_);
        f_vwap_denom.push((double)(ask.observed_transaction_volume_ + bid.observed_transaction_volume_));
        spread_window.push(std::max(0.0, sp));
        // target_price_->update: MidPrice / MicroPrice (src/market/target_price.cpp:44-71)
        tp_mp.push(P.target_price == LOB_TP_MICROPRICE ? microprice(ask, bid) : midprice(ask, bid));
        tp_val = tp_mp.mean();
        return_ups.push(std::max(0.0, mpm));
        return_downs.push(fabs(std::min(0.0, mpm)));
        f_ask_tx.push((double)ask.observed_transaction_volume_);
        f_bid_tx.push((double)bid.observed_transaction_volume_);
        return true;
    }

    void place_orders(int al, int bl) {  // intraday.cpp:164-173, l2p_ :64-82
        ask_level = al;
        bid_level = bl;
        if (P.quote_mode == LOB_QUOTE_BOOK) {
            ask_quote = market.ToPrice(market.ToTicks(ask.price(0)) + al);
            bid_quote = market.ToPrice(market.ToTicks(bid.price(0)) - bl);
        } else {
            double tp = tp_val, half_spd = std::max(0.0, spread_window.mean() / 2.0);
            ask_quote = market.ToPrice(market.ToTicks(tp + al * half_spd));
            bid_quote = market.ToPrice(market.ToTicks(tp - bl * half_spd));
        }
        rmPlace(ask, ask_quote, P.order_size);
        rmPlace(bid, bid_quote, P.order_size);
    }

    void ClearInventory() {  // base.cpp:339-349 + risk_manager.cpp:101-113
        Fill out = MarketOrder(-position, ask, bid);
        position += std::get<0>(out);
        pnl_step += std::get<1>(out);
        lo_vol_step += (int)labs(std::get<0>(out));
        ep_pnl += std::get<2>(out);
        if (std::get<0>(out) > 0) market_buys++;
        else if (std::get<0>(out) < 0) market_sells++;
    }

    void DoAction(int action) {  // intraday.cpp:176-220
        switch (action) {
            case 0: place_orders(1, 1); break;
            case 1: ClearInventory(); place_orders(ask_level, bid_level); break;
            case 2: place_orders(2, 2); break;
            case 3: place_orders(3, 3); break;
            case 4: place_orders(0, 2); break;
            case 5: place_orders(2, 0); break;
            case 6: place_orders(1, 4); break;
            case 7: place_orders(4, 1); break;
            case 8: place_orders(5, 5); break;
        }
    }

    bool Initialise() {  // base.cpp:123-135 + intraday.cpp:103-138
        ask_quote = bid_quote = 0.0;
        ask.Reset();
        bid.Reset();
        ep_reward = ep_pnl = ep_bandh = 0;  // ClearStats
        total_ticks = market_buys = market_sells = 0;
        ticks_with_ask = ticks_with_bid = ticks_with_both = ticks_with_position = ticks_long = ticks_short = 0;
        ask_transactions = bid_transactions = 0;
        spread_window.clear(); tp_mp.clear(); f_midprice.clear(); f_volatility.clear();
        f_vwap_numer.clear(); f_vwap_denom.clear(); pnl_ups.clear(); pnl_downs.clear();
        f_ask_tx.clear(); f_bid_tx.clear();
        cursor = 0;
        exhausted = false;
        market.time_ = 0;
        TradeMap none;
        while (!market.IsOpen())
            if (!UpdateBookProfiles(none)) return false;
        trades_from = cursor;  // time_and_sales.SkipUntil(market time), intraday.cpp:116
        while (!(f_ask_tx.full() && f_bid_tx.full() && f_vwap_numer.full() && f_vwap_denom.full() &&
                 f_volatility.full() && f_midprice.full() && tp_mp.full() && spread_window.full()))
            if (!NextState()) return false;
        // the second time_and_sales.SkipUntil(market time), intraday.cpp:130: when the last of these NextStates went
        // through more than one depth row (an invalid state in between), the trades up to the last of them are dropped
        trades_from = cursor;
        place_orders(1, 1);
        return true;
    }

    bool performAction(int action) {  // base.cpp:254-337
        last_action = action;
        lo_vol_step = 0;
        pnl_step = 0.0;
        momentum_pnl_step = 0.0;
        DoAction(action);
        CheckOrders();
        ask_transactions = ask.n_transacted_; bid_transactions = bid.n_transacted_;  // UpdateStats (base.cpp:412-442)
        total_ticks++;
        {
            const bool has_ask = ask.order_count() > 0, has_bid = bid.order_count() > 0;
            if (has_ask) ticks_with_ask++;
            if (has_bid) ticks_with_bid++;
            if (has_ask && has_bid) ticks_with_both++;
            const long exposure = position;
            if (exposure != 0) ticks_with_position++;
            if (exposure > 0) ticks_long++;
            else if (exposure < 0) ticks_short++;
        }
        double agg_r = getReward();
        double agg_pnl = pnl_step;
        double agg_mpm = 0.0;
        unsigned step_or = 0;
        int n_pass = 0;
        do {
            pnl_step = 0.0;
            if (!NextState()) return false;
            g_pass_stats[0]++;                                   // passes
            for (int b = 0; b < 9; b++) g_pass_stats[1 + b] += (g_pass >> b) & 1;
            if (!(g_pass & ~(unsigned)PS_VOL_BEHIND)) g_pass_stats[10]++;   // quiet pass: nothing but more volume behind
            step_or |= g_pass;
            n_pass++;
            double mpm = midprice_move(ask, bid);
            pnl_step += position * mpm;
            momentum_pnl_step += position * mpm;
            agg_r += getReward();
            agg_pnl += pnl_step;
            agg_mpm += mpm;
        } while (!isTerminal() && fabs(agg_mpm) < 1e-5);
        g_pass_stats[11]++;                                      // steps
        if (!(step_or & ~(unsigned)PS_VOL_BEHIND)) g_pass_stats[12]++;      // steps all of whose passes were quiet
        if (!(step_or & (PS_FILL | PS_EXECUTED | PS_ADVERSE | PS_ERASED | PS_LEVEL_GONE))) g_pass_stats[13]++;  // no fill / structural change
        g_pass_stats[14] += n_pass > 1;
        pnl_step = agg_pnl;
        pnl_ups.push(std::max(0.0, pnl_step));
        pnl_downs.push(fabs(std::min(0.0, pnl_step)));
        ep_reward += agg_r;
        ep_bandh += agg_mpm;
        return true;
    }

    double getVariable(int v) {  // intraday.cpp:316-409
        auto ulb = [](double val, double lb, double ub) { return std::max(std::min(val, ub), lb); };
        switch (v) {
            case LOB_VAR_POS: return double(position) / P.order_size;
            case LOB_VAR_SPD:
                return ulb((double)(market.ToTicks(ask.price(0)) - market.ToTicks(bid.price(0))), 0.0, 20.0);
            case LOB_VAR_MPM:
                return ulb((double)(market.ToTicks(f_midprice.front()) - market.ToTicks(f_midprice.back())), -10.0, 10.0);
            case LOB_VAR_IMB: {
                double v_a = (double)ask.total_volume_, v_b = (double)bid.total_volume_;
                return ((v_a + v_b) > 0 ? 5 * (v_b - v_a) / (v_b + v_a) : 0.0);
            }
            case LOB_VAR_SVL: {
                double q_a = f_ask_tx.sum(), q_b = f_bid_tx.sum();
                return ((q_a + q_b) > 0 ? 5 * (q_b - q_a) / (q_a + q_b) : 0.0);
            }
            case LOB_VAR_VOL: return ulb(5.0 * f_volatility.std(), 0.0, 10.0);
            case LOB_VAR_RSI: {
                double u = return_ups.mean(), d = return_downs.mean();
                return (u + d) != 0.0 ? 5.0 * (u - d) / (u + d) : 0.0;
            }
            case LOB_VAR_VWAP: {
                double d = f_vwap_numer.sum() / f_vwap_denom.sum();
                return ulb(d / spread_window.mean(), -10.0, 10.0);
            }
            case LOB_VAR_A_DIST:
                if (ask.order_count() > 0)
                    return ((double)market.ToTicks(ask.best_open_order_price()) - (double)market.ToTicks(ask.price(0)));
                return -100.0;
            case LOB_VAR_A_QUEUE:
                if (ask.order_count() > 0) return 10.0 * ask.queue_progress();
                return -1.0;
            case LOB_VAR_B_DIST:
                if (bid.order_count() > 0)
                    return ((double)market.ToTicks(bid.price(0)) - (double)market.ToTicks(bid.best_open_order_price()));
                return -100.0;
            case LOB_VAR_B_QUEUE:
                if (bid.order_count() > 0) return 10.0 * bid.queue_progress();
                return -1.0;
            case LOB_VAR_LAST_ACTION: return last_action;
        }
        throw std::invalid_argument("unknown state variable");
    }
    void getState(std::vector<float>& out) {
        for (int i = 0; i < P.n_vars; i++) out.push_back((float)getVariable(P.vars[i]));
    }

    void fill(lob_book_dump& d) {
        memset(&d, 0, sizeof d);
        for (int l = 0; l < D; l++) {
            if (ask.prices[l] != 0.0) { d.ask_px[l] = ask.prices[l]; d.ask_vol[l] = ask.volume(ask.prices[l]); }
            if (bid.prices[l] != 0.0) { d.bid_px[l] = bid.prices[l]; d.bid_vol[l] = bid.volume(bid.prices[l]); }
            if (ask.last_prices[l] != 0.0) { d.ask_last_px[l] = ask.last_prices[l]; d.ask_last_vol[l] = ask.last_volume(ask.last_prices[l]); }
            if (bid.last_prices[l] != 0.0) { d.bid_last_px[l] = bid.last_prices[l]; d.bid_last_vol[l] = bid.last_volume(bid.last_prices[l]); }
        }
        d.ask_total_volume = ask.total_volume_; d.bid_total_volume = bid.total_volume_;
        d.ask_last_total_volume = ask.last_total_volume_; d.bid_last_total_volume = bid.last_total_volume_;
        d.ask_n_transacted = ask.n_transacted_; d.bid_n_transacted = bid.n_transacted_;
        d.ask_has_order = ask.order_count(); d.bid_has_order = bid.order_count();
        if (d.ask_has_order) {
            Order& o = *ask.open_orders.begin()->second;
            d.ask_order_px = ask.open_orders.begin()->first; d.ask_order_rem = o.remaining();
            d.ask_q_head = o.q_head; d.ask_q_tail = o.q_tail;
        }
        if (d.bid_has_order) {
            Order& o = *bid.open_orders.begin()->second;
            d.bid_order_px = bid.open_orders.begin()->first; d.bid_order_rem = o.remaining();
            d.bid_q_head = o.q_head; d.bid_q_tail = o.q_tail;
        }
        d.position = position;
        d.ask_quote = ask_quote; d.bid_quote = bid_quote;
        d.ask_level = ask_level; d.bid_level = bid_level;
        d.pnl_step = pnl_step; d.momentum_pnl_step = momentum_pnl_step;
        d.lo_vol_step = lo_vol_step; d.last_action = last_action;
        d.episode_reward = ep_reward; d.episode_pnl = ep_pnl; d.episode_bandh = ep_bandh;
        d.spread_mean = spread_window.mean(); d.target_price = tp_val;
        d.time_ms = market.time_;
        d.cursor = cursor;
        d.terminal = exhausted ? 2 : (isTerminal() ? 1 : 0);
        d.total_ticks = total_ticks;
        d.market_buys = market_buys; d.market_sells = market_sells;
        d.ticks_with_ask = ticks_with_ask; d.ticks_with_bid = ticks_with_bid; d.ticks_with_both = ticks_with_both;
        d.ticks_with_position = ticks_with_position; d.ticks_long = ticks_long; d.ticks_short = ticks_short;
        d.ask_transactions = ask_transactions; d.bid_transactions = bid_transactions;
    }
};

}  // namespace

// ---------------------------------------------------------------------------
// Learner: rl::Agent (SARSA / QLearn) + experiment::serial::Learner::_step
struct oracle_learner {
    lob_params P;
    int B;
    long M;
    std::vector<std::unique_ptr<Env>> env;
    std::vector<std::vector<double>> theta;    // 1 (shared) or B (private)
    std::vector<std::vector<double>> theta_b;  // DoubleAgent::theta_b (agent.cpp:188)
    double ml_agg = 0.0;                        // Agent::_agg_delta / _update_counter and the rows handed to "model_log" (agent.cpp:93-100)
    int64_t ml_cnt = 0;
    std::vector<double> ml_rows;
    std::vector<std::mt19937_64> agent_gen;    // Agent::gen (agent.cpp:31): the DoubleQLearn coin
    std::vector<std::uniform_real_distribution<double>> agent_unif;
    std::vector<Traces> traces;
    // state / last_state per book (Runner::state1/state2)
    std::vector<std::vector<float>> vars, last_vars;
    std::vector<std::vector<std::vector<int>>> feats, last_feats;
    std::vector<uint64_t> rng_ctr;
    std::vector<int> done;  // 0 live, 1 terminal, 2 out of data
    std::vector<oracle_step_rec> recs;
    double alpha, epsilon, tau;
    std::vector<double> rho;  // RLearn / OnlineRLearn::rho (include/rl/agent.h:131,145): one per agent (1 shared, B private)
    double& rh(int b) { return rho[P.theta_mode == LOB_THETA_PRIVATE ? b : 0]; }
    int64_t n_steps_done = 0, n_updates = 0;

    double* th(int b) { return theta[P.theta_mode == LOB_THETA_PRIVATE ? b : 0].data(); }
    double* thb(int b) { return theta_b[P.theta_mode == LOB_THETA_PRIVATE ? b : 0].data(); }

    uint64_t raw(int b) { return lob_rng(P.seed, P.book_id_offset + (uint64_t)b, rng_ctr[b]++); }
    int rnd(int b) { return (int)(raw(b) >> 33); }  // interposed libc rand()

    // a step run in two halves: the weights of its first half, the per-book hand-over, the step's accumulators
    std::vector<std::vector<double>> theta_from, theta_b_from;
    bool have_from = false;
    std::vector<int> pend_a;
    std::vector<double> pend_reward;
    std::vector<char> pend_live;
    std::vector<double> st_upd, st_rl_q, st_rl_t, st_rl_r;
    std::vector<char> st_has;
    double getQ_from(int b, const std::vector<std::vector<int>>& f, int action, bool use_b = false) {
        if (!have_from) return getQ(b, f, action, use_b);
        const int i = P.theta_mode == LOB_THETA_PRIVATE ? b : 0;
        return getQ_t((use_b ? theta_b_from : theta_from)[i].data(), f, action);
    }
    int argmaxQ_from(int b, const std::vector<std::vector<int>>& f, bool use_b = false) {
        if (!have_from) return argmaxQ(b, f, use_b);
        int index = 0, n_ties = 1;
        double cur = getQ_from(b, f, 0, use_b);
        for (int a = 1; a < 9; a++) {
            double val = getQ_from(b, f, a, use_b);
            if (val >= cur) {
                if (val > cur) { cur = val; index = a; }
                else {
                    n_ties++;
                    if (0 == rnd(b) % n_ties) { cur = val; index = a; }
                }
            }
        }
        return index;
    }
    double getQ_t(const double* t, const std::vector<std::vector<int>>& f, int action) {
        const std::vector<int>& ft = f[action];
        double Q = 0.0;
        double w = P.group_weights[0];
        for (int i = 0; i < 32; i++) Q += w * t[ft[i]];
        w = P.group_weights[1];
        for (int i = 32; i < 64; i++) Q += w * t[ft[i]];
        w = P.group_weights[2];
        for (int i = 32; i < 96; i++) Q += w * t[ft[i]];
        return Q;
    }
    double getQ(int b, const std::vector<std::vector<int>>& f, int action, bool use_b = false) {  // agent.cpp:117-135 / getQb :206-227 (quirk Q3)
        const double* t = use_b ? thb(b) : th(b);
        const std::vector<int>& ft = f[action];
        double Q = 0.0;
        double w = P.group_weights[0];
        for (int i = 0; i < 32; i++) Q += w * t[ft[i]];
        w = P.group_weights[1];
        for (int i = 32; i < 64; i++) Q += w * t[ft[i]];
        w = P.group_weights[2];
        for (int i = 32; i < 96; i++) Q += w * t[ft[i]];
        return Q;
    }
    int argmaxQ(int b, const std::vector<std::vector<int>>& f, bool use_b = false) {  // agent.cpp:144-169 / argmaxQb :236-262
        int index = 0, n_ties = 1;
        double cur = getQ(b, f, 0, use_b);
        for (int a = 1; a < 9; a++) {
            double val = getQ(b, f, a, use_b);
            if (val >= cur) {
                if (val > cur) { cur = val; index = a; }
                else {
                    n_ties++;
                    if (0 == rnd(b) % n_ties) { cur = val; index = a; }
                }
            }
        }
        return index;
    }
    int greedy_sample(int b, const double* qs) {  // Greedy::Sample, policy.cpp:37-55
        int argmax = 0, n_ties = 1;
        for (int a = 1; a < 9; a++) {
            if (qs[a] > qs[argmax]) argmax = a;
            else if (qs[a] >= qs[argmax]) {
                n_ties++;
                if (0 == rnd(b) % n_ties) argmax = a;
            }
        }
        return argmax;
    }
    int policy_sample(int b, const double* qs, bool greedy) {  // EpsilonGreedy::Sample, policy.cpp:69-75
        if (!greedy && P.policy == LOB_POLICY_BOLTZMANN) {  // Boltzmann::Sample, policy.cpp:98-117
            double prob[9], z = 0.0;
            for (int a = 0; a < 9; a++) {
                prob[a] = std::exp(qs[a] / tau);
                z += prob[a];
            }
            double acc = 0.0;
            const double r = (double)(raw(b) >> 11) * (1.0 / 9007199254740992.0);
            for (int a = 0; a < 9; a++) {
                acc += prob[a] / z;
                if (r < acc) return a;
            }
            return 8;
        }
        if (!greedy) {
            double u = (double)(raw(b) >> 11) * (1.0 / 9007199254740992.0);
            if (u < epsilon) return (int)(((raw(b) >> 32) * 9ull) >> 32);
        }
        return greedy_sample(b, qs);
    }
    int action(int b, const std::vector<std::vector<int>>& f, bool greedy = false) {  // Agent::action, agent.cpp:67-74
        double qs[9];
        if (P.algo == LOB_ALGO_DOUBLE_Q || P.algo == LOB_ALGO_DOUBLE_R_LEARN)  // DoubleAgent::action, agent.cpp:196-204
            for (int a = 0; a < 9; a++) qs[a] = (getQ(b, f, a) + getQ(b, f, a, true)) / 2.0f;
        else
            for (int a = 0; a < 9; a++) qs[a] = getQ(b, f, a);
        return policy_sample(b, qs, greedy);
    }
    void new_state(int b) {  // State::newState(env), state.cpp:35-43
        vars[b].clear();
        env[b]->getState(vars[b]);
        populate_features(M, 32, 9, vars[b], feats[b]);
    }
    void record(int b, int action, double reward, double td) {
        oracle_step_rec& r = recs[b];
        memset(&r, 0, sizeof r);
        r.action = action;
        r.reward = reward;
        r.td = td;
        r.n_vars = (int)vars[b].size();
        for (size_t i = 0; i < vars[b].size(); i++) r.vars[i] = vars[b][i];
        r.rng_ctr = rng_ctr[b];
        env[b]->fill(r.book);
        r.book.n_traces = (int)traces[b].nonzero.size();
    }
};

extern "C" {

oracle_learner* oracle_create(const lob_params* p, int32_t n_books, const uint32_t* records, int32_t n_events) {
    if (p->n_tilings != 32 || p->n_actions != 9) return nullptr;
    oracle_learner* o = new oracle_learner();
    o->P = *p;
    o->B = n_books;
    o->M = p->memory_size;
    const int W = lob_rec_words(p->depth, p->max_trades);
    for (int b = 0; b < n_books; b++)
        o->env.emplace_back(new Env(*p, records + (size_t)b * n_events * W, n_events));
    int nt = p->theta_mode == LOB_THETA_PRIVATE ? n_books : 1;
    o->theta.assign(nt, std::vector<double>((size_t)o->M, 0.0));
    if (p->algo == LOB_ALGO_DOUBLE_Q || p->algo == LOB_ALGO_DOUBLE_R_LEARN) {
        o->theta_b.assign(nt, std::vector<double>((size_t)o->M, 0.0));
        for (int b = 0; b < n_books; b++) {
            // gen(c["debug"]["random_seed"].as<unsigned>()), one agent per book: seed + global book id
            o->agent_gen.emplace_back((unsigned)(p->seed + p->book_id_offset + (uint64_t)b));
            o->agent_unif.emplace_back(0.0, 1.0);
        }
    }
    if (p->random_init) {
        // learning.random_init (src/rl/agent.cpp:37-39; DoubleAgent::DoubleAgent, agent.cpp:190-192):
        //   generate(&theta[0], &theta[MEMORY_SIZE], [this]() { return 2.0*unif_dist(gen)-1.0; });
        // theta first, theta_b after it, from the agent's own generator, which then goes on to toss DoubleQLearn's coin.
        // Private theta: every book's agent draws its own vectors.  Shared theta (no counterpart in the reference beyond one
        // book): the one vector is global book 0's agent's -- the same on every shard --, and that agent's generator moves on
        // where the book lives (book_id_offset == 0).
        const bool dq = !o->theta_b.empty();
        for (int t = 0; t < nt; t++) {
            const bool own = p->theta_mode == LOB_THETA_PRIVATE || p->book_id_offset == 0;
            std::mt19937_64 local((unsigned)(p->seed + (p->theta_mode == LOB_THETA_PRIVATE ? p->book_id_offset + (uint64_t)t : 0ull)));
            std::mt19937_64& gen = dq && own ? o->agent_gen[t] : local;
            std::uniform_real_distribution<double> unif_dist(0.0, 1.0);
            std::generate(o->theta[t].begin(), o->theta[t].end(), [&]() { return 2.0 * unif_dist(gen) - 1.0; });
            if (dq) std::generate(o->theta_b[t].begin(), o->theta_b[t].end(), [&]() { return 2.0 * unif_dist(gen) - 1.0; });
        }
    }
    o->traces.resize(n_books);
    o->vars.resize(n_books);
    o->last_vars.resize(n_books);
    o->feats.assign(n_books, std::vector<std::vector<int>>(9, std::vector<int>(96, 0)));
    o->last_feats = o->feats;
    o->rng_ctr.assign(n_books, 0);
    o->done.assign(n_books, 0);
    o->recs.resize(n_books);
    o->pend_a.assign(n_books, 0); o->pend_reward.assign(n_books, 0.0); o->pend_live.assign(n_books, 0);
    o->alpha = p->alpha;
    o->epsilon = p->epsilon;
    o->tau = p->tau;
    o->rho.assign(nt, 0.0);
    return o;
}
void oracle_destroy(oracle_learner* o) { delete o; }
void oracle_debug_pass_stats(long long* out16, int reset) {
    for (int i = 0; i < 16; i++) { out16[i] = g_pass_stats[i]; if (reset) g_pass_stats[i] = 0; }
}

int oracle_reset(oracle_learner* o) {
    o->have_from = false;  // a step abandoned after its first half (no oracle_td_step_end) ends with its episode: lob_reset does the same
    for (int b = 0; b < o->B; b++) {
        bool ok = o->env[b]->Initialise();
        o->done[b] = ok ? 0 : 2;
        // last_state->newState(environment), serial.cpp:25: the object `last_state` points
        // to is overwritten, the one `state` points to keeps whatever it held.  NB:
        // Learner::_step starts with swap(state, last_state) (serial.cpp:55), so the
        // state just extracted becomes `state` and the first action / first TD update of
        // an episode use the OTHER State object: all-zero features on the first episode
        // (State ctor, src/rl/state.cpp:10-19), the previous episode's final state
        // afterwards.  Reproduced as is.
        o->vars[b].swap(o->last_vars[b]);
        o->feats[b].swap(o->last_feats[b]);
        o->new_state(b);             // fills vars/feats = the object last_state points to
        o->record(b, -1, 0.0, 0.0);
        o->vars[b].swap(o->last_vars[b]);
        o->feats[b].swap(o->last_feats[b]);
    }
    return 0;
}

static int oracle_td_step_impl(oracle_learner* o, int32_t n_steps, int half) {
    const float rate = (float)(o->P.gamma * o->P.lambda);
    for (int s = 0; s < n_steps; s++) {
        std::vector<double>& upd = o->st_upd;
        std::vector<char>& has = o->st_has;  // 1: update theta, 2: update theta_b
        if (half != 2) { upd.assign(o->B, 0.0); has.assign(o->B, 0); }
        // R-learning (agent.cpp:357-412): what the rho update after updateQ still needs of the step
        const bool r_learn = o->P.algo == LOB_ALGO_R_LEARN || o->P.algo == LOB_ALGO_ONLINE_R_LEARN || o->P.algo == LOB_ALGO_DOUBLE_R_LEARN;
        std::vector<double>&rl_q = o->st_rl_q, &rl_t = o->st_rl_t, &rl_r = o->st_rl_r;
        if (half != 2) { rl_q.assign(o->B, 0.0); rl_t.assign(o->B, 0.0); rl_r.assign(o->B, 0.0); }
        // read phase: the books are independent (every one reads theta_t / rho_t, draws from its own counter-based
        // stream, writes its own traces and record), so a test may spread them over host threads (ORACLE_THREADS=n;
        // the write phase below stays serial, in book order)
        // first half of a book's step: swap, isTerminal, action, performAction, newState; false: the book does not learn this step
        auto phase1 = [&](int b) -> bool {
            o->pend_live[b] = 0;
            if (o->done[b]) return false;
            Env& e = *o->env[b];
            // swap(state, last_state)
            o->vars[b].swap(o->last_vars[b]);
            o->feats[b].swap(o->last_feats[b]);
            if (e.isTerminal()) { o->done[b] = 1; return false; }
            int a = o->action(b, o->last_feats[b]);
            if (!e.performAction(a)) {
                o->done[b] = 2;
                o->recs[b].rng_ctr = o->rng_ctr[b];
                e.fill(o->recs[b].book);
                o->recs[b].book.n_traces = (int)o->traces[b].nonzero.size();
                return false;
            }
            o->new_state(b);
            o->pend_a[b] = a;
            o->pend_reward[b] = e.getReward();
            o->pend_live[b] = 1;
            return true;
        };
        // second half: HandleTransition.  Q(from_state, .) is what it was when the action was chosen (getQ_from: the weights
        // of the first half, when a weight exchange has come in between -- oracle_td_step_begin / _end).
        auto phase2 = [&](int b, int64_t* n_done) {
            Env& e = *o->env[b];
            (void)e;
            const int a = o->pend_a[b];
            const double reward = o->pend_reward[b];
            // HandleTransition: UpdateTraces, UpdateWeights (agent.cpp:86-115)
            double delta;
            int target = 1;
            if (o->P.algo == LOB_ALGO_DOUBLE_R_LEARN) {
                int amax = o->argmaxQ_from(b, o->last_feats[b]);  // DoubleRLearn::UpdateTraces, agent.cpp:422-430
                if (a != amax) o->traces[b].decay(0.0f);
                else o->traces[b].decay(rate);
                o->traces[b].update(o->last_feats[b], a, 9, 32);
                double Q, mQ;
                if (o->agent_unif[b](o->agent_gen[b]) > 0.5) {  // UPDATE(A), agent.cpp:436-443
                    Q = o->getQ_from(b, o->last_feats[b], a);
                    mQ = o->getQ(b, o->feats[b], o->argmaxQ(b, o->feats[b]), true);
                } else {  // UPDATE(B)
                    Q = o->getQ_from(b, o->last_feats[b], a, true);
                    mQ = o->getQ(b, o->feats[b], o->argmaxQ(b, o->feats[b], true));
                    target = 2;
                }
                delta = reward - o->rh(b) + mQ - Q;
                rl_q[b] = Q; rl_r[b] = reward;
            } else if (o->P.algo == LOB_ALGO_DOUBLE_Q) {
                int amax = o->argmaxQ_from(b, o->last_feats[b]);  // DoubleQLearn::UpdateTraces, agent.cpp:319-327
                if (a != amax) o->traces[b].decay(0.0f);
                else o->traces[b].decay(rate);
                o->traces[b].update(o->last_feats[b], a, 9, 32);
                double F_term = o->P.gamma * 0.0 - 0.0;
                if (o->agent_unif[b](o->agent_gen[b]) > 0.5) {  // UPDATE(A), agent.cpp:334-342
                    double Qa = o->getQ_from(b, o->last_feats[b], a);
                    delta = reward + F_term + o->P.gamma * o->getQ(b, o->feats[b], o->argmaxQ(b, o->feats[b]), true) - Qa;
                } else {  // UPDATE(B)
                    double Qb = o->getQ_from(b, o->last_feats[b], a, true);
                    delta = reward + F_term + o->P.gamma * o->getQ(b, o->feats[b], o->argmaxQ(b, o->feats[b], true)) - Qb;
                    target = 2;
                }
            } else if (o->P.algo == LOB_ALGO_R_LEARN) {
                int amax = o->argmaxQ_from(b, o->last_feats[b]);  // RLearn::UpdateTraces, agent.cpp:363-371
                if (a != amax) o->traces[b].decay(0.0f);
                else o->traces[b].decay(rate);
                o->traces[b].update(o->last_feats[b], a, 9, 32);
                double Q = o->getQ_from(b, o->last_feats[b], a);  // RLearn::UpdateWeights, agent.cpp:373-380
                double mQ = o->getQ(b, o->feats[b], o->argmaxQ(b, o->feats[b]));
                delta = reward - o->rh(b) + mQ - Q;
                rl_q[b] = Q; rl_t[b] = mQ; rl_r[b] = reward;
            } else if (o->P.algo == LOB_ALGO_ONLINE_R_LEARN) {
                o->traces[b].decay(rate);  // Agent::UpdateTraces, agent.cpp:111-115
                o->traces[b].update(o->last_feats[b], a, 9, 32);
                double Q = o->getQ_from(b, o->last_feats[b], a);  // OnlineRLearn::UpdateWeights, agent.cpp:398-405
                int a2 = o->action(b, o->feats[b]);
                double gQ = o->getQ(b, o->feats[b], a2);
                delta = reward - o->rh(b) + gQ - Q;
                rl_q[b] = Q; rl_t[b] = gQ; rl_r[b] = reward;
            } else if (o->P.algo == LOB_ALGO_QLAMBDA) {
                int amax = o->argmaxQ_from(b, o->last_feats[b]);  // QLearn::UpdateTraces, agent.cpp:272-280
                if (a != amax) o->traces[b].decay(0.0f);
                else o->traces[b].decay(rate);
                o->traces[b].update(o->last_feats[b], a, 9, 32);
                double Q = o->getQ_from(b, o->last_feats[b], a);
                double F_term = o->P.gamma * 0.0 - 0.0;
                int am2 = o->argmaxQ(b, o->feats[b]);  // maxQ(to_state)
                double mq = o->getQ(b, o->feats[b], am2);
                delta = reward + F_term + o->P.gamma * mq - Q;
            } else {
                o->traces[b].decay(rate);  // Agent::UpdateTraces, agent.cpp:111-115
                o->traces[b].update(o->last_feats[b], a, 9, 32);
                double Q1 = o->getQ_from(b, o->last_feats[b], a);
                int a2 = o->action(b, o->feats[b]);
                double Q2 = o->getQ(b, o->feats[b], a2);
                double F = o->P.gamma * 0.0 - 0.0;
                delta = reward + F + o->P.gamma * Q2 - Q1;
            }
            upd[b] = o->alpha * delta;
            has[b] = (char)target;
            ++*n_done;
            o->record(b, a, reward, delta);
        };
        auto read_phase = [&](int b_lo, int b_hi, int64_t* n_done) {
            for (int b = b_lo; b < b_hi; b++) {
                if (half != 2) { if (!phase1(b)) continue; }
                else if (!o->pend_live[b]) continue;
                if (half != 1) phase2(b, n_done);
            }
        };
        {
            static const int n_thr_env = getenv("ORACLE_THREADS") ? atoi(getenv("ORACLE_THREADS")) : 1;
            const int n_thr = std::max(1, std::min(n_thr_env, o->B / 64));
            std::vector<int64_t> done_cnt(n_thr, 0);
            if (n_thr == 1) read_phase(0, o->B, &done_cnt[0]);
            else {
                std::vector<std::thread> pool;
                for (int t = 0; t < n_thr; t++)
                    pool.emplace_back(read_phase, (int)((int64_t)o->B * t / n_thr), (int)((int64_t)o->B * (t + 1) / n_thr), &done_cnt[t]);
                for (auto& th : pool) th.join();
            }
            for (int64_t c : done_cnt) o->n_steps_done += c;
        }
        if (half == 1) {  // the weights Q(from_state, .) was evaluated under
            o->theta_from = o->theta;
            o->theta_b_from = o->theta_b;
            o->have_from = true;
            continue;
        }
        // model_log (Agent::HandleTransition, agent.cpp:93-100): _agg_delta += abs(delta); every 1000 updates a row _agg_delta /
        // 1000 and both start again.  The batch adds a step's |delta| in book order and writes a row once the aggregate holds 1000
        // updates or more (one book: the count reaches 1000 one update at a time -- the reference's rows exactly)
        {
            double s = 0.0;
            int64_t n = 0;
            for (int b = 0; b < o->B; b++)
                if (has[b]) { s += std::fabs(o->recs[b].td); n++; }
            o->ml_agg += s;
            o->ml_cnt += n;
            if (o->ml_cnt >= 1000) {
                o->ml_rows.push_back(o->ml_agg / (double)o->ml_cnt);
                o->ml_agg = 0.0;
                o->ml_cnt = 0;
            }
        }
        // write phase: updateQ (agent.cpp:137-142) for every book, book order
        for (int b = 0; b < o->B; b++) {
            if (!has[b]) continue;
            double scaled = upd[b] / 32;
            double* t = has[b] == 2 ? o->thb(b) : o->th(b);
            for (int f : o->traces[b].nonzero) t[f] += scaled * o->traces[b].get(f);
            o->n_updates++;
        }
        // R-learning: nQ = Q + update; if (nQ - maxQ(from_state) < 1e-7) rho += beta * (reward - rho + target - nQ)
        // (agent.cpp:382-385, 407-410) -- maxQ under the weights AFTER updateQ, its argmax draws last in the book's
        // stream; every book reads rho_t, the increments are summed (the batch semantic of theta)
        if (r_learn) {
            std::vector<double> inc(o->rho.size(), 0.0);
            std::vector<int> cnt(o->rho.size(), 0);  // rho moves by the MEAN of the step's increments
            for (int b = 0; b < o->B; b++) {
                if (!has[b]) continue;
                const double nQ = rl_q[b] + upd[b];
                if (o->P.algo == LOB_ALGO_DOUBLE_R_LEARN) {
                    // agent.cpp:453-464: mQ = max_i (getQ + getQb) / 2.0 of from_state (first maximum), and THAT mQ in the increment
                    double mQ = -DBL_MAX;
                    for (int i = 0; i < 9; i++) {
                        double val = (o->getQ(b, o->last_feats[b], i) + o->getQ(b, o->last_feats[b], i, true)) / 2.0;
                        if (val > mQ) mQ = val;
                    }
                    if (nQ - mQ < 1e-7) { inc[o->P.theta_mode == LOB_THETA_PRIVATE ? b : 0] += o->P.beta * (rl_r[b] - o->rh(b) + mQ - nQ); cnt[o->P.theta_mode == LOB_THETA_PRIVATE ? b : 0]++; }
                    continue;
                }
                const double mq_from = o->getQ(b, o->last_feats[b], o->argmaxQ(b, o->last_feats[b]));
                o->recs[b].rng_ctr = o->rng_ctr[b];
                if (nQ - mq_from < 1e-7) { inc[o->P.theta_mode == LOB_THETA_PRIVATE ? b : 0] += o->P.beta * (rl_r[b] - o->rh(b) + rl_t[b] - nQ); cnt[o->P.theta_mode == LOB_THETA_PRIVATE ? b : 0]++; }
            }
            for (size_t i = 0; i < inc.size(); i++) if (cnt[i] > 0) o->rho[i] += inc[i] / (double)cnt[i];
        }
    }
    return 0;
}

int oracle_td_step(oracle_learner* o, int32_t n_steps) { return oracle_td_step_impl(o, n_steps, 0); }
// the model_log rows so far (never cleared)
int32_t oracle_model_log(oracle_learner* o, double* rows, int32_t cap) {
    const int32_t n = (int32_t)std::min<size_t>(o->ml_rows.size(), (size_t)cap);
    for (int32_t i = 0; i < n; i++) rows[i] = o->ml_rows[i];
    return (int32_t)o->ml_rows.size();
}
// One step in two halves (lob_td_step_begin / lob_td_step_end): whatever changes the weights in between (a multi-GPU
// exchange) is seen by the second half's evaluations of the NEW state only.
int oracle_td_step_begin(oracle_learner* o) { return oracle_td_step_impl(o, 1, 1); }
int oracle_td_step_end(oracle_learner* o) {
    const int rc = oracle_td_step_impl(o, 1, 2);
    o->have_from = false;
    o->theta_from.clear(); o->theta_from.shrink_to_fit();
    o->theta_b_from.clear(); o->theta_b_from.shrink_to_fit();
    return rc;
}

int oracle_eval_step(oracle_learner* o, int32_t n_steps) {  // Backtester::_step, serial.cpp:124-137
    for (int s = 0; s < n_steps; s++)
        for (int b = 0; b < o->B; b++) {
            if (o->done[b]) continue;
            Env& e = *o->env[b];
            if (e.isTerminal()) { o->done[b] = 1; continue; }
            o->new_state(b);
            int a = o->action(b, o->feats[b], true);
            if (!e.performAction(a)) { o->done[b] = 2; e.fill(o->recs[b].book); continue; }
            o->n_steps_done++;
            o->record(b, a, e.getReward(), 0.0);
        }
    return 0;
}

int oracle_env_step(oracle_learner* o, const int32_t* actions) {
    for (int b = 0; b < o->B; b++) {
        if (o->done[b]) continue;
        Env& e = *o->env[b];
        if (e.isTerminal()) { o->done[b] = 1; continue; }
        if (!e.performAction(actions[b])) { o->done[b] = 2; e.fill(o->recs[b].book); continue; }
        o->new_state(b);
        o->n_steps_done++;
        o->record(b, actions[b], e.getReward(), 0.0);
    }
    return 0;
}

int oracle_clear_inventory(oracle_learner* o) {
    for (int b = 0; b < o->B; b++) {
        o->env[b]->ClearInventory();
        o->record(b, -2, 0.0, 0.0);
    }
    return 0;
}
int oracle_handle_terminal(oracle_learner* o) {
    for (int b = 0; b < o->B; b++) o->traces[b].decay(0.0f);
    return 0;
}
void oracle_set_alpha(oracle_learner* o, double a) { o->alpha = a; }
void oracle_set_epsilon(oracle_learner* o, double e) { o->epsilon = e; }
void oracle_set_tau(oracle_learner* o, double t) { o->tau = t; }
int oracle_get_rho(oracle_learner* o, double* out, int32_t n) {
    if ((size_t)n > o->rho.size()) return -1;
    for (int i = 0; i < n; i++) out[i] = o->rho[i];
    return 0;
}
void oracle_get_rec(oracle_learner* o, int32_t book, oracle_step_rec* out) { *out = o->recs[book]; }
double* oracle_theta(oracle_learner* o, int32_t which) { return o->theta[which].data(); }
double* oracle_theta_b(oracle_learner* o, int32_t which) { return o->theta_b.empty() ? nullptr : o->theta_b[which].data(); }
int32_t oracle_get_traces(oracle_learner* o, int32_t b, int32_t* idx, float* e, int32_t cap) {
    int n = (int)o->traces[b].nonzero.size();
    for (int i = 0; i < n && i < cap; i++) {
        idx[i] = o->traces[b].nonzero[i];
        e[i] = o->traces[b].get(idx[i]);
    }
    return n;
}
void oracle_get_counters(oracle_learner* o, int64_t out[4]) {
    int64_t ev = 0, live = 0;
    for (int b = 0; b < o->B; b++) { ev += o->env[b]->events_consumed; live += o->done[b] == 0; }
    out[0] = o->n_steps_done; out[1] = ev; out[2] = live; out[3] = o->n_updates;
}

// ---- unit-level -----------------------------------------------------------
void oracle_tiles(int64_t M, const float* vars, int32_t n_vars, int32_t n, int32_t* out) {
    std::vector<std::vector<int>> f(9, std::vector<int>(96, 0));
    for (int i = 0; i < n; i++) {
        std::vector<float> v(vars + (size_t)i * n_vars, vars + (size_t)(i + 1) * n_vars);
        populate_features(M, 32, 9, v, f);
        for (int a = 0; a < 9; a++) memcpy(out + ((size_t)i * 9 + a) * 96, f[a].data(), 96 * 4);
    }
}
int32_t oracle_hash_unh(const int32_t* ints, int32_t n, int64_t m, int32_t inc) { return hash_UNH(ints, n, m, inc); }
void oracle_rndseq(uint32_t* out) { memcpy(out, rndseq().t, sizeof(uint32_t) * 2048); }
int32_t oracle_to_ticks(const lob_market* m, double price) { return Market(*m).ToTicks(price); }
double oracle_to_price(const lob_market* m, int32_t ticks) { return Market(*m).ToPrice(ticks); }
double oracle_tick_size(const lob_market* m, double price) { return Market(*m).tick_size(price); }

void oracle_order_script(double price, int64_t size, int64_t q_head, const int64_t* ops, int32_t n_ops, int64_t* out) {
    Order o(price, size, q_head);
    for (int i = 0; i < n_ops; i++) {
        long ret = 0;
        switch (ops[2 * i]) {
            case 0: ret = o.doTransaction(ops[2 * i + 1]); break;
            case 1: o.doCancellation(ops[2 * i + 1]); break;
            case 2: o.addVolumeBehind(ops[2 * i + 1]); break;
            case 3: o.clearQueues(); break;
        }
        out[4 * i + 0] = o.q_head;
        out[4 * i + 1] = o.q_tail;
        out[4 * i + 2] = o.remaining();
        out[4 * i + 3] = ret;
    }
}
void oracle_rolling_mean(int32_t window, const double* vals, int32_t n, double* out) {
    RollingMean r(window);
    for (int i = 0; i < n; i++) {
        r.push(vals[i]);
        out[5 * i + 0] = r.mean();
        out[5 * i + 1] = r.var();
        out[5 * i + 2] = r.std();
        out[5 * i + 3] = r.sum();
        out[5 * i + 4] = r.full() ? 1.0 : 0.0;
    }
}

// Book script interpreter for the known answers of test/test_Book.cpp.
// script words (doubles): opcode, args...
//   1 side D p[D] v[D] nT (tp tv)*nT : ApplyChanges(side)        -> (none)
//   2 side                           : StashState(side)          -> (none)
//   3 side price size                : PlaceOrder                -> placed(0/1)
//   4 side nT (tp tv)*nT ref         : ApplyTransactions         -> volume, proxy, value
//   5 side ref size                  : WalkTheBook               -> executed, proxy, value
//   6 side price                     : queue_ahead, queue_behind, remaining (-1 if none)
//   7 side level                     : price(level), volume(price(level))
//   8                                : HandleAdverseSelection    -> volume, proxy, value
//   9 side price                     : CancelOrder               -> (none)
//  10 side                           : observed_value, observed_volume, n_transacted, order_count
//  11 side price                     : price_level(price)
//  12 side level                     : last_price(level), last_volume(last_price(level))
// side: 0 ask, 1 bid.  Returns the number of doubles written, or -1 on throw.
int oracle_book_script(int32_t depth, const double* s, int32_t n, double* out, int32_t cap) {
    AskBook ask(depth);
    BidBook bid(depth);
    int i = 0, k = 0;
    auto put = [&](double v) { if (k < cap) out[k] = v; k++; };
    try {
        while (i < n) {
            int op = (int)s[i++];
            if (op == 8) {
                Fill f = HandleAdverseSelection(ask, bid);
                put((double)std::get<0>(f)); put(std::get<1>(f)); put(std::get<2>(f));
                continue;
            }
            int side = (int)s[i++];
            switch (op) {
                case 1: {
                    int D = (int)s[i++];
                    std::vector<double> p(s + i, s + i + D); i += D;
                    std::vector<long> v(D);
                    for (int l = 0; l < D; l++) v[l] = (long)s[i++];
                    int nT = (int)s[i++];
                    TradeMap tm;
                    for (int t = 0; t < nT; t++) { tm[s[i]] += (long)s[i + 1]; i += 2; }
                    if (side == 0) ask.ApplyChanges(p.data(), v.data(), tm);
                    else bid.ApplyChanges(p.data(), v.data(), tm);
                    break;
                }
                case 2: if (side == 0) ask.StashState(); else bid.StashState(); break;
                case 3: {
                    double p = s[i++]; long sz = (long)s[i++];
                    put(side == 0 ? ask.PlaceOrder(p, sz) : bid.PlaceOrder(p, sz));
                    break;
                }
                case 4: {
                    int nT = (int)s[i++];
                    TradeMap tm;
                    for (int t = 0; t < nT; t++) { tm[s[i]] += (long)s[i + 1]; i += 2; }
                    double ref = s[i++];
                    Fill f = side == 0 ? ask.ApplyTransactions(tm, ref) : bid.ApplyTransactions(tm, ref);
                    put((double)std::get<0>(f)); put(std::get<1>(f)); put(std::get<2>(f));
                    break;
                }
                case 5: {
                    double ref = s[i++]; long sz = (long)s[i++];
                    Fill f = side == 0 ? ask.WalkTheBook(ref, sz) : bid.WalkTheBook(ref, sz);
                    put((double)std::get<0>(f)); put(std::get<1>(f)); put(std::get<2>(f));
                    break;
                }
                case 6: {
                    double p = s[i++];
                    Order* o = nullptr;
                    if (side == 0) { auto it = ask.open_orders.find(p); if (it != ask.open_orders.end()) o = it->second.get(); }
                    else { auto it = bid.open_orders.find(p); if (it != bid.open_orders.end()) o = it->second.get(); }
                    put(o ? (double)o->q_head : -1); put(o ? (double)o->q_tail : -1); put(o ? (double)o->remaining() : -1);
                    break;
                }
                case 7: {
                    int l = (int)s[i++];
                    double p = side == 0 ? ask.price(l) : bid.price(l);
                    put(p); put((double)(side == 0 ? ask.volume(p) : bid.volume(p)));
                    break;
                }
                case 9: {
                    double p = s[i++];
                    if (side == 0) ask.open_orders.erase(p); else bid.open_orders.erase(p);
                    break;
                }
                case 10:
                    if (side == 0) { put(ask.observed_transaction_value_); put((double)ask.observed_transaction_volume_); put(ask.n_transacted_); put(ask.order_count()); }
                    else { put(bid.observed_transaction_value_); put((double)bid.observed_transaction_volume_); put(bid.n_transacted_); put(bid.order_count()); }
                    break;
                case 11: { double p = s[i++]; put(side == 0 ? ask.price_level(p) : bid.price_level(p)); break; }
                case 12: {
                    int l = (int)s[i++];
                    double p = side == 0 ? ask.last_price(l) : bid.last_price(l);
                    put(p); put((double)(side == 0 ? ask.last_volume(p) : bid.last_volume(p)));
                    break;
                }
                default: return -2;
            }
        }
    } catch (std::exception&) {
        return -1;
    }
    return k;
}

}  // extern "C"
