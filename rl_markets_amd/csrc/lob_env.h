// Device-side environment: one lane = one book.  Hand-written restatement of
// the reference's single-book step for the batched SoA layout of lob_state.h:
//   market::Book / AskBook / BidBook / BookUtils  (src/market/book.cpp)
//   market::Order                                  (src/market/order.cpp)
//   environment::RiskManager                       (src/environment/risk_manager.cpp)
//   environment::Base / Intraday                   (src/environment/base.cpp, intraday.cpp)
//   Accumulator / RollingMean / EWMA               (src/utilities/accumulators.cpp)
// Because the order limit is hard-wired to 1 per side (quirk Q13) the
// reference's order map degenerates to one optional order per side, and because
// the stream precondition is "levels sorted best->worst, unique 1e-4 keys"
// (lob_validate_stream) the level map degenerates to the sorted level arrays.
// All arithmetic that feeds book state is integer or IEEE f64 evaluated in the
// reference's order (compile with -ffp-contract=off).
#ifndef LOB_ENV_H
#define LOB_ENV_H

#include <hip/hip_runtime.h>

#include "lob_internal.h"
#include "lob_state.h"

struct EnvR {
#define X(t, n) t n;
    LOB_ENV_FIELDS(X)
#undef X
    // cached best prices of the current / stashed snapshot (0.0 = undefined)
    f64 ap0, bp0, lap0, lbp0;
};

struct EnvCtx {
    const DevParams& P;
    const DevState& S;
    int b;
    __device__ EnvCtx(const DevParams& p, const DevState& s, int book) : P(p), S(s), b(book) {}
    __device__ size_t lvl(int sel, int side, int l) const {
        return ((size_t)((sel * 2 + side) * P.D + l)) * (size_t)S.B + (size_t)b;
    }
    __device__ void err(int bit) const { atomicOr(S.error_flag, bit); }
};

__device__ inline f64 key4(f64 p) { return rint(p * 10000.0); }  // utilities/comparison.h:4-34

// double -> long with the x86 `cvttsd2si` result for out-of-range / NaN
// (quirk Q2: the reference's `long -= double` in Order::doCancellation).
__device__ inline i64 cvt_long_x86(f64 d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return (i64)0x8000000000000000ull;
    return (i64)d;
}

// ---- market::Order (src/market/order.cpp:36-107) ---------------------------
struct OrderR {
    i64 size, qh, qt, ex;
};
__device__ inline i64 ord_remaining(const OrderR& o) { i64 r = o.size - o.ex; return r > 0 ? r : 0; }
__device__ inline bool ord_executed(const OrderR& o) { return o.ex >= o.size; }
__device__ inline i64 ord_transaction(OrderR& o, i64 volume) {
    i64 remaining_volume = volume - o.qh;
    if (remaining_volume > 0) {
        o.qh = 0;
        if (ord_remaining(o) <= remaining_volume) {
            o.ex = o.size;
            remaining_volume -= o.size;
        } else {
            o.ex += remaining_volume;
            remaining_volume = 0;
        }
    } else {
        o.qh -= volume;
    }
    return remaining_volume > 0 ? remaining_volume : 0;
}
__device__ inline void ord_cancellation(OrderR& o, i64 volume) {
    if (o.qt == 0) {
        o.qh -= volume;
    } else {
        f64 total = (f64)(o.qh + o.qt);
        i64 nh = cvt_long_x86((f64)o.qh - ceil((f64)(volume * o.qh) / total));
        i64 nt = cvt_long_x86((f64)o.qt - floor((f64)(volume * o.qt) / total));
        o.qh = nh;
        o.qt = nt;
    }
    if (o.qh < 0) {
        o.qt = (i64)((u64)o.qt + (u64)o.qh);
        o.qh = 0;
    }
    if (o.qt < 0) o.qt = 0;
}

// ---- rolling windows (src/utilities/accumulators.cpp:17-131) ---------------
__device__ inline void rm_push(const RMPtrs& r, int B, int b, f64 val) {
    i32 cnt = r.cnt[b], head = r.head[b];
    f64 sum = r.sum[b], mean = r.mean[b], s = r.s[b];
    sum += val;
    if (cnt < r.w) {
        head = (cnt == 0) ? 0 : (head + 1 == r.w ? 0 : head + 1);
        cnt++;
        r.ring[(size_t)head * B + b] = val;
        f64 n = (f64)cnt;
        f64 old_mean = mean;
        mean += (val - mean) / n;
        s += (val - mean) * (val - old_mean);
    } else {
        // window full: push_front then pop_back of the oldest
        i32 oldest = head + 1 == r.w ? 0 : head + 1;
        f64 old = r.ring[(size_t)oldest * B + b];
        r.ring[(size_t)oldest * B + b] = val;
        head = oldest;
        f64 n = (f64)(cnt + 1);
        f64 old_mean = mean;
        mean += (val - mean) / n;
        s += (val - mean) * (val - old_mean);
        sum -= old;
        f64 n2 = (f64)cnt;
        f64 old_mean2 = mean;
        mean -= (old - mean) / n2;
        s -= (old - mean) * (old - old_mean2);
    }
    r.cnt[b] = cnt;
    r.head[b] = head;
    r.sum[b] = sum;
    r.mean[b] = mean;
    r.s[b] = s;
}
__device__ inline f64 rm_front(const RMPtrs& r, int B, int b) { return r.ring[(size_t)r.head[b] * B + b]; }
__device__ inline f64 rm_back(const RMPtrs& r, int B, int b) {
    i32 idx = r.head[b] - r.cnt[b] + 1;
    if (idx < 0) idx += r.w;
    return r.ring[(size_t)idx * B + b];
}
__device__ inline bool rm_full(const RMPtrs& r, int b) { return r.cnt[b] == r.w; }
__device__ inline f64 rm_std(const RMPtrs& r, int b) {
    f64 v = r.s[b] / (f64)(r.cnt[b] - 1);
    return v > 0 ? sqrt(v) : 0.0;
}
__device__ inline void acc_push(const AccPtrs& r, int B, int b, f64 val) {
    i32 cnt = r.cnt[b], head = r.head[b];
    f64 sum = r.sum[b];
    sum += val;
    if (cnt < r.w) {
        head = (cnt == 0) ? 0 : (head + 1 == r.w ? 0 : head + 1);
        cnt++;
        r.ring[(size_t)head * B + b] = val;
    } else {
        i32 oldest = head + 1 == r.w ? 0 : head + 1;
        f64 old = r.ring[(size_t)oldest * B + b];
        r.ring[(size_t)oldest * B + b] = val;
        head = oldest;
        sum -= old;
    }
    r.cnt[b] = cnt;
    r.head[b] = head;
    r.sum[b] = sum;
}

// ---- batched window updates --------------------------------------------------
// One market event pushes into ten windows.  Issuing every window's loads
// before the first store lets all of them be in flight together (the compiler
// must otherwise order each push's loads behind the previous push's stores):
// two memory round trips per event instead of ten-plus.
struct RMReg {
    i32 cnt, head, slot;
    f64 sum, mean, s, old;
};
__device__ inline void rm_load(const RMPtrs& r, int b, RMReg& g) {
    g.cnt = r.cnt[b]; g.head = r.head[b]; g.sum = r.sum[b]; g.mean = r.mean[b]; g.s = r.s[b];
}
__device__ inline void rm_prep(const RMPtrs& r, int B, int b, RMReg& g) {
    if (g.cnt < r.w) {
        g.slot = (g.cnt == 0) ? 0 : (g.head + 1 == r.w ? 0 : g.head + 1);
        g.old = 0.0;
    } else {
        g.slot = g.head + 1 == r.w ? 0 : g.head + 1;  // the oldest entry is replaced
        g.old = r.ring[(size_t)g.slot * B + b];
    }
}
__device__ inline void rm_apply(const RMPtrs& r, int B, int b, RMReg& g, f64 val) {
    g.sum += val;
    r.ring[(size_t)g.slot * B + b] = val;
    if (g.cnt < r.w) {
        g.cnt++;
        f64 n = (f64)g.cnt;
        f64 old_mean = g.mean;
        g.mean += (val - g.mean) / n;
        g.s += (val - g.mean) * (val - old_mean);
    } else {
        f64 n = (f64)(g.cnt + 1);
        f64 old_mean = g.mean;
        g.mean += (val - g.mean) / n;
        g.s += (val - g.mean) * (val - old_mean);
        g.sum -= g.old;
        f64 n2 = (f64)g.cnt;
        f64 old_mean2 = g.mean;
        g.mean -= (g.old - g.mean) / n2;
        g.s -= (g.old - g.mean) * (g.old - old_mean2);
    }
    g.head = g.slot;
    r.cnt[b] = g.cnt; r.head[b] = g.head; r.sum[b] = g.sum; r.mean[b] = g.mean; r.s[b] = g.s;
}
struct AccReg {
    i32 cnt, head, slot;
    f64 sum, old;
};
__device__ inline void acc_load(const AccPtrs& r, int b, AccReg& g) { g.cnt = r.cnt[b]; g.head = r.head[b]; g.sum = r.sum[b]; }
__device__ inline void acc_prep(const AccPtrs& r, int B, int b, AccReg& g) {
    if (g.cnt < r.w) { g.slot = (g.cnt == 0) ? 0 : (g.head + 1 == r.w ? 0 : g.head + 1); g.old = 0.0; }
    else { g.slot = g.head + 1 == r.w ? 0 : g.head + 1; g.old = r.ring[(size_t)g.slot * B + b]; }
}
__device__ inline void acc_apply(const AccPtrs& r, int B, int b, AccReg& g, f64 val) {
    g.sum += val;
    r.ring[(size_t)g.slot * B + b] = val;
    if (g.cnt < r.w) g.cnt++;
    else g.sum -= g.old;
    g.head = g.slot;
    r.cnt[b] = g.cnt; r.head[b] = g.head; r.sum[b] = g.sum;
}

// ---- measures (include/market/measures.h:9-55) ------------------------------
__device__ inline f64 e_mid(const EnvCtx& c, const EnvR& e) {
    if (e.ap0 == 0.0 || e.bp0 == 0.0) c.err(LOB_ERR_UNDEF_PRICE);
    return (e.ap0 + e.bp0) / 2.0;
}
__device__ inline f64 e_last_mid(const EnvCtx& c, const EnvR& e) {
    if (e.lap0 == 0.0 || e.lbp0 == 0.0) c.err(LOB_ERR_UNDEF_PRICE);
    return (e.lap0 + e.lbp0) / 2.0;
}
__device__ inline f64 e_micro(const EnvR& e) {
    f64 div = (f64)(e.a_tv + e.b_tv);
    f64 mpm_a = (f64)e.a_tv * e.bp0, mpm_b = e.ap0 * (f64)e.b_tv;
    return (mpm_a + mpm_b) / div;
}

// ---- Book lookups over the sorted level arrays ------------------------------
__device__ inline i64 book_volume(const EnvCtx& c, int sel, int side, f64 price) {
    const f64 k = key4(price);
    i64 v = 0;
    for (int l = 0; l < c.P.D; l++) {
        f32 p = c.S.px[c.lvl(sel, side, l)];
        if (p != 0.0f && key4((f64)p) == k) v = (i64)c.S.vol[c.lvl(sel, side, l)];
    }
    return v;
}

// RiskManager::CheckOrders (src/environment/risk_manager.cpp:26-32)
__device__ inline void check_orders(const DevParams& P, EnvR& e) {
    if (e.position >= P.pos_ub) e.b_on = 0;
    else if (e.position <= P.pos_lb) e.a_on = 0;
}

// RiskManager::PlaceOrder with ORDER_LIMIT == 1 (risk_manager.cpp:61-99) +
// Book::PlaceOrder (book.cpp:250-261): cancel whatever rests, place a new
// order queued behind the displayed volume at that price.
__device__ inline void place_one(const EnvCtx& c, EnvR& e, int side, f64 price) {
    if (!(price > 0.0)) c.err(LOB_ERR_BAD_ORDER_PRICE);
    i64 qh = book_volume(c, e.sel, side, price);
    if (side == 0) {
        e.a_on = 1; e.a_opx = price; e.a_osz = c.P.order_size; e.a_oqh = qh; e.a_oqt = 0; e.a_oex = 0; e.a_oiq = qh;
    } else {
        e.b_on = 1; e.b_opx = price; e.b_osz = c.P.order_size; e.b_oqh = qh; e.b_oqt = 0; e.b_oex = 0; e.b_oiq = qh;
    }
}

// Intraday::_place_orders + l2p_ (src/environment/intraday.cpp:64-82,164-173)
__device__ inline void place_orders(const EnvCtx& c, EnvR& e, int al, int bl) {
    const DevParams& P = c.P;
    e.ask_level = al;
    e.bid_level = bl;
    if (P.quote_mode == LOB_QUOTE_BOOK) {
        if (e.ap0 == 0.0 || e.bp0 == 0.0) c.err(LOB_ERR_UNDEF_PRICE);
        e.ask_quote = lobh::to_price_t(P_tick(P), lobh::to_ticks_t(P_tick(P), e.ap0) + al);
        e.bid_quote = lobh::to_price_t(P_tick(P), lobh::to_ticks_t(P_tick(P), e.bp0) - bl);
    } else {
        f64 tp = e.tp_val;
        f64 half = c.S.spread_window.mean[c.b] / 2.0;
        f64 half_spd = 0.0 > half ? 0.0 : half;  // std::max(0.0, x)
        e.ask_quote = lobh::to_price_t(P_tick(P), lobh::to_ticks_t(P_tick(P), tp + (f64)al * half_spd));
        e.bid_quote = lobh::to_price_t(P_tick(P), lobh::to_ticks_t(P_tick(P), tp - (f64)bl * half_spd));
    }
    place_one(c, e, 0, e.ask_quote);
    place_one(c, e, 1, e.bid_quote);
}

// AskBook/BidBook::WalkTheBook via BookUtils::MarketOrder (book.cpp:431-456,514-539,595-610)
__device__ inline void market_order(const EnvCtx& c, EnvR& e, i64 size, i64& out_vol, f64& out_proxy, f64& out_value) {
    out_vol = 0; out_proxy = 0.0; out_value = 0.0;
    f64 mip = e_mid(c, e);
    if (size == 0) return;
    const int side = size > 0 ? 0 : 1;
    i64 abs_size = size < 0 ? -size : size;
    i64 tv = side == 0 ? e.a_tv : e.b_tv;  // cumulative (quirk Q1)
    if (abs_size > tv) return;
    i64 executed = 0;
    f64 proxy = 0.0, value = 0.0;
    for (int l = 0; l < c.P.D; l++) {
        f32 pf = c.S.px[c.lvl(e.sel, side, l)];
        if (pf == 0.0f) continue;  // empty map
        f64 p = (f64)pf;
        i64 lvol = (i64)c.S.vol[c.lvl(e.sel, side, l)];
        i64 left = abs_size - executed;
        i64 l_ex = lvol < left ? lvol : left;
        executed += l_ex;
        proxy -= (f64)l_ex * fabs(p - mip);
        if (side == 0) value -= (f64)l_ex * p;
        else value += (f64)l_ex * p;
        if (executed >= abs_size) {
            if (side == 0) e.a_ntr++; else e.b_ntr++;
            break;
        }
    }
    out_vol = side == 0 ? executed : -executed;
    out_proxy = proxy;
    out_value = value;
}

// Base::ClearInventory (base.cpp:339-349) + RiskManager::ClearInventory/MarketOrder
__device__ inline void clear_inventory(const EnvCtx& c, EnvR& e) {
    i64 v; f64 proxy, value;
    market_order(c, e, -e.position, v, proxy, value);
    e.position += v;
    e.pnl_step += proxy;
    e.lo_vol_step += (i32)(v < 0 ? -v : v);
    e.ep_pnl += value;
    if (v > 0) e.market_buys++;
    else if (v < 0) e.market_sells++;
}

// Intraday::DoAction (intraday.cpp:176-220)
__device__ inline void do_action(const EnvCtx& c, EnvR& e, int action) {
    switch (action) {
        case 0: place_orders(c, e, 1, 1); break;
        case 1: clear_inventory(c, e); place_orders(c, e, e.ask_level, e.bid_level); break;
        case 2: place_orders(c, e, 2, 2); break;
        case 3: place_orders(c, e, 3, 3); break;
        case 4: place_orders(c, e, 0, 2); break;
        case 5: place_orders(c, e, 2, 0); break;
        case 6: place_orders(c, e, 1, 4); break;
        case 7: place_orders(c, e, 4, 1); break;
        case 8: place_orders(c, e, 5, 5); break;
        default: break;
    }
}

__device__ inline bool is_open(const DevParams& P, i32 t) {  // Market::IsOpen, market.cpp:67-70
    return ((i64)t > P.open_ms + 30 * 60000LL) && ((i64)t < P.close_ms - 30 * 60000LL);
}

// Base::getReward (base.cpp:166-237)
__device__ inline f64 get_reward(const EnvCtx& c, const EnvR& e) {
    const DevParams& P = c.P;
    f64 r = 0.0;
    i64 ap = e.position < 0 ? -e.position : e.position;
    i32 abs_pos = (i32)ap;
    switch (P.reward_measure) {
        case LOB_REWARD_NONE: break;
        case LOB_REWARD_PNL: r = e.pnl_step; break;
        case LOB_REWARD_PNL_DAMPED: {
            f64 m = 0.0 > e.momentum_pnl_step ? 0.0 : e.momentum_pnl_step;
            r = e.pnl_step - (f64)P.damping_factor * m;
            break;
        }
        case LOB_REWARD_SPREAD: r = e.pnl_step / c.S.spread_window.mean[c.b]; break;
        case LOB_REWARD_LOVOL: r = (f64)e.lo_vol_step; break;
        case LOB_REWARD_MM_LINEAR: {
            f32 pen = -P.pos_weight * (f32)abs_pos;  // float product in the reference
            r = (f64)pen;
            r += (f64)P.pnl_weight * e.pnl_step;
            break;
        }
        case LOB_REWARD_MM_DIV:
            if (e.pnl_step > 0) {
                f64 d = 1.0 > (f64)abs_pos ? 1.0 : (f64)abs_pos;
                r = e.pnl_step / d;
            } else r = e.pnl_step;
            break;
        case LOB_REWARD_NORMED: {
            const RMPtrs &u_ = c.S.pnl_ups, &d_ = c.S.pnl_downs;
            if (!(rm_full(u_, c.b) && rm_full(d_, c.b))) r = 0.0;
            else {
                f64 u = u_.mean[c.b], d = d_.mean[c.b], su = rm_std(u_, c.b), sd = rm_std(d_, c.b);
                f64 numer = (u * sd - d * su), denom = (su + sd);
                if (isnan(numer) || isinf(numer)) numer = 0.0;
                if (isnan(denom) || isinf(denom)) denom = 0.0;
                r = (fabs(denom) < 1e-5) ? numer : (numer / denom);
            }
            break;
        }
        default: break;  // mm_exp: not on the north-star path
    }
    return r * 100;
}

// Book::ApplyChanges + UpdateOrder for one side (book.cpp:64-141).
// `rec_px/rec_vol`: the new snapshot of this side; tp/tv: the event's trades.
__device__ inline void apply_changes(const EnvCtx& c, EnvR& e, int side, const uint32_t* rec_px,
                                     const uint32_t* rec_vol, const f64* tp, const i64* tv) {
    const DevParams& P = c.P;
    i64 tot = side == 0 ? e.a_tv : e.b_tv;
    if (side == 0) e.a_ltv = tot; else e.b_ltv = tot;
    for (int l = 0; l < P.D; l++) {
        f32 p = __uint_as_float(rec_px[l]);
        i32 v = (i32)rec_vol[l];
        if (!(p > 0.0f) || v <= 0) c.err(LOB_ERR_BAD_LEVEL);
        c.S.px[c.lvl(e.sel, side, l)] = p;
        c.S.vol[c.lvl(e.sel, side, l)] = v;
        tot += (i64)v;
    }
    if (side == 0) { e.a_tv = tot; e.ap0 = (f64)__uint_as_float(rec_px[0]); }
    else { e.b_tv = tot; e.bp0 = (f64)__uint_as_float(rec_px[0]); }

    i32 on = side == 0 ? e.a_on : e.b_on;
    if (!on) return;
    OrderR o;
    f64 opx;
    if (side == 0) { o.size = e.a_osz; o.qh = e.a_oqh; o.qt = e.a_oqt; o.ex = e.a_oex; opx = e.a_opx; }
    else { o.size = e.b_osz; o.qh = e.b_oqh; o.qt = e.b_oqt; o.ex = e.b_oex; opx = e.b_opx; }
    if (ord_executed(o)) {
        if (side == 0) e.a_on = 0; else e.b_on = 0;
        return;
    }
    i64 lv = book_volume(c, e.sel ^ 1, side, opx);
    if (lv == 0) return;
    i64 v = 0;  // volume(price) in the snapshot just applied: scan the record itself
    {
        const f64 k = key4(opx);
        for (int l = 0; l < P.D; l++) {
            f32 p = __uint_as_float(rec_px[l]);
            if (p != 0.0f && key4((f64)p) == k) v = (i64)(i32)rec_vol[l];
        }
    }
    if (v == 0) {
        o.qh = 0; o.qt = 0;
    } else {
        i64 vol_diff = lv - v;
        if (vol_diff >= 0) {
            i64 trade_vol = 0;
            const f64 k = key4(opx);
            for (int i = 0; i < P.T; i++)
                if (tv[i] > 0 && key4(tp[i]) == k) trade_vol = tv[i];
            i64 cancelled = vol_diff - trade_vol;
            if (cancelled > 0) ord_cancellation(o, cancelled);
        } else {
            o.qt += vol_diff;  // quirk Q2: addVolumeBehind(negative)
        }
    }
    if (side == 0) { e.a_oqh = o.qh; e.a_oqt = o.qt; }
    else { e.b_oqh = o.qh; e.b_oqt = o.qt; }
}

// BookUtils::IsValidState (book.cpp:612-625)
__device__ inline bool is_valid_state(const EnvCtx& c, const EnvR& e) {
    f64 mp = e_mid(c, e);
    bool has_a = key4(e.lap0) != key4(0.0), has_b = key4(e.lbp0) != key4(0.0);
    if (has_a && has_b) {
        f64 lm = (e.lap0 + e.lbp0) / 2.0;
        return ((e.ap0 - e.bp0) >= 0.0) && (mp > 0.0) && (fabs(mp - lm) < mp);
    }
    return true;
}

__device__ inline const uint32_t* rec_row(const EnvCtx& c, int i) {
    return c.S.records + ((size_t)c.b * (size_t)c.S.n_events + (size_t)i) * (size_t)c.P.W;
}

// Intraday::UpdateBookProfiles (intraday.cpp:275-313).  Returns false when
// the depth stream has no row after the one being made current (the
// reference's Streamer::LoadNext fails, src/data/streamer.cpp:42-49).
__device__ inline bool update_book_profiles(const EnvCtx& c, EnvR& e, const f64* tp, const i64* tv) {
    const DevParams& P = c.P;
    // StashState on both books: parity flip
    e.sel ^= 1;
    { f64 t = e.ap0; e.ap0 = e.lap0; e.lap0 = t; }
    { f64 t = e.bp0; e.bp0 = e.lbp0; e.lbp0 = t; }
    while (true) {
        if (e.cursor + 1 >= c.S.n_events) { e.done = 2; return false; }
        const uint32_t* r = rec_row(c, e.cursor);
        e.cursor++;
        e.events++;
        e.time_ms = (i32)r[LOB_REC_TIME];
        apply_changes(c, e, 0, r + lob_rec_ask_px(P.D, P.T), r + lob_rec_ask_vol(P.D, P.T), tp, tv);
        apply_changes(c, e, 1, r + lob_rec_bid_px(P.D, P.T), r + lob_rec_bid_vol(P.D, P.T), tp, tv);
        if ((i32)rec_row(c, e.cursor)[LOB_REC_TIME] == e.time_ms) continue;  // !WillTimeChange()
        if (is_valid_state(c, e)) break;
    }
    return true;
}

// Intraday::NextState (intraday.cpp:225-272): one market event.
__device__ inline bool next_state(const EnvCtx& c, EnvR& e) {
    const DevParams& P = c.P;
    const DevState& S = c.S;
    const int b = c.b, B = S.B;
    if (e.cursor >= S.n_events) { e.done = 2; return false; }
    f64 tp[LOB_MAX_TRADES];
    i64 tv[LOB_MAX_TRADES];
    {
        const uint32_t* r = rec_row(c, e.cursor);
#pragma unroll
        for (int i = 0; i < LOB_MAX_TRADES; i++) {
            if (i < P.T) {
                f32 p = __uint_as_float(r[lob_rec_trade_px(P.D, P.T) + i]);
                i32 v = (i32)r[lob_rec_trade_vol(P.D, P.T) + i];
                bool ok = (p > 0.0f) && (v > 0);
                tp[i] = ok ? (f64)p : 0.0;
                tv[i] = ok ? (i64)v : 0;
            } else { tp[i] = 0.0; tv[i] = 0; }
        }
    }
    const f64 mp = e_mid(c, e);
    // AskBook::ApplyTransactions (book.cpp:383-427): trades ascending
    i64 au_vol = 0; f64 au_proxy = 0.0, au_value = 0.0;
    e.a_obsval = 0.0; e.a_obsvol = 0;
#pragma unroll
    for (int i = 0; i < LOB_MAX_TRADES; i++) {
        if (i >= P.T || tv[i] <= 0) continue;
        if (tp[i] < mp) continue;
        i64 vol = tv[i];
        e.a_obsval += tp[i] * (f64)vol;
        e.a_obsvol += vol;
        if (e.a_on && e.a_opx <= tp[i]) {
            OrderR o{e.a_osz, e.a_oqh, e.a_oqt, e.a_oex};
            i64 rem0 = ord_remaining(o);
            vol = ord_transaction(o, vol);
            i64 exec = rem0 - ord_remaining(o);
            au_vol -= exec;
            au_proxy += (e.a_opx - mp) * (f64)exec;
            au_value += e.a_opx * (f64)exec;
            e.a_oqh = o.qh; e.a_oex = o.ex;
            if (ord_executed(o)) { e.a_on = 0; e.a_ntr++; }
        }
    }
    // BidBook::ApplyTransactions (book.cpp:468-510): trades descending
    i64 bu_vol = 0; f64 bu_proxy = 0.0, bu_value = 0.0;
    e.b_obsval = 0.0; e.b_obsvol = 0;
#pragma unroll
    for (int ii = 0; ii < LOB_MAX_TRADES; ii++) {
        const int i = LOB_MAX_TRADES - 1 - ii;
        if (i >= P.T || tv[i] <= 0) continue;
        if (tp[i] > mp) continue;
        i64 vol = tv[i];
        e.b_obsval += tp[i] * (f64)vol;
        e.b_obsvol += vol;
        if (e.b_on && e.b_opx >= tp[i]) {
            OrderR o{e.b_osz, e.b_oqh, e.b_oqt, e.b_oex};
            i64 rem0 = ord_remaining(o);
            vol = ord_transaction(o, vol);
            i64 exec = rem0 - ord_remaining(o);
            bu_vol += exec;
            bu_proxy += (mp - e.b_opx) * (f64)exec;
            bu_value -= e.b_opx * (f64)exec;
            e.b_oqh = o.qh; e.b_oex = o.ex;
            if (ord_executed(o)) { e.b_on = 0; e.b_ntr++; }
        }
    }
    if (!update_book_profiles(c, e, tp, tv)) return false;

    // BookUtils::HandleAdverseSelection (book.cpp:551-592)
    i64 ad_vol = 0; f64 ad_proxy = 0.0, ad_value = 0.0;
    {
        const f64 bap = e.ap0, bbp = e.bp0, rp = e_last_mid(c, e);
        if (e.a_on && e.a_opx <= bbp) {
            OrderR o{e.a_osz, e.a_oqh, e.a_oqt, e.a_oex};
            i64 rem = ord_remaining(o);
            ad_vol -= rem;
            ad_proxy += (f64)rem * (e.a_opx - rp);
            ad_value += (f64)rem * e.a_opx;
            e.a_on = 0; e.a_ntr++;
        }
        if (e.b_on && e.b_opx >= bap) {
            OrderR o{e.b_osz, e.b_oqh, e.b_oqt, e.b_oex};
            i64 rem = ord_remaining(o);
            ad_vol += rem;
            ad_proxy += (f64)rem * (rp - e.b_opx);
            ad_value -= (f64)rem * e.b_opx;
            e.b_on = 0; e.b_ntr++;
        }
    }
    e.pnl_step += au_proxy + bu_proxy + ad_proxy;
    e.lo_vol_step += (i32)(bu_vol - au_vol + (ad_vol < 0 ? -ad_vol : ad_vol));
    e.ep_pnl += au_value + bu_value + ad_value;
    e.position += bu_vol + au_vol + ad_vol;  // RiskManager::Update
    check_orders(P, e);

    const f64 mid = e_mid(c, e);
    const i64 mpt = (i64)lobh::to_ticks_t(P_tick(P), mid);
    const f64 mpm = mid - e_last_mid(c, e), sp = e.ap0 - e.bp0;
    // ten window pushes, batched: all loads, then all stores
    RMReg w_mid, w_vol, w_spr, w_tp, w_atx, w_btx;
    AccReg w_vn, w_vd;
    rm_load(S.f_midprice, b, w_mid); rm_load(S.f_volatility, b, w_vol); rm_load(S.spread_window, b, w_spr);
    rm_load(S.tp_mp, b, w_tp); rm_load(S.f_ask_tx, b, w_atx); rm_load(S.f_bid_tx, b, w_btx);
    acc_load(S.f_vwap_numer, b, w_vn); acc_load(S.f_vwap_denom, b, w_vd);
    rm_prep(S.f_midprice, B, b, w_mid); rm_prep(S.f_volatility, B, b, w_vol); rm_prep(S.spread_window, B, b, w_spr);
    rm_prep(S.tp_mp, B, b, w_tp); rm_prep(S.f_ask_tx, B, b, w_atx); rm_prep(S.f_bid_tx, B, b, w_btx);
    acc_prep(S.f_vwap_numer, B, b, w_vn); acc_prep(S.f_vwap_denom, B, b, w_vd);
    rm_apply(S.f_midprice, B, b, w_mid, (f64)mpt);
    rm_apply(S.f_volatility, B, b, w_vol, (f64)mpt);
    acc_apply(S.f_vwap_numer, B, b, w_vn, e.a_obsval + e.b_obsval);
    acc_apply(S.f_vwap_denom, B, b, w_vd, (f64)(e.a_obsvol + e.b_obsvol));
    rm_apply(S.spread_window, B, b, w_spr, 0.0 > sp ? 0.0 : sp);
    // TargetPrice::update (src/market/target_price.cpp:44-71)
    rm_apply(S.tp_mp, B, b, w_tp, P.target_price == LOB_TP_MICROPRICE ? e_micro(e) : mid);
    e.tp_val = w_tp.mean;
    {   // EWMA<double>::push (accumulators.cpp:157-163)
        f64 up = 0.0 > mpm ? 0.0 : mpm;
        f64 dn = fabs(0.0 < mpm ? 0.0 : mpm);
        e.ret_ups_mean = (P.ewma_alpha * up) + ((1 - P.ewma_alpha) * e.ret_ups_mean);
        e.ret_downs_mean = (P.ewma_alpha * dn) + ((1 - P.ewma_alpha) * e.ret_downs_mean);
    }
    rm_apply(S.f_ask_tx, B, b, w_atx, (f64)e.a_obsvol);
    rm_apply(S.f_bid_tx, B, b, w_btx, (f64)e.b_obsvol);
    return true;
}

// Base::performAction (base.cpp:254-337)
__device__ inline bool perform_action(const EnvCtx& c, EnvR& e, int action) {
    const DevParams& P = c.P;
    e.last_action = action;
    e.lo_vol_step = 0;
    e.pnl_step = 0.0;
    e.momentum_pnl_step = 0.0;
    do_action(c, e, action);
    check_orders(P, e);
    e.total_ticks++;  // UpdateStats
    f64 agg_r = get_reward(c, e);
    f64 agg_pnl = e.pnl_step;
    f64 agg_mpm = 0.0;
    do {
        e.pnl_step = 0.0;
        if (!next_state(c, e)) return false;
        f64 mpm = e_mid(c, e) - e_last_mid(c, e);
        e.pnl_step += (f64)e.position * mpm;
        e.momentum_pnl_step += (f64)e.position * mpm;
        agg_r += get_reward(c, e);
        agg_pnl += e.pnl_step;
        agg_mpm += mpm;
    } while (is_open(P, e.time_ms) && fabs(agg_mpm) < 1e-5);
    e.pnl_step = agg_pnl;
    {
        RMReg wu, wd;
        rm_load(c.S.pnl_ups, c.b, wu); rm_load(c.S.pnl_downs, c.b, wd);
        rm_prep(c.S.pnl_ups, c.S.B, c.b, wu); rm_prep(c.S.pnl_downs, c.S.B, c.b, wd);
        rm_apply(c.S.pnl_ups, c.S.B, c.b, wu, 0.0 > e.pnl_step ? 0.0 : e.pnl_step);
        rm_apply(c.S.pnl_downs, c.S.B, c.b, wd, fabs(0.0 < e.pnl_step ? 0.0 : e.pnl_step));
    }
    e.ep_reward += agg_r;
    e.ep_bandh += agg_mpm;
    return true;
}

// Intraday::getVariable (intraday.cpp:316-409)
__device__ inline f64 ulb(f64 val, f64 lb, f64 ub) {
    f64 m = val < ub ? val : ub;    // std::min(val, ub)
    return m < lb ? lb : m;         // std::max(., lb)
}
__device__ inline f64 get_variable(const EnvCtx& c, const EnvR& e, int v) {
    const DevParams& P = c.P;
    const DevState& S = c.S;
    const int b = c.b, B = S.B;
    switch (v) {
        case LOB_VAR_POS: return (f64)e.position / (f64)P.order_size;
        case LOB_VAR_SPD:
            return ulb((f64)(lobh::to_ticks_t(P_tick(P), e.ap0) - lobh::to_ticks_t(P_tick(P), e.bp0)), 0.0, 20.0);
        case LOB_VAR_MPM:
            return ulb((f64)(lobh::to_ticks_t(P_tick(P), rm_front(S.f_midprice, B, b)) -
                             lobh::to_ticks_t(P_tick(P), rm_back(S.f_midprice, B, b))), -10.0, 10.0);
        case LOB_VAR_IMB: {
            f64 v_a = (f64)e.a_tv, v_b = (f64)e.b_tv;
            return ((v_a + v_b) > 0 ? 5 * (v_b - v_a) / (v_b + v_a) : 0.0);
        }
        case LOB_VAR_SVL: {
            f64 q_a = S.f_ask_tx.sum[b], q_b = S.f_bid_tx.sum[b];
            return ((q_a + q_b) > 0 ? 5 * (q_b - q_a) / (q_a + q_b) : 0.0);
        }
        case LOB_VAR_VOL: return ulb(5.0 * rm_std(S.f_volatility, b), 0.0, 10.0);
        case LOB_VAR_RSI: {
            f64 u = e.ret_ups_mean, d = e.ret_downs_mean;
            return (u + d) != 0.0 ? 5.0 * (u - d) / (u + d) : 0.0;
        }
        case LOB_VAR_VWAP: {
            f64 d = S.f_vwap_numer.sum[b] / S.f_vwap_denom.sum[b];
            return ulb(d / S.spread_window.mean[b], -10.0, 10.0);
        }
        case LOB_VAR_A_DIST:
            if (e.a_on) return ((f64)lobh::to_ticks_t(P_tick(P), e.a_opx) - (f64)lobh::to_ticks_t(P_tick(P), e.ap0));
            return -100.0;
        case LOB_VAR_A_QUEUE:
            if (e.a_on) {
                f32 iq = (f32)e.a_oiq;
                f32 prog = (f32)e.a_oqh / (1.0f > iq ? 1.0f : iq);
                return 10.0 * (f64)(i64)prog;  // Book::queue_progress returns long
            }
            return -1.0;
        case LOB_VAR_B_DIST:
            if (e.b_on) return ((f64)lobh::to_ticks_t(P_tick(P), e.bp0) - (f64)lobh::to_ticks_t(P_tick(P), e.b_opx));
            return -100.0;
        case LOB_VAR_B_QUEUE:
            if (e.b_on) {
                f32 iq = (f32)e.b_oiq;
                f32 prog = (f32)e.b_oqh / (1.0f > iq ? 1.0f : iq);
                return 10.0 * (f64)(i64)prog;
            }
            return -1.0;
        case LOB_VAR_LAST_ACTION: return (f64)e.last_action;
        default: return 0.0;
    }
}

// Load / store the per-book scalars.
__device__ inline void env_load(const DevState& S, int b, EnvR& e) {
#define X(t, n) e.n = S.n[b];
    LOB_ENV_FIELDS(X)
#undef X
}
__device__ inline void env_store(const DevState& S, int b, const EnvR& e) {
#define X(t, n) S.n[b] = e.n;
    LOB_ENV_FIELDS(X)
#undef X
}
__device__ inline void env_load_best(const EnvCtx& c, EnvR& e) {
    e.ap0 = (f64)c.S.px[c.lvl(e.sel, 0, 0)];
    e.bp0 = (f64)c.S.px[c.lvl(e.sel, 1, 0)];
    e.lap0 = (f64)c.S.px[c.lvl(e.sel ^ 1, 0, 0)];
    e.lbp0 = (f64)c.S.px[c.lvl(e.sel ^ 1, 1, 0)];
}

#endif
