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
//
// Split in two:
//  * market pre-pass (`market_prepass`, once per episode, lane per book): every
//    quantity of Intraday::NextState that does not depend on the agent -- which
//    rows are applied (same-timestamp rows, invalid states), midprices,
//    cumulative volumes, observed trade volumes, the ten rolling windows, target
//    price, the state variables spd/mpm/imb/svl/vol/rsi/vwap -- written to the
//    per-event `Track`.  Even the step boundaries are agent-independent: a step
//    ends when the accumulated midprice move is >= 1e-5 (base.cpp:285-305).
//  * agent step (`perform_action`, every step): order matching against the
//    trades, queue updates against consecutive snapshots (read straight from the
//    immutable event records), adverse selection, inventory, PnL, reward,
//    quoting, and the agent-dependent state variables.
#ifndef LOB_ENV_H
#define LOB_ENV_H

#include <hip/hip_runtime.h>

#include "lob_internal.h"
#include "lob_state.h"

// ---- records in HBM: the device layout --------------------------------------------------------------
// The ABI's record (lob_engine.h / lob_stream.h: time, flags, ask_px[D], ask_vol[D], bid_px[D], bid_vol[D],
// trade_px[T], trade_vol[T]) is re-packed on the way into HBM so that every array starts on a 16-byte
// boundary and the trades are (price, volume) pairs: a level array is ceil(D / 4) 16-byte loads instead of
// D 4-byte ones.  The lane-per-book kernels are bound by the number of divergent loads a lane issues
// (every book reads its own records), not by bytes.
//   [0] time_ms [1] flags [2] 0 [3] 0 | ask_px[D4] | ask_vol[D4] | bid_px[D4] | bid_vol[D4] | (trade_px, trade_vol)[T] padded to 4
LOB_HD int drec_pad4(int n) { return (n + 3) & ~3; }
LOB_HD int drec_words(int D, int T) { return 4 + 4 * drec_pad4(D) + drec_pad4(2 * T); }
LOB_HD int drec_ask_px(int, int) { return 4; }
LOB_HD int drec_ask_vol(int D, int) { return 4 + drec_pad4(D); }
LOB_HD int drec_bid_px(int D, int) { return 4 + 2 * drec_pad4(D); }
LOB_HD int drec_bid_vol(int D, int) { return 4 + 3 * drec_pad4(D); }
LOB_HD int drec_trades(int D, int) { return 4 + 4 * drec_pad4(D); }
// ABI record -> device record (host upload path: repack_kernel; the synthetic generator writes it directly)
LOB_HD void drec_from_abi(const uint32_t* src, int D, int T, uint32_t* dst) {
    const int Wd = drec_words(D, T);
    for (int i = 0; i < Wd; i++) dst[i] = 0;
    dst[0] = src[LOB_REC_TIME];
    dst[1] = src[LOB_REC_FLAGS];
    for (int l = 0; l < D; l++) {
        dst[drec_ask_px(D, T) + l] = src[lob_rec_ask_px(D, T) + l];
        dst[drec_ask_vol(D, T) + l] = src[lob_rec_ask_vol(D, T) + l];
        dst[drec_bid_px(D, T) + l] = src[lob_rec_bid_px(D, T) + l];
        dst[drec_bid_vol(D, T) + l] = src[lob_rec_bid_vol(D, T) + l];
    }
    for (int i = 0; i < T; i++) {
        dst[drec_trades(D, T) + 2 * i] = src[lob_rec_trade_px(D, T) + i];
        dst[drec_trades(D, T) + 2 * i + 1] = src[lob_rec_trade_vol(D, T) + i];
    }
}
#if defined(__HIPCC__)
// One level array (16-byte aligned, D <= LOB_MAX_DEPTH words used); entries >= D come back 0.  All (LOB_MAX_DEPTH + 3) / 4
// 16-byte loads are issued unconditionally -- a load under a branch on D splits the batch of loads it belongs to (the
// compiler waits for everything in flight at the join), and these kernels are chains of memory round trips.  For D <= 8 the
// last quad lies in the next array of the record, after the last array in the trade slots / the next record: read and
// masked out (the stream buffer is allocated with a tail pad, lob_engine.hip set_records).
#define LOB_LEVEL_QUADS ((LOB_MAX_DEPTH + 3) / 4)
__device__ inline void drec_unpack(const uint4* v, int D, uint32_t* out) {
#pragma unroll
    for (int q = 0; q < LOB_LEVEL_QUADS; q++) {
        if (q * 4 + 0 < LOB_MAX_DEPTH) out[q * 4 + 0] = q * 4 + 0 < D ? v[q].x : 0u;
        if (q * 4 + 1 < LOB_MAX_DEPTH) out[q * 4 + 1] = q * 4 + 1 < D ? v[q].y : 0u;
        if (q * 4 + 2 < LOB_MAX_DEPTH) out[q * 4 + 2] = q * 4 + 2 < D ? v[q].z : 0u;
        if (q * 4 + 3 < LOB_MAX_DEPTH) out[q * 4 + 3] = q * 4 + 3 < D ? v[q].w : 0u;
    }
}
__device__ inline void drec_levels(const uint32_t* arr, int D, uint32_t* out) {
    const uint4* a4 = reinterpret_cast<const uint4*>(arr);
    uint4 v[(LOB_MAX_DEPTH + 3) / 4];
#pragma unroll
    for (int q = 0; q < (LOB_MAX_DEPTH + 3) / 4; q++) v[q] = a4[q];
#pragma unroll
    for (int q = 0; q < (LOB_MAX_DEPTH + 3) / 4; q++) {
        if (q * 4 + 0 < LOB_MAX_DEPTH) out[q * 4 + 0] = q * 4 + 0 < D ? v[q].x : 0u;
        if (q * 4 + 1 < LOB_MAX_DEPTH) out[q * 4 + 1] = q * 4 + 1 < D ? v[q].y : 0u;
        if (q * 4 + 2 < LOB_MAX_DEPTH) out[q * 4 + 2] = q * 4 + 2 < D ? v[q].z : 0u;
        if (q * 4 + 3 < LOB_MAX_DEPTH) out[q * 4 + 3] = q * 4 + 3 < D ? v[q].w : 0u;
    }
}
#endif

struct EnvR {
#define X(t, n) t n;
    LOB_ENV_FIELDS(X)
#undef X
};

struct EnvCtx {
    const DevParams& P;
    const DevState& S;
    int b;
    const uint32_t* rows;  // this book's first record: its own stream, or its window of the replayed one
    const TickLds* tk;     // the venue's tick table (LDS)
    // what the caller has fetched already, in the same round trip as the rest of the step's inputs (env_step_kernel): the
    // track entry of event k - 1 at the start of the step (target price, spread mean, cumulative volumes: the quotes, the market
    // order's guard) and BookMeta's n_track / complete.  Null / -1: looked up where needed.
    const Track* pre_prev = nullptr;
    int pre_n_track = -1, pre_complete = 0;
#ifdef LOB_PROF
    // phase clocks of the lane-per-book kernels (tools/exp_prof.py): the first lane of a wave stamps for the wave.  The phases are
    // summed in registers and written once, at the end (a read-modify-write of the counters per stamp put a memory round trip
    // into every phase it was meant to measure).
    mutable long long pt_ = 0;
    mutable long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    mutable i64* prow_ = nullptr;
    __device__ void prof_start(i64* prof, int lane, long long t0 = 0) const { prow_ = (prof && lane == 0) ? prof + (size_t)b * LOB_PROF_N : nullptr; pt_ = t0 ? t0 : clock64(); }
    __device__ void mark(int i) const { const long long n = clock64(); acc_[i - 20] += n - pt_; pt_ = n; }
    __device__ void flush() const {
        if (prow_) {
#pragma unroll
            for (int i = 0; i < 12; i++) prow_[20 + i] += acc_[i];
        }
    }
#else
    __device__ void prof_start(i64*, int, long long = 0) const {}
    __device__ void mark(int) const {}
    __device__ void flush() const {}
#endif
    __device__ EnvCtx(const DevParams& p, const DevState& s, int book, const TickLds* t) : P(p), S(s), b(book), tk(t) {
        const size_t first = s.rec_phase ? (size_t)s.rec_phase[book] : (size_t)book * (size_t)s.n_events;
        rows = s.records + first * (size_t)p.Wd;
    }
    __device__ EnvCtx(const DevParams& p, const DevState& s, int book, const TickLds* t, const uint32_t* rows_) : P(p), S(s), b(book), rows(rows_), tk(t) {}
    __device__ void err(int bit) const { atomicOr(S.error_flag, bit); }
    __device__ const uint32_t* row(int i) const { return rows + (size_t)i * (size_t)P.Wd; }
    __device__ const Track& track(int k) const { return S.track[(size_t)b * (size_t)S.track_len + (size_t)(k & S.track_mask)]; }
    __device__ const TrackHead& track_head(int k) const { return *reinterpret_cast<const TrackHead*>(&track(k)); }
    __device__ const TrackHead64& track_head64(int k) const { return *reinterpret_cast<const TrackHead64*>(&track(k)); }
    __device__ Track& track_w(int k) const { return S.track[(size_t)b * (size_t)S.track_len + (size_t)(k & S.track_mask)]; }
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
// Register-resident variants for the pre-pass, whose event loop pushes into the same windows thousands of
// times in a row: the running state (count, head, sum, mean, S) stays in registers from rm_load to
// rm_store, only the ring slots go through memory.
__device__ inline void rm_apply_reg(const RMPtrs& r, int B, int b, RMReg& g, f64 val) {
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
}
__device__ inline void rm_store(const RMPtrs& r, int b, const RMReg& g) {
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

__device__ inline void acc_apply_reg(const AccPtrs& r, int B, int b, AccReg& g, f64 val) {
    g.sum += val;
    r.ring[(size_t)g.slot * B + b] = val;
    if (g.cnt < r.w) g.cnt++;
    else g.sum -= g.old;
    g.head = g.slot;
}
__device__ inline void acc_store(const AccPtrs& r, int b, const AccReg& g) { r.cnt[b] = g.cnt; r.head[b] = g.head; r.sum[b] = g.sum; }

// ---- snapshots are read straight from the (immutable) event records ---------
__device__ inline f64 rec_price(const EnvCtx& c, int rec, int side, int l) {
    if (rec < 0) return 0.0;
    const uint32_t* r = c.row(rec);
    return (f64)__uint_as_float(r[(side == 0 ? drec_ask_px(c.P.D, c.P.T) : drec_bid_px(c.P.D, c.P.T)) + l]);
}
// Book::volume(price) / last_volume(price) (book.cpp:200-214) on the snapshot held by record `rec`.
// All level prices are fetched before the first compare (the kernel is a chain of dependent loads
// otherwise: a scan that waits for each level in turn costs D memory round trips per look-up).
__device__ inline int book_level_of(const uint32_t* px, int D, f64 k) {
    uint32_t w[LOB_MAX_DEPTH];
    drec_levels(px, D, w);
    f32 p[LOB_MAX_DEPTH];
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) p[l] = l < D ? __uint_as_float(w[l]) : 0.0f;
    int hit = -1;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++)
        if (p[l] != 0.0f && key4((f64)p[l]) == k) hit = l;  // price keys are unique per side (lob_validate_stream)
    return hit;
}
__device__ inline i64 book_volume(const EnvCtx& c, int rec, int side, f64 price) {
    if (rec < 0) return 0;
    const uint32_t* r = c.row(rec);
    const uint32_t* px = r + (side == 0 ? drec_ask_px(c.P.D, c.P.T) : drec_bid_px(c.P.D, c.P.T));
    const uint32_t* vol = r + (side == 0 ? drec_ask_vol(c.P.D, c.P.T) : drec_bid_vol(c.P.D, c.P.T));
    const int hit = book_level_of(px, c.P.D, key4(price));
    return hit >= 0 ? (i64)(i32)vol[hit] : 0;
}
// volume at `price` in two snapshots at once (UpdateOrder needs last_volume and volume): both
// level scans in flight together
__device__ inline void book_volume2(const EnvCtx& c, int rec_a, int rec_b, int side, f64 price, i64& va, i64& vb) {
    const int opx = side == 0 ? drec_ask_px(c.P.D, c.P.T) : drec_bid_px(c.P.D, c.P.T);
    const int ovol = side == 0 ? drec_ask_vol(c.P.D, c.P.T) : drec_bid_vol(c.P.D, c.P.T);
    const uint32_t* ra = c.row(rec_a < 0 ? 0 : rec_a);
    const uint32_t* rb = c.row(rec_b < 0 ? 0 : rec_b);
    const int D = c.P.D;
    const f64 k = key4(price);
    uint32_t wa[LOB_MAX_DEPTH], wb[LOB_MAX_DEPTH];
    drec_levels(ra + opx, D, wa);
    drec_levels(rb + opx, D, wb);
    f32 pa[LOB_MAX_DEPTH], pb[LOB_MAX_DEPTH];
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        pa[l] = l < D ? __uint_as_float(wa[l]) : 0.0f;
        pb[l] = l < D ? __uint_as_float(wb[l]) : 0.0f;
    }
    int ha = -1, hb = -1;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        if (pa[l] != 0.0f && key4((f64)pa[l]) == k) ha = l;
        if (pb[l] != 0.0f && key4((f64)pb[l]) == k) hb = l;
    }
    va = (rec_a >= 0 && ha >= 0) ? (i64)(i32)ra[ovol + ha] : 0;
    vb = (rec_b >= 0 && hb >= 0) ? (i64)(i32)rb[ovol + hb] : 0;
}

// The four look-ups of one applied depth row -- last_volume and volume at the ask order's price, the
// same at the bid order's -- with all 4 x D level prices in flight together.
__device__ inline void order_volumes(const EnvCtx& c, const EnvR& e, int last_rec, int row_rec, i64& a_lv, i64& a_v,
                                     i64& b_lv, i64& b_v) {
    a_lv = a_v = b_lv = b_v = 0;
    const int D = c.P.D;
    const int apx = drec_ask_px(D, c.P.T), avol = drec_ask_vol(D, c.P.T);
    const int bpx = drec_bid_px(D, c.P.T), bvol = drec_bid_vol(D, c.P.T);
    const uint32_t* rl = c.row(last_rec < 0 ? 0 : last_rec);
    const uint32_t* rr = c.row(row_rec < 0 ? 0 : row_rec);
    const bool a_on = e.a_on != 0, b_on = e.b_on != 0;
    uint32_t wal[LOB_MAX_DEPTH], war[LOB_MAX_DEPTH], wbl[LOB_MAX_DEPTH], wbr[LOB_MAX_DEPTH];
    drec_levels(rl + apx, D, wal); drec_levels(rr + apx, D, war);
    drec_levels(rl + bpx, D, wbl); drec_levels(rr + bpx, D, wbr);
    f32 pal[LOB_MAX_DEPTH], par[LOB_MAX_DEPTH], pbl[LOB_MAX_DEPTH], pbr[LOB_MAX_DEPTH];
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        pal[l] = (a_on && l < D) ? __uint_as_float(wal[l]) : 0.0f;
        par[l] = (a_on && l < D) ? __uint_as_float(war[l]) : 0.0f;
        pbl[l] = (b_on && l < D) ? __uint_as_float(wbl[l]) : 0.0f;
        pbr[l] = (b_on && l < D) ? __uint_as_float(wbr[l]) : 0.0f;
    }
    const f64 ka = key4(e.a_opx), kb = key4(e.b_opx);
    int hal = -1, har = -1, hbl = -1, hbr = -1;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        if (pal[l] != 0.0f && key4((f64)pal[l]) == ka) hal = l;
        if (par[l] != 0.0f && key4((f64)par[l]) == ka) har = l;
        if (pbl[l] != 0.0f && key4((f64)pbl[l]) == kb) hbl = l;
        if (pbr[l] != 0.0f && key4((f64)pbr[l]) == kb) hbr = l;
    }
    if (last_rec >= 0 && hal >= 0) a_lv = (i64)(i32)rl[avol + hal];
    if (row_rec >= 0 && har >= 0) a_v = (i64)(i32)rr[avol + har];
    if (last_rec >= 0 && hbl >= 0) b_lv = (i64)(i32)rl[bvol + hbl];
    if (row_rec >= 0 && hbr >= 0) b_v = (i64)(i32)rr[bvol + hbr];
}

// One snapshot's four level arrays in registers, ONE memory round trip (row_full_load).
//  * DoAction reads the best prices, the displayed volume at both quotes and -- for a market order -- walks
//    one side of the CURRENT snapshot (looked up one after the other they were three to twelve dependent
//    round trips per step);
//  * an event pass requests the arrays of the depth row it applies at its top, together with the trade slots,
//    and finds the volumes resting at the two order prices in them (row_volumes; the volumes come with the
//    prices: fetched after the level is known they are one more round trip, 0.103 -> 0.101 ms).  The matching
//    last_volume()s are not looked up at all: inside a step the order prices do not change, and the stashed
//    snapshot of one event is the current snapshot of the event before, so last_volume(price) of this pass is
//    volume(price) of the previous one (StepAgg::cv_*; at the first pass of a step: the displayed volume the
//    order was queued behind, Book::PlaceOrder).
struct RowFull {
    uint32_t apx[LOB_MAX_DEPTH], avol[LOB_MAX_DEPTH], bpx[LOB_MAX_DEPTH], bvol[LOB_MAX_DEPTH];
};
__device__ inline void row_full_load(const EnvCtx& c, int rec, RowFull& R) {
    const int D = c.P.D;
    const uint32_t* r = c.row(rec < 0 ? 0 : rec);
    drec_levels(r + drec_ask_px(D, c.P.T), D, R.apx);
    drec_levels(r + drec_ask_vol(D, c.P.T), D, R.avol);
    drec_levels(r + drec_bid_px(D, c.P.T), D, R.bpx);
    drec_levels(r + drec_bid_vol(D, c.P.T), D, R.bvol);
    // no snapshot: every price reads as "undefined" (0), like rec_price / book_volume
    const uint32_t keep = rec < 0 ? 0u : ~0u;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) { R.apx[l] &= keep; R.avol[l] &= keep; R.bpx[l] &= keep; R.bvol[l] &= keep; }
}
__device__ inline void row_volumes(const EnvCtx& c, const EnvR& e, const RowFull& L, i64& a_v, i64& b_v) {
    const int D = c.P.D;
    const bool a_on = e.a_on != 0, b_on = e.b_on != 0;
    const f64 ka = key4(e.a_opx), kb = key4(e.b_opx);
    int ha = -1, hb = -1;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        const f32 pa = (a_on && l < D) ? __uint_as_float(L.apx[l]) : 0.0f;
        const f32 pb = (b_on && l < D) ? __uint_as_float(L.bpx[l]) : 0.0f;
        if (pa != 0.0f && key4((f64)pa) == ka) ha = l;  // price keys are unique per side (lob_validate_stream)
        if (pb != 0.0f && key4((f64)pb) == kb) hb = l;
    }
    uint32_t va = 0, vb = 0;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        if (l == ha) va = L.avol[l];
        if (l == hb) vb = L.bvol[l];
    }
    a_v = ha >= 0 ? (i64)(i32)va : 0;
    b_v = hb >= 0 ? (i64)(i32)vb : 0;
}

// Book::volume(price) (book.cpp:200-214) on that snapshot
__device__ inline i64 full_volume(const EnvCtx& c, const RowFull& R, int side, f64 price) {
    const int D = c.P.D;
    const f64 k = key4(price);
    bool hit = false;
    uint32_t v = 0;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        const f32 p = l < D ? __uint_as_float(side == 0 ? R.apx[l] : R.bpx[l]) : 0.0f;
        if (p != 0.0f && key4((f64)p) == k) { hit = true; v = side == 0 ? R.avol[l] : R.bvol[l]; }  // price keys are unique per side
    }
    return hit ? (i64)(i32)v : 0;
}

__device__ inline void row_resolve(const EnvCtx&, const RowFull&) {}
// best prices of a snapshot held as RowFull (the quotes in LOB_QUOTE_BOOK mode)
__device__ inline f64 row_best_px(const RowFull& R, int side) { return (f64)__uint_as_float(side == 0 ? R.apx[0] : R.bpx[0]); }

#if defined(__HIP__)
// ---- the same snapshot with its LEVELS ACROSS LANES: 16 lanes per book (env_step16_kernel, lob_envstep.h) ---------------------
// Small batches leave most of the chip idle under the lane-per-book kernels (4 096 books: 64 waves on 1 024 SIMDs, each a chain
// of ~10 000 dependent instructions for 64 books).  Here a book is 16 lanes of a wave: the 14 quads of its 224-byte record are ONE
// coalesced load (lane i takes quad i: a wave fetches four records as four contiguous runs), a 64-word LDS row per book turns
// them into "lane l holds level l of the four arrays", and what Book does level by level becomes wave arithmetic:
//   Book::volume(price) / last_volume (book.cpp:200-222)   every level's lane compares its 1e-4 key, the group's 16 bits of the
//                                                          wave ballot name the level, its lane hands the volume over;
//   Ask/BidBook::WalkTheBook (book.cpp:431-456,514-539)    executed-so-far = an exclusive prefix sum of the level volumes across
//                                                          the group's lanes, l_ex = min(lvol, size - prefix) in every lane at once,
//                                                          "filled" = the ballot of prefix + lvol >= size; only the two f64 sums
//                                                          (proxy, value) are then added up in level order, lane by lane, so that
//                                                          they round as the reference's loop does.
// Everything of the step that is not per level (orders, inventory, PnL, reward, state variables) is replicated in the group's 16
// lanes: same instructions, same values, lane 0 stores.  Control flow is uniform within a group, so its lanes reach every
// cross-lane operation together.
struct RowLev {
    mutable uint4 q;                           // this lane's quad of the record, as loaded (lane li < Wd / 4)
    mutable uint32_t apx, avol, bpx, bvol;     // level li of the four arrays (0 beyond the depth / without a snapshot), once resolved
    uint32_t* stage;                           // the book's LDS staging row (64 words)
    int li;                                    // lane within the group
    mutable bool pending;                      // `q` has arrived (or is on its way) and is not transposed yet
    bool none;                                 // no snapshot (record < 0): every price reads as undefined
    __device__ __forceinline__ void resolve(const EnvCtx& c) const {
        if (!pending) return;
        const int D = c.P.D, D4 = drec_pad4(D);
        if (li < (c.P.Wd >> 2)) reinterpret_cast<uint4*>(stage)[li] = q;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        const bool in = li < D && !none;
        const int l = li < D ? li : 0;
        const uint32_t a = stage[4 + l], av = stage[4 + D4 + l], b = stage[4 + 2 * D4 + l], bv = stage[4 + 3 * D4 + l];
        apx = in ? a : 0u; avol = in ? av : 0u; bpx = in ? b : 0u; bvol = in ? bv : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");   // (the row may be written again right away)
        __builtin_amdgcn_wave_barrier();
        pending = false;
    }
};
// the group's 16 bits of a wave ballot
__device__ __forceinline__ uint32_t grp_ballot(bool p) { return (uint32_t)(__ballot(p) >> (threadIdx.x & 48)) & 0xffffu; }
__device__ inline void row_full_load(const EnvCtx& c, int rec, RowLev& R) {
    const uint4* r = reinterpret_cast<const uint4*>(c.row(rec < 0 ? 0 : rec));
    const int nq = c.P.Wd >> 2;
    R.q = r[R.li < nq ? R.li : nq - 1];   // one 16-byte load per lane: the record's quads, contiguous
    R.none = rec < 0;
    R.pending = true;
}
__device__ inline void row_resolve(const EnvCtx& c, const RowLev& R) { R.resolve(c); }
__device__ inline f64 row_best_px(const RowLev& R, int side) {   // (resolved by the caller: place_orders below)
    return (f64)__uint_as_float(__shfl(side == 0 ? R.apx : R.bpx, 0, 16));
}
// Book::volume(price) by ballot: the level whose key matches (unique per side: lob_validate_stream; the sequential form keeps
// the LAST match, so does this)
__device__ inline i64 rowlev_volume(const EnvCtx& c, const RowLev& R, int side, f64 k) {
    R.resolve(c);
    const f32 p = __uint_as_float(side == 0 ? R.apx : R.bpx);
    const uint32_t m = grp_ballot(p != 0.0f && key4((f64)p) == k);
    const uint32_t v = __shfl(side == 0 ? R.avol : R.bvol, m ? 31 - __clz(m) : 0, 16);
    return m ? (i64)(i32)v : 0;
}
__device__ inline i64 full_volume(const EnvCtx& c, const RowLev& R, int side, f64 price) { return rowlev_volume(c, R, side, key4(price)); }
__device__ inline void row_volumes_k(const EnvCtx& c, bool a_on, bool b_on, f64 ka, f64 kb, const RowLev& L, i64& a_v, i64& b_v) {
    const i64 va = rowlev_volume(c, L, 0, ka), vb = rowlev_volume(c, L, 1, kb);
    a_v = a_on ? va : 0;
    b_v = b_on ? vb : 0;
}
#endif

// RiskManager::CheckOrders (src/environment/risk_manager.cpp:26-32)
__device__ inline void check_orders(const DevParams& P, EnvR& e) {
    if (e.position >= P.pos_ub) e.b_on = 0;
    else if (e.position <= P.pos_lb) e.a_on = 0;
}

// RiskManager::PlaceOrder with ORDER_LIMIT == 1 (risk_manager.cpp:61-99) +
// Book::PlaceOrder (book.cpp:250-261): cancel whatever rests, place a new
// order queued behind the displayed volume at that price.
template <class ROW>
__device__ inline void place_one(const EnvCtx& c, EnvR& e, int side, f64 price, i32 price_ticks, const ROW& cur) {
    if (!(price > 0.0)) c.err(LOB_ERR_BAD_ORDER_PRICE);
    i64 qh = full_volume(c, cur, side, price);
    if (side == 0) {
        e.a_on = 1; e.a_opx = price; e.a_osz = c.P.order_size; e.a_oqh = qh; e.a_oqt = 0; e.a_oex = 0; e.a_oiq = qh; e.a_otk = price_ticks;
    } else {
        e.b_on = 1; e.b_opx = price; e.b_osz = c.P.order_size; e.b_oqh = qh; e.b_oqt = 0; e.b_oex = 0; e.b_oiq = qh; e.b_otk = price_ticks;
    }
}

// Intraday::_place_orders + l2p_ (src/environment/intraday.cpp:64-82,164-173)
// `cur` = the level arrays of e.rec_cur (row_full_load)
template <class ROW>
__device__ inline void place_orders(const EnvCtx& c, EnvR& e, int al, int bl, const ROW& cur) {
    const DevParams& P = c.P;
    e.ask_level = al;
    e.bid_level = bl;
    int ta, tb;
    int band = 0, band_t = 0;  // (to_ticks_hint / to_price_hint: the six conversions of a re-quote fall in one band)
    if (P.quote_mode == LOB_QUOTE_BOOK) {
        row_resolve(c, cur);
        const f64 ap0 = row_best_px(cur, 0), bp0 = row_best_px(cur, 1);
        if (ap0 == 0.0 || bp0 == 0.0) c.err(LOB_ERR_UNDEF_PRICE);
        ta = lobh::to_ticks_hint((*c.tk), ap0, band) + al;
        tb = lobh::to_ticks_hint((*c.tk), bp0, band) - bl;
    } else {
        const Track& t = c.pre_prev ? *c.pre_prev : c.track(e.k - 1);
        f64 tp = t.tp_val;
        f64 half = t.spread_mean / 2.0;
        f64 half_spd = 0.0 > half ? 0.0 : half;  // std::max(0.0, x)
        ta = lobh::to_ticks_hint((*c.tk), tp + (f64)al * half_spd, band);
        tb = lobh::to_ticks_hint((*c.tk), tp - (f64)bl * half_spd, band);
    }
    e.ask_quote = lobh::to_price_hint((*c.tk), ta, band_t);
    e.bid_quote = lobh::to_price_hint((*c.tk), tb, band_t);
    // a_dist / b_dist need ToTicks(order price): computed once, here
    place_one(c, e, 0, e.ask_quote, lobh::to_ticks_hint((*c.tk), e.ask_quote, band), cur);
    place_one(c, e, 1, e.bid_quote, lobh::to_ticks_hint((*c.tk), e.bid_quote, band), cur);
}
__device__ inline void place_orders(const EnvCtx& c, EnvR& e, int al, int bl) {
    RowFull cur;
    row_full_load(c, e.rec_cur, cur);
    place_orders(c, e, al, bl, cur);
}

// AskBook/BidBook::WalkTheBook via BookUtils::MarketOrder (book.cpp:431-456,514-539,595-610)
__device__ inline void market_order(const EnvCtx& c, EnvR& e, i64 size, i64& out_vol, f64& out_proxy, f64& out_value, const RowFull& cur) {
    out_vol = 0; out_proxy = 0.0; out_value = 0.0;
    if (e.rec_cur < 0) c.err(LOB_ERR_UNDEF_PRICE);
    const f64 mip = e.mid;
    if (size == 0) return;
    const int side = size > 0 ? 0 : 1;
    i64 abs_size = size < 0 ? -size : size;
    // cumulative total_volume_ (quirk Q1) of the side being walked
    i64 tv;
    if (e.done == 2) {
        // out of data: the totals where the pre-pass stopped -- the last complete event's plus the rows an abandoned
        // event still applied before the stream ran dry (it went through same-timestamp rows / invalid states)
        tv = side == 0 ? c.S.prep[c.b].a_tv : c.S.prep[c.b].b_tv;
    } else {
        tv = c.pre_prev ? (side == 0 ? c.pre_prev->a_tv : c.pre_prev->b_tv) : (side == 0 ? c.track(e.k - 1).a_tv : c.track(e.k - 1).b_tv);
    }
    if (abs_size > tv) return;
    i64 executed = 0;
    f64 proxy = 0.0, value = 0.0;
    bool filled = false;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        const f32 pf = __uint_as_float(side == 0 ? cur.apx[l] : cur.bpx[l]);
        if (l >= c.P.D || filled || pf == 0.0f) continue;
        f64 p = (f64)pf;
        i64 lvol = (i64)(i32)(side == 0 ? cur.avol[l] : cur.bvol[l]);
        i64 left = abs_size - executed;
        i64 l_ex = lvol < left ? lvol : left;
        executed += l_ex;
        proxy -= (f64)l_ex * fabs(p - mip);
        if (side == 0) value -= (f64)l_ex * p;
        else value += (f64)l_ex * p;
        if (executed >= abs_size) {
            if (side == 0) e.a_ntr++; else e.b_ntr++;
            filled = true;
        }
    }
    out_vol = side == 0 ? executed : -executed;
    out_proxy = proxy;
    out_value = value;
}

#if defined(__HIP__)
// ... the same walk with the levels across the group's lanes (RowLev above): prefix sum + ballot instead of the loop
__device__ inline void market_order(const EnvCtx& c, EnvR& e, i64 size, i64& out_vol, f64& out_proxy, f64& out_value, const RowLev& cur) {
    out_vol = 0; out_proxy = 0.0; out_value = 0.0;
    if (e.rec_cur < 0) c.err(LOB_ERR_UNDEF_PRICE);
    const f64 mip = e.mid;
    if (size == 0) return;
    const int side = size > 0 ? 0 : 1;
    const i64 abs_size = size < 0 ? -size : size;
    i64 tv;  // cumulative total_volume_ (quirk Q1) of the side being walked: as in the lane-per-book form
    if (e.done == 2) tv = side == 0 ? c.S.prep[c.b].a_tv : c.S.prep[c.b].b_tv;
    else tv = c.pre_prev ? (side == 0 ? c.pre_prev->a_tv : c.pre_prev->b_tv) : (side == 0 ? c.track(e.k - 1).a_tv : c.track(e.k - 1).b_tv);
    if (abs_size > tv) return;
    cur.resolve(c);
    const f32 pf = __uint_as_float(side == 0 ? cur.apx : cur.bpx);
    const bool level = cur.li < c.P.D && pf != 0.0f;                  // (a level the loop does not skip)
    const i64 lvol = level ? (i64)(i32)(side == 0 ? cur.avol : cur.bvol) : 0;
    // executed before this level = exclusive prefix sum of the level volumes over the group's lanes
    i64 incl = lvol;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const i64 up = __shfl_up(incl, d, 16);
        if (cur.li >= d) incl += up;
    }
    const i64 before = incl - lvol;
    // the loop stops taking levels once `executed >= abs_size`: a level takes part while the volume before it is short of the size
    const bool takes = level && before < abs_size;
    const i64 left = abs_size - before;
    const i64 l_ex = takes ? (lvol < left ? lvol : left) : 0;
    const uint32_t fills = grp_ballot(takes && before + l_ex >= abs_size);   // the level at which the order is complete
    // total executed: the last lane's inclusive sum, capped
    const i64 total = __shfl(incl, 15, 16);
    const i64 executed = total < abs_size ? total : abs_size;
    // the two f64 sums in level order (a level that does not take part contributes an exact +0.0: x - 0.0 == x, x + 0.0 == x for
    // every x these sums can hold -- they start at +0.0 and `value` of a bid walk only grows)
    const f64 p = (f64)pf;
    const f64 t_proxy = takes ? (f64)l_ex * fabs(p - mip) : 0.0;
    const f64 t_value = takes ? (f64)l_ex * p : 0.0;
    f64 proxy = 0.0, value = 0.0;
    for (int l = 0; l < c.P.D; l++) {
        const f64 tp = __shfl(t_proxy, l, 16), tvl = __shfl(t_value, l, 16);
        proxy -= tp;
        if (side == 0) value -= tvl;
        else value += tvl;
    }
    if (fills) { if (side == 0) e.a_ntr++; else e.b_ntr++; }
    out_vol = side == 0 ? executed : -executed;
    out_proxy = proxy;
    out_value = value;
}
#endif

// Base::ClearInventory (base.cpp:339-349) + RiskManager::ClearInventory/MarketOrder
template <class ROW>
__device__ inline void clear_inventory(const EnvCtx& c, EnvR& e, const ROW& cur) {
    i64 v; f64 proxy, value;
    market_order(c, e, -e.position, v, proxy, value, cur);
    e.position += v;
    e.pnl_step += proxy;
    e.lo_vol_step += (i32)(v < 0 ? -v : v);
    e.ep_pnl += value;
    if (v > 0) e.market_buys++;
    else if (v < 0) e.market_sells++;
}
__device__ inline void clear_inventory(const EnvCtx& c, EnvR& e) {
    RowFull cur;
    row_full_load(c, e.rec_cur, cur);
    clear_inventory(c, e, cur);
}

// Intraday::DoAction (intraday.cpp:176-220)
template <class ROW>
__device__ inline void do_action(const EnvCtx& c, EnvR& e, int action, const ROW& cur) {
    int al, bl;
    switch (action) {
        case 0: al = 1; bl = 1; break;
        case 1: clear_inventory(c, e, cur); al = e.ask_level; bl = e.bid_level; break;
        case 2: al = 2; bl = 2; break;
        case 3: al = 3; bl = 3; break;
        case 4: al = 0; bl = 2; break;
        case 5: al = 2; bl = 0; break;
        case 6: al = 1; bl = 4; break;
        case 7: al = 4; bl = 1; break;
        case 8: al = 5; bl = 5; break;
        default: return;
    }
    place_orders(c, e, al, bl, cur);
}

__device__ inline bool is_open(const DevParams& P, i32 t) {  // Market::IsOpen, market.cpp:67-70
    return ((i64)t > P.open_ms + 30 * 60000LL) && ((i64)t < P.close_ms - 30 * 60000LL);
}

// std::exp(float) as the reference's libm computes it (glibc 2.35 x86-64, sysdeps/ieee754/flt-32/e_expf.c,
// the FMA build its ifunc selects on every CPU this runs beside): exp(x) = 2^(k/32) * 2^(r/32) with a
// 32-entry table and a cubic in double precision, every multiply-add fused.  tools/check_expf.c compares
// this restatement with libm's expf over ALL 2^32 - 2^24 non-NaN inputs: 0 differences.
__device__ inline f32 expf_glibc(f32 x) {
    static const u64 T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
        0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
        0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
        0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
        0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    const f64 N = 32.0;
    const f64 InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
    const f64 C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    const uint32_t ux = __float_as_uint(x), ax = ux & 0x7fffffffu;
    if (ax >= 0x42b00000u) {  // |x| >= 88 or NaN
        if (ux == 0xff800000u) return 0.0f;
        if (ax >= 0x7f800000u) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    const f64 xd = (f64)x;
    f64 kd = fma(InvLn2N, xd, SHIFT);
    const u64 ki = (u64)__double_as_longlong(kd);
    kd -= SHIFT;
    const f64 r = fma(InvLn2N, xd, -kd);
    const f64 s = __longlong_as_double((long long)(T[ki % 32] + (ki << 47)));
    const f64 z = fma(C0, r, C1);
    const f64 r2 = r * r;
    f64 y = fma(C2, r, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (f32)y;
}

// Base::getReward (base.cpp:166-237)
// `spread_mean`: spread_window.mean() as of the last completed event (track entry e.k - 1), when the caller holds it
__device__ inline f64 get_reward(const EnvCtx& c, const EnvR& e, const f64* spread_mean = nullptr) {
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
        case LOB_REWARD_SPREAD: r = e.pnl_step / (spread_mean ? *spread_mean : c.track(e.k - 1).spread_mean); break;
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
        case LOB_REWARD_MM_EXP: {
            // -pow(1.0 - exp(r_pos_weight * abs_pos), 2) + r_pnl_weight * pnl_step: the product and the exp are float
            // (std::exp(float)), pow(x, 2) is x * x in the compiled reference (and exactly rounded either way)
            const f32 ex = expf_glibc(P.pos_weight * (f32)abs_pos);
            const f64 t = 1.0 - (f64)ex;
            r = -(t * t);
            r += (f64)P.pnl_weight * e.pnl_step;
            break;
        }
        default: break;
    }
    return r * 100;
}

// Book::UpdateOrder (book.cpp:102-141) of one side for ONE applied depth row:
// `last_rec` = the stashed snapshot, `row_rec` = the row just applied.
__device__ inline void update_order(const EnvCtx& c, EnvR& e, int side, i64 lv, i64 v, const f64* tp, const i64* tv) {
    const DevParams& P = c.P;
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
    if (lv == 0) return;
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

// The trades one NextState sees: TimeAndSales::LoadUntil(next depth time)
// (src/data/streamer.cpp:57-80, src/data/basic.cpp:148-181) hands over everything
// up to the timestamp of the event's FIRST depth row that has not been consumed
// yet, merged per 1e-4 price key.  Rows carry the trades of their own interval,
// so an event whose predecessor swallowed several rows (same timestamp, invalid
// states) merges the slots of rows lo..hi; normally lo == hi.
// (merge_trade_slots: one record's trade slots -- `w4`: its (price, volume) pairs as loaded -- merged into the list tp / tv of n entries)
template <int TM>
__device__ inline void merge_trade_slots(const EnvCtx& c, const uint4* w4, int& n, f64* tp, i64* tv) {
    const DevParams& P = c.P;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        if (i >= P.T) continue;
        const uint4 w = w4[i >> 1];
        f32 p = __uint_as_float((i & 1) ? w.z : w.x);
        i32 v = (i32)((i & 1) ? w.w : w.y);
        if (!((p > 0.0f) && (v > 0))) continue;
        const f64 pd = (f64)p, k = key4(pd);
        // std::map<double,long,FloatComparator>::operator[] += : find the key or insert in order
        int pos = 0;
        bool found = false;
#pragma unroll
        for (int q = 0; q < TM; q++) {
            if (q < n) {
                const f64 kq = key4(tp[q]);
                if (kq == k) { tv[q] += (i64)v; found = true; }
                if (kq < k) pos = q + 1;
            }
        }
        if (found) continue;
        if (n >= P.T) { c.err(LOB_ERR_TRADE_OVERFLOW); continue; }
#pragma unroll
        for (int q = TM - 1; q > 0; q--) {
            if (q > pos && q <= n) { tp[q] = tp[q - 1]; tv[q] = tv[q - 1]; }
        }
#pragma unroll
        for (int q = 0; q < TM; q++)
            if (q == pos) { tp[q] = pd; tv[q] = (i64)v; }
        n++;
    }
}
template <int TM>
__device__ inline void load_trades(const EnvCtx& c, int lo, int hi, f64* tp, i64* tv) {
    const DevParams& P = c.P;
#pragma unroll
    for (int i = 0; i < TM; i++) { tp[i] = 0.0; tv[i] = 0; }
    int n = 0;
    for (int rec = lo; rec <= hi; rec++) {
        const uint4* r4 = reinterpret_cast<const uint4*>(c.row(rec) + drec_trades(P.D, P.T));  // 16-byte aligned: two (price, volume) pairs per load
        uint4 w4[(TM + 1) / 2];
#pragma unroll
        for (int q = 0; q < (TM + 1) / 2; q++) w4[q] = 2 * q < P.T ? r4[q] : make_uint4(0, 0, 0, 0);  // wave-uniform; the pad reads as "no trade"
        merge_trade_slots<TM>(c, w4, n, tp, tv);
    }
}

// AskBook / BidBook::ApplyTransactions (book.cpp:383-427, 468-510) for the agent's
// orders (the observed-volume bookkeeping of the same functions is agent
// independent and lives in the pre-pass).
template <int TM>
__device__ inline void match_orders(const EnvCtx& c, EnvR& e, const f64* tp, const i64* tv, f64 mp, i64& au_vol,
                                    f64& au_proxy, f64& au_value, i64& bu_vol, f64& bu_proxy, f64& bu_value) {
    const DevParams& P = c.P;
    au_vol = 0; au_proxy = 0.0; au_value = 0.0;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        if (i >= P.T || tv[i] <= 0) continue;
        if (tp[i] < mp) continue;
        i64 vol = tv[i];
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
    bu_vol = 0; bu_proxy = 0.0; bu_value = 0.0;
#pragma unroll
    for (int ii = 0; ii < TM; ii++) {
        const int i = TM - 1 - ii;
        if (i >= P.T || tv[i] <= 0) continue;
        if (tp[i] > mp) continue;
        i64 vol = tv[i];
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
}

struct StepAgg {
    f64 r, pnl, mpm;
    i32 n_track, complete;  // BookMeta's, as of this step
    // volume(order price) in the current snapshot, per side: the next pass's last_volume (row_volumes)
    i64 cv_a, cv_b;
    i32 cv_valid, _pad;
};
// Agent-dependent part of Intraday::NextState (intraday.cpp:225-272) for event e.k.
// Returns false when the stream is exhausted (the abandoned event still matches
// its trades and stashes the books, like the reference).
// `t` = the track entry of event e.k, fetched by the caller (one pass ahead where it can: the entries are
// agent-independent, and a pass is a chain of dependent look-ups: track entry -> rows -> volumes);
// `n_track` / `complete`: BookMeta's, fetched once per step.
template <int TM>
__device__ inline bool next_state(const EnvCtx& c, EnvR& e, const TrackHead& t, StepAgg& g) {
    const int n_track = g.n_track, complete = g.complete;
    const DevParams& P = c.P;
    f64 tp[TM];
    i64 tv[TM];
    i64 au_vol, bu_vol; f64 au_proxy, au_value, bu_proxy, bu_value;
    if (e.k >= n_track) {
        const BookMeta& M = c.S.meta[c.b];
        // the ring of a long stream ran dry before its next refill: the run is void (reported as LOB_ESTATE)
        if (!complete) c.err(LOB_ERR_TRACK_UNDERRUN);
        // out of data inside this event (Streamer::LoadNext fails, src/data/streamer.cpp:42-49)
        if (M.ex_first >= 0) {
            load_trades<TM>(c, e.pf + 1, M.ex_first, tp, tv);
            match_orders<TM>(c, e, tp, tv, e.mid, au_vol, au_proxy, au_value, bu_vol, bu_proxy, bu_value);
            // rows applied before the stream ran dry still update the queue model
            for (int r = M.ex_first; r <= M.ex_cur && M.ex_cur >= M.ex_first; r++) {
                i64 a_lv, a_v, b_lv, b_v;
                order_volumes(c, e, M.ex_last, r, a_lv, a_v, b_lv, b_v);
                update_order(c, e, 0, a_lv, a_v, tp, tv);
                update_order(c, e, 1, b_lv, b_v, tp, tv);
            }
        }
        e.rec_cur = M.ex_cur; e.rec_last = M.ex_last; e.time_ms = M.ex_time;
        e.mid_prev = e.rec_last >= 0 ? (rec_price(c, e.rec_last, 0, 0) + rec_price(c, e.rec_last, 1, 0)) / 2.0 : 0.0;
        e.mid = e.rec_cur >= 0 ? (rec_price(c, e.rec_cur, 0, 0) + rec_price(c, e.rec_cur, 1, 0)) / 2.0 : 0.0;
        e.events += M.ex_records;
        e.done = 2;
        return false;
    }
    // everything this pass reads of the record stream is requested here, together: the level prices of its
    // (first) row and the trade slots
    RowFull L;
    row_full_load(c, t.rec_first, L);
    load_trades<TM>(c, e.pf + 1, t.rec_first, tp, tv);
    e.pf = t.rec_first;
    c.mark(22);  // track entry, trade slots
    const f64 mp = e.mid;
    match_orders<TM>(c, e, tp, tv, mp, au_vol, au_proxy, au_value, bu_vol, bu_proxy, bu_value);
    c.mark(23);  // match_orders
    // UpdateBookProfiles: StashState, then ApplyChanges (-> UpdateOrder) for every applied row
    const int last_rec = e.rec_cur;
    i64 a_lv = g.cv_a, b_lv = g.cv_b;
    if (!g.cv_valid) {  // no order was placed this step (an action outside 0..8): look the stashed snapshot up
        i64 a_v0, b_v0;
        order_volumes(c, e, last_rec, t.rec_first, a_lv, a_v0, b_lv, b_v0);
    }
    for (int r = t.rec_first; r <= t.rec_last; r++) {
        if (r != t.rec_first) row_full_load(c, r, L);
        i64 a_v, b_v;
        row_volumes(c, e, L, a_v, b_v);
        update_order(c, e, 0, a_lv, a_v, tp, tv);
        update_order(c, e, 1, b_lv, b_v, tp, tv);
        g.cv_a = a_v;
        g.cv_b = b_v;
    }
    g.cv_valid = 1;
    c.mark(24);  // order volumes + update_order
    e.rec_last = last_rec;
    e.rec_cur = t.rec_last;
    e.mid_prev = e.mid;
    e.mid = t.mid;
    e.time_ms = t.time_ms;
    e.events += (i64)(t.rec_last - t.rec_first + 1);
    e.k++;

    // BookUtils::HandleAdverseSelection (book.cpp:551-592); the best prices of the new snapshot are L's
    i64 ad_vol = 0; f64 ad_proxy = 0.0, ad_value = 0.0;
    if (e.a_on || e.b_on) {
        const f64 bap = (f64)__uint_as_float(L.apx[0]), bbp = (f64)__uint_as_float(L.bpx[0]), rp = e.mid_prev;
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
    c.mark(25);  // adverse selection, PnL, position
    return true;
}

// Base::performAction (base.cpp:254-337) in three pieces, so that the event loop can be driven either by
// the lane that owns the book (perform_action) or by whichever lane of the block is free
// (env_compact_kernel): the running sums of the loop live in `StepAgg`.
// up to the first NextState: DoAction, CheckOrders, UpdateStats, the reward of the action itself
template <class ROW>
__device__ inline void step_prologue(const EnvCtx& c, EnvR& e, int action, StepAgg& g, const ROW& cur) {
    const DevParams& P = c.P;
    if (c.pre_n_track >= 0) {
        g.n_track = c.pre_n_track;
        g.complete = c.pre_complete;
    } else {
        const BookMeta& M = c.S.meta[c.b];
        g.n_track = M.n_track;
        g.complete = M.complete;
    }
    e.last_action = action;
    e.lo_vol_step = 0;
    e.pnl_step = 0.0;
    e.momentum_pnl_step = 0.0;
    do_action(c, e, action, cur);
    // Book::PlaceOrder queued both orders behind the displayed volume at their prices (oiq): that is
    // last_volume(order price) of the first event of this step
    g.cv_valid = action >= 0 && action <= 8;
    g.cv_a = e.a_oiq;
    g.cv_b = e.b_oiq;
    check_orders(P, e);
    c.mark(21);  // DoAction: quotes, tick conversions, queue position
    e.total_ticks++;  // UpdateStats (base.cpp:412-442): occupancy of the two quotes and of the inventory at decision time
    {
        const i64 ha = e.a_on ? 1 : 0, hb = e.b_on ? 1 : 0;
        e.tick_ab += ha | (hb << 32);
        e.tick_both += (i32)(ha & hb);
        e.tick_pos += (i64)(e.position > 0) | ((i64)(e.position < 0) << 32);
        e.ntr_snap = (i64)(uint32_t)e.a_ntr | ((i64)(uint32_t)e.b_ntr << 32);
    }
    g.r = get_reward(c, e);
    g.pnl = e.pnl_step;
    g.mpm = 0.0;
}
// one pass of the do-while: 0 = another event follows, 1 = the step is complete, 2 = out of data
template <int TM>
__device__ inline int step_event(const EnvCtx& c, EnvR& e, StepAgg& g, const TrackHead& t) {
    const DevParams& P = c.P;
    e.pnl_step = 0.0;
    if (!next_state<TM>(c, e, t, g)) return 2;
    const f64 mpm = e.mid - e.mid_prev;
    e.pnl_step += (f64)e.position * mpm;
    e.momentum_pnl_step += (f64)e.position * mpm;
    g.r += get_reward(c, e);
    g.pnl += e.pnl_step;
    g.mpm += mpm;
    c.mark(26);  // reward, loop bookkeeping
    c.mark(31);  // (count of passes: ~0 clocks each)
    return (is_open(P, e.time_ms) && fabs(g.mpm) < 1e-5) ? 0 : 1;
}
// after the loop: PnL windows, episode totals
__device__ inline void step_epilogue(const EnvCtx& c, EnvR& e, const StepAgg& g) {
    e.pnl_step = g.pnl;
    {
        RMReg wu, wd;
        rm_load(c.S.pnl_ups, c.b, wu); rm_load(c.S.pnl_downs, c.b, wd);
        rm_prep(c.S.pnl_ups, c.S.B, c.b, wu); rm_prep(c.S.pnl_downs, c.S.B, c.b, wd);
        rm_apply(c.S.pnl_ups, c.S.B, c.b, wu, 0.0 > e.pnl_step ? 0.0 : e.pnl_step);
        rm_apply(c.S.pnl_downs, c.S.B, c.b, wd, fabs(0.0 < e.pnl_step ? 0.0 : e.pnl_step));
    }
    e.ep_reward += g.r;
    e.ep_bandh += g.mpm;
    c.mark(27);  // PnL windows
}
// the same with the two windows' registers already loaded and prepared (rm_load + rm_prep) by the caller
__device__ inline void step_epilogue_pre(const EnvCtx& c, EnvR& e, const StepAgg& g, RMReg& wu, RMReg& wd) {
    e.pnl_step = g.pnl;
    rm_apply(c.S.pnl_ups, c.S.B, c.b, wu, 0.0 > e.pnl_step ? 0.0 : e.pnl_step);
    rm_apply(c.S.pnl_downs, c.S.B, c.b, wd, fabs(0.0 < e.pnl_step ? 0.0 : e.pnl_step));
    e.ep_reward += g.r;
    e.ep_bandh += g.mpm;
    c.mark(27);  // PnL windows
}
// `t` = the track entry of event e.k, `cur` = the level arrays of e.rec_cur: requested by the caller from the
// two scalars k and rec_cur BEFORE it loads the rest of the book's state, so that three round trips (state,
// track entry, snapshot) overlap.
template <int TM>
__device__ inline bool perform_action(const EnvCtx& c, EnvR& e, int action, TrackHead t, const RowFull& cur) {
    // From then on every pass fetches the NEXT event's track entry (its first 32 bytes: all a pass reads)
    // before it starts on its own: wasted once per step, hidden every time.  Touching the next pass's record
    // as well (one dword per 64-byte sector, a pass ahead) bought nothing once a pass requested all it reads
    // of the record in one go (next_state).
    StepAgg g;
    step_prologue(c, e, action, g, cur);
    int st;
    do {
        const TrackHead tn = c.track_head(e.k + 1);
        st = step_event<TM>(c, e, g, t);
        t = tn;
    } while (st == 0);
    if (st == 2) return false;
    step_epilogue(c, e, g);
    return true;
}

// ---- the event pass again, for streams with at most two trade slots per record (TM == 2): "pass_fast" -----------
// Same arithmetic, statement for statement, as next_state + step_event above -- AskBook / BidBook::ApplyTransactions,
// Book::UpdateOrder, BookUtils::HandleAdverseSelection, RiskManager::Update / CheckOrders and the loop body of
// Base::performAction (base.cpp:285-305) -- written for the latency of ONE pass, which is what bounds env_kernel (a
// wave runs until its unluckiest book's midprice has moved, and each pass was ~20 000 clocks of dependent work):
//  * the book's hot scalars (`EnvR h`, a register copy of the LDS slot) stay in registers across the loop;
//  * the merged trade list and the touch after the event come with the track entry (TrackHead64, written by the
//    pre-pass: they do not depend on the agent), so a pass reads nothing of the record stream but the level arrays
//    of the rows it applies -- and those are requested ONE PASS AHEAD (an event's first row is the record after the
//    previous event's last), like the track entry itself: a pass waits for no memory at all;
//  * the order prices are fixed inside a step, so their 1e-4 keys are computed once (ka, kb);
//  * the two sides are straight-line select form instead of nested divergent branches (only the pro-rata
//    cancellation, with its two IEEE divisions, stays behind a branch).
// Anything else (a step that starts without freshly placed orders, the end of the stream) takes the general pass.
// the fields of EnvR an event pass can write (everything else of the register copy is dead after the loop)
#define LOB_ENV_PASS_FIELDS(X) \
    X(k) X(time_ms) X(rec_cur) X(rec_last) X(pf) X(mid) X(mid_prev) X(position) X(lo_vol_step) X(pnl_step) X(momentum_pnl_step) \
    X(ep_pnl) X(events) X(a_ntr) X(a_on) X(a_oqh) X(a_oqt) X(a_oex) X(b_ntr) X(b_on) X(b_oqh) X(b_oqt) X(b_oex)
struct FastKeys {
    f64 ka, kb;  // key4 of the two order prices
};
// Order::doTransaction (order.cpp:84-107) under a predicate; returns the volume executed
__device__ inline i64 tx_sel(bool act, i64 size, i64& qh, i64& ex, i64 volume) {
    const i64 rem0 = size - ex > 0 ? size - ex : 0;
    const i64 rv = volume - qh;
    const bool pos = rv > 0;
    const bool full = pos && rem0 <= rv;
    const i64 n_ex = full ? size : (pos ? ex + rv : ex);
    const i64 n_qh = pos ? 0 : qh - volume;
    const i64 rem1 = size - n_ex > 0 ? size - n_ex : 0;
    qh = act ? n_qh : qh;
    ex = act ? n_ex : ex;
    return act ? rem0 - rem1 : 0;
}
// Book::UpdateOrder (book.cpp:102-141) of one side for one applied row: `lv` / `v` = last_volume / volume at the order's price
__device__ inline void update_order_sel(i32& on, i64 size, i64& qh, i64& qt, i64 ex, f64 k, i64 lv, i64 v, f64 tk0, f64 tk1, i64 tv0, i64 tv1) {
    const bool dead = on && ex >= size;
    on = dead ? 0 : on;
    const bool act = on && lv != 0;
    const i64 diff = lv - v;
    i64 trade_vol = 0;
    if (tv0 > 0 && tk0 == k) trade_vol = tv0;
    if (tv1 > 0 && tk1 == k) trade_vol = tv1;
    const i64 cancelled = diff - trade_vol;
    const bool gone = act && v == 0;
    const bool can = act && v != 0 && diff >= 0 && cancelled > 0;
    const bool behind = act && v != 0 && diff < 0;
    i64 nh = qh, nt = qt;
    if (can) {  // Order::doCancellation (order.cpp:51-82)
        if (qt == 0) {
            nh = qh - cancelled;
        } else {
            const f64 total = (f64)(qh + qt);
            nh = cvt_long_x86((f64)qh - ceil((f64)(cancelled * qh) / total));
            nt = cvt_long_x86((f64)qt - floor((f64)(cancelled * qt) / total));
        }
        if (nh < 0) { nt = (i64)((u64)nt + (u64)nh); nh = 0; }
        if (nt < 0) nt = 0;
    }
    nt = behind ? qt + diff : nt;  // quirk Q2: addVolumeBehind(negative)
    qh = gone ? 0 : nh;
    qt = gone ? 0 : nt;
}
// volume resting at the two order keys in one row
__device__ inline void row_volumes_k(const EnvCtx& c, bool a_on, bool b_on, f64 ka, f64 kb, const RowFull& L, i64& a_v, i64& b_v) {
    const int D = c.P.D;
    uint32_t va = 0, vb = 0;
    bool fa = false, fb = false;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        const f32 pa = l < D ? __uint_as_float(L.apx[l]) : 0.0f;
        const f32 pb = l < D ? __uint_as_float(L.bpx[l]) : 0.0f;
        const bool ha = pa != 0.0f && key4((f64)pa) == ka;  // price keys are unique per side (lob_validate_stream)
        const bool hb = pb != 0.0f && key4((f64)pb) == kb;
        va = ha ? L.avol[l] : va; fa = fa || ha;
        vb = hb ? L.bvol[l] : vb; fb = fb || hb;
    }
    a_v = (a_on && fa) ? (i64)(i32)va : 0;
    b_v = (b_on && fb) ? (i64)(i32)vb : 0;
}
// one pass: 0 = another event follows, 1 = the step is complete.  `L` = the level arrays of row t.rec_first.
template <class TH, class ROW>  // TrackHead64, or the whole Track entry; RowFull, or RowLev (levels across 16 lanes)
__device__ inline int pass_fast(const EnvCtx& c, EnvR& h, StepAgg& g, const TH& t, ROW& L, const FastKeys& K, const f64* spread_mean = nullptr) {
    const DevParams& P = c.P;
    h.pnl_step = 0.0;
    const f64 tp0 = (f64)t.tr_px[0], tp1 = (f64)t.tr_px[1];
    const i64 tv0 = (t.info & 3) > 0 ? t.tr_vol[0] : 0, tv1 = (t.info & 3) > 1 ? t.tr_vol[1] : 0;
    const f64 tk0 = key4(tp0), tk1 = key4(tp1);
    h.pf = t.rec_first;
    const f64 mp = h.mid;
    i64 au_vol = 0; f64 au_proxy = 0.0, au_value = 0.0;
    i64 bu_vol = 0; f64 bu_proxy = 0.0, bu_value = 0.0;
    // AskBook::ApplyTransactions: a trade at or above the reference price runs through the ask order priced at or below it
#define LOB_FP_ASK(TP, TV)                                                                                  \
    {                                                                                                       \
        const bool act = (TV) > 0 && !((TP) < mp) && h.a_on && h.a_opx <= (TP);                             \
        const i64 exec = tx_sel(act, h.a_osz, h.a_oqh, h.a_oex, (TV));                                      \
        au_vol -= exec;                                                                                     \
        const f64 pr = au_proxy + (h.a_opx - mp) * (f64)exec, va = au_value + h.a_opx * (f64)exec;          \
        au_proxy = act ? pr : au_proxy;                                                                     \
        au_value = act ? va : au_value;                                                                     \
        const bool done = act && h.a_oex >= h.a_osz;                                                        \
        h.a_on = done ? 0 : h.a_on;                                                                         \
        h.a_ntr += done ? 1 : 0;                                                                            \
    }
    // BidBook::ApplyTransactions: at or below it, through the bid order priced at or above
#define LOB_FP_BID(TP, TV)                                                                                  \
    {                                                                                                       \
        const bool act = (TV) > 0 && !((TP) > mp) && h.b_on && h.b_opx >= (TP);                             \
        const i64 exec = tx_sel(act, h.b_osz, h.b_oqh, h.b_oex, (TV));                                      \
        bu_vol += exec;                                                                                     \
        const f64 pr = bu_proxy + (mp - h.b_opx) * (f64)exec, va = bu_value - h.b_opx * (f64)exec;          \
        bu_proxy = act ? pr : bu_proxy;                                                                     \
        bu_value = act ? va : bu_value;                                                                     \
        const bool done = act && h.b_oex >= h.b_osz;                                                        \
        h.b_on = done ? 0 : h.b_on;                                                                         \
        h.b_ntr += done ? 1 : 0;                                                                            \
    }
    LOB_FP_ASK(tp0, tv0) LOB_FP_ASK(tp1, tv1)   // ascending prices
    LOB_FP_BID(tp1, tv1) LOB_FP_BID(tp0, tv0)   // descending
#undef LOB_FP_ASK
#undef LOB_FP_BID
    // UpdateBookProfiles: StashState, then ApplyChanges (-> UpdateOrder) for every applied row
    const int last_rec = h.rec_cur;
    const i64 a_lv = g.cv_a, b_lv = g.cv_b;
    for (int r = t.rec_first; r <= t.rec_last; r++) {
        if (r != t.rec_first) row_full_load(c, r, L);
        i64 a_v, b_v;
        row_volumes_k(c, h.a_on != 0, h.b_on != 0, K.ka, K.kb, L, a_v, b_v);
        update_order_sel(h.a_on, h.a_osz, h.a_oqh, h.a_oqt, h.a_oex, K.ka, a_lv, a_v, tk0, tk1, tv0, tv1);
        update_order_sel(h.b_on, h.b_osz, h.b_oqh, h.b_oqt, h.b_oex, K.kb, b_lv, b_v, tk0, tk1, tv0, tv1);
        g.cv_a = a_v;
        g.cv_b = b_v;
    }
    h.rec_last = last_rec;
    h.rec_cur = t.rec_last;
    h.mid_prev = h.mid;
    h.mid = t.mid;
    h.time_ms = t.time_ms;
    h.events += (i64)(t.rec_last - t.rec_first + 1);
    h.k++;
    // BookUtils::HandleAdverseSelection (book.cpp:551-592) against the touch of the new snapshot
    i64 ad_vol = 0; f64 ad_proxy = 0.0, ad_value = 0.0;
    {
        const f64 bap = (f64)t.bap, bbp = (f64)t.bbp, rp = h.mid_prev;
        const bool ha = h.a_on && h.a_opx <= bbp;
        {
            const i64 rem = h.a_osz - h.a_oex > 0 ? h.a_osz - h.a_oex : 0;
            const f64 pr = ad_proxy + (f64)rem * (h.a_opx - rp), va = ad_value + (f64)rem * h.a_opx;
            ad_vol -= ha ? rem : 0;
            ad_proxy = ha ? pr : ad_proxy;
            ad_value = ha ? va : ad_value;
            h.a_on = ha ? 0 : h.a_on;
            h.a_ntr += ha ? 1 : 0;
        }
        const bool hb = h.b_on && h.b_opx >= bap;
        {
            const i64 rem = h.b_osz - h.b_oex > 0 ? h.b_osz - h.b_oex : 0;
            const f64 pr = ad_proxy + (f64)rem * (rp - h.b_opx), va = ad_value - (f64)rem * h.b_opx;
            ad_vol += hb ? rem : 0;
            ad_proxy = hb ? pr : ad_proxy;
            ad_value = hb ? va : ad_value;
            h.b_on = hb ? 0 : h.b_on;
            h.b_ntr += hb ? 1 : 0;
        }
    }
    h.pnl_step += au_proxy + bu_proxy + ad_proxy;
    h.lo_vol_step += (i32)(bu_vol - au_vol + (ad_vol < 0 ? -ad_vol : ad_vol));
    h.ep_pnl += au_value + bu_value + ad_value;
    h.position += bu_vol + au_vol + ad_vol;  // RiskManager::Update
    check_orders(P, h);
    // the rest of the loop body of Base::performAction
    const f64 mpm = h.mid - h.mid_prev;
    h.pnl_step += (f64)h.position * mpm;
    h.momentum_pnl_step += (f64)h.position * mpm;
    g.r += get_reward(c, h, spread_mean);
    g.pnl += h.pnl_step;
    g.mpm += mpm;
    return (is_open(P, h.time_ms) && fabs(g.mpm) < 1e-5) ? 0 : 1;
}
// The event loop of a step with the fast pass.  `t` = the track entry of event e.k, `L` = the level arrays of record
// e.rec_cur + 1 (the first row the step applies), both requested by the caller ahead of time; every pass requests the next
// entry and the next row before it starts on its own.  TE = TrackHead64, or Track when the caller wants the last completed
// event's whole entry back in `t` (the state extraction reads its second half).  Returns the last pass's status (1: step
// complete, 2: out of data).
template <class TE, class ROW>
__device__ inline int event_loop_fast(const EnvCtx& c, EnvR& e, StepAgg& g, TE& t, ROW L) {
    EnvR h = e;  // registers from here to the end of the loop
    const FastKeys K{key4(h.a_opx), key4(h.b_opx)};
    c.mark(22);  // hot copy
    const int last_row = c.S.n_events - 1;
    int st;
    while (true) {
        // the next event's entry and first row are requested a pass ahead (wasted on a step's last pass: skipping them there on a
        // "step ends here" bit computed by the pre-pass measured SLOWER, 0.104 vs 0.098 ms -- the conditional requests split the
        // batch of loads)
        const TE tn = *reinterpret_cast<const TE*>(&c.track(h.k + 1));
        // the fast pass needs: the event inside the track, freshly placed orders behind it, its trade list whole, and the row
        // it was handed (a step's first row is the record after the current snapshot -- anything else is reloaded)
        const bool fast = h.k < g.n_track && g.cv_valid && (t.info & LOB_TRK_TRADES_OK);
        bool have_next_row = false;
        if (fast) {
            if (t.rec_first != h.rec_cur + 1) row_full_load(c, t.rec_first, L);
            ROW Ln = L;  // next pass's first row, in flight during this one
            { const int rn = t.rec_last + 1; row_full_load(c, rn < last_row ? rn : last_row, Ln); }
            c.mark(23);  // loop top: next entry / next row requested
            st = pass_fast(c, h, g, t, L, K, sizeof(TE) == sizeof(Track) ? &reinterpret_cast<const Track*>(&t)->spread_mean : nullptr);
            L = Ln;
            have_next_row = true;
            c.mark(24);  // the pass itself
        } else {
#define X(n) e.n = h.n;
            LOB_ENV_PASS_FIELDS(X)
#undef X
            TrackHead t32;
            t32.rec_first = t.rec_first; t32.rec_last = t.rec_last; t32.time_ms = t.time_ms; t32.tick_ap0 = t.tick_ap0;
            t32.tick_bp0 = t.tick_bp0; t32.info = t.info; t32.mid = t.mid;
            st = step_event<2>(c, e, g, t32);
            h = e;
        }
        if (st != 0) break;
        if (!have_next_row) row_full_load(c, h.rec_cur + 1 < last_row ? h.rec_cur + 1 : last_row, L);
        t = tn;
    }
    c.mark(25);  // waiting for the wave's slowest lane
#define X(n) e.n = h.n;
    LOB_ENV_PASS_FIELDS(X)
#undef X
    e.done = h.done;  // (the general pass sets it when the stream runs dry)
    c.mark(26);  // hot copy back
    return st;
}
// perform_action with the fast pass.  `first` = the level arrays of record e.rec_cur + 1, requested by the caller together with `cur`.
__device__ inline bool perform_action_fast(const EnvCtx& c, EnvR& e, int action, TrackHead64 t, const RowFull& cur, const RowFull& first) {
    StepAgg g;
    step_prologue(c, e, action, g, cur);
    if (event_loop_fast(c, e, g, t, first) == 2) return false;
    step_epilogue(c, e, g);
    return true;
}
// Intraday::getVariable (intraday.cpp:316-409): market variables come from the
// track entry of the last completed event, agent variables are computed here.
// `t` = state_track(c, e), fetched ONCE by the caller (by value: six 16-byte loads in flight) rather
// than once per variable.
__device__ inline Track state_track(const EnvCtx& c, const EnvR& e) { return c.track(e.k > 0 ? e.k - 1 : 0); }
__device__ inline f64 get_variable(const EnvCtx& c, const EnvR& e, int v, const Track& t) {
    const DevParams& P = c.P;
    switch (v) {
        case LOB_VAR_POS: return (f64)e.position / (f64)P.order_size;
        case LOB_VAR_SPD: return (f64)t.mv[LOB_MV_SPD];
        case LOB_VAR_MPM: return (f64)t.mv[LOB_MV_MPM];
        case LOB_VAR_IMB: return (f64)t.mv[LOB_MV_IMB];
        case LOB_VAR_SVL: return (f64)t.mv[LOB_MV_SVL];
        case LOB_VAR_VOL: return (f64)t.mv[LOB_MV_VOL];
        case LOB_VAR_RSI: return (f64)t.mv[LOB_MV_RSI];
        case LOB_VAR_VWAP: return (f64)t.mv[LOB_MV_VWAP];
        case LOB_VAR_A_DIST:
            if (e.a_on) return ((f64)e.a_otk - (f64)t.tick_ap0);
            return -100.0;
        case LOB_VAR_A_QUEUE:
            if (e.a_on) {
                f32 iq = (f32)e.a_oiq;
                f32 prog = (f32)e.a_oqh / (1.0f > iq ? 1.0f : iq);
                return 10.0 * (f64)(i64)prog;  // Book::queue_progress returns long
            }
            return -1.0;
        case LOB_VAR_B_DIST:
            if (e.b_on) return ((f64)t.tick_bp0 - (f64)e.b_otk);
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

// ---------------------------------------------------------------------------
// Market pre-pass: the agent-independent part of Intraday::Initialise and of
// every Intraday::NextState of the stream, once per episode.
struct MarketR {
    int cursor, time_ms, rec_cur, rec_last;
    f64 ap0, bp0, lap0, lbp0;   // best prices of the current / stashed snapshot (0 = undefined)
    i64 a_tv, b_tv;
    f64 a_obsval, b_obsval;
    i64 a_obsvol, b_obsvol;
    f64 ewma_up, ewma_down, tp_val;
    i64 records;
};

// utilities/maths.h:4-8: max(min(val, ub), lb) with std::min(a, b) = (b < a) ? b : a and
// std::max(a, b) = (a < b) ? b : a -- a NaN `val` (e.g. vwap over a window without trades: 0/0)
// comes back as NaN, exactly as in the reference.
__device__ inline f64 ulb(f64 val, f64 lb, f64 ub) {
    const f64 m = (ub < val) ? ub : val;
    return (m < lb) ? lb : m;
}

// Book::ApplyChanges (book.cpp:64-99) without the order part.
// (mk_apply_levels: the same on a row whose level arrays are in registers already)
__device__ inline void mk_apply_levels(const EnvCtx& c, MarketR& m, int rec, i32 time_ms, const uint32_t* wpa, const uint32_t* wva, const uint32_t* wpb, const uint32_t* wvb);
__device__ inline void mk_apply_row(const EnvCtx& c, MarketR& m, int rec) {
    const DevParams& P = c.P;
    const uint32_t* r = c.row(rec);
    // the whole row is fetched before it is looked at, every level array in 16-byte loads
    uint32_t wpa[LOB_MAX_DEPTH], wpb[LOB_MAX_DEPTH], wva[LOB_MAX_DEPTH], wvb[LOB_MAX_DEPTH];
    drec_levels(r + drec_ask_px(P.D, P.T), P.D, wpa);
    drec_levels(r + drec_ask_vol(P.D, P.T), P.D, wva);
    drec_levels(r + drec_bid_px(P.D, P.T), P.D, wpb);
    drec_levels(r + drec_bid_vol(P.D, P.T), P.D, wvb);
    mk_apply_levels(c, m, rec, (i32)r[LOB_REC_TIME], wpa, wva, wpb, wvb);
}
__device__ inline void mk_apply_levels(const EnvCtx& c, MarketR& m, int rec, i32 time_ms, const uint32_t* wpa, const uint32_t* wva, const uint32_t* wpb, const uint32_t* wvb) {
    const DevParams& P = c.P;
    f32 pa[LOB_MAX_DEPTH], pb[LOB_MAX_DEPTH];
    i32 va[LOB_MAX_DEPTH], vb[LOB_MAX_DEPTH];
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        const bool in = l < P.D;
        pa[l] = in ? __uint_as_float(wpa[l]) : 1.0f;
        pb[l] = in ? __uint_as_float(wpb[l]) : 1.0f;
        va[l] = in ? (i32)wva[l] : 0;
        vb[l] = in ? (i32)wvb[l] : 0;
    }
    bool bad = false;
#pragma unroll
    for (int l = 0; l < LOB_MAX_DEPTH; l++) {
        if (l < P.D) {
            bad |= !(pa[l] > 0.0f) || va[l] <= 0 || !(pb[l] > 0.0f) || vb[l] <= 0;
            m.a_tv += (i64)va[l];
            m.b_tv += (i64)vb[l];
        }
    }
    if (bad) c.err(LOB_ERR_BAD_LEVEL);
    m.ap0 = (f64)pa[0];
    m.bp0 = (f64)pb[0];
    m.rec_cur = rec;
    m.time_ms = time_ms;
}

// One record of the stream in registers (the pre-pass keeps the row it is about to apply and the one after it there: every
// book advances event by event, so the addresses are known an event ahead and no load is waited for where it is issued).
template <int TM>
struct PreRow {
    uint4 hdr;                          // time, flags
    uint4 lv[4][LOB_LEVEL_QUADS];       // ask px, ask vol, bid px, bid vol
    uint4 tr[(TM + 1) / 2];             // (price, volume) pairs
};
template <int TM>
__device__ inline void pre_row_issue(const EnvCtx& c, int rec, PreRow<TM>& R) {
    const DevParams& P = c.P;
    const int last = c.S.n_events - 1;
    const uint4* r4 = reinterpret_cast<const uint4*>(c.row(rec < last ? rec : last));  // (past the stream: the last row again, never looked at)
    const int d4 = drec_pad4(P.D) / 4;
    R.hdr = r4[0];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int q = 0; q < LOB_LEVEL_QUADS; q++) R.lv[a][q] = r4[1 + a * d4 + q];
#pragma unroll
    for (int q = 0; q < (TM + 1) / 2; q++) R.tr[q] = 2 * q < P.T ? r4[1 + 4 * d4 + q] : make_uint4(0, 0, 0, 0);
}
__device__ inline i32 pre_row_time(const EnvCtx& c, int rec) {
    const int last = c.S.n_events - 1;
    return (i32)c.row(rec < last ? rec : last)[LOB_REC_TIME];
}
// Intraday::UpdateBookProfiles as mk_update_book_profiles, the first row it applies taken from registers (`R0` = row m.cursor,
// `t1` = the time of the row after it); further rows of the event (same timestamp, invalid states) come from memory.
template <int TM>
__device__ inline bool mk_update_book_profiles_pre(const EnvCtx& c, MarketR& m, const PreRow<TM>& R0, i32 t1) {
    { int t = m.rec_cur; m.rec_cur = m.rec_last; m.rec_last = t; }
    { f64 t = m.ap0; m.ap0 = m.lap0; m.lap0 = t; }
    { f64 t = m.bp0; m.bp0 = m.lbp0; m.lbp0 = t; }
    bool regs = true;
    while (true) {
        if (m.cursor + 1 >= c.S.n_events) return false;
        const int rec = m.cursor;
        m.cursor++;
        m.records++;
        i32 next_time;
        if (regs) {
            uint32_t wpa[LOB_MAX_DEPTH], wpb[LOB_MAX_DEPTH], wva[LOB_MAX_DEPTH], wvb[LOB_MAX_DEPTH];
            drec_unpack(R0.lv[0], c.P.D, wpa); drec_unpack(R0.lv[1], c.P.D, wva);
            drec_unpack(R0.lv[2], c.P.D, wpb); drec_unpack(R0.lv[3], c.P.D, wvb);
            mk_apply_levels(c, m, rec, (i32)R0.hdr.x, wpa, wva, wpb, wvb);
            next_time = t1;
            regs = false;
        } else {
            mk_apply_row(c, m, rec);
            next_time = (i32)c.row(m.cursor)[LOB_REC_TIME];
        }
        if (next_time == m.time_ms) continue;  // !WillTimeChange()
        // BookUtils::IsValidState (book.cpp:612-625)
        const f64 mp = (m.ap0 + m.bp0) / 2.0;
        const bool has_a = key4(m.lap0) != key4(0.0), has_b = key4(m.lbp0) != key4(0.0);
        bool valid = true;
        if (has_a && has_b) {
            const f64 lm = (m.lap0 + m.lbp0) / 2.0;
            valid = ((m.ap0 - m.bp0) >= 0.0) && (mp > 0.0) && (fabs(mp - lm) < mp);
        }
        if (valid) break;
    }
    return true;
}

// Intraday::UpdateBookProfiles (intraday.cpp:275-313), market part.  Returns
// false when the depth stream has no row after the one being made current
// (Streamer::LoadNext fails, src/data/streamer.cpp:42-49).
__device__ inline bool mk_update_book_profiles(const EnvCtx& c, MarketR& m) {
    // StashState on both books: current <-> stashed
    { int t = m.rec_cur; m.rec_cur = m.rec_last; m.rec_last = t; }
    { f64 t = m.ap0; m.ap0 = m.lap0; m.lap0 = t; }
    { f64 t = m.bp0; m.bp0 = m.lbp0; m.lbp0 = t; }
    while (true) {
        if (m.cursor + 1 >= c.S.n_events) return false;
        const int rec = m.cursor;
        m.cursor++;
        m.records++;
        mk_apply_row(c, m, rec);
        if ((i32)c.row(m.cursor)[LOB_REC_TIME] == m.time_ms) continue;  // !WillTimeChange()
        // BookUtils::IsValidState (book.cpp:612-625)
        const f64 mp = (m.ap0 + m.bp0) / 2.0;
        const bool has_a = key4(m.lap0) != key4(0.0), has_b = key4(m.lbp0) != key4(0.0);
        bool valid = true;
        if (has_a && has_b) {
            const f64 lm = (m.lap0 + m.lbp0) / 2.0;
            valid = ((m.ap0 - m.bp0) >= 0.0) && (mp > 0.0) && (fabs(mp - lm) < mp);
        }
        if (valid) break;
    }
    return true;
}

// Window sums that survive ClearWindows() (quirk Q7), saved at episode start so that
// the exact sums at the point where the episode later stops can be regenerated.
__device__ inline void persist_io(const DevState& S, int b, bool save) {
    int i = 0;
    const int B = S.B;
#define X(n)                                                                                   \
    if (save) { S.persist[(size_t)(i + 0) * B + b] = S.n.sum[b]; S.persist[(size_t)(i + 1) * B + b] = S.n.mean[b]; S.persist[(size_t)(i + 2) * B + b] = S.n.s[b]; } \
    else { S.n.sum[b] = S.persist[(size_t)(i + 0) * B + b]; S.n.mean[b] = S.persist[(size_t)(i + 1) * B + b]; S.n.s[b] = S.persist[(size_t)(i + 2) * B + b]; }   \
    i += 3;
    X(f_midprice) X(f_volatility) X(f_ask_tx) X(f_bid_tx) X(spread_window) X(tp_mp)
#undef X
#define X(n)                                                            \
    if (save) S.persist[(size_t)i * B + b] = S.n.sum[b];                \
    else S.n.sum[b] = S.persist[(size_t)i * B + b];                     \
    i += 1;
    X(f_vwap_numer) X(f_vwap_denom)
#undef X
    if (save) {
        S.persist[(size_t)(i + 0) * B + b] = S.ewma_up[b];
        S.persist[(size_t)(i + 1) * B + b] = S.ewma_down[b];
        S.persist[(size_t)(i + 2) * B + b] = S.tp_val[b];
    } else {
        S.ewma_up[b] = S.persist[(size_t)(i + 0) * B + b];
        S.ewma_down[b] = S.persist[(size_t)(i + 1) * B + b];
        S.tp_val[b] = S.persist[(size_t)(i + 2) * B + b];
    }
}

// The agent-independent evolution of one book, resumable: `prepass_begin` is Initialise's ClearWindows
// and skip to market open; `prepass_run` then writes one Track entry per NextState until `k_stop`
// events exist or the stream runs dry, and leaves its registers in `st` for the next call
// (prepass_extend_kernel).  `write_track` false: only the window arithmetic (finalize_kernel's replay
// of the events an episode consumed, quirk Q7: sums survive ClearWindows()).
__device__ inline void prepass_begin(const EnvCtx& c, PrepState& st, BookMeta& M) {
    const DevParams& P = c.P;
    const DevState& S = c.S;
    const int b = c.b;
    MarketR m;
    m.cursor = 0; m.time_ms = 0; m.rec_cur = -1; m.rec_last = -1;
    m.ap0 = m.bp0 = m.lap0 = m.lbp0 = 0.0;
    m.a_tv = m.b_tv = 0;
    m.a_obsval = m.b_obsval = 0.0; m.a_obsvol = m.b_obsvol = 0;
    m.ewma_up = S.ewma_up[b]; m.ewma_down = S.ewma_down[b]; m.tp_val = S.tp_val[b];
    m.records = 0;
    // ClearWindows (base.cpp:145-163): deques emptied, running sums kept (quirk Q7)
#define X(n) S.n.cnt[b] = 0;
    LOB_ROLLING_MEANS(X)
    LOB_ACCUMULATORS(X)
#undef X
    M.n_track = 0; M.k_warm = -1; M.init_ok = 0; M.complete = 0; M._pad = 0;
    M.ex_first = -1; M.ex_cur = -1; M.ex_last = -1; M.ex_time = 0; M.ex_records = 0;
    bool ok = true;
    while (ok && !is_open(P, m.time_ms)) ok = mk_update_book_profiles(c, m);
    M.rec_cur0 = m.rec_cur; M.rec_last0 = m.rec_last; M.time0 = m.time_ms;
    M.mid0 = (m.ap0 + m.bp0) / 2.0; M.mid_prev0 = (m.lap0 + m.lbp0) / 2.0;
    if (!ok) { M.ex_cur = m.rec_cur; M.ex_last = m.rec_last; M.ex_time = m.time_ms; M.complete = 1; }
    st.cursor = m.cursor; st.time_ms = m.time_ms; st.rec_cur = m.rec_cur; st.rec_last = m.rec_last;
    st.ap0 = m.ap0; st.bp0 = m.bp0; st.lap0 = m.lap0; st.lbp0 = m.lbp0;
    st.a_tv = m.a_tv; st.b_tv = m.b_tv;
    st.ewma_up = m.ewma_up; st.ewma_down = m.ewma_down; st.tp_val = m.tp_val;
    st.records = m.records;
    st.k = 0;
    st.prev_first = m.rec_cur;  // Initialise: time_and_sales.SkipUntil(market time) drops everything up to here
}

// (prepass_begin without the windows: the two-wave pre-pass's book role; the window role clears the windows itself)
__device__ inline void prepass_begin_books(const EnvCtx& c, PrepState& st, BookMeta& M) {
    const DevParams& P = c.P;
    MarketR m;
    m.cursor = 0; m.time_ms = 0; m.rec_cur = -1; m.rec_last = -1;
    m.ap0 = m.bp0 = m.lap0 = m.lbp0 = 0.0;
    m.a_tv = m.b_tv = 0;
    m.a_obsval = m.b_obsval = 0.0; m.a_obsvol = m.b_obsvol = 0;
    m.ewma_up = m.ewma_down = m.tp_val = 0.0;
    m.records = 0;
    M.n_track = 0; M.k_warm = -1; M.init_ok = 0; M.complete = 0; M._pad = 0;
    M.ex_first = -1; M.ex_cur = -1; M.ex_last = -1; M.ex_time = 0; M.ex_records = 0;
    bool ok = true;
    while (ok && !is_open(P, m.time_ms)) ok = mk_update_book_profiles(c, m);
    M.rec_cur0 = m.rec_cur; M.rec_last0 = m.rec_last; M.time0 = m.time_ms;
    M.mid0 = (m.ap0 + m.bp0) / 2.0; M.mid_prev0 = (m.lap0 + m.lbp0) / 2.0;
    if (!ok) { M.ex_cur = m.rec_cur; M.ex_last = m.rec_last; M.ex_time = m.time_ms; M.complete = 1; }
    st.cursor = m.cursor; st.time_ms = m.time_ms; st.rec_cur = m.rec_cur; st.rec_last = m.rec_last;
    st.ap0 = m.ap0; st.bp0 = m.bp0; st.lap0 = m.lap0; st.lbp0 = m.lbp0;
    st.a_tv = m.a_tv; st.b_tv = m.b_tv;
    st.ewma_up = st.ewma_down = st.tp_val = 0.0;
    st.records = m.records;
    st.k = 0;
    st.prev_first = m.rec_cur;  // Initialise: time_and_sales.SkipUntil(market time) drops everything up to here
}

template <int TM>
__device__ inline void prepass_run(const EnvCtx& c, PrepState& st, BookMeta& M, int k_stop, bool write_track) {
    const DevParams& P = c.P;
    const DevState& S = c.S;
    const int b = c.b, B = S.B;
    MarketR m;
    m.cursor = st.cursor; m.time_ms = st.time_ms; m.rec_cur = st.rec_cur; m.rec_last = st.rec_last;
    m.ap0 = st.ap0; m.bp0 = st.bp0; m.lap0 = st.lap0; m.lbp0 = st.lbp0;
    m.a_tv = st.a_tv; m.b_tv = st.b_tv;
    m.a_obsval = m.b_obsval = 0.0; m.a_obsvol = m.b_obsvol = 0;
    m.ewma_up = st.ewma_up; m.ewma_down = st.ewma_down; m.tp_val = st.tp_val;
    m.records = st.records;
    int k = st.k;
    int prev_first = st.prev_first;
    // the ten windows of intraday.cpp:253-269: state in registers for the whole run
    RMReg w_mid, w_vol, w_spr, w_tp, w_atx, w_btx;
    AccReg w_vn, w_vd;
    rm_load(S.f_midprice, b, w_mid); rm_load(S.f_volatility, b, w_vol); rm_load(S.spread_window, b, w_spr);
    rm_load(S.tp_mp, b, w_tp); rm_load(S.f_ask_tx, b, w_atx); rm_load(S.f_bid_tx, b, w_btx);
    acc_load(S.f_vwap_numer, b, w_vn); acc_load(S.f_vwap_denom, b, w_vd);
    // Software pipeline: every book of the wave goes through its stream event by event, so all addresses of an event are known
    // one event ahead.  `R0` = the row the event applies first (row m.cursor), `t1` = the time of the row after it; the row of
    // the NEXT event and the time behind it are requested at the top of this one, the ring slots that fall out of the windows at
    // the next push right after this event's pushes.  An event that runs through more rows (same timestamp, invalid states)
    // reads them where it needs them and the pipeline restarts.
#ifndef LOB_PRE_TOUCH
#define LOB_PRE_TOUCH 0
#endif
#if LOB_PRE_TOUCH
    // (-DLOB_PRE_TOUCH=1, measured and not taken: holding the NEXT row in registers as well needs 128 registers for the two rows
    // and spills 164 bytes per lane to scratch -- 16.4 ms; with the row two events ahead only TOUCHED (one word of each of its
    // four 64-byte sectors, so that it is in L2 when its turn comes) and the row an event applies read at the top of that event
    // nothing spills, but the L2 round trip of 64 lanes x 14 divergent 16-byte loads is exposed in every event: 19.8 ms)
    PreRow<TM> R0;
    uint32_t nt[4] = {0u, 0u, 0u, 0u};
#else
    PreRow<TM> R0, R1;
#endif
    i32 t1 = 0, t2 = 0;
    bool piped = false;
    int band_px = 0, band_tk = 0;  // to_ticks_hint: the band of the last price / of the last tick count converted as a price
    rm_prep(S.f_midprice, B, b, w_mid); rm_prep(S.f_volatility, B, b, w_vol); rm_prep(S.spread_window, B, b, w_spr);
    rm_prep(S.tp_mp, B, b, w_tp); rm_prep(S.f_ask_tx, B, b, w_atx); rm_prep(S.f_bid_tx, B, b, w_btx);
    acc_prep(S.f_vwap_numer, B, b, w_vn); acc_prep(S.f_vwap_denom, B, b, w_vd);
    while (!M.complete && k < k_stop) {
        const int first = m.cursor;
#if LOB_PRE_TOUCH
        {
            const int last = c.S.n_events - 1;
            const uint32_t* r2 = c.row(first + 2 < last ? first + 2 : last);
            const int wl = P.Wd - 1;
            nt[0] = r2[0]; nt[1] = r2[16 < wl ? 16 : wl]; nt[2] = r2[32 < wl ? 32 : wl]; nt[3] = r2[wl];
        }
        pre_row_issue<TM>(c, first, R0);
        if (!piped) t1 = pre_row_time(c, first + 1);
#else
        if (!piped) { pre_row_issue<TM>(c, first, R0); t1 = pre_row_time(c, first + 1); }
        pre_row_issue<TM>(c, first + 1, R1);
        t2 = pre_row_time(c, first + 2);
#endif
        if (first < c.S.n_events && (R0.hdr.y & LOB_EVT_FLAG_TAS_DRY)) {
            // the time-and-sales stream has run dry (Streamer::LoadUntil fails, streamer.cpp:61-85): NextState returns
            // false before it touches the books -- out of data with nothing of this event applied (ex_first < 0)
            M.ex_first = -1; M.ex_cur = m.rec_cur; M.ex_last = m.rec_last; M.ex_time = m.time_ms; M.ex_records = 0;
            M.complete = 1;
            break;
        }
        f64 tp[TM];
        i64 tv[TM];
        if (prev_first + 1 == first) {  // (the usual case: the trades of the event's own first row)
#pragma unroll
            for (int i = 0; i < TM; i++) { tp[i] = 0.0; tv[i] = 0; }
            int ntr = 0;
            merge_trade_slots<TM>(c, R0.tr, ntr, tp, tv);
        } else {
            load_trades<TM>(c, prev_first + 1, first, tp, tv);
        }
        prev_first = first;
        const f64 mp = (m.ap0 + m.bp0) / 2.0;
        // observed transaction value / volume of Ask/BidBook::ApplyTransactions (book.cpp:394-400, 479-485)
        m.a_obsval = 0.0; m.a_obsvol = 0; m.b_obsval = 0.0; m.b_obsvol = 0;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            if (i >= P.T || tv[i] <= 0) continue;
            if (tp[i] < mp) continue;
            m.a_obsval += tp[i] * (f64)tv[i];
            m.a_obsvol += tv[i];
        }
#pragma unroll
        for (int ii = 0; ii < TM; ii++) {
            const int i = TM - 1 - ii;
            if (i >= P.T || tv[i] <= 0) continue;
            if (tp[i] > mp) continue;
            m.b_obsval += tp[i] * (f64)tv[i];
            m.b_obsvol += tv[i];
        }
        const i64 rec0 = m.records;
        if (!mk_update_book_profiles_pre<TM>(c, m, R0, t1)) {
            M.ex_first = first; M.ex_cur = m.rec_cur; M.ex_last = m.rec_last; M.ex_time = m.time_ms;
            M.ex_records = m.records - rec0;
            M.complete = 1;
            break;
        }
        if (m.lap0 == 0.0 || m.lbp0 == 0.0) c.err(LOB_ERR_UNDEF_PRICE);
        const f64 mid = (m.ap0 + m.bp0) / 2.0, lmid = (m.lap0 + m.lbp0) / 2.0;
        const int tick_ap0 = lobh::to_ticks_hint((*c.tk), m.ap0, band_px), tick_bp0 = lobh::to_ticks_hint((*c.tk), m.bp0, band_px);
        const i64 mpt = (i64)lobh::to_ticks_hint((*c.tk), mid, band_px);
        const f64 mpm = mid - lmid, sp = m.ap0 - m.bp0;
        // ten window pushes (intraday.cpp:253-269): the ring slots that fall out were fetched an event ago
        rm_apply_reg(S.f_midprice, B, b, w_mid, (f64)mpt);
        rm_apply_reg(S.f_volatility, B, b, w_vol, (f64)mpt);
        acc_apply_reg(S.f_vwap_numer, B, b, w_vn, m.a_obsval + m.b_obsval);
        acc_apply_reg(S.f_vwap_denom, B, b, w_vd, (f64)(m.a_obsvol + m.b_obsvol));
        rm_apply_reg(S.spread_window, B, b, w_spr, 0.0 > sp ? 0.0 : sp);
        // TargetPrice::update (src/market/target_price.cpp:44-71)
        f64 micro;
        {   // measure::microprice (measures.h:40-55)
            f64 div = (f64)(m.a_tv + m.b_tv);
            f64 mpm_a = (f64)m.a_tv * m.bp0, mpm_b = m.ap0 * (f64)m.b_tv;
            micro = (mpm_a + mpm_b) / div;
        }
        rm_apply_reg(S.tp_mp, B, b, w_tp, P.target_price == LOB_TP_MICROPRICE ? micro : mid);
        m.tp_val = w_tp.mean;
        {   // EWMA<double>::push (accumulators.cpp:157-163)
            f64 up = 0.0 > mpm ? 0.0 : mpm;
            f64 dn = fabs(0.0 < mpm ? 0.0 : mpm);
            m.ewma_up = (P.ewma_alpha * up) + ((1 - P.ewma_alpha) * m.ewma_up);
            m.ewma_down = (P.ewma_alpha * dn) + ((1 - P.ewma_alpha) * m.ewma_down);
        }
        rm_apply_reg(S.f_ask_tx, B, b, w_atx, (f64)m.a_obsvol);
        rm_apply_reg(S.f_bid_tx, B, b, w_btx, (f64)m.b_obsvol);
        // ... and those of the next event's pushes leave now
        rm_prep(S.f_midprice, B, b, w_mid); rm_prep(S.f_volatility, B, b, w_vol); rm_prep(S.spread_window, B, b, w_spr);
        rm_prep(S.tp_mp, B, b, w_tp); rm_prep(S.f_ask_tx, B, b, w_atx); rm_prep(S.f_bid_tx, B, b, w_btx);
        acc_prep(S.f_vwap_numer, B, b, w_vn); acc_prep(S.f_vwap_denom, B, b, w_vd);

        if (write_track) {
            Track t;
            t.rec_first = first; t.rec_last = m.rec_cur; t.time_ms = m.time_ms;
            t.tick_ap0 = tick_ap0; t.tick_bp0 = tick_bp0;
            {   // the event's merged trade list (entries are packed at the front, ascending key) and the touch it leaves behind
                int ntr = 0;
#pragma unroll
                for (int i = 0; i < TM; i++) ntr += (i < P.T && tv[i] > 0) ? 1 : 0;
                t.info = (ntr < 2 ? ntr : 2) | (ntr <= 2 ? LOB_TRK_TRADES_OK : 0);
                t.tr_px[0] = (f32)tp[0]; t.tr_vol[0] = tv[0];
                t.tr_px[1] = TM > 1 ? (f32)tp[TM > 1 ? 1 : 0] : 0.0f; t.tr_vol[1] = TM > 1 ? tv[TM > 1 ? 1 : 0] : 0;
                t.bap = (f32)m.ap0; t.bbp = (f32)m.bp0;
            }
            t.mid = mid; t.tp_val = m.tp_val; t.spread_mean = w_spr.mean;
            t.a_tv = m.a_tv; t.b_tv = m.b_tv;
            // Intraday::getVariable (intraday.cpp:316-409), the stream-only variables
            t.mv[LOB_MV_SPD] = (f32)ulb((f64)(tick_ap0 - tick_bp0), 0.0, 20.0);
            {
                const f64 front = (f64)mpt;  // just pushed
                i32 bi = w_mid.head - w_mid.cnt + 1;
                if (bi < 0) bi += S.f_midprice.w;
                // (a full window's oldest entry is the one the next push replaces: rm_prep has just asked for it)
                const f64 back = w_mid.cnt == S.f_midprice.w ? w_mid.old : S.f_midprice.ring[(size_t)bi * B + b];
                t.mv[LOB_MV_MPM] = (f32)ulb((f64)(lobh::to_ticks_hint((*c.tk), front, band_tk) - lobh::to_ticks_hint((*c.tk), back, band_tk)), -10.0, 10.0);
            }
            {
                f64 v_a = (f64)m.a_tv, v_b = (f64)m.b_tv;
                t.mv[LOB_MV_IMB] = (f32)((v_a + v_b) > 0 ? 5 * (v_b - v_a) / (v_b + v_a) : 0.0);
                f64 q_a = w_atx.sum, q_b = w_btx.sum;
                t.mv[LOB_MV_SVL] = (f32)((q_a + q_b) > 0 ? 5 * (q_b - q_a) / (q_a + q_b) : 0.0);
            }
            {
                f64 var = w_vol.s / (f64)(w_vol.cnt - 1);
                f64 sd = var > 0 ? sqrt(var) : 0.0;
                t.mv[LOB_MV_VOL] = (f32)ulb(5.0 * sd, 0.0, 10.0);
                f64 u = m.ewma_up, d = m.ewma_down;
                t.mv[LOB_MV_RSI] = (f32)((u + d) != 0.0 ? 5.0 * (u - d) / (u + d) : 0.0);
                f64 vw = w_vn.sum / w_vd.sum;
                t.mv[LOB_MV_VWAP] = (f32)ulb(vw / w_spr.mean, -10.0, 10.0);
                t.mv[7] = 0.0f;
            }
            c.track_w(k) = t;
            if (M.k_warm < 0 && w_atx.cnt == S.f_ask_tx.w && w_btx.cnt == S.f_bid_tx.w && w_vn.cnt == S.f_vwap_numer.w &&
                w_vd.cnt == S.f_vwap_denom.w && w_vol.cnt == S.f_volatility.w && w_mid.cnt == S.f_midprice.w &&
                w_tp.cnt == S.tp_mp.w && w_spr.cnt == S.spread_window.w)
                M.k_warm = k + 1;  // Intraday::Initialise stops pulling events here (intraday.cpp:119-128)
        }
        // ... and calls time_and_sales.SkipUntil(market time) once more (intraday.cpp:130): if this event went through
        // more than one depth row, the trades up to the last of them are dropped, not handed to the next event
        if (k + 1 == M.k_warm) prev_first = m.rec_cur;
        k++;
        piped = m.cursor == first + 1;  // one row applied: what was requested at the top is what the next event starts with
#if LOB_PRE_TOUCH
        t2 = (i32)nt[0];
        asm volatile("" ::"v"(nt[1]), "v"(nt[2]), "v"(nt[3]));  // (the touches are loads somebody waits for -- here, an event after they left)
        if (piped) t1 = t2;
#else
        if (piped) { R0 = R1; t1 = t2; }
#endif
    }
    rm_store(S.f_midprice, b, w_mid); rm_store(S.f_volatility, b, w_vol); rm_store(S.spread_window, b, w_spr);
    rm_store(S.tp_mp, b, w_tp); rm_store(S.f_ask_tx, b, w_atx); rm_store(S.f_bid_tx, b, w_btx);
    acc_store(S.f_vwap_numer, b, w_vn); acc_store(S.f_vwap_denom, b, w_vd);
    st.cursor = m.cursor; st.time_ms = m.time_ms; st.rec_cur = m.rec_cur; st.rec_last = m.rec_last;
    st.ap0 = m.ap0; st.bp0 = m.bp0; st.lap0 = m.lap0; st.lbp0 = m.lbp0;
    st.a_tv = m.a_tv; st.b_tv = m.b_tv;
    st.ewma_up = m.ewma_up; st.ewma_down = m.ewma_down; st.tp_val = m.tp_val;
    st.records = m.records;
    st.k = k;
    st.prev_first = prev_first;
    S.ewma_up[b] = m.ewma_up; S.ewma_down[b] = m.ewma_down; S.tp_val[b] = m.tp_val;
    if (write_track) {
        M.n_track = k;
        M.init_ok = M.k_warm > 0 ? 1 : 0;
    }
}


// ---- the pre-pass on TWO waves per 64 books ----------------------------------------------------------------------------
// prepass_run is one wave executing ~2 100 dependent instructions per event at ~9 clocks each, one wave per SIMD: the chip
// issues an instruction every ninth clock.  Half of those instructions are the book (row, trades, validity, tick conversions),
// half the ten windows and the state variables, and the second half needs nine numbers of the first.  So a block is two
// waves over the same 64 books: wave 0 ("books") runs Intraday::UpdateBookProfiles and hands every event's numbers over
// through LDS, wave 1 ("windows") is one event behind with the pushes and the variables; two such blocks share a SIMD
// (registers: the larger of the two roles) and fill each other's stalls.  One block barrier per event.  The arithmetic is
// prepass_run's, statement for statement; each role writes its own 64-byte half of the track entry.
// Block barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and store the wave has in flight
// -- here the row touches, the ring slots requested for the NEXT event and the track stores, all of which are meant to stay in
// flight across the hand-over.  (This compiler emits s_waitcnt lgkmcnt(0) + s_barrier for __syncthreads() in these kernels as
// well; the explicit form states the requirement.)
__device__ inline void lds_block_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
struct PreXch {  // what the book role hands the window role for one event (one buffer; struct-of-arrays: conflict-free)
    i64 mpt[64], a_obsvol[64], b_obsvol[64], a_tv[64], b_tv[64];
    f64 sp[64], mpm[64], tp_in[64], obsval[64];
    i32 spd[64], live[64];
    i32 go, _pad;
};
// The two roles are loops of their own (one loop body holding both kept both roles' state live: 340 registers spilled; as separate
// loops their live ranges are disjoint); they run the same number of block barriers.
// `on`: this lane takes part (block-uniform control flow: every lane runs the loop).
template <int TM>
__device__ inline void prepass_books_role(const DevParams* Pp, const DevState* Sp, int b, const uint32_t* rows, const TickLds* tk, PrepState* stp, BookMeta* Mp,
                                                int k_stop, bool on, PreXch* xch, int lane) {
    const DevParams& P = *Pp;
    const DevState& S = *Sp;
    const EnvCtx c(P, S, b, tk, rows);
    PrepState& st = *stp;
    BookMeta& M = *Mp;
    MarketR m;
    m.cursor = st.cursor; m.time_ms = st.time_ms; m.rec_cur = st.rec_cur; m.rec_last = st.rec_last;
    m.ap0 = st.ap0; m.bp0 = st.bp0; m.lap0 = st.lap0; m.lbp0 = st.lbp0;
    m.a_tv = st.a_tv; m.b_tv = st.b_tv;
    m.a_obsval = m.b_obsval = 0.0; m.a_obsvol = m.b_obsvol = 0;
    m.ewma_up = m.ewma_down = m.tp_val = 0.0;
    m.records = st.records;
    int k = st.k;
    int prev_first = st.prev_first;
    PreRow<TM> R0;
    uint32_t nt[4] = {0u, 0u, 0u, 0u};
    i32 t1 = 0;
    bool piped = false;
    int band_px = 0;
    // every window is pushed once per event from a count of zero (prepass_begin): all are full after max(w) events -- where
    // Intraday::Initialise stops pulling events (intraday.cpp:119-128).  The book role needs that event (the closing
    // SkipUntil, below) and does not see the windows.
    int w_max = S.f_midprice.w;
    {
        const int ws[] = {S.f_volatility.w, S.f_ask_tx.w, S.f_bid_tx.w, S.spread_window.w, S.tp_mp.w, S.f_vwap_numer.w, S.f_vwap_denom.w};
        for (int w : ws) w_max = w > w_max ? w : w_max;
    }
#pragma unroll 1
    for (int it = 0;; it++) {
        PreXch& X = xch[it & 1];
        int live = 0;
        if (on && !M.complete && k < k_stop) {
            do {  // (one event; `break` = the stream ends here, nothing produced)
                const int first = m.cursor;
                {   // the row two events ahead is touched (it is in L2 when its turn comes; its first word is the time the event
                    // before it compares with); the other wave of the SIMD runs while this one waits for its own row
                    const int last = c.S.n_events - 1;
                    const uint32_t* r2 = c.row(first + 2 < last ? first + 2 : last);
                    const int wl = P.Wd - 1;
                    nt[0] = r2[0]; nt[1] = r2[16 < wl ? 16 : wl]; nt[2] = r2[32 < wl ? 32 : wl]; nt[3] = r2[wl];
                }
                pre_row_issue<TM>(c, first, R0);
                if (!piped) t1 = pre_row_time(c, first + 1);
                if (first < c.S.n_events && (R0.hdr.y & LOB_EVT_FLAG_TAS_DRY)) {
                    M.ex_first = -1; M.ex_cur = m.rec_cur; M.ex_last = m.rec_last; M.ex_time = m.time_ms; M.ex_records = 0;
                    M.complete = 1;
                    break;
                }
                f64 tp[TM];
                i64 tv[TM];
                if (prev_first + 1 == first) {
#pragma unroll
                    for (int i = 0; i < TM; i++) { tp[i] = 0.0; tv[i] = 0; }
                    int ntr = 0;
                    merge_trade_slots<TM>(c, R0.tr, ntr, tp, tv);
                } else {
                    load_trades<TM>(c, prev_first + 1, first, tp, tv);
                }
                prev_first = first;
                const f64 mp = (m.ap0 + m.bp0) / 2.0;
                m.a_obsval = 0.0; m.a_obsvol = 0; m.b_obsval = 0.0; m.b_obsvol = 0;
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    if (i >= P.T || tv[i] <= 0) continue;
                    if (tp[i] < mp) continue;
                    m.a_obsval += tp[i] * (f64)tv[i];
                    m.a_obsvol += tv[i];
                }
#pragma unroll
                for (int ii = 0; ii < TM; ii++) {
                    const int i = TM - 1 - ii;
                    if (i >= P.T || tv[i] <= 0) continue;
                    if (tp[i] > mp) continue;
                    m.b_obsval += tp[i] * (f64)tv[i];
                    m.b_obsvol += tv[i];
                }
                const i64 rec0 = m.records;
                if (!mk_update_book_profiles_pre<TM>(c, m, R0, t1)) {
                    M.ex_first = first; M.ex_cur = m.rec_cur; M.ex_last = m.rec_last; M.ex_time = m.time_ms;
                    M.ex_records = m.records - rec0;
                    M.complete = 1;
                    break;
                }
                if (m.lap0 == 0.0 || m.lbp0 == 0.0) c.err(LOB_ERR_UNDEF_PRICE);
                const f64 mid = (m.ap0 + m.bp0) / 2.0, lmid = (m.lap0 + m.lbp0) / 2.0;
                const int tick_ap0 = lobh::to_ticks_hint((*c.tk), m.ap0, band_px), tick_bp0 = lobh::to_ticks_hint((*c.tk), m.bp0, band_px);
                const i64 mpt = (i64)lobh::to_ticks_hint((*c.tk), mid, band_px);
                const f64 mpm = mid - lmid, sp = m.ap0 - m.bp0;
                f64 micro;
                {   // measure::microprice (measures.h:40-55)
                    f64 div = (f64)(m.a_tv + m.b_tv);
                    f64 mpm_a = (f64)m.a_tv * m.bp0, mpm_b = m.ap0 * (f64)m.b_tv;
                    micro = (mpm_a + mpm_b) / div;
                }
                X.mpt[lane] = mpt; X.a_obsvol[lane] = m.a_obsvol; X.b_obsvol[lane] = m.b_obsvol; X.a_tv[lane] = m.a_tv; X.b_tv[lane] = m.b_tv;
                X.sp[lane] = sp; X.mpm[lane] = mpm; X.tp_in[lane] = P.target_price == LOB_TP_MICROPRICE ? micro : mid;
                X.obsval[lane] = m.a_obsval + m.b_obsval;
                X.spd[lane] = tick_ap0 - tick_bp0;
                {   // the entry's first half: which rows, time, touch, midprice, the merged trade list
                    TrackHead64 t;
                    t.rec_first = first; t.rec_last = m.rec_cur; t.time_ms = m.time_ms;
                    t.tick_ap0 = tick_ap0; t.tick_bp0 = tick_bp0;
                    int ntr = 0;
#pragma unroll
                    for (int i = 0; i < TM; i++) ntr += (i < P.T && tv[i] > 0) ? 1 : 0;
                    t.info = (ntr < 2 ? ntr : 2) | (ntr <= 2 ? LOB_TRK_TRADES_OK : 0);
                    t.mid = mid;
                    t.tr_px[0] = (f32)tp[0]; t.tr_vol[0] = tv[0];
                    t.tr_px[1] = TM > 1 ? (f32)tp[TM > 1 ? 1 : 0] : 0.0f; t.tr_vol[1] = TM > 1 ? tv[TM > 1 ? 1 : 0] : 0;
                    t.bap = (f32)m.ap0; t.bbp = (f32)m.bp0;
                    *reinterpret_cast<TrackHead64*>(&c.track_w(k)) = t;
                }
                if (M.k_warm < 0 && k + 1 >= w_max) M.k_warm = k + 1;
                if (k + 1 == M.k_warm) prev_first = m.rec_cur;  // Initialise's closing SkipUntil (intraday.cpp:130)
                k++;
                live = 1;
                piped = m.cursor == first + 1;
                asm volatile("" ::"v"(nt[1]), "v"(nt[2]), "v"(nt[3]));
                if (piped) t1 = (i32)nt[0];
            } while (false);
        }
        X.live[lane] = live;
        const int more = __any(on && !M.complete && k < k_stop) ? 1 : 0;
        if (lane == 0) X.go = more;
        lds_block_barrier();
        if (!more) break;
    }
    st.cursor = m.cursor; st.time_ms = m.time_ms; st.rec_cur = m.rec_cur; st.rec_last = m.rec_last;
    st.ap0 = m.ap0; st.bp0 = m.bp0; st.lap0 = m.lap0; st.lbp0 = m.lbp0;
    st.a_tv = m.a_tv; st.b_tv = m.b_tv;
    st.records = m.records;
    st.k = k;
    st.prev_first = prev_first;
    M.n_track = k;
    M.init_ok = M.k_warm > 0 ? 1 : 0;
}
__device__ inline void prepass_windows_role(const DevParams* Pp, const DevState* Sp, int b, const TickLds* tk, PrepState* stp, bool on, PreXch* xch, int lane) {
    const DevParams& P = *Pp;
    const DevState& S = *Sp;
    const int B = S.B;
    PrepState& st = *stp;
    f64 ewma_up = st.ewma_up, ewma_down = st.ewma_down, tp_val = st.tp_val;
    int k = st.k;
    int band_tk = 0;
    Track* track_b = S.track + (size_t)b * (size_t)S.track_len;
    RMReg w_mid, w_vol, w_spr, w_tp, w_atx, w_btx;
    AccReg w_vn, w_vd;
    if (on) {
        rm_load(S.f_midprice, b, w_mid); rm_load(S.f_volatility, b, w_vol); rm_load(S.spread_window, b, w_spr);
        rm_load(S.tp_mp, b, w_tp); rm_load(S.f_ask_tx, b, w_atx); rm_load(S.f_bid_tx, b, w_btx);
        acc_load(S.f_vwap_numer, b, w_vn); acc_load(S.f_vwap_denom, b, w_vd);
        rm_prep(S.f_midprice, B, b, w_mid); rm_prep(S.f_volatility, B, b, w_vol); rm_prep(S.spread_window, B, b, w_spr);
        rm_prep(S.tp_mp, B, b, w_tp); rm_prep(S.f_ask_tx, B, b, w_atx); rm_prep(S.f_bid_tx, B, b, w_btx);
        acc_prep(S.f_vwap_numer, B, b, w_vn); acc_prep(S.f_vwap_denom, B, b, w_vd);
    }
#pragma unroll 1
    for (int it = 0;; it++) {
        PreXch& X = xch[it & 1];
        lds_block_barrier();
        const int go = X.go;
        if (on && X.live[lane]) {
            const i64 mpt = X.mpt[lane], a_obsvol = X.a_obsvol[lane], b_obsvol = X.b_obsvol[lane], a_tv = X.a_tv[lane], b_tv = X.b_tv[lane];
            const f64 sp = X.sp[lane], mpm = X.mpm[lane], tp_in = X.tp_in[lane], obsval = X.obsval[lane];
            const int spd_ticks = X.spd[lane];
            rm_apply_reg(S.f_midprice, B, b, w_mid, (f64)mpt);
            rm_apply_reg(S.f_volatility, B, b, w_vol, (f64)mpt);
            acc_apply_reg(S.f_vwap_numer, B, b, w_vn, obsval);
            acc_apply_reg(S.f_vwap_denom, B, b, w_vd, (f64)(a_obsvol + b_obsvol));
            rm_apply_reg(S.spread_window, B, b, w_spr, 0.0 > sp ? 0.0 : sp);
            rm_apply_reg(S.tp_mp, B, b, w_tp, tp_in);  // TargetPrice::update (src/market/target_price.cpp:44-71)
            tp_val = w_tp.mean;
            {   // EWMA<double>::push (accumulators.cpp:157-163)
                f64 up = 0.0 > mpm ? 0.0 : mpm;
                f64 dn = fabs(0.0 < mpm ? 0.0 : mpm);
                ewma_up = (P.ewma_alpha * up) + ((1 - P.ewma_alpha) * ewma_up);
                ewma_down = (P.ewma_alpha * dn) + ((1 - P.ewma_alpha) * ewma_down);
            }
            rm_apply_reg(S.f_ask_tx, B, b, w_atx, (f64)a_obsvol);
            rm_apply_reg(S.f_bid_tx, B, b, w_btx, (f64)b_obsvol);
            rm_prep(S.f_midprice, B, b, w_mid); rm_prep(S.f_volatility, B, b, w_vol); rm_prep(S.spread_window, B, b, w_spr);
            rm_prep(S.tp_mp, B, b, w_tp); rm_prep(S.f_ask_tx, B, b, w_atx); rm_prep(S.f_bid_tx, B, b, w_btx);
            acc_prep(S.f_vwap_numer, B, b, w_vn); acc_prep(S.f_vwap_denom, B, b, w_vd);
            {   // the entry's second half: target price, spread mean, cumulative volumes, the stream-only state variables
                struct __attribute__((aligned(16))) Half { f64 tp_val, spread_mean; i64 a_tv, b_tv; f32 mv[8]; } h;
                h.tp_val = tp_val; h.spread_mean = w_spr.mean; h.a_tv = a_tv; h.b_tv = b_tv;
                h.mv[LOB_MV_SPD] = (f32)ulb((f64)spd_ticks, 0.0, 20.0);
                {
                    const f64 front = (f64)mpt;
                    i32 bi = w_mid.head - w_mid.cnt + 1;
                    if (bi < 0) bi += S.f_midprice.w;
                    const f64 back = w_mid.cnt == S.f_midprice.w ? w_mid.old : S.f_midprice.ring[(size_t)bi * B + b];
                    h.mv[LOB_MV_MPM] = (f32)ulb((f64)(lobh::to_ticks_hint((*tk), front, band_tk) - lobh::to_ticks_hint((*tk), back, band_tk)), -10.0, 10.0);
                }
                {
                    f64 v_a = (f64)a_tv, v_b = (f64)b_tv;
                    h.mv[LOB_MV_IMB] = (f32)((v_a + v_b) > 0 ? 5 * (v_b - v_a) / (v_b + v_a) : 0.0);
                    f64 q_a = w_atx.sum, q_b = w_btx.sum;
                    h.mv[LOB_MV_SVL] = (f32)((q_a + q_b) > 0 ? 5 * (q_b - q_a) / (q_a + q_b) : 0.0);
                }
                {
                    f64 var = w_vol.s / (f64)(w_vol.cnt - 1);
                    f64 sd = var > 0 ? sqrt(var) : 0.0;
                    h.mv[LOB_MV_VOL] = (f32)ulb(5.0 * sd, 0.0, 10.0);
                    f64 u = ewma_up, d = ewma_down;
                    h.mv[LOB_MV_RSI] = (f32)((u + d) != 0.0 ? 5.0 * (u - d) / (u + d) : 0.0);
                    f64 vw = w_vn.sum / w_vd.sum;
                    h.mv[LOB_MV_VWAP] = (f32)ulb(vw / w_spr.mean, -10.0, 10.0);
                    h.mv[7] = 0.0f;
                }
                static_assert(sizeof(Half) == 64, "the second half of a Track entry");
                *reinterpret_cast<Half*>(reinterpret_cast<char*>(&track_b[(size_t)(k & S.track_mask)]) + 64) = h;
            }
            k++;
        }
        if (!go) break;
    }
    if (on) {
        rm_store(S.f_midprice, b, w_mid); rm_store(S.f_volatility, b, w_vol); rm_store(S.spread_window, b, w_spr);
        rm_store(S.tp_mp, b, w_tp); rm_store(S.f_ask_tx, b, w_atx); rm_store(S.f_bid_tx, b, w_btx);
        acc_store(S.f_vwap_numer, b, w_vn); acc_store(S.f_vwap_denom, b, w_vd);
        st.ewma_up = ewma_up; st.ewma_down = ewma_down; st.tp_val = tp_val;
        S.ewma_up[b] = ewma_up; S.ewma_down[b] = ewma_down; S.tp_val[b] = tp_val;
    }
}
template <int TM>
__device__ inline void prepass_run2(const EnvCtx& c, PrepState& st, BookMeta& M, int k_stop, bool on, int role, PreXch* xch, int lane) {
    if (role == 0) prepass_books_role<TM>(&c.P, &c.S, c.b, c.rows, c.tk, &st, &M, k_stop, on, xch, lane);
    else prepass_windows_role(&c.P, &c.S, c.b, c.tk, &st, on, xch, lane);
}

#endif
