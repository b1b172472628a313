// Event-stream record layout + the synthetic stream generator
// (SURVEY.md §8d, configs C1-C4).  Header-only; the same integer-only code
// runs on the host (lob_gen_stream_host, the oracle harnesses) and on the
// device (gen_events_kernel) so both produce bit-identical records.
//
// A record is what one reference Intraday::NextState consumes
// (src/environment/intraday.cpp:225-272): the per-price aggregated trades
// since the previous depth snapshot (include/data/records.h:30-37) and the
// new depth snapshot (include/data/records.h:20-28).
#ifndef LOB_STREAM_H
#define LOB_STREAM_H

#include <stdint.h>
#include <string.h>

#include "../../include/lob_engine.h"

#if defined(__HIPCC__)
#define LOB_HD __host__ __device__ inline
#else
#define LOB_HD inline
#endif

// ---- record layout (32-bit words) -----------------------------------------
LOB_HD int lob_rec_words(int depth, int max_trades) {
    int w = 2 + 4 * depth + 2 * max_trades;
    return (w + 3) & ~3;  // 16-byte aligned records: dwordx4 loads
}
#define LOB_REC_TIME 0
#define LOB_REC_FLAGS 1
LOB_HD int lob_rec_ask_px(int, int) { return 2; }
LOB_HD int lob_rec_ask_vol(int D, int) { return 2 + D; }
LOB_HD int lob_rec_bid_px(int D, int) { return 2 + 2 * D; }
LOB_HD int lob_rec_bid_vol(int D, int) { return 2 + 3 * D; }
LOB_HD int lob_rec_trade_px(int D, int) { return 2 + 4 * D; }
LOB_HD int lob_rec_trade_vol(int D, int T) { return 2 + 4 * D + T; }

// ---- counter-based RNG ------------------------------------------------------
// splitmix64 finaliser; used both for the generator (sequential stream per
// book) and for the policy draws of the learner (DESIGN.md "RNG").
LOB_HD uint64_t lob_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// draw number `ctr` of stream `stream` under `seed`
LOB_HD uint64_t lob_rng(uint64_t seed, uint64_t stream, uint64_t ctr) {
    return lob_mix64(seed + stream * 0xD1B54A32D192ED03ull + (ctr + 1) * 0x9E3779B97F4A7C15ull);
}

LOB_HD float lob_tick_to_price_f32(int ticks) {
    // float32 price on the 0.1 grid, exactly what `stof("700.1")` yields in the
    // reference CSV reader (quirk Q8, src/data/basic.cpp:51-52).
    return (float)((double)ticks / 10.0);
}

LOB_HD uint32_t lob_f32_bits(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
LOB_HD float lob_bits_f32(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

LOB_HD int lob_uniform(uint32_t r, int lo, int hi) {  // U{lo..hi}
    uint32_t span = (uint32_t)(hi - lo + 1);
    return lo + (int)(((uint64_t)r * span) >> 32);
}

// Generator state carried from event to event for one book.
struct lob_gen_state {
    uint64_t ctr;
    int bb;      // best-bid tick
    int spread;  // current spread in ticks
};

LOB_HD void lob_gen_init(const lob_gen_params& g, lob_gen_state& s) {
    s.ctr = 0;
    s.bb = g.start_ticks;
    s.spread = 1;
}

// Emit event `e` of book `book_id` into rec[0..rec_words).
LOB_HD void lob_gen_event(const lob_gen_params& g, int D, int T, uint64_t book_id, int e,
                          lob_gen_state& s, uint32_t* rec) {
    const int W = lob_rec_words(D, T);
    for (int i = 0; i < W; i++) rec[i] = 0;
    rec[LOB_REC_TIME] = (uint32_t)(g.t0_ms + e * g.dt_ms);
    rec[LOB_REC_FLAGS] = 0;

    // fixed number of draws per event keeps streams aligned across branches
    uint64_t r0 = lob_rng(g.seed, book_id, s.ctr++);
    uint64_t r1 = lob_rng(g.seed, book_id, s.ctr++);
    uint64_t r2 = lob_rng(g.seed, book_id, s.ctr++);

    // ---- trades against the PREVIOUS snapshot (none before the first) ----
    int n_tr = 0;
    int tpx[2];
    int tvol[2];
    if (e > 0 && (int)(r0 & 0xFFFF) < g.trade_prob_q16) {
        int side_ask = (int)((r0 >> 16) & 1);  // 1: buyer-initiated, prints on the ask side
        int lvl = ((int)((r0 >> 17) & 0xFFFF) < g.touch_prob_q16) ? 0 : 1;
        int px = side_ask ? (s.bb + s.spread + lvl) : (s.bb - lvl);
        int vol = lob_uniform((uint32_t)(r1 & 0xFFFFFFFFu), g.trade_min, g.trade_max);
        tpx[0] = px;
        tvol[0] = vol;
        n_tr = 1;
        if (T >= 2 && (int)((r0 >> 33) & 0xFFFF) < g.trade2_prob_q16) {
            int lvl2 = ((int)((r0 >> 49) & 0x7FFF) * 2 < g.touch_prob_q16) ? 0 : 1;
            int px2 = side_ask ? (s.bb - lvl2) : (s.bb + s.spread + lvl2);
            int vol2 = lob_uniform((uint32_t)(r1 >> 32), g.trade_min, g.trade_max);
            // keep slots ascending by price
            if (px2 < px) {
                tpx[1] = tpx[0];
                tvol[1] = tvol[0];
                tpx[0] = px2;
                tvol[0] = vol2;
            } else {
                tpx[1] = px2;
                tvol[1] = vol2;
            }
            n_tr = 2;
        }
    }
    const int o_tp = lob_rec_trade_px(D, T), o_tv = lob_rec_trade_vol(D, T);
    for (int i = 0; i < n_tr && i < T; i++) {
        rec[o_tp + i] = lob_f32_bits(lob_tick_to_price_f32(tpx[i]));
        rec[o_tv + i] = (uint32_t)tvol[i];
    }

    // ---- evolve the book -------------------------------------------------
    if (e > 0 && (int)(r2 & 0xFFFF) < g.move_prob_q16) {
        s.bb += ((r2 >> 16) & 1) ? 1 : -1;
        if (s.bb < g.min_ticks) s.bb = g.min_ticks;
        if (s.bb > g.max_ticks) s.bb = g.max_ticks;
    }
    s.spread = ((int)((r2 >> 17) & 0xFFFF) < g.spread2_prob_q16) ? 2 : 1;

    const int o_ap = lob_rec_ask_px(D, T), o_av = lob_rec_ask_vol(D, T);
    const int o_bp = lob_rec_bid_px(D, T), o_bv = lob_rec_bid_vol(D, T);
    for (int l = 0; l < D; l++) {
        uint64_t rv = lob_rng(g.seed, book_id, s.ctr++);
        rec[o_ap + l] = lob_f32_bits(lob_tick_to_price_f32(s.bb + s.spread + l));
        rec[o_bp + l] = lob_f32_bits(lob_tick_to_price_f32(s.bb - l));
        rec[o_av + l] = (uint32_t)lob_uniform((uint32_t)(rv & 0xFFFFFFFFu), g.vol_min, g.vol_max);
        rec[o_bv + l] = (uint32_t)lob_uniform((uint32_t)(rv >> 32), g.vol_min, g.vol_max);
    }
}

#endif  // LOB_STREAM_H
