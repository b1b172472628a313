// Internal declarations shared by the host and device halves of the engine.
#ifndef LOB_INTERNAL_H
#define LOB_INTERNAL_H

#include <stdint.h>

#include <string>

#include "../../include/lob_engine.h"
#include "lob_stream.h"

void lob_set_error(const std::string& s);

// Tick maths shared by host (lob_to_ticks...) and device (state extraction,
// quoting).  Restates reference Market::ToTicks / ToPrice / tick_size
// (src/market/market.cpp:78-138) over the ascending band table; the
// int/long "+= double" truncations of quirk Q16 are kept.
namespace lobh {

struct TickTable {
    int n;
    double lb[LOB_MAX_BANDS];
    double tick[LOB_MAX_BANDS];
    int64_t cum[LOB_MAX_BANDS];  // tts_ keys: cumulative ticks at each band's lower bound
    // What the reference's loops have accumulated when they REACH band i, having walked through the
    // full bands below it -- independent of the argument, so ToTicks / ToPrice can start at the
    // argument's own band (one or two divisions instead of one per band below the price):
    int pt[LOB_MAX_BANDS];       // ToTicks: `int ticks += double` over bands 0..i-1
    double pp[LOB_MAX_BANDS];    // ToPrice: `double price +=` over bands 0..i-1
};

LOB_HD void build_tick_table(const lob_market& m, TickTable& t) {
    t.n = m.n_bands;
    for (int i = 0; i < LOB_MAX_BANDS; i++) {
        t.lb[i] = i < m.n_bands ? m.band_lb[i] : 0.0;
        t.tick[i] = i < m.n_bands ? m.band_tick[i] : 1.0;
        t.cum[i] = 0;
    }
    // Market ctor, src/market/market.cpp:27-36: long acc_ticks += double
    int64_t acc = 0;
    for (int i = 1; i < m.n_bands; i++) {
        acc = (int64_t)((double)acc + (m.band_lb[i] - m.band_lb[i - 1]) / m.band_tick[i - 1]);
        t.cum[i] = acc;
    }
    t.pt[0] = 0;
    t.pp[0] = 0.0;
    for (int i = 1; i < LOB_MAX_BANDS; i++) {
        if (i < m.n_bands) {
            // the loop bodies of to_ticks_t / to_price_t for a full band i-1 (ub = next lower bound)
            t.pt[i] = (int)((double)t.pt[i - 1] + (t.lb[i] - t.lb[i - 1]) / t.tick[i - 1]);
            t.pp[i] = t.pp[i - 1] + ((double)t.cum[i] - (double)t.cum[i - 1]) * t.tick[i - 1];
        } else {
            t.pt[i] = t.pt[i - 1];
            t.pp[i] = t.pp[i - 1];
        }
    }
}

template <class TT> LOB_HD double tick_size_t(const TT& t, double price) {
    // prev(upper_bound(price)): the last band whose lower bound is <= price
    double ts = t.tick[0];
    for (int i = 1; i < t.n; i++)
        if (t.lb[i] <= price) ts = t.tick[i];
    return ts;
}

template <class TT> LOB_HD int to_ticks_t(const TT& t, double price) {
    // i0: the last band whose lower bound is <= price.  Every band below it is a full band for this
    // price (the loop condition holds and ub = the next lower bound), so the loop can start at i0
    // with the prefix it would have accumulated; bands i0, i0 + 1, ... then run exactly as written
    // in the reference (src/market/market.cpp:78-102).  A NaN price gives i0 = 0 and breaks at once.
    int i0 = 0;
    for (int i = 1; i < t.n; i++)
        if (t.lb[i] <= price) i0 = i;
    const double half = t.tick[i0] / 2.0;  // tick_size(price) / 2
    int ticks = t.pt[i0];
    for (int i = i0; i < t.n; i++) {
        const double lb = t.lb[i], tk = t.tick[i];
        if (!(price + tk / 2.0 > lb)) break;
        double ub;
        if (i == t.n - 1 || price < t.lb[i + 1]) ub = price + half;
        else ub = t.lb[i + 1];
        ticks = (int)((double)ticks + (ub - lb) / tk);  // `int += double`
    }
    return ticks;
}

// to_ticks_t with the band search replaced by a check of `hint` (the band the caller's previous price fell in; prices of one
// book stay in one band for hours) and the loop's first iteration written out: the same operations on the same operands.
// i0 = h exactly when (h == 0 or lb[h] <= price) and (h is the last band or not lb[h + 1] <= price), the lower bounds being
// strictly ascending (std::map keys); a NaN price fails the first test unless h == 0, where the search would also leave 0.
template <class TT> LOB_HD int to_ticks_hint(const TT& t, double price, int& hint) {
    int i0 = hint;
    const bool ok = (i0 == 0 || t.lb[i0] <= price) && (i0 + 1 >= t.n || !(t.lb[i0 + 1] <= price));
    if (!ok) {
        i0 = 0;
        for (int i = 1; i < t.n; i++)
            if (t.lb[i] <= price) i0 = i;
        hint = i0;
    }
    const double tk0 = t.tick[i0], lb0 = t.lb[i0];
    const double half = tk0 / 2.0;
    int ticks = t.pt[i0];
    if (!(price + tk0 / 2.0 > lb0)) return ticks;
    {
        double ub;
        if (i0 == t.n - 1 || price < t.lb[i0 + 1]) ub = price + half;
        else ub = t.lb[i0 + 1];
        ticks = (int)((double)ticks + (ub - lb0) / tk0);
    }
    for (int i = i0 + 1; i < t.n; i++) {  // (entered only by a price within half a tick of the next band)
        const double lb = t.lb[i], tk = t.tick[i];
        if (!(price + tk / 2.0 > lb)) break;
        double ub;
        if (i == t.n - 1 || price < t.lb[i + 1]) ub = price + half;
        else ub = t.lb[i + 1];
        ticks = (int)((double)ticks + (ub - lb) / tk);
    }
    return ticks;
}

template <class TT> LOB_HD double to_price_t(const TT& t, int ticks) {
    // same idea: bands whose upper tick count is <= ticks are full, start after them
    int i0 = 0;
    for (int i = 1; i < t.n; i++)
        if (t.cum[i] <= (int64_t)ticks) i0 = i;
    double price = t.pp[i0];
    for (int i = i0; i < t.n; i++) {
        if (!((int64_t)ticks > t.cum[i])) break;
        double ub;
        if (i == t.n - 1 || (int64_t)ticks < t.cum[i + 1]) ub = (double)ticks;
        else ub = (double)t.cum[i + 1];
        price += (ub - (double)t.cum[i]) * t.tick[i];
    }
    return price;
}

// to_price_t with the band search replaced by a check of `hint` (see to_ticks_hint): i0 = h exactly when (h == 0 or
// cum[h] <= ticks) and (h is the last band or not cum[h + 1] <= ticks), the cumulative tick counts being non-decreasing --
// where two bands share a count the search keeps the LAST of them, so a hint is only accepted when the band after it starts
// strictly above `ticks`.
template <class TT> LOB_HD double to_price_hint(const TT& t, int ticks, int& hint) {
    int i0 = hint;
    const bool ok = (i0 == 0 || t.cum[i0] <= (int64_t)ticks) && (i0 + 1 >= t.n || !(t.cum[i0 + 1] <= (int64_t)ticks));
    if (!ok) {
        i0 = 0;
        for (int i = 1; i < t.n; i++)
            if (t.cum[i] <= (int64_t)ticks) i0 = i;
        hint = i0;
    }
    double price = t.pp[i0];
    for (int i = i0; i < t.n; i++) {
        if (!((int64_t)ticks > t.cum[i])) break;
        double ub;
        if (i == t.n - 1 || (int64_t)ticks < t.cum[i + 1]) ub = (double)ticks;
        else ub = (double)t.cum[i + 1];
        price += (ub - (double)t.cum[i]) * t.tick[i];
    }
    return price;
}

inline double tick_size(const lob_market& m, double price) {
    TickTable t;
    build_tick_table(m, t);
    return tick_size_t(t, price);
}
inline int to_ticks(const lob_market& m, double price) {
    TickTable t;
    build_tick_table(m, t);
    return to_ticks_t(t, price);
}
inline double to_price(const lob_market& m, int ticks) {
    TickTable t;
    build_tick_table(m, t);
    return to_price_t(t, ticks);
}

}  // namespace lobh

#endif
