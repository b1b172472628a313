// TEST INFRASTRUCTURE ONLY.  lobh::to_ticks_hint (the pre-pass's tick conversion: band hint + first loop iteration written
// out) against lobh::to_ticks_t (the restatement of Market::ToTicks pinned on the reference) on the CPU, both compiled from the
// engine's own header: every venue table, prices on / next to every band boundary and every tick of a window, random prices,
// tick counts used as prices (the mpm variable), NaN / inf / negative / zero, and EVERY possible hint for each price.
//   g++ -std=c++17 -O1 -ffp-contract=off -o ticks_diff tests/host_env/ticks_diff.cpp rl_markets_amd/csrc/lob_host.cpp && ./ticks_diff
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>

#include "../../rl_markets_amd/csrc/lob_internal.h"

int main() {
    const char* tickers[] = {"HSBA.L", "AAL.L", "VOD.L", "BNP.PA", "ASML.AS", "NOVO.CO", "NOKIA.HE", "ERIC.ST", "EQNR.OL", "ISP.MI", "SAN.MC", "SAP.DE", "NESN.VX", "ABI.BR"};
    std::mt19937_64 rng(7);
    long n = 0, bad = 0;
    int venues = 0;
    for (const char* tk : tickers) {
        lob_market m;
        if (lob_market_preset(tk, &m) != LOB_OK) continue;
        venues++;
        lobh::TickTable t;
        lobh::build_tick_table(m, t);
        auto check = [&](double price) {
            const int want = lobh::to_ticks_t(t, price);
            for (int h = 0; h < t.n; h++) {
                int hint = h;
                const int got = lobh::to_ticks_hint(t, price, hint);
                n++;
                if (got != want || hint < 0 || hint >= t.n) {
                    if (bad++ < 10) printf("MISMATCH %s price %.17g hint %d: %d vs %d\n", tk, price, h, got, want);
                }
                // the hint the call leaves must be good for the same price again
                int h2 = hint;
                if (lobh::to_ticks_hint(t, price, h2) != want) bad++;
            }
        };
        const double specials[] = {0.0, -0.0, -1.0, 1e-300, 1e300, INFINITY, -INFINITY, NAN, 5e-5, 1e-4};
        for (double s : specials) check(s);
        for (int i = 0; i < t.n; i++) {
            const double lb = t.lb[i], tick = t.tick[i];
            for (int k = -3; k <= 3; k++) {
                check(lb + k * tick); check(lb + k * tick * 0.5); check(lb + k * tick * 0.25);
                check(nextafter(lb + k * tick * 0.5, INFINITY)); check(nextafter(lb + k * tick * 0.5, -INFINITY));
                check((double)(float)(lb + k * tick));   // prices are float32 in the stream (quirk Q8)
            }
            const double ub = i + 1 < t.n ? t.lb[i + 1] : lb * 4 + 10;
            for (int r = 0; r < 4000; r++) {
                const double u = (double)(rng() >> 11) * (1.0 / 9007199254740992.0);
                const double p = lb + u * (ub - lb);
                check(p); check((double)(float)p);
                check(floor(p / tick) * tick); check((double)(float)(floor(p / tick) * tick));
            }
        }
        for (int r = 0; r < 20000; r++) check((double)(int)(rng() % 200000));  // tick counts converted as prices (Intraday::getVariable mpm)
        // ToPrice: every band boundary +- a few ticks, random and negative tick counts, every hint
        auto check_p = [&](int ticks) {
            const double want = lobh::to_price_t(t, ticks);
            for (int h = 0; h < t.n; h++) {
                int hint = h;
                const double got = lobh::to_price_hint(t, ticks, hint);
                n++;
                if (memcmp(&got, &want, 8) != 0 || hint < 0 || hint >= t.n) {
                    if (bad++ < 10) printf("MISMATCH %s ticks %d hint %d: %.17g vs %.17g\n", tk, ticks, h, got, want);
                }
            }
        };
        for (int i = 0; i < t.n; i++)
            for (int k = -4; k <= 4; k++) check_p((int)t.cum[i] + k);
        for (int r = 0; r < 40000; r++) check_p((int)(rng() % 300000) - 1000);
        check_p(0); check_p(-1); check_p(2147483647); check_p(-2147483647 - 1);
    }
    printf("ticks_diff: %d venues, %ld conversions, %ld mismatches\n", venues, n, bad);
    if (bad || venues < 5) return 1;
    printf("ticks_diff OK\n");
    return 0;
}
