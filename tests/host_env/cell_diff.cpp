// TEST INFRASTRUCTURE ONLY.  The arithmetic trace_lane_kernel relies on (rl_markets_amd/csrc/lob_tiles.h), on the CPU, compiled
// from the engine's own device header through tests/host_env/shim:
//   (1) tile_same_cell_mask (cell numbers, early exit) == tile_same_cell tiling by tiling (through tile_coord) for random pairs of
//       group-0 triples -- near each other, 2 048 k apart, across zero, at the ends of the plain range;
//   (2) same cell  =>  same weight index for every action (the tile IS the same table-term sum), with the real hash
//       (tile_base_m / tile_index over a random table and random table sizes);
//   (3) the converse is NOT claimed: different cells may share an index (the tile registry's business) -- counted, not an error.
//   g++ -std=c++17 -O1 -Itests/host_env/shim -o cell_diff tests/host_env/cell_diff.cpp && ./cell_diff [cases]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../rl_markets_amd/csrc/lob_tiles.h"

static std::mt19937_64 rng(987654321);
static int ri(int lo, int hi) { return lo + (int)(rng() % (uint64_t)((long long)hi - lo + 1)); }

// the index of tile (q0, q1, q2, action a, tiling j) as memo_kernel / apply_kernel compute it
static i32 index_of(uint32_t M, const uint32_t* rnd, const int q[3], int a, int j) {
    uint32_t sum = 0;
    int base = j;
    for (int i = 0; i < 3; i++) {
        sum = mod_add(sum, rnd[(tile_coord(q[i], base) + 449 * i) & 2047], M);
        base += 2 * j;
    }
    sum = mod_add(sum, rnd[(j + 449 * 3) & 2047], M);
    return tile_index(sum, rnd[2048 + a], M);
}

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 300000;
    long n_same = 0, n_reject = 0, n_coincide = 0, bad = 0;
    for (long cs = 0; cs < cases && bad < 10; cs++) {
        const uint32_t Ms[] = {4099u, 1u << 16, 100003u, 20000000u, 2147483647u};
        const uint32_t M = Ms[ri(0, 4)];
        uint32_t rnd[2048 + 32];
        for (int i = 0; i < 2048 + 32; i++) rnd[i] = (uint32_t)(rng() % M);
        int a[3], b[3];
        const int kind = ri(0, 5);
        for (int i = 0; i < 3; i++) {
            const int centre = kind == 4 ? LOB_TILE_PLAIN_MIN + ri(0, 4000) : kind == 5 ? 2147483647 - ri(0, 4000) : ri(-5000, 5000);
            a[i] = centre;
            int d = kind == 0 ? ri(-40, 40) : kind == 1 ? ri(-2, 2) : kind == 2 ? 2048 * ri(-3, 3) + ri(-33, 33) : ri(-3000, 3000);
            if (kind == 4 && d < 0) d = -d;
            if (kind == 5 && d > 0) d = -d;
            b[i] = centre + d;
        }
        const uint32_t mask = tile_same_cell_mask(a[0], a[1], a[2], b[0], b[1], b[2]);
        if (mask == 0) n_reject++;
        const int4 ia = make_int4(a[0], a[1], a[2], 0), ib = make_int4(b[0], b[1], b[2], 0);
        for (int j = 0; j < 32; j++) {
            const bool same = tile_same_cell(ia, ib, j);
            if (same != (((mask >> j) & 1u) != 0)) {
                printf("MISMATCH mask: a=(%d,%d,%d) b=(%d,%d,%d) tiling %d: mask bit %u, tile_coord says %d\n", a[0], a[1], a[2], b[0], b[1], b[2], j, (mask >> j) & 1u, (int)same);
                bad++;
            }
            for (int act = 0; act < 9; act++) {
                const bool eq = index_of(M, rnd, a, act, j) == index_of(M, rnd, b, act, j);
                if (same && !eq) {
                    printf("MISMATCH index: same cell but different indices: a=(%d,%d,%d) b=(%d,%d,%d) tiling %d action %d M %u\n", a[0], a[1], a[2], b[0], b[1], b[2], j, act, M);
                    bad++;
                }
                if (!same && eq) n_coincide++;
            }
            n_same += same;
        }
    }
    printf("cases %ld: same-cell tilings %ld, pairs rejected at once %ld, index coincidences across cells %ld (the registry's)\n", cases, n_same, n_reject, n_coincide);
    if (bad) { printf("cell_diff FAILED\n"); return 1; }
    printf("cell_diff OK\n");
    return 0;
}
