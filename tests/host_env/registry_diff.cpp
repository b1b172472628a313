// TEST INFRASTRUCTURE ONLY.  The tile registry's invariant (rl_markets_amd/csrc/lob_tiles.h tile_register, the engine's own
// function compiled as host code through tests/host_env/shim; registry_block's per-slot loop (lob_fast.h) and registry_scan_block (lob_kernels.h) are
// restated serially here): after every step's registrations and scan,
//     two registered tiles share a weight index and are NOT the same tile (tiling, action, cell)
//         =>  both carry their bit in mk_amb,
// which is what lets trace_lane_kernel compare tile indices only where those bits are set.  Random batches of memo slots
// (triples clustered so that cells coincide, twins 2 048 apart, tiny tables where every index is shared), registered in
// random order over several "steps"; three tiles on one index in every arrival order are the interesting case.
//   g++ -std=c++17 -O1 -Itests/host_env/shim -o registry_diff tests/host_env/registry_diff.cpp && ./registry_diff [trials]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <random>
#include <vector>

#include "../../rl_markets_amd/csrc/lob_tiles.h"

static std::mt19937_64 rng(20260926);
static int ri(int lo, int hi) { return lo + (int)(rng() % (uint64_t)((long long)hi - lo + 1)); }

static i32 index_of(uint32_t M, const uint32_t* rnd, const int4& q, int a, int j) {
    uint32_t sum = 0;
    int base = j;
    sum = mod_add(sum, rnd[(tile_coord(q.x, base) + 449 * 0) & 2047], M); base += 2 * j;
    sum = mod_add(sum, rnd[(tile_coord(q.y, base) + 449 * 1) & 2047], M); base += 2 * j;
    sum = mod_add(sum, rnd[(tile_coord(q.z, base) + 449 * 2) & 2047], M);
    sum = mod_add(sum, rnd[(j + 449 * 3) & 2047], M);
    return tile_index(sum, rnd[2048 + a], M);
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 60;
    long n_amb_tiles = 0, n_tiles = 0, bad = 0;
    for (int tr = 0; tr < trials && bad == 0; tr++) {
        const uint32_t Ms[] = {4099u, 1u << 14, 100003u, 1u << 20};
        const uint32_t M = Ms[ri(0, 3)];
        std::vector<uint32_t> rnd(2048 + 32);
        for (auto& x : rnd) x = (uint32_t)(rng() % M);
        const int n_slots = ri(8, 60), n_steps = ri(1, 5);
        DevState S;
        memset(&S, 0, sizeof S);
        S.ow_slots = 1 << 16;
        std::vector<u64> ow((size_t)S.ow_slots, ~0ull);
        std::vector<uint32_t> amb_bits(M / 32 + 1, 0), mk_amb((size_t)n_slots * 9, 0);
        S.amb_cap = 1 << 16;
        std::vector<i32> amb_new(2 * (size_t)S.amb_cap), amb_new_n(2, 0), amb_flag(1, 0), mk_ident((size_t)n_slots * 4), mk_tiles((size_t)n_slots * 288), mk_all;
        std::vector<i64> counters(8, 0);
        S.ow_tab = ow.data(); S.amb_bits = amb_bits.data(); S.amb_new = amb_new.data(); S.amb_new_n = amb_new_n.data(); S.amb_flag = amb_flag.data();
        S.counters = counters.data(); S.mk_ident = mk_ident.data(); S.mk_amb = mk_amb.data();
        // distinct triples, clustered; some are another one's twin 2 048 k away (same cells everywhere: same indices)
        std::vector<int4> ids;
        while ((int)ids.size() < n_slots) {
            int4 q = make_int4(ri(-40, 40), ri(-40, 40), ri(-40, 40), 0);
            if (!ids.empty() && ri(0, 5) == 0) {
                q = ids[ri(0, (int)ids.size() - 1)];
                (ri(0, 2) == 0 ? q.x : ri(0, 1) ? q.y : q.z) += 2048 * ri(-2, 2);
            }
            bool dup = false;
            for (auto& o : ids) dup |= o.x == q.x && o.y == q.y && o.z == q.z;
            if (!dup) ids.push_back(q);
        }
        for (int s = 0; s < n_slots; s++) { mk_ident[s * 4] = ids[s].x; mk_ident[s * 4 + 1] = ids[s].y; mk_ident[s * 4 + 2] = ids[s].z; mk_ident[s * 4 + 3] = 0; }
        std::vector<int> order(n_slots);
        for (int s = 0; s < n_slots; s++) order[s] = s;
        std::shuffle(order.begin(), order.end(), rng);
        int done = 0, par = 0;
        for (int step = 0; step < n_steps; step++) {
            const int upto = step == n_steps - 1 ? n_slots : std::min(n_slots, done + ri(1, n_slots));
            // ---- registry_block: a "wave" per new slot, its 288 tiles in (action, tiling) order of the lanes' loop ----
            for (; done < upto; done++) {
                const int s = order[done];
                uint32_t bits[9] = {0};
                for (int k = 0; k < 5; k++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int a = (lane >= 32 ? 5 : 0) + k, j = lane & 31;
                        if (a >= 9) continue;
                        const i32 tile = index_of(M, rnd.data(), ids[s], a, j);
                        mk_tiles[(size_t)s * 288 + a * 32 + j] = tile;
                        const int r = tile_register(S, ids[s], s, a, j, tile, par);
                        if (r < 0) { printf("registry full\n"); return 2; }
                        if (r > 0) bits[a] |= 1u << j;
                    }
                for (int a = 0; a < 9; a++) mk_amb[(size_t)s * 9 + a] = bits[a];
                mk_all.push_back(s);
            }
            // ---- registry_scan_block ----
            const int n_new = amb_new_n[par];
            amb_new_n[par ^ 1] = 0;
            for (int e = 0; e < n_new; e++) {
                const i32 f = amb_new[(size_t)par * S.amb_cap + e];
                for (int s : mk_all)
                    for (int t = 0; t < 288; t++)
                        if (mk_tiles[(size_t)s * 288 + t] == f) mk_amb[(size_t)s * 9 + t / 32] |= 1u << (t % 32);
            }
            par ^= 1;
            // ---- the invariant, over everything registered so far ----
            std::map<i32, std::vector<int>> by_index;  // index -> tiles (s * 288 + a * 32 + j)
            for (int s : mk_all)
                for (int t = 0; t < 288; t++) by_index[mk_tiles[(size_t)s * 288 + t]].push_back(s * 288 + t);
            for (auto& kv : by_index) {
                bool ambiguous = false;
                for (size_t x = 0; x < kv.second.size() && !ambiguous; x++)
                    for (size_t y = x + 1; y < kv.second.size() && !ambiguous; y++) {
                        const int tx = kv.second[x], ty = kv.second[y];
                        const bool same = tx % 288 == ty % 288 && tile_same_cell(ids[tx / 288], ids[ty / 288], tx % 32);
                        ambiguous = !same;
                    }
                for (int t : kv.second) {
                    const bool flagged = (mk_amb[(size_t)(t / 288) * 9 + (t % 288) / 32] >> (t % 32)) & 1u;
                    n_tiles++;
                    n_amb_tiles += flagged;
                    if (ambiguous && !flagged) {
                        printf("MISSING flag: trial %d step %d index %d tile (slot %d, action %d, tiling %d) shares it with a different tile\n", tr, step, kv.first,
                               t / 288, (t % 288) / 32, t % 32);
                        bad++;
                    }
                    if (!ambiguous && flagged) {
                        printf("SPURIOUS flag: trial %d step %d index %d tile (slot %d, action %d, tiling %d)\n", tr, step, kv.first, t / 288, (t % 288) / 32, t % 32);
                        bad++;
                    }
                }
            }
        }
    }
    printf("trials %d: %ld tile checks, %ld flagged ambiguous\n", trials, n_tiles, n_amb_tiles);
    if (bad) { printf("registry_diff FAILED\n"); return 1; }
    printf("registry_diff OK\n");
    return 0;
}
