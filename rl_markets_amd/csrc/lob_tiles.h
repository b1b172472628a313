// CMAC tile arithmetic of the group-0 (and every other) tile group: pure integer functions, no wave intrinsics, so that
// tests/host_env can compile them as host code (tests/host_env/cell_diff.cpp) -- the product includes this file through
// lob_learn.h only.   tiles()/hash_UNH: src/rl/tiles.cpp:31-75,130-169.
#ifndef LOB_TILES_H
#define LOB_TILES_H

#include <hip/hip_runtime.h>

#include "lob_state.h"

// The hash of tiles.cpp:152-168 is (sum of table terms) mod M.  The table is stored already reduced
// mod M (lob_engine.hip), and the running sum is kept reduced: s, x < M < 2^31, so s + x fits 32 bits
// and one conditional subtraction (min with the wrapped difference) restores s < M.  Same residue as
// reducing the 64-bit sum once, no 64-bit arithmetic and no division on the device.
__device__ inline uint32_t mod_add(uint32_t s, uint32_t x, uint32_t M) {
    s += x;
    const uint32_t d = s - M;  // wraps to >= 2^31 when s < M
    return d < s ? d : s;
}

// (int) floor(x * num_tilings) with x86 `cvttsd2si` semantics (NaN / out of range -> INT_MIN), the
// quantised coordinate of tiles.cpp:50-53.
__device__ inline int tile_quant(f32 x) {
    const f32 fq = floorf(x * 32.0f);
    return (fq >= -2147483648.0f && fq < 2147483648.0f) ? (int)fq : (int)0x80000000;
}

// Coordinate of quantised value q in the tiling whose offset for this variable is `base`
// (tiles.cpp:61-64):  q >= base: q - ((q - base) % 32);  else: q + 1 + ((base - q - 1) % 32) - 32.
// Without overflow both branches are base + 32 floor((q - base) / 32) = base + ((q - base) & ~31)
// (write base - q - 1 = 32 m + r in the second).  base <= 31 * 25, so the subtractions can only
// overflow for q within 1024 of INT_MIN -- in practice q == INT_MIN, a NaN variable -- and there the
// compiled reference wraps (two's complement) and takes a signed remainder: spelt out.
__device__ inline int tile_coord(int q, int base) {
    if (__builtin_expect(q < (int)0x80000400, 0)) {
        if (q >= base) return (int)((uint32_t)q - (uint32_t)((int)((uint32_t)q - (uint32_t)base) % 32));
        return (int)((uint32_t)q + 1u + (uint32_t)((int)((uint32_t)base - (uint32_t)q - 1u) % 32) - 32u);
    }
    return base + ((q - base) & ~31);
}

// Reduced sum of the table terms of tiling j that do not depend on the action: the nf float
// coordinates and the tiling index (tiles.cpp:50-70).  `v` = the group's float sub-array
// (State::populateFeatures passes &state_vars[0] or &state_vars[3]); `rndM` = the table mod M.
// General form (any lane, any group): used where speed does not matter.
__device__ inline uint32_t tile_base_m(uint32_t M, const f32* v, int nf, int j, const uint32_t* rndM) {
    uint32_t sum = 0;
    int base = j;  // j * (1 + 2 i), built up by adding 2 j per coordinate
    for (int i = 0; i < nf; i++) {
        sum = mod_add(sum, rndM[(tile_coord(tile_quant(v[i]), base) + 449 * i) & 2047], M);
        base += 2 * j;
    }
    return mod_add(sum, rndM[(j + 449 * nf) & 2047], M);
}
// (base + term) mod M with both operands already reduced: one add, one compare, one select.
// hash_UNH sums the table terms and reduces once (tiles.cpp:165-168); reducing the
// action-independent partial sum and the action term separately gives the same residue.
__device__ inline i32 tile_index(uint32_t base_m, uint32_t term_m, uint32_t M) {
    const uint32_t s = base_m + term_m;  // < 2^32: M < 2^31
    return (i32)(s >= M ? s - M : s);
}

// Two group-0 triples (quantised coordinates a, b; every coordinate >= LOB_TILE_PLAIN_MIN: tile_coord's plain branch) fall in
// the same cell of tiling j when, for the three coordinates i, the cell numbers ((q_i - (1 + 2 i) j) >> 5) agree mod 64: the
// hash reads the cell's coordinate & 2047 (tile_base_m), so two tiles of one tiling and one action whose cells agree that
// far ARE one table term sum, one weight index.  Bit j of the result: same cell in tiling j.  (trace_lane_kernel: which tiles
// of an old generation the new state re-sets or clears, without the indices; the registry's tile_same_cell is the same
// test through tile_coord, for any coordinate.)
#define LOB_TILE_PLAIN_MIN ((int)0x80000400)
__device__ inline uint32_t tile_same_cell_mask(int a0, int a1, int a2, int b0, int b1, int b2) {
    const uint32_t d0 = (uint32_t)(a0 - b0) & 2047u, d1 = (uint32_t)(a1 - b1) & 2047u, d2 = (uint32_t)(a2 - b2) & 2047u;
    // (a coordinate 32 .. 2015 away mod 2048 is in another cell of every tiling: the cell numbers differ by 1 .. 63 mod 64)
    if (d0 - 32u <= 1983u || d1 - 32u <= 1983u || d2 - 32u <= 1983u) return 0;
    uint32_t hit = 0;
#pragma unroll 4
    for (int j = 0; j < 32; j++) {
        const int x = ((a0 - j) >> 5) ^ ((b0 - j) >> 5), y = ((a1 - 3 * j) >> 5) ^ ((b1 - 3 * j) >> 5), z = ((a2 - 5 * j) >> 5) ^ ((b2 - 5 * j) >> 5);
        if (((x | y | z) & 63) == 0) hit |= 1u << j;
    }
    return hit;
}
// ... for any coordinates (the wrap-around branch of tile_coord included), one tiling
__device__ inline bool tile_same_cell(const int4& x, const int4& y, int j) {
    int base = j;
    bool same = ((tile_coord(x.x, base) ^ tile_coord(y.x, base)) & 2047) == 0; base += 2 * j;
    same = same && ((tile_coord(x.y, base) ^ tile_coord(y.y, base)) & 2047) == 0; base += 2 * j;
    return same && ((tile_coord(x.z, base) ^ tile_coord(y.z, base)) & 2047) == 0;
}

// Tile registry (lob_state.h ow_tab): enter tile (slot s of triple `id`, action a, tiling j) with weight index `tile`.  Returns 1 if the index
// is (now) known to be ambiguous, 0 if not, -1 if the table had no room (the slot then stays unregistered: the lane path
// leaves every book that meets it to the wave-per-book kernel).
// FIRST: the first probe's compare-and-swap and the index's word of the ambiguity bitmap have been asked for by the caller
// (tile_register_ask: registry_block puts a lane's five tiles in flight at once instead of waiting for them one after the
// other); `first_word` may be older than the swap -- an index that turns ambiguous between the two is on amb_new, and the scan
// behind the registry marks it in every registered slot.
template <bool FIRST>
__device__ __forceinline__ int tile_register_impl(const DevState& S, const int4& id, int s, int a, int j, i32 tile, int par, u64 first_old, uint32_t first_word) {
    const u64 want = ((u64)(uint32_t)tile << 32) | (u64)(uint32_t)(s * (LOB_N_ACTIONS * 32) + a * 32 + j);
    const uint32_t mask = (uint32_t)(S.ow_slots - 1);
    uint32_t h = ((uint32_t)tile * 2654435761u) & mask;
    const uint32_t bit = 1u << ((uint32_t)tile & 31);
    uint32_t* word = S.amb_bits + ((uint32_t)tile >> 5);
    for (int probe = 0; probe < 64; probe++) {
        const u64 old = (FIRST && probe == 0) ? first_old : (u64)atomicCAS((unsigned long long*)&S.ow_tab[h], ~0ull, (unsigned long long)want);
        if (old == ~0ull) return (((FIRST && probe == 0) ? first_word : *word) & bit) ? 1 : 0;  // first tile on this index (the bit is clear unless the table lost an entry)
        if ((uint32_t)(old >> 32) == (uint32_t)tile) {
            const uint32_t ref = (uint32_t)old;
            const int s2 = (int)(ref / (LOB_N_ACTIONS * 32)), r2 = (int)(ref % (LOB_N_ACTIONS * 32));
            bool same = (r2 >> 5) == a && (r2 & 31) == j;
            if (same && s2 != s) same = tile_same_cell(id, *reinterpret_cast<const int4*>(S.mk_ident + (size_t)s2 * 4), j);
            if (!same) {
                const uint32_t was = atomicOr(word, bit);
                if (!(was & bit)) {
                    const int pos = atomicAdd(&S.amb_new_n[par], 1);
                    cnt_add(S, 7, 1ull);
                    if (pos < S.amb_cap) S.amb_new[(size_t)par * S.amb_cap + pos] = tile;
                    else S.amb_flag[0] = 1;
                }
                return 1;
            }
            return (*word & bit) ? 1 : 0;
        }
        h = (h + 1) & mask;
    }
    return -1;
}
__device__ inline int tile_register(const DevState& S, const int4& id, int s, int a, int j, i32 tile, int par) {
    return tile_register_impl<false>(S, id, s, a, j, tile, par, 0ull, 0u);
}
__device__ __forceinline__ void tile_register_ask(const DevState& S, int s, int a, int j, i32 tile, u64& first_old, uint32_t& first_word) {
    const u64 want = ((u64)(uint32_t)tile << 32) | (u64)(uint32_t)(s * (LOB_N_ACTIONS * 32) + a * 32 + j);
    const uint32_t h = ((uint32_t)tile * 2654435761u) & (uint32_t)(S.ow_slots - 1);
    first_old = (u64)atomicCAS((unsigned long long*)&S.ow_tab[h], ~0ull, (unsigned long long)want);
    first_word = S.amb_bits[(uint32_t)tile >> 5];
}

#endif
