// rp_grid.h — the broad phase's hashed grid as its readers see it (rp_broadphase.hip builds and maintains it; the continuous-collision
// pass of rp_ccd.h looks its targets up in it).  Moved out of rp_broadphase.hip in round 4, unchanged.
#pragma once
#include "rp_pairs.h"

__device__ __forceinline__ unsigned long long rp_hash64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ unsigned long long cell_key(int cx, int cy, int cz) {
    const int off = 1 << 20;
    return ((unsigned long long)((cx + off) & 0x1fffff)) | ((unsigned long long)((cy + off) & 0x1fffff) << 21) |
           ((unsigned long long)((cz + off) & 0x1fffff) << 42);
}
// Sub-worlds (rp_world_begin_subworld: a batch of small worlds in one device world) may overlap in space: a collider's cells are keyed
// with its sub-world, so every sub-world fills buckets of its own (what still collides in the hash is told apart by pair_allowed).
__device__ __forceinline__ unsigned long long cell_key_of(const DevWorld &w, int collider, int cx, int cy, int cz) {
    unsigned long long k = cell_key(cx, cy, cz);
    if (w.n_sub > 1) k ^= (unsigned long long)(unsigned)w.c_sub[collider] * 0x9E3779B97F4A7C15ull;
    return k;
}
// ... and the brute-force list of large colliders (ground slabs: one per sub-world in a batch of small worlds) is kept SORTED by
// sub-world (bp_large_by_sub, after every build pass): a collider walks its own sub-world's segment, not n slabs it can never meet.
__device__ __forceinline__ void large_range_of(const DevWorld &w, int collider, int nl, int &q0, int &q1) {
    q0 = 0; q1 = nl;
    if (w.n_sub > 1) {
        const int s = w.c_sub[collider];
        q0 = w.large_sub_begin[s]; q1 = w.large_sub_begin[s + 1];
        if (q0 > nl) q0 = nl;
        if (q1 > nl) q1 = nl;
    }
}
__device__ __forceinline__ int cell_coord(float x, float inv_cell) { return (int)floorf(x * inv_cell); }

struct CellRange { int lo[3], hi[3]; bool large; };
__device__ __forceinline__ CellRange cell_range_of(const DevWorld &w, float4 mn, float4 mx);
__device__ __forceinline__ CellRange cell_range(const DevWorld &w, int i) { return cell_range_of(w, w.c_fatmin[i], w.c_fatmax[i]); }
__device__ __forceinline__ CellRange cell_range_of(const DevWorld &w, float4 mn, float4 mx) {
    CellRange r;
    float ic = w.prm.inv_cell_size;
    r.lo[0] = cell_coord(mn.x, ic); r.lo[1] = cell_coord(mn.y, ic); r.lo[2] = cell_coord(mn.z, ic);
    r.hi[0] = cell_coord(mx.x, ic); r.hi[1] = cell_coord(mx.y, ic); r.hi[2] = cell_coord(mx.z, ic);
    // (an unbounded AABB — a half-space — saturates the cell coordinates: sized from the floats, not from their difference)
    const bool unbounded = (mx.x - mn.x) * ic > 1.0e6f || (mx.y - mn.y) * ic > 1.0e6f || (mx.z - mn.z) * ic > 1.0e6f;
    r.large = unbounded || (r.hi[0] - r.lo[0] > 2) || (r.hi[1] - r.lo[1] > 2) || (r.hi[2] - r.lo[2] > 2);
    return r;
}

// ---- the grid, and how it follows a collider whose fat AABB was rewritten ------------------------------------------------------------
// The grid: grid_cap hash buckets of RP_BP_BUCKET fixed slots, two copies (the one in service: epoch parity).  A slot is ONE word:
// collider (24 bits) | which of its at most 27 cells the entry stands for (5 bits, x fastest inside the collider's cell range) | the
// collider's RANGE VERSION when the entry was written (3 bits).  Cells that share a bucket are told apart without a stored key: a
// reader at cell X accepts an entry only if the cell it stands for — recomputed from the partner's fat AABB, which the overlap test
// loads anyway — is X (bp_entry_is_cell).
// Round 4: the grid in service FOLLOWS the colliders between full rebuilds.  A rewritten fat AABB that still covers the same cells needs
// nothing (its entries decode to the same cells).  One that covers other cells bumps the collider's range version — which kills every
// entry written before, wherever it sits — and appends an entry for every cell of the new range (k_collider_update / k_fast_front, the
// kernel before the broad-phase pass).  So no collider is ever "stale": the incremental pass finds every partner through the grid, and
// the stale list of round 3 (2,048 colliders, brute force, full rebuild when full: one pass in three on b3d_large_pyramid, five in
// six on b3d_joint_grid) is gone.  A bucket that fills up, or a collider that changes cells for the 7th time since the last full
// rebuild (3 version bits), clears FL_BP_GRID_OK: the next pass is a full rebuild, which starts every version from zero.
// Which grid copy is in service: its own parity (lay_state[8]), not the pair tables' epoch — a full rebuild that finds the grid in order
// (it follows the colliders, see above) keeps it and skips the build pass; lay_state[9] counts such rebuilds since the last build.
#define BP_GPAR(w) ((w).lay_state[8] & 1)
RP_DEV int bp_entry(int collider, int ordinal, int version) { return collider | (ordinal << 24) | ((version & 7) << 29); }
RP_DEV bool bp_entry_is_cell(const DevWorld &w, int entry, int x, int y, int z) {
    const int j = entry & 0xffffff;
    if ((int)((unsigned)entry >> 29) != (w.c_rver[j] & 7)) return false; // written for a cell range the collider has left since
    const CellRange r = cell_range(w, j);
    const int o = (entry >> 24) & 31, nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1;
    return r.lo[0] + o % nx == x && r.lo[1] + (o / nx) % ny == y && r.lo[2] + o / (nx * ny) == z;
}
