// rp_gridbar.h — grid-wide barrier for the fused "rebuild" kernels (broad-phase rebuild, solver-graph / island layout).
//
// Those rebuilds are chains of 8-9 dependent passes that run only when something changed (FL_BP_DIRTY / FL_LAYOUT_DIRTY).  As
// separate launches every pass costs a launch (~2.4 us kernel floor + ~1.5 us boundary on MI355X) even when it exits at once; as
// ONE launch the clean step pays one early exit, and a rebuild pays one counter barrier (~1.3-3.3 us at 64-256 workgroups,
// tools/ubench/flatbar.hip) per pass instead of a launch.
//
// Protocol (MI355X guide, Guideline 16, counter form): every wave drains its stores, the workgroup meets, ONE lane releases at
// agent scope (per-XCD L2 write-back), arrives on a monotonic device counter, polls it relaxed, then acquires at agent scope
// (stale L1 / L2 lines dropped) and the workgroup meets again: plain loads and stores on either side are then safe on any
// XCD placement.  The counter never resets: a launch starts from `base` (published by the previous launch after its last
// barrier) and barrier k completes at base + k * gridDim.x; comparisons are wrap-safe.  Every workgroup of the launch must be
// resident: the grids are capped by DevWorld::gbar_blocks, which the host derives from the occupancy of these very kernels on the
// device it runs on (rp_api.hip: CUs x the smallest occupancy answer, a quarter left free), not from a CU count.  A bounded spin
// turns a violation into RP_OVF_GRID instead of a hang, and the launch then ENDS there: the waiter that gave up raises the bit, every
// workgroup looks at it behind every barrier and returns — no later pass ever runs on half-built scans (a late arrival cannot
// slip through either: it finds the bit the early leavers raised).  The error is sticky (rp_last_error); the world must be rebuilt.
#pragma once
#include "rp_world.h"

struct GridBar { unsigned *word; unsigned target; int *ovf; };
RP_DEV GridBar gbar_begin(const DevWorld &w, int which) {
    GridBar b;
    b.word = w.bar + 2 * which; // [0] arrivals, [1] base of the next launch
    b.target = __hip_atomic_load(b.word + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.ovf = &w.flags[FL_OVERFLOW];
    return b;
}
// true = the launch is dead (some barrier of it timed out): the caller returns
RP_DEV bool gbar_sync(GridBar &b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave: its stores have left the CU
    __syncthreads();
    b.target += gridDim.x;
    int dead = 0;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-back has completed before the arrival is visible
        __hip_atomic_fetch_add(b.word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(b.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - b.target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) { atomicOr(b.ovf, RP_OVF_GRID); break; } // ~2 s: a workgroup of this launch is not resident
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        dead = (__hip_atomic_load(b.ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & RP_OVF_GRID) != 0;
    }
    return __syncthreads_or(dead) != 0;
}
#define GBAR_SYNC(bar) do { if (gbar_sync(bar)) return; } while (0)
// Item index of this thread for the grid-stride passes: consecutive 64-item groups go to DIFFERENT workgroups (wave w of workgroup b
// takes group w * gridDim + b), so a pass over a few thousand items still spreads over every CU of the launch instead of filling
// the first few 1024-thread workgroups; a wavefront keeps 64 consecutive items (coalesced).  Stride = gridDim.x * blockDim.x.
RP_DEV int gbar_item(void) { return (int)((((threadIdx.x >> 6) * gridDim.x + blockIdx.x) << 6) + (threadIdx.x & 63)); }
// after the last barrier of the launch: the next launch starts from here
RP_DEV void gbar_end(const GridBar &b) {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(b.word + 1, b.target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -DRP_PASS_PROFILE (tools/pass_profile.py): thread 0 of the launch accumulates the time between stamps (10 ns ticks) into
// DevWorld::dbg[base + k], so the passes of a fused rebuild kernel can be told apart.  Stamps sit BEHIND barriers: what is measured is
// the slowest workgroup of each pass plus its barrier.
#ifdef RP_PASS_PROFILE
#define RP_PASS_BEGIN() long long rp_pp_t = wall_clock64(); int rp_pp_k = 0
#define RP_PASS_STAMP(w, base) do { if (blockIdx.x == 0 && threadIdx.x == 0) { long long n_ = wall_clock64(); (w).dbg[(base) + rp_pp_k] += n_ - rp_pp_t; rp_pp_t = n_; (w).dbg[(base) + 15] += (rp_pp_k == 0); } rp_pp_k++; } while (0)
#else
#define RP_PASS_BEGIN() do { } while (0)
#define RP_PASS_STAMP(w, base) do { } while (0)
#endif
