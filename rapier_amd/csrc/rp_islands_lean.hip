// rp_islands_lean.hip — k_island_solve_dense, the register-lean form of the per-island megakernel (rp_islands_lean.h): its own
// translation unit because it is built with -mllvm -disable-machine-licm (Makefile): hoisting the per-lane LDS addresses and pointer
// selects out of the island loop costs this kernel, which lives on a 168-VGPR budget, its scratch-free steady state.
#include "rp_island_stages.h"
#include "rp_islands_lean.h"

// (twelve wavefronts = three per SIMD at 168 VGPRs: ONE 768-thread workgroup per CU that holds TWO islands, 135 KB of LDS)
__global__ void __launch_bounds__(LEAN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) k_island_solve_dense(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_lean<false>(w, has_restitution, fast, retire, fused); }
// ... as a launch of `nsteps` fused steps (island_solve_body, rp_islands.hip)
__global__ void __launch_bounds__(LEAN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) k_island_solve_dense_steps(DevWorld w, int has_restitution, int nsteps) { island_solve_lean<false>(w, has_restitution, 1, 1, 1, nsteps); }
// (the WIDE validators of rp_island_stages.h: worlds with compound bodies or sleeping enabled)
__global__ void __launch_bounds__(LEAN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) k_island_solve_dense_wide(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_lean<true>(w, has_restitution, fast, retire, fused); }

// Most workgroups of the lean form a fused fast step may launch (every workgroup resident at once, like rp_fused_grid): one 640-thread
// workgroup = two islands per CU, 1/16 of the CUs left free; 0 = the device does not hold such a workgroup.
int rp_fused_grid_dense(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device] > 0 ? cached[device] : 0;
    hipDeviceProp_t prop;
    int per_cu = 0, cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_island_solve_dense, LEAN_THREADS, 0) != hipSuccess) per_cu = 0;
    { int pw = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_island_solve_dense_wide, LEAN_THREADS, 0) != hipSuccess) pw = 0; if (pw < per_cu) per_cu = pw; }
    int g = 0;
    if (per_cu >= 1 && cus >= 1) { g = cus - (cus + 15) / 16; if (g < 1) g = 1; }
    if (device >= 0 && device < 64) cached[device] = g > 0 ? g : -1;
    return g;
}
void rp_launch_island_solve_dense_steps(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int nsteps) {
    hipLaunchKernelGGL(k_island_solve_dense_steps, dim3(grid < 1 ? 1 : grid), dim3(LEAN_THREADS), 0, st, w, has_restitution, nsteps);
}
// `grid` = workgroups (each takes islands 2b and 2b + 1, then strides by 2 x grid)
void rp_launch_island_solve_dense(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int fast, int retire, int fused, int wide) {
    if (wide) hipLaunchKernelGGL(k_island_solve_dense_wide, dim3(grid < 1 ? 1 : grid), dim3(LEAN_THREADS), 0, st, w, has_restitution, fast, retire, fused);
    else hipLaunchKernelGGL(k_island_solve_dense, dim3(grid < 1 ? 1 : grid), dim3(LEAN_THREADS), 0, st, w, has_restitution, fast, retire, fused);
}
