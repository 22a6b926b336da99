// rp_islands_lean.hip — k_island_solve_dense, the register-lean form of the per-island megakernel (rp_islands_lean.h): its own
// translation unit because it is built with -mllvm -disable-machine-licm (Makefile): hoisting the per-lane LDS addresses and pointer
// selects out of the island loop costs this kernel, which lives on a 168-VGPR budget, its scratch-free steady state.
#include "rp_island_stages.h"
#include "rp_islands_lean.h"

// (three waves per SIMD = 168 VGPRs: with 320-thread workgroups and < 80 KB of LDS that is two islands per CU)
__global__ void __launch_bounds__(ISL_THREADS_DENSE) __attribute__((amdgpu_waves_per_eu(3, 3))) k_island_solve_dense(DevWorld w, int has_restitution, int fast, int retire, int fused) { island_solve_lean(w, has_restitution, fast, retire, fused); }

// the same for the dense form of the kernel (two islands per CU when the occupancy answer allows it; 0 = not better than the other form)
int rp_fused_grid_dense(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device] > 0 ? cached[device] : 0;
    hipDeviceProp_t prop;
    int per_cu = 0, cus = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_island_solve_dense, ISL_THREADS_DENSE, 0) != hipSuccess) per_cu = 0;
    int g = 0;
    if (per_cu >= 2 && cus >= 1) { g = cus * 2 - (cus + 15) / 16; }
    if (device >= 0 && device < 64) cached[device] = g > 0 ? g : -1;
    return g;
}
void rp_launch_island_solve_dense(const DevWorld &w, hipStream_t st, int grid, int has_restitution, int fast, int retire, int fused) {
    hipLaunchKernelGGL(k_island_solve_dense, dim3(grid < 1 ? 1 : grid), dim3(ISL_THREADS_DENSE), 0, st, w, has_restitution, fast, retire, fused);
}
