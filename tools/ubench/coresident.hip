// Which resource keeps two 320-thread workgroups of the register-lean island kernel from sharing a CU?  
// Every workgroup arrives on a counter and waits (bounded) for the whole grid: a launch whose workgroups are all resident sees
// everybody arrive; otherwise the waiters time out.  Variants: LDS bytes x scratch bytes x VGPR budget (waves_per_eu).
//   hipcc --offload-arch=gfx950 -O3 -o coresident coresident.hip && ./coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS, int SCRATCH_DW, int THREADS>
__device__ __forceinline__ void body(unsigned *counter, unsigned *ok, unsigned *hw, int sel) {
    __shared__ int lds[LDS / 4];
    volatile int scr[SCRATCH_DW > 0 ? SCRATCH_DW : 1];
    if (SCRATCH_DW > 0) { for (int i = 0; i < SCRATCH_DW; ++i) scr[i] = i + sel; }
    lds[threadIdx.x] = threadIdx.x; lds[LDS / 4 - 1 - threadIdx.x] = sel;
    __syncthreads();
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hw[blockIdx.x] = (id & 0xffff) | ((xcc & 0xf) << 16);
        atomicAdd(counter, 1u);
        long long t0 = wall_clock64(); // 100 MHz
        int good = 0;
        while (wall_clock64() - t0 < 2000000) { // 20 ms
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { good = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        s_ok = good;
    }
    __syncthreads();
    if (threadIdx.x == 0) ok[blockIdx.x] = s_ok + (SCRATCH_DW > 0 ? (scr[(sel + 3) % SCRATCH_DW] & 0) : 0) + (lds[(sel * 7) & 63] & 0);
}
#define VARIANT(NAME, LDS, SCR, THREADS, WPE) \
    __global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) NAME(unsigned *c, unsigned *ok, unsigned *hw, int sel) { \
        if (WPE == 3) asm volatile("v_mov_b32 v167, 0" ::: "v167"); \
        if (WPE == 2) asm volatile("v_mov_b32 v255, 0" ::: "v255"); \
        if (WPE == 4) asm volatile("v_mov_b32 v127, 0" ::: "v127"); \
        if (WPE == 8) asm volatile("v_mov_b32 v63, 0" ::: "v63"); \
        body<LDS, SCR, THREADS>(c, ok, hw, sel); }
VARIANT(k_t64_v168, 4096, 0, 64, 3)
VARIANT(k_t64_v128, 4096, 0, 64, 4)
VARIANT(k_t64_v64, 4096, 0, 64, 8)
VARIANT(k_t128_v168, 4096, 0, 128, 3)
VARIANT(k_t128_v128, 4096, 0, 128, 4)
VARIANT(k_t128_v64, 4096, 0, 128, 8)
VARIANT(k_t192_v168, 4096, 0, 192, 3)
VARIANT(k_t192_v128, 4096, 0, 192, 4)
VARIANT(k_t192_v64, 4096, 0, 192, 8)
VARIANT(k_t256_v168, 4096, 0, 256, 3)
VARIANT(k_t256_v128, 4096, 0, 256, 4)
VARIANT(k_t256_v64, 4096, 0, 256, 8)
VARIANT(k_t320_v168, 4096, 0, 320, 3)
VARIANT(k_t320_v128, 4096, 0, 320, 4)
VARIANT(k_t320_v64, 4096, 0, 320, 8)
VARIANT(k_t384_v168, 4096, 0, 384, 3)
VARIANT(k_t384_v128, 4096, 0, 384, 4)
VARIANT(k_t384_v64, 4096, 0, 384, 8)
VARIANT(k_t448_v168, 4096, 0, 448, 3)
VARIANT(k_t448_v128, 4096, 0, 448, 4)
VARIANT(k_t448_v64, 4096, 0, 448, 8)
VARIANT(k_t512_v168, 4096, 0, 512, 3)
VARIANT(k_t512_v128, 4096, 0, 512, 4)
VARIANT(k_t512_v64, 4096, 0, 512, 8)
VARIANT(k_t640_v168, 4096, 0, 640, 3)
VARIANT(k_t640_v128, 4096, 0, 640, 4)
VARIANT(k_t640_v64, 4096, 0, 640, 8)
VARIANT(k_lds67_scr72_v168, 67328, 72, 320, 3)
template <typename K> void run(const char *name, K k, int threads) {
    unsigned *c, *ok, *hw; hipMalloc(&c, 4); hipMalloc(&ok, 4096 * 4); hipMalloc(&hw, 4096 * 4);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, threads, 0);
    for (int grid : {256, 512, 768, 1024}) {
        hipMemset(c, 0, 4); hipMemset(ok, 0, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, c, ok, hw, 1);
        hipDeviceSynchronize();
        std::vector<unsigned> h(grid), w(grid); hipMemcpy(h.data(), ok, grid * 4, hipMemcpyDeviceToHost); hipMemcpy(w.data(), hw, grid * 4, hipMemcpyDeviceToHost);
        int good = 0; for (unsigned x : h) good += x;
        // distinct (xcc, se, sh, cu) ids seen
        std::vector<int> cnt(1 << 20, 0); int distinct = 0, maxper = 0;
        for (unsigned x : w) { int key = (int)(((x >> 8) & 0xff) | ((x >> 16) << 8)); if (cnt[key]++ == 0) distinct++; if (cnt[key] > maxper) maxper = cnt[key]; }
        printf("%-28s occ_api=%d grid=%3d  all-arrived workgroups=%3d  distinct CUs=%3d  max workgroups on one CU=%d\n", name, occ, grid, good, distinct, maxper);
    }
    hipFree(c); hipFree(ok); hipFree(hw);
}
int main() {
    run("thr=64 vgpr<=168 lds4K", k_t64_v168, 64);
    run("thr=64 vgpr<=128 lds4K", k_t64_v128, 64);
    run("thr=64 vgpr<=64 lds4K", k_t64_v64, 64);
    run("thr=128 vgpr<=168 lds4K", k_t128_v168, 128);
    run("thr=128 vgpr<=128 lds4K", k_t128_v128, 128);
    run("thr=128 vgpr<=64 lds4K", k_t128_v64, 128);
    run("thr=192 vgpr<=168 lds4K", k_t192_v168, 192);
    run("thr=192 vgpr<=128 lds4K", k_t192_v128, 192);
    run("thr=192 vgpr<=64 lds4K", k_t192_v64, 192);
    run("thr=256 vgpr<=168 lds4K", k_t256_v168, 256);
    run("thr=256 vgpr<=128 lds4K", k_t256_v128, 256);
    run("thr=256 vgpr<=64 lds4K", k_t256_v64, 256);
    run("thr=320 vgpr<=168 lds4K", k_t320_v168, 320);
    run("thr=320 vgpr<=128 lds4K", k_t320_v128, 320);
    run("thr=320 vgpr<=64 lds4K", k_t320_v64, 320);
    run("thr=384 vgpr<=168 lds4K", k_t384_v168, 384);
    run("thr=384 vgpr<=128 lds4K", k_t384_v128, 384);
    run("thr=384 vgpr<=64 lds4K", k_t384_v64, 384);
    run("thr=448 vgpr<=168 lds4K", k_t448_v168, 448);
    run("thr=448 vgpr<=128 lds4K", k_t448_v128, 448);
    run("thr=448 vgpr<=64 lds4K", k_t448_v64, 448);
    run("thr=512 vgpr<=168 lds4K", k_t512_v168, 512);
    run("thr=512 vgpr<=128 lds4K", k_t512_v128, 512);
    run("thr=512 vgpr<=64 lds4K", k_t512_v64, 512);
    run("thr=640 vgpr<=168 lds4K", k_t640_v168, 640);
    run("thr=640 vgpr<=128 lds4K", k_t640_v128, 640);
    run("thr=640 vgpr<=64 lds4K", k_t640_v64, 640);
    return 0;
}
