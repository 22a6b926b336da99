// Cost of one wave-wide poll round (rp_flow.hip: up to 64 lanes x 4 sc1 16-byte loads, then a wave-wide compare) as a function of
// the address pattern, the number of polling lanes and the number of polling wavefronts on the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int LOADS>
__global__ void k_poll(float4 *rec, int iters, int stride_items, int lanes, unsigned *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rec, 0, 1 << 26, 0x00020000);
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    if (lane < lanes) {
        // item of this lane: consecutive (stride_items = 1) or scattered records; body 2 = a neighbour a few hundred records away
        const int i1 = ((wave * 64 + lane) * stride_items) & 0xfffff, i2 = (i1 + 201 * stride_items) & 0xfffff;
        for (int it = 0; it < iters; ++it) {
            u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, i1 * 32, 0, 16);
            u4 b = LOADS > 1 ? __builtin_amdgcn_raw_buffer_load_b128(r, i1 * 32 + 16, 0, 16) : a;
            u4 c = LOADS > 2 ? __builtin_amdgcn_raw_buffer_load_b128(r, i2 * 32, 0, 16) : a;
            u4 d = LOADS > 3 ? __builtin_amdgcn_raw_buffer_load_b128(r, i2 * 32 + 16, 0, 16) : a;
            acc += a.w + b.w + c.w + d.w;
            if (__any(acc == 0xdeadbeefu)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (acc == 12345u) out[0] = acc;
}
int main() {
    float4 *rec; unsigned *out;
    hipMalloc(&rec, 1 << 26); hipMalloc(&out, 4); hipMemset(rec, 0, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    for (int loads : {4, 2, 1}) for (int stride : {1, 97}) for (int lanes : {64, 16, 1}) for (int blocks : {1, 256, 512}) {
        hipDeviceSynchronize(); hipEventRecord(e0);
        if (loads == 4) hipLaunchKernelGGL(k_poll<4>, dim3(blocks), dim3(256), 0, 0, rec, iters, stride, lanes, out);
        else if (loads == 2) hipLaunchKernelGGL(k_poll<2>, dim3(blocks), dim3(256), 0, 0, rec, iters, stride, lanes, out);
        else hipLaunchKernelGGL(k_poll<1>, dim3(blocks), dim3(256), 0, 0, rec, iters, stride, lanes, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("loads/lane=%d  %s  lanes=%2d  waves=%4d: %.3f us per poll round\n", loads, stride == 1 ? "consecutive" : "scattered  ", lanes, blocks * 4, ms * 1e3 / iters);
    }
    return 0;
}
