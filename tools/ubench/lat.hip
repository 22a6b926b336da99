// Single-wave VALU latency microbenchmarks (gfx950): cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__global__ void k(float *out, long long *cyc, float c) {
    float x = out[threadIdx.x], y0 = x + 1, y1 = x + 2, y2 = x + 3, y3 = x + 4, y4 = x + 5, y5 = x + 6, y6 = x + 7, y7 = x + 8;
    long long t0, t1;
    // (a) dependent v_add chain
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // (b) 8 independent chains
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                     : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(c));
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // (c) dependent mul -> add alternating (axpy chain)
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 2; ++i) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // (d) dependent DPP mov + add
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 2; ++i) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // (e) dependent pk_add chain
    {
        float2 p = make_float2(x, y0), q = make_float2(c, c);
        t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
        t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[4] = t1 - t0;
        x += p.x + p.y;
    }
    // (f) LDS write -> barrier -> read round trip
    __shared__ float4 sh[512];
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 64; ++i) {
        sh[threadIdx.x] = make_float4(x, x, x, x);
        __syncthreads();
        x += sh[(threadIdx.x + 1) % blockDim.x].x;
        __syncthreads();
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // (g) s_memtime overhead
    t0 = __builtin_readcyclecounter();
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // (h) uniform branch chain: not-taken s_cbranch every other instruction
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) asm volatile("v_add_f32 %0, %0, %1\n s_cmp_eq_u32 %2, 12345\n s_cbranch_scc1 1\n v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c), "s"(i) : "scc");
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[7] = t1 - t0;
    out[threadIdx.x] = x + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}
int main() {
    float *o; long long *c;
    hipMalloc(&o, 4096 * 4); hipMalloc(&c, 64 * 8); hipMemset(o, 0, 4096 * 4);
    for (int threads : {64, 320}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, o, c, 1.0f); hipDeviceSynchronize(); }
        long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("threads=%d  dep add %.2f cyc/instr | 8-indep add %.2f | dep mul/add %.2f | nop+dpp+add %.2f per pair-instr | dep pk_add %.2f | lds rt %.1f cyc/iter | memtime %lld | add+branch %.2f per 4 instr\n", threads,
               h[0] / (double)N, h[1] / (double)N, h[2] / (double)N, h[3] / (double)(N / 2), h[4] / (double)N, h[5] / 64.0, h[6], h[7] / (double)(N / 4));
    }
    return 0;
}
