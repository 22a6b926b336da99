// Hand-off latency between workgroups through tagged 16-byte write-through records (the primitive of rp_flow.hip):
// a token travels round a ring of G single-wave workgroups; time per hop = kernel time / (G * laps).
//   hipcc --offload-arch=gfx950 -O3 -o hop hop.hip && ./hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define SC1 16
template <int aux> __global__ void k_ring(float4 *rec, int laps, int sleep_n, int payload_loads, float4 *junk) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rec, 0, gridDim.x * 16 * 2, 0x00020000);
    const int g = blockIdx.x, G = gridDim.x, prev = (g + G - 1) % G;
    if (threadIdx.x != 0) return;
    float acc = 0.0f;
    for (int lap = 0; lap < laps; ++lap) {
        // WG 0 starts lap `lap` when WG G-1 finished lap-1; others wait for their predecessor in this lap
        unsigned want = (g == 0) ? (unsigned)lap : (unsigned)(lap + 1);
        for (;;) {
            u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, prev * 32, 0, aux);
            u4 b = __builtin_amdgcn_raw_buffer_load_b128(r, prev * 32 + 16, 0, aux);
            if (a.w == want && b.w == want) { acc += __int_as_float((int)a.x); break; }
            if (sleep_n) __builtin_amdgcn_s_sleep(1);
        }
        for (int k = 0; k < payload_loads; ++k) acc += junk[(g * 64 + lap * 7 + k * 131) & 0xffff].x; // dependent-ish plain loads (constraint planes)
        u4 v; v.x = (unsigned)__float_as_int(acc); v.y = 1; v.z = 2; v.w = (unsigned)(lap + 1);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, g * 32, 0, aux);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, g * 32 + 16, 0, aux);
    }
}
// background pollers: every other workgroup polls a record that never changes (the load the dataflow kernel puts on the fabric)
__global__ void k_ring_loaded(float4 *rec, int laps, int ring, float4 *junk, int *stop) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rec, 0, 1 << 24, 0x00020000);
    const int g = blockIdx.x;
    const int aux = 16;
    if (g >= ring) { // poller wave: 64 lanes x 4 scattered 16-byte sc1 loads per round until the ring is done
        unsigned s = 0;
        while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            int idx = ring * 2 + ((g * 64 + threadIdx.x) * 37 + (int)s * 11) % 60000;
            u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, idx * 16, 0, aux);
            u4 b = __builtin_amdgcn_raw_buffer_load_b128(r, ((idx + 977) % 60000 + ring * 2) * 16, 0, aux);
            s += a.w + b.w + 1;
            __builtin_amdgcn_s_sleep(2);
        }
        return;
    }
    if (threadIdx.x != 0) return;
    const int G = ring, prev = (g + G - 1) % G;
    for (int lap = 0; lap < laps; ++lap) {
        unsigned want = (g == 0) ? (unsigned)lap : (unsigned)(lap + 1);
        for (;;) {
            u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, prev * 32, 0, aux);
            u4 b = __builtin_amdgcn_raw_buffer_load_b128(r, prev * 32 + 16, 0, aux);
            if (a.w == want && b.w == want) break;
            __builtin_amdgcn_s_sleep(1);
        }
        u4 v; v.x = 0; v.y = 1; v.z = 2; v.w = (unsigned)(lap + 1);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, g * 32, 0, aux);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, g * 32 + 16, 0, aux);
    }
    if (g == ring - 1) __hip_atomic_store(stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
    float4 *rec, *junk; int *stop;
    hipMalloc(&rec, 1 << 24); hipMalloc(&junk, 65536 * 16); hipMalloc(&stop, 4);
    hipMemset(junk, 0, 65536 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int laps = 50;
    for (int aux : {16, 17, 0}) for (int G : {2, 8, 64, 240}) for (int pl : {0, 8}) {
        hipMemset(rec, 0, 1 << 24);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (aux == 16) hipLaunchKernelGGL(k_ring<16>, dim3(G), dim3(64), 0, 0, rec, laps, 1, pl, junk);
        else if (aux == 17) hipLaunchKernelGGL(k_ring<17>, dim3(G), dim3(64), 0, 0, rec, laps, 1, pl, junk);
        else continue;
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("idle  aux=%d ring=%3d payload_loads=%d: %.3f us per hop\n", aux, G, pl, ms * 1e3 / (G * laps));
    }
    for (int pollers : {0, 256, 768}) {
        hipMemset(rec, 0, 1 << 24); hipMemset(stop, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ring_loaded, dim3(64 + pollers), dim3(64), 0, 0, rec, laps, 64, junk, stop);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("loaded ring=64 + %d polling waves: %.3f us per hop\n", pollers, ms * 1e3 / (64 * laps));
    }
    return 0;
}
