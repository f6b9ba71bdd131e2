// Standalone probe (not part of the library): device-to-device copy variants on the MI355X, read + write bytes per second.
// hipcc -O3 --offload-arch=gfx950 -o hbm_copy_probe tools/probes/hbm_copy_probe.hip && ./hbm_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// A: one 16-B element per thread, no loop (one workgroup per 4 KB)
__global__ __launch_bounds__(256) void copy_flat(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
// B: grid-stride, U elements in flight per lane, plain loads / stores
template <int U> __global__ __launch_bounds__(256) void copy_gs(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = s[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) d[i + k * stride] = v[k];
    }
    for (; i < n; i += stride) d[i] = s[i];
}
// C: as B with nontemporal loads and stores
template <int U> __global__ __launch_bounds__(256) void copy_gs_nt(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = __builtin_nontemporal_load(&s[i + k * stride]);
#pragma unroll
        for (int k = 0; k < U; k++) __builtin_nontemporal_store(v[k], &d[i + k * stride]);
    }
    for (; i < n; i += stride) d[i] = s[i];
}
// D: each workgroup owns a contiguous chunk (consecutive 4-KB rounds), U rounds in flight
template <int U> __global__ __launch_bounds__(256) void copy_chunk(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, size_t n, size_t per_block) {
    size_t i = (size_t)blockIdx.x * per_block + threadIdx.x;
    const size_t end = (size_t)(blockIdx.x + 1) * per_block < n ? (size_t)(blockIdx.x + 1) * per_block : n;
    for (; i + (U - 1) * 256 < end; i += U * 256) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = s[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; k++) d[i + k * 256] = v[k];
    }
    for (; i < end; i += 256) d[i] = s[i];
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    u32x4 *s, *d;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %7.1f us  %6.2f TB/s (read + write)\n", name, best * 1e3, 2.0 * bytes / (best * 1e-3) / 1e12);
    };
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
    run("flat: one 16-B element per thread", [&] { hipLaunchKernelGGL(copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, s, d, n); });
    for (int bpc : {4, 8, 16, 32}) {
        char nm[96];
        const unsigned g = 256u * bpc;
        snprintf(nm, sizeof nm, "grid-stride x1, %d wg/CU", bpc); run(nm, [&] { hipLaunchKernelGGL(copy_gs<1>, dim3(g), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, sizeof nm, "grid-stride x4, %d wg/CU", bpc); run(nm, [&] { hipLaunchKernelGGL(copy_gs<4>, dim3(g), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, sizeof nm, "grid-stride x4 nontemporal, %d wg/CU", bpc); run(nm, [&] { hipLaunchKernelGGL(copy_gs_nt<4>, dim3(g), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, sizeof nm, "grid-stride x8, %d wg/CU", bpc); run(nm, [&] { hipLaunchKernelGGL(copy_gs<8>, dim3(g), dim3(256), 0, 0, s, d, n); });
        const size_t per = ((n + g - 1) / g + 255) / 256 * 256;
        snprintf(nm, sizeof nm, "contiguous chunk per wg x4, %d wg/CU", bpc); run(nm, [&] { hipLaunchKernelGGL(copy_chunk<4>, dim3(g), dim3(256), 0, 0, s, d, n, per); });
    }
    return 0;
}
