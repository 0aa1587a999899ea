// What does it cost to stream the step kernel's bytes (2 x 10.24 MB in, 10.24 MB out) at all?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_gs(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ o, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint4 x = a[i], y = b[i];
        x.x ^= y.y;
        o[i] = x;
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_unroll(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ o, int n) {
    const int base = blockIdx.x * 256 * U + threadIdx.x;
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + 256 * u < n) { x[u] = a[base + 256 * u]; y[u] = b[base + 256 * u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + 256 * u < n) { x[u].x ^= y[u].y; o[base + 256 * u] = x[u]; }
}
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ a, uint4 *__restrict__ o, int n) {
    uint4 acc = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { uint4 x = a[i]; acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w; }
    if (acc.x == 0x12345678) o[0] = acc;
}

template <typename F> float time_it(F launch, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / n;
}

int main() {
    const size_t big = 1ull << 30;
    uint4 *a, *b, *c;
    hipMalloc(&a, big); hipMalloc(&b, big); hipMalloc(&c, big);
    hipMemset(a, 1, big); hipMemset(b, 2, big); hipMemset(c, 0, big);
    for (size_t bytes : {10240000ull, 40960000ull, 163840000ull, 1ull << 30}) {
        const int n = (int)(bytes / 16);
        printf("---- %zu bytes per array (2 read + 1 written)\n", bytes);
        for (int grid : {256, 1024, 2048, 4096, 8192}) {
            float us = time_it([&] { hipLaunchKernelGGL(k_gs, dim3(grid), dim3(256), 0, 0, a, b, c, n); }, 200);
            printf("  grid-stride grid=%5d : %8.2f us  %7.1f GB/s\n", grid, us, 3.0 * bytes / us * 1e-3);
        }
        {
            float us = time_it([&] { hipLaunchKernelGGL(k_unroll<3>, dim3((n + 767) / 768), dim3(256), 0, 0, a, b, c, n); }, 200);
            printf("  unroll3 (1 block / 768 vec): %8.2f us  %7.1f GB/s\n", us, 3.0 * bytes / us * 1e-3);
            us = time_it([&] { hipLaunchKernelGGL(k_unroll<1>, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, c, n); }, 200);
            printf("  unroll1 (1 block / 256 vec): %8.2f us  %7.1f GB/s\n", us, 3.0 * bytes / us * 1e-3);
            us = time_it([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, a, c, n); }, 200);
            printf("  read-only grid=2048        : %8.2f us  %7.1f GB/s\n", us, 1.0 * bytes / us * 1e-3);
        }
    }
    return 0;
}
