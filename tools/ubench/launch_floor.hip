// Launch floor: duration of (nearly) empty kernels with the fused step kernel's launch shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_small(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x;
}
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_vgpr(128))) void k_fat(uint32_t *out) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) { smem[0] = 1; out[blockIdx.x] = blockIdx.x + smem[0]; }
}
// same, but every thread moves 40 bytes in and 40 bytes out (the step kernel's HBM traffic)
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_vgpr(128))) void k_copy(const uint4 *in, const uint4 *in2, uint4 *o) {
    extern __shared__ unsigned char smem[];
    const int base = blockIdx.x * 625, t = threadIdx.x;
    uint4 a[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) if (t + 256 * i < 625) { a[i] = in[base + t + 256 * i]; b[i] = in2[base + t + 256 * i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) if (t + 256 * i < 625) { a[i].x ^= b[i].y; o[base + t + 256 * i] = a[i]; }
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
    uint32_t *out; hipMalloc(&out, 1 << 20);
    uint4 *a, *b, *c; size_t n = 1024 * 625 * 16;
    hipMalloc(&a, n); hipMalloc(&b, n); hipMalloc(&c, n);
    hipMemset(a, 1, n); hipMemset(b, 2, n);
    hipFuncSetAttribute((const void *)k_fat, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k_copy, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("small  1024x256            : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_small, dim3(1024), dim3(256), 0, 0, out); }, 400));
    printf("fat    1024x256 24KB LDS   : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_fat, dim3(1024), dim3(256), 24576, 0, out); }, 400));
    printf("fat    1024x256 0KB LDS    : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_fat, dim3(1024), dim3(256), 0, 0, out); }, 400));
    printf("copy 20MB in / 10MB out    : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 24576, 0, a, b, c); }, 400));
    return 0;
}
