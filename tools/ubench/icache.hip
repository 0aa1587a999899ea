// Does straight-line code (every instruction executed once per launch) pay an instruction-fetch
// penalty compared with the same work in a loop?  Shape: 1024 x 256 threads like the step kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))
#define R256(X) R4(R64(X))
#define BODY "v_xor_b32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_and_b32 %0, 0x00ff00ff, %0\n v_or_b32 %1, %1, %0\n"

__global__ __launch_bounds__(256) void k_straight(uint32_t *out) {        // 4 * 1024 = 4096 instructions, ~24 KB
    uint32_t a = threadIdx.x, b = blockIdx.x;
    asm volatile(R256(R4(BODY)) : "+v"(a), "+v"(b));
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b;
}
__global__ __launch_bounds__(256) void k_loop(uint32_t *out) {            // same 4096 instructions, 64-instruction body
    uint32_t a = threadIdx.x, b = blockIdx.x;
    for (int i = 0; i < 64; ++i) asm volatile(R16(BODY) : "+v"(a), "+v"(b));
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b;
}
__global__ __launch_bounds__(256) void k_straight1k(uint32_t *out) {      // 1024 instructions straight
    uint32_t a = threadIdx.x, b = blockIdx.x;
    asm volatile(R256(BODY) : "+v"(a), "+v"(b));
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b;
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
    uint32_t *out; hipMalloc(&out, 1024 * 256 * 4);
    printf("straight 4096 instr : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_straight, dim3(1024), dim3(256), 0, 0, out); }, 300));
    printf("loop 64x64 instr    : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_loop, dim3(1024), dim3(256), 0, 0, out); }, 300));
    printf("straight 1024 instr : %.2f us\n", time_it([&] { hipLaunchKernelGGL(k_straight1k, dim3(1024), dim3(256), 0, 0, out); }, 300));
    return 0;
}
