// Host cost of a kernel launch as a function of the by-value argument size (run with HIP_FORCE_DEV_KERNARG=1 and 0).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

template <int N> struct Blob { uint32_t w[N]; };
template <int N> __global__ void k_args(uint32_t *out, Blob<N> b) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = b.w[N - 1];
}

template <int N> double host_us(hipStream_t s0, hipStream_t s1, uint32_t *out) {
    Blob<N> b = {};
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_args<N>, dim3(64), dim3(256), 0, (i & 1) ? s1 : s0, out, b);
    hipDeviceSynchronize();
    const int n = 2000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_args<N>, dim3(64), dim3(256), 0, (i & 1) ? s1 : s0, out, b);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    hipDeviceSynchronize();
    return us;
}

int main() {
    uint32_t *out; hipMalloc(&out, 4096);
    hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    printf("args   16 B: %.2f us per launch\n", host_us<2>(s0, s1, out));
    printf("args  128 B: %.2f us per launch\n", host_us<30>(s0, s1, out));
    printf("args  256 B: %.2f us per launch\n", host_us<62>(s0, s1, out));
    printf("args  512 B: %.2f us per launch\n", host_us<126>(s0, s1, out));
    printf("args  800 B: %.2f us per launch\n", host_us<198>(s0, s1, out));
    printf("args 1600 B: %.2f us per launch\n", host_us<398>(s0, s1, out));
    return 0;
}
