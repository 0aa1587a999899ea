// How long does the dispatcher take to START the workgroups of a launch, as a function of the
// launch footprint?  Every workgroup stamps s_memrealtime (100 MHz, chip-wide) when it starts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define STAMP_KERNEL(NAME, ATTR)                                                                   \
    __global__ __launch_bounds__(256) ATTR void NAME(long long *t, int spin) {                         \
        extern __shared__ unsigned char smem[];                                                         \
        long long t0 = __builtin_amdgcn_s_memrealtime();                                                \
        if (threadIdx.x == 0) { smem[0] = 1; t[blockIdx.x] = t0; }                                      \
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8); /* stay resident for a while */     \
    }
STAMP_KERNEL(k_stamp32, __attribute__((amdgpu_num_vgpr(32))))
STAMP_KERNEL(k_stamp64, __attribute__((amdgpu_num_vgpr(64))))
STAMP_KERNEL(k_stamp128, __attribute__((amdgpu_num_vgpr(128))))
template <int VGPR> struct Pick;
template <> struct Pick<32> { static constexpr auto fn = k_stamp32; };
template <> struct Pick<64> { static constexpr auto fn = k_stamp64; };
template <> struct Pick<128> { static constexpr auto fn = k_stamp128; };
template <int VGPR>
void run(const char *name, int blocks, int lds, int spin) {
    long long *t; hipMalloc(&t, blocks * 8);
    hipFuncSetAttribute((const void *)Pick<VGPR>::fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<long long> h(blocks);
    double spread = 0, p90 = 0;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(Pick<VGPR>::fn, dim3(blocks), dim3(256), lds, 0, t, spin);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        if (rep) { spread += (h.back() - h.front()) * 10.0; p90 += (h[blocks * 9 / 10] - h.front()) * 10.0; }
    }
    printf("%-34s blocks=%4d lds=%5d : last start %.0f ns, p90 %.0f ns\n", name, blocks, lds, spread / 5, p90 / 5);
    hipFree(t);
}
int main() {
    run<32>("32 VGPR", 1024, 0, 200);
    run<32>("32 VGPR", 1024, 24576, 200);
    run<32>("32 VGPR", 1024, 38912, 200);
    run<128>("128 VGPR", 1024, 0, 200);
    run<128>("128 VGPR", 1024, 24576, 200);
    run<128>("128 VGPR", 1024, 38912, 200);
    run<64>("64 VGPR", 1024, 38912, 200);
    run<128>("128 VGPR", 512, 38912, 200);
    run<128>("128 VGPR", 2048, 16384, 200);
    return 0;
}
