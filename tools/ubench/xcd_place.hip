// xcd_place.hip -- where do the workgroups of a dispatch run?  Prints the XCC_ID of the first workgroups of a series
// of launches: same grid again and again, grids that are not multiples of the XCD count, a one-workgroup kernel in
// between, two streams at once; tiny kernels and kernels shaped like the step (256 threads, 37 KB LDS, a few us).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xcd_place.hip -o tools/ubench/xcd_place.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned *out, long long spin) {
    extern __shared__ unsigned char smem[];
    if (spin > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
        if (threadIdx.x == 1000) smem[0] = 1;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
}

static void show(const char *tag, const std::vector<unsigned> &h, int grid) {
    printf("%-44s grid %4d :", tag, grid);
    for (int i = 0; i < grid && i < 12; ++i) printf(" %u", h[i]);
    bool rr = true;
    for (int i = 1; i < grid; ++i) rr = rr && h[i] == (h[0] + i) % 8;
    printf("  %s\n", rr ? "(round robin from the first)" : "(NOT round robin)");
}

int main() {
    unsigned *out;
    hipMalloc(&out, 1 << 20);
    hipStream_t s[2];
    hipStreamCreate(&s[0]);
    hipStreamCreate(&s[1]);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    std::vector<unsigned> h(65536);
    for (int shaped = 0; shaped < 2; ++shaped) {
        const int lds = shaped ? 37 * 1024 : 0;
        const long long spin = shaped ? 300 : 0;        // wall_clock64 ticks at 100 MHz: 3 us
        printf("---- %s\n", shaped ? "step-shaped kernels (256 threads, 37 KB LDS, 3 us)" : "tiny kernels");
        for (int grid : {8, 256, 10, 10, 10, 13, 256, 256, 1, 256, 3, 256}) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, s[0], out, spin);
            hipStreamSynchronize(s[0]);
            hipMemcpy(h.data(), out, grid * 4, hipMemcpyDeviceToHost);
            show("one stream, synchronised", h, grid);
        }
        // back to back on one stream without synchronising: 6 launches of 256 with a 1-workgroup kernel between
        for (int k = 0; k < 6; ++k) {
            hipLaunchKernelGGL(probe, dim3(256), dim3(256), lds, s[0], out + 4096 * k, spin);
            if (k == 2) hipLaunchKernelGGL(probe, dim3(1), dim3(256), lds, s[0], out + 60000, spin);
        }
        hipStreamSynchronize(s[0]);
        for (int k = 0; k < 6; ++k) {
            hipMemcpy(h.data(), out + 4096 * k, 256 * 4, hipMemcpyDeviceToHost);
            show(k == 3 ? "back to back (1-workgroup kernel before)" : "back to back", h, 256);
        }
        // two streams at once, 20 launches each of 256
        for (int k = 0; k < 8; ++k)
            for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(probe, dim3(256), dim3(256), lds, s[q], out + 4096 * (2 * k + q), spin);
        hipDeviceSynchronize();
        for (int k = 0; k < 16; ++k) {
            hipMemcpy(h.data(), out + 4096 * k, 256 * 4, hipMemcpyDeviceToHost);
            show(k & 1 ? "two streams at once: stream 1" : "two streams at once: stream 0", h, 256);
        }
        // ... of 250 (not a multiple of 8)
        for (int k = 0; k < 4; ++k)
            for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(probe, dim3(250), dim3(256), lds, s[q], out + 4096 * (2 * k + q), spin);
        hipDeviceSynchronize();
        for (int k = 0; k < 8; ++k) {
            hipMemcpy(h.data(), out + 4096 * k, 250 * 4, hipMemcpyDeviceToHost);
            show(k & 1 ? "two streams, grid 250: stream 1" : "two streams, grid 250: stream 0", h, 250);
        }
    }
    return 0;
}
