// Micro-benchmark: issue rate of the VALU instructions the SWAR stencil is made of (gfx950).
// Each kernel runs N_ITER iterations of 32 independent instructions of one kind per wave; with
// 16 waves per CU (4 per SIMD) the issue rate, not latency, bounds the time.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITER 2048
#define REP8(X) X X X X X X X X

#define KERNEL(name, ASM)                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t *out) {                            \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 9,     \
                 a5 = a0 + 11, a6 = a0 * 13, a7 = a0 + 17, b = blockIdx.x + 1, c = 0x55;    \
        for (int i = 0; i < N_ITER; ++i) {                                                  \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4),      \
                         "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));                    \
        }                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }

#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OPB(op) op " %0, %0, %8, %9 bitop3:0xe8\n" op " %1, %1, %8, %9 bitop3:0xe8\n" op " %2, %2, %8, %9 bitop3:0xe8\n" \
                op " %3, %3, %8, %9 bitop3:0xe8\n" op " %4, %4, %8, %9 bitop3:0xe8\n" op " %5, %5, %8, %9 bitop3:0xe8\n" \
                op " %6, %6, %8, %9 bitop3:0xe8\n" op " %7, %7, %8, %9 bitop3:0xe8\n"
#define OPD(op) op " %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                op " %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                op " %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                op " %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n" op " %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"

KERNEL(k_and, OP2("v_and_b32"))
KERNEL(k_or3, OP3("v_or3_b32"))
KERNEL(k_bitop3, OPB("v_bitop3_b32"))
KERNEL(k_alignbit, OP3("v_alignbit_b32"))
KERNEL(k_lshl_or, OP3("v_lshl_or_b32"))
KERNEL(k_and_or, OP3("v_and_or_b32"))
KERNEL(k_add3, OP3("v_add3_u32"))
KERNEL(k_lshr, OP2("v_lshrrev_b32"))
KERNEL(k_pk_add, OP2("v_pk_add_u16"))
KERNEL(k_pk_lshr, OP2("v_pk_lshrrev_b16"))
KERNEL(k_mul24, OP2("v_mul_u32_u24"))
KERNEL(k_mad24, OP3("v_mad_u32_u24"))
KERNEL(k_mul_lo, OP2("v_mul_lo_u32"))
KERNEL(k_fma, OP3("v_fma_f32"))
KERNEL(k_perm, OP3("v_perm_b32"))
KERNEL(k_mov_dpp, OPD("v_mov_b32_dpp"))
KERNEL(k_add_u32, OP2("v_add_u32"))
KERNEL(k_xor, OP2("v_xor_b32"))
KERNEL(k_bfe, OP3("v_bfe_u32"))
KERNEL(k_sdwa, "v_and_b32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
               "v_and_b32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")


#define OPL(op, lit) op " %0, " lit ", %0\n" op " %1, " lit ", %1\n" op " %2, " lit ", %2\n" op " %3, " lit ", %3\n" \
                     op " %4, " lit ", %4\n" op " %5, " lit ", %5\n" op " %6, " lit ", %6\n" op " %7, " lit ", %7\n"
#define OPS(op) op " %0, s4, %0\n" op " %1, s4, %1\n" op " %2, s4, %2\n" op " %3, s4, %3\n" \
                op " %4, s4, %4\n" op " %5, s4, %5\n" op " %6, s4, %6\n" op " %7, s4, %7\n"
#define OPBL(lit) "v_bitop3_b32 %0, %0, %8, " lit " bitop3:0xe8\n v_bitop3_b32 %1, %1, %8, " lit " bitop3:0xe8\n" \
                  "v_bitop3_b32 %2, %2, %8, " lit " bitop3:0xe8\n v_bitop3_b32 %3, %3, %8, " lit " bitop3:0xe8\n" \
                  "v_bitop3_b32 %4, %4, %8, " lit " bitop3:0xe8\n v_bitop3_b32 %5, %5, %8, " lit " bitop3:0xe8\n" \
                  "v_bitop3_b32 %6, %6, %8, " lit " bitop3:0xe8\n v_bitop3_b32 %7, %7, %8, " lit " bitop3:0xe8\n"
KERNEL(k_and_lit, OPL("v_and_b32", "0x0f000f00"))
KERNEL(k_and_inl, OPL("v_and_b32", "15"))
KERNEL(k_and_sgpr, OPS("v_and_b32"))
KERNEL(k_or, OP2("v_or_b32"))
KERNEL(k_lshl_inl, OPL("v_lshlrev_b32", "4"))
KERNEL(k_lshr_inl, OPL("v_lshrrev_b32", "4"))
KERNEL(k_bitop3_sgpr, OPBL("s4"))
KERNEL(k_cndmask, OP2("v_cndmask_b32"))
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
KERNEL(k_sub, OP2("v_sub_u32"))
KERNEL(k_max, OP2("v_max_u32"))
KERNEL(k_lshl_add, OP3("v_lshl_add_u32"))
KERNEL(k_xad, OP3("v_xad_u32"))
KERNEL(k_and_e64, OP2("v_and_b32_e64"))
KERNEL(k_pk_mul, OP2("v_pk_mul_lo_u16"))
KERNEL(k_ashr, OPL("v_ashrrev_i32", "4"))
KERNEL(k_add_lit, OPL("v_add_u32", "0x00e100e1"))
KERNEL(k_mul24_inl, OPL("v_mul_u32_u24", "15"))
KERNEL(k_cvt_pk, OP2("v_cvt_pk_u16_u32"))
KERNEL(k_msad, OP3("v_msad_u8"))
KERNEL(k_sad, OP3("v_sad_u16"))

template <typename K>
void run(const char *name, K kern, int per_iter, uint32_t *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 4;   // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 4 waves * N_ITER * per_iter
    double insts = 4.0 * N_ITER * per_iter;
    double cycles_at_2p4 = ms * 1e-3 * 2.4e9;
    printf("%-14s %8.3f ms  -> %.2f cycles @2.4GHz per wave-instruction per SIMD\n", name, ms, cycles_at_2p4 / insts);
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    run("v_and_b32", k_and, 64, out);
    run("v_xor_b32", k_xor, 64, out);
    run("v_add_u32", k_add_u32, 64, out);
    run("v_lshrrev_b32", k_lshr, 64, out);
    run("v_or3_b32", k_or3, 64, out);
    run("v_bitop3_b32", k_bitop3, 64, out);
    run("v_alignbit", k_alignbit, 64, out);
    run("v_lshl_or", k_lshl_or, 64, out);
    run("v_and_or", k_and_or, 64, out);
    run("v_add3_u32", k_add3, 64, out);
    run("v_bfe_u32", k_bfe, 64, out);
    run("v_perm_b32", k_perm, 64, out);
    run("v_pk_add_u16", k_pk_add, 64, out);
    run("v_pk_lshr_b16", k_pk_lshr, 64, out);
    run("v_mul_u32_u24", k_mul24, 64, out);
    run("v_mad_u32_u24", k_mad24, 64, out);
    run("v_mul_lo_u32", k_mul_lo, 64, out);
    run("v_fma_f32", k_fma, 64, out);
    run("v_mov_dpp", k_mov_dpp, 64, out);
    run("v_and_sdwa", k_sdwa, 64, out);
    run("and literal", k_and_lit, 64, out);
    run("and inline", k_and_inl, 64, out);
    run("and sgpr", k_and_sgpr, 64, out);
    run("v_or_b32", k_or, 64, out);
    run("lshl inline", k_lshl_inl, 64, out);
    run("lshr inline", k_lshr_inl, 64, out);
    run("ashr inline", k_ashr, 64, out);
    run("bitop3 sgpr", k_bitop3_sgpr, 64, out);
    run("v_cndmask", k_cndmask, 64, out);
    run("v_mov_b32", k_mov, 64, out);
    run("v_sub_u32", k_sub, 64, out);
    run("v_max_u32", k_max, 64, out);
    run("v_lshl_add_u32", k_lshl_add, 64, out);
    run("v_xad_u32", k_xad, 64, out);
    run("v_and_e64", k_and_e64, 64, out);
    run("v_pk_mul_lo_u16", k_pk_mul, 64, out);
    run("add literal", k_add_lit, 64, out);
    run("mul24 inline", k_mul24_inl, 64, out);
    run("v_cvt_pk_u16_u32", k_cvt_pk, 64, out);
    run("v_msad_u8", k_msad, 64, out);
    run("v_sad_u16", k_sad, 64, out);
    return 0;
}
