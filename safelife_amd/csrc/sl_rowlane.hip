// sl_rowlane.hip -- the fast gfx950 path: one board ROW per lane, two cells per 32-bit register.
//
// Mapping (template on the board shape H x W, H, W <= 64; see struct Geom / LaneMap):
//   * a 256-thread workgroup owns NB = 4*G consecutive boards, each wavefront G of them; a lane holds
//     one row of one board entirely in VGPRs as WS = ceil(W/2) words in the "split halves" layout:
//     word k = (cell k, cell k+WS), so the horizontal neighbours of both cells of a word are the two
//     cells of the adjacent word and only the row seam needs byte permutes;
//   * vertical neighbours: a DPP wave shift when the board can afford two halo lanes (25, 26 rows), a
//     DPP wave rotate for 64 rows, ds_bpermute otherwise;
//   * HBM -> LDS: the workgroup's boards are one contiguous, 16-byte aligned span moved by the
//     global_load_lds DMA (no VGPR staging); LDS -> HBM by lane-linear 16-byte stores.  The flat LDS
//     image is what the agent's action, the exit repaint, the on-device reset and the observation
//     epilogue work on; rows wider than 32 cells are stored bank-swizzled (Geom::cell);
//   * the 3x3 neighbourhood reduction is the commutative merge of sl_device.h in SWAR form: OR ("seen
//     once"), majority ("seen twice") and integer add (alive count) on whole registers, bitop3 for every
//     three-input function;
//   * random draws (spawners) are rare: eligible cells are flagged in the fast pass and resolved in a
//     wave-uniform slow path (DPP prefix count per board, PCG64 jump to the lane's first draw, the lane
//     then steps through its own flagged cells), preserving the reference's row-major draw order
//     (advance_board.c:115); kernels for spawner-free batches drop all of it;
//   * the score sum(points_table * alive_counts) is a per-cell byte gather from a table indexed by the
//     cell's relevant bits and the goal colour (slhip_env_prepare builds it);
//   * the training wrappers of env_wrappers.py run in the WRAP variants (leader lane, float64);
//   * the only workgroup barriers are the two around the HBM <-> LDS moves.
//
// Reference behaviour restated: advance_board.c:34-125 (CA step), :153-189 (occupancy), :217-300
// (actions), safelife_env.py:148-218 + safelife_game.py:505-552,684-719,746-761 (step / reset glue),
// safelife_env.py:105-146 + helper_utils.py:42-75 (observation), env_wrappers.py:32-213 (wrappers).
#include "sl_device.h"
#include <atomic>
#include <cstddef>
#include <cstring>
#include <type_traits>

#include "sl_kernels.h"
#include "sl_planes.h"

// A/B knobs of the span moves (cache policy of the LDS DMA loads, flavour of the span stores)
#ifndef SL_LOAD_AUX
#define SL_LOAD_AUX 0
#endif
#ifndef SL_STORE
#define SL_STORE 1   /* 0 plain, 1 non-temporal, 2 sc1 (write-through), 3 sc0 sc1.  Round 3, two-slice C3 step: 8.2-8.3 us
                        plain, 7.9 non-temporal, 7.95-8.0 either write-through form.  What the flavours change is the
                        kernel BOUNDARY: in-kernel clocks (tools/trace_overlap.py) put 2.0-2.5 us between a slice's last
                        acknowledged store and the first wave of its next launch, of which ~0.9 us are the write-back
                        of the 5.6 MB the launch left dirty in L2; streaming lines leave earlier.  Non-temporal stores
                        keep the ordinary coherence rules.  The write-through forms produced WRONG boards now and then
                        under concurrent launches in round 2 (64x64 spawner levels, ~1 run in 3) and stay a knob. */
#endif

namespace sl {
namespace rl {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 4;

// -DSL_TRACE: phase timestamps (s_memrealtime) of every wave are written to the `reward_t`
// argument, reinterpreted as long long [grid * WAVES, 16] (profiling builds only).
#ifdef SL_TRACE
#define SL_STAMP(i)                                                                     \
    do {                                                                                \
        if ((threadIdx.x & 63) == 0 && reward_t)                                        \
            ((long long *)reward_t)[(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SL_STAMP(i) do {} while (0)
#endif

// ---- instruction selection notes (measured with tools/ubench/valu_rate.hip on gfx950) -----------
// Per wave64 instruction a SIMD spends 2 cycles on v_and/or/xor/add/sub/lshrrev/mov and on
// v_bitop3_b32 with VGPR operands, and 4 cycles on everything else this kernel could use
// (v_lshlrev, v_alignbit, v_perm, v_or3/v_and_or/v_lshl_or/v_add3, v_pk_*, v_mul_u32_u24, DPP, SDWA)
// and on ANY VALU op with an SGPR operand.  Hence: three-input logic goes through bitop3, masks
// live in VGPRs (vreg()), shifts are right shifts, and the cell layout below needs no funnel shifts.
constexpr unsigned TA = 0xF0, TB = 0xCC, TC = 0xAA;      // truth-table columns of bitop3 operands
#define SL_BO3(expr, a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned)((expr)&0xFF))
#define BO3_OR3(a, b, c) SL_BO3(TA | TB | TC, a, b, c)
#define BO3_MAJ(a, b, c) SL_BO3((TA & TB) | (TA & TC) | (TB & TC), a, b, c)
#define BO3_AND_OR(a, b, c) SL_BO3((TA & TB) | TC, a, b, c)            /* (a & b) | c */
#define BO3_OR_AND(a, b, c) SL_BO3(TA | (TB & TC), a, b, c)            /* a | (b & c) */
#define BO3_ORAND(a, b, c) SL_BO3((TA | TB) & TC, a, b, c)             /* (a | b) & c */
#define BO3_INSERT(a, b, c) SL_BO3((TA & ~TC) | (TB & TC), a, b, c)    /* c ? b : a   */

__device__ __forceinline__ u32 vreg(u32 c) {     // a constant the compiler must keep in a VGPR
    u32 r = c;
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ u32 bperm(int byte_addr, u32 v) {
    return (u32)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
}
__device__ __forceinline__ u32 rot16(u32 x) { return __builtin_amdgcn_alignbit(x, x, 16); }
__device__ __forceinline__ u32 pack_lo_lo(u32 lo_src, u32 hi_src) {        // (lo(lo_src), lo(hi_src))
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u);
}
__device__ __forceinline__ u32 pack_hi_lo(u32 lo_src, u32 hi_src) {        // (hi(lo_src), lo(hi_src))
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040302u);
}
// Workgroup barrier that orders LDS traffic.  The explicit wait is not redundant: on the back edge of the step loop
// hipcc (ROCm 7.2) emitted the barrier of a plain __syncthreads() WITHOUT a preceding s_waitcnt lgkmcnt(0) although
// LDS stores and an LDS atomic were outstanding on that path -- a wave could read the mailbox before the leader's
// writes had landed and take a different number of barriers than the rest of its workgroup.
__device__ __forceinline__ void wg_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Inclusive prefix sum across the 64 lanes with DPP row shifts / row broadcasts (no LDS traffic).
__device__ __forceinline__ int wave_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// Register layout of a row ("split halves"): WS = ceil(W/2) words; word k holds cell k in its low
// half and cell k+WS in its high half (unused for k = WS-1 when W is odd).  The left / right
// neighbours of both cells of word k are the two cells of word k-1 / k+1, so the horizontal pass
// needs no intra-register shifts; only the seams (cells 0, WS-1, WS, W-1) need a byte permute.
enum { V_BPERM = 0, V_SHIFT = 1, V_ROTATE = 2 };

template <int H, int W>
struct Geom {
    static constexpr int HW = H * W;
    static constexpr int WS = (W + 1) / 2;
    static constexpr bool ODD = (W & 1) != 0;
    // Rows above / below a lane's row (VERT):
    //   V_SHIFT   every board also keeps copies of its last and first row in a lane in front of and a
    //             lane behind its H row lanes, so both neighbours are one DPP wave shift away (no LDS
    //             traffic).  Chosen when the two extra lanes per board do not cost a board per wave.
    //   V_ROTATE  H == 64: one board fills the wave, the torus is a DPP wave rotate.
    //   V_BPERM   ds_bpermute with per-lane addresses that wrap inside the board's lane group.
    static constexpr int VERT = H == 64 ? V_ROTATE : (64 / (H + 2) == 64 / H ? V_SHIFT : V_BPERM);
    static constexpr int GL = H + (VERT == V_SHIFT ? 2 : 0);   // lanes per board
    static constexpr int G = 64 / GL;                      // boards per wave
    static_assert(G >= 1, "H must be <= 64");
    static_assert(WS <= 32 && W >= 4 && H >= 4, "row-per-lane path: 4 <= W <= 64, H >= 4");
    static constexpr int WAVES_PER_SIMD = WS <= 16 ? 4 : 2;   // VGPR budget: 128 / 256 registers
    // fused step, LEAN variants of the narrow shapes: a fifth wavefront that holds no rows leads the workgroup
    // (four workgroups per CU -> five waves per SIMD -> 96 registers)
    // -- measured and switched off: the wave's lifetime drops (4.7 -> 4.57 us at 4096 envs) but the two-slice step does
    //    not move (8.26 us) and the one-launch step loses (9.6 -> 10.9 us: twenty waves per CU spread unevenly over
    //    the SIMDs)
#ifndef SL_LEADX
#define SL_LEADX 0              /* A/B knob: 1 = the LEAN variants of the 25 / 26-cell shapes get a fifth, row-less leader wave */
#endif
    static constexpr bool LEADX_OK = SL_LEADX && (W == 25 || W == 26);
    static constexpr int NL = G * GL;                      // lanes in use
    static constexpr int NB = WAVES * G;                   // boards per workgroup
    static_assert((NB * HW) % 8 == 0, "workgroup span must be a multiple of 16 bytes");
    static constexpr int SPAN = NB * HW * 2;               // bytes
    static constexpr int PAD = 16;
    static constexpr int REGION = SPAN + 2 * PAD;
    static constexpr int OFF_BOARD = 0;
    static constexpr int OFF_GOALS = REGION;
    static constexpr int OFF_RNG = 2 * REGION;             // NB x 4 u64
    static constexpr int OFF_BOX = OFF_RNG + NB * 32;      // NB x BoardBox: rows <-> leader lanes (fused step)
    static constexpr int OFF_REC = OFF_BOX + NB * 32;      // NB x sl_env_scalars: the leaders' working copies
    static constexpr int OFF_GSH = OFF_REC + NB * 64;      // per lane WS words: goal colours, pre-shifted
    // (boards wider than 32 cells keep the goal words in registers in every variant: the region shrinks
    //  to the few hundred bytes the observation epilogue parks its per-board parameters in)
    static constexpr int GSH_BYTES = WAVES_PER_SIMD < 4 ? 512 : WAVES * 64 * WS * 4;
    static constexpr int OFF_LUT = OFF_GSH + GSH_BYTES;             // 4 KiB compact score table
    static constexpr int OFF_MOVE = OFF_LUT + 4096;                 // NB x 16 bytes: the agents' moves, leader -> rows
    static constexpr int LDS_BYTES = OFF_MOVE + NB * 16;
    // training wrappers (WRAP variants only): per-board sl_wrap_state, the movement table, and -- when the
    // goal words occupy OFF_GSH -- the baseline rows of the side-effect count
    static constexpr int OFF_WST = LDS_BYTES;
    static constexpr int MVT_N = (H + W + SL_WRAP_MAX_PERIOD + 1) & ~1;        // doubles staged
    static constexpr int OFF_MVT = OFF_WST + NB * (int)sizeof(sl_wrap_state);
    static constexpr int OFF_BASE = OFF_MVT + MVT_N * 8;
    static constexpr int LDS_WRAP_GSHREG = OFF_BASE;                           // baseline rows reuse OFF_GSH
    static constexpr int LDS_WRAP_GSHLDS = OFF_BASE + WAVES * 64 * WS * 4;
    // "inaction" baseline of SimpleSideEffectPenalty (WRAP variants, when the flag is set): a third board image and
    // the baselines' generators, behind whichever wrapper layout the variant uses
    static constexpr int INACTION_BYTES = REGION + NB * 32;
    static constexpr int LDS_ADVANCE = OFF_RNG + NB * 32;  // advance_board needs no score state
    // LDS image of a board: row-major cells, except that for 128-byte rows (W = 64) the 16-byte chunks of
    // row y are XOR-swizzled with (y >> 1) & 7.  Unswizzled, all 64 row lanes would hit the same two banks
    // (row pitch = half the bank sweep: measured 95 % of the LDS cycles were bank conflicts); swizzled,
    // "chunk j of every row" spreads over all 64 banks.  The HBM<->LDS DMA applies the permutation on the
    // global side (an LDS slot is fixed per lane, the global address is free), so HBM keeps the plain layout.
    // Round 4: the same for EVERY row that is a whole number of 16-byte chunks (W a multiple of 8) -- the rows are then
    // read and written as aligned 16-byte chunks instead of cell by cell (the 16-bit accesses of the other shapes are
    // conflict-free only where the row pitch is an odd number of dwords, 25 and 26 cells: at 32 cells the 64 row lanes
    // hit two banks, at 48 four, at 40 eight), and where the chunk count CH = W / 8 is even the chunks of a row are
    // XOR-permuted with a key that runs through its low bits over sixteen rows: 64 cells (y >> 1) & 7, 32 cells
    // (y >> 2) & 3, 16 and 48 cells (y >> 3) & 1; an odd chunk count (8, 24, 40 cells) sweeps the banks by itself.
    // (The key has a period of sixteen rows and the DMA works on rows of the whole span: H must be a multiple of 16
    //  wherever there is a key.)
#ifndef SL_SWZ_ALL
#define SL_SWZ_ALL 1            /* A/B knob: 0 = only 64-cell rows are chunked and swizzled (rounds 1-3) */
#endif
    static constexpr int CH = W / 8;
    static constexpr bool SWZ = W == 64 || (SL_SWZ_ALL && W % 8 == 0 && (W == 8 || W == 24 || W == 40 || H % 16 == 0));
    static constexpr int KM = !SWZ ? 0 : W == 64 ? 7 : W == 32 ? 3 : (W == 16 || W == 48) ? 1 : 0;
    static constexpr int KS = W == 64 ? 1 : W == 32 ? 2 : 3;
    static_assert(KM == 0 || H % 16 == 0, "swizzled images: the key's period is sixteen rows");
    // Even rows that are not whole chunks (10, 12, 20, 26, 30 cells): every row of the image still starts on a dword
    // (the image starts 16-byte aligned, a board and a row are an even number of cells), so the rows are read and
    // written as W / 2 aligned dwords -- half the LDS instructions of the cell-by-cell form for the same number of
    // vector instructions (one v_perm per word of the split layout either way).
#ifndef SL_ROW_DWORDS
#define SL_ROW_DWORDS 1         /* A/B knob: 0 = such rows cell by cell (rounds 1-3) */
#endif
    static constexpr bool DW = SL_ROW_DWORDS && !SWZ && (W & 1) == 0;
    // Odd rows, READS only (a row's first or last cell shares its dword with a neighbouring row, which another lane
    // writes): the WS + 1 aligned dwords that cover the row, shifted into place per lane.
#ifndef SL_ROW_ODD_DWORDS
#define SL_ROW_ODD_DWORDS 1     /* A/B knob: 0 = odd rows are read cell by cell (rounds 1-3) */
#endif
    static constexpr bool ODW = SL_ROW_ODD_DWORDS && !SWZ && (W & 1) == 1;
    static __device__ __forceinline__ int key(int y) { return KM ? (y >> KS) & KM : 0; }
    static __device__ __forceinline__ int swz_chunk(int s) {           // LDS slot (16-byte chunk of the span) -> global chunk
        return KM ? s ^ key(s / CH) : s;
    }
    static __device__ __forceinline__ int cell(int y, int x) {          // (row, col) -> cell index in the image
        // (24-bit multiply: a quarter of the issue cycles of v_mul_lo_u32, and the operands are tiny)
        return KM ? __mul24(y, W) + (((x >> 3) ^ key(y)) << 3) + (x & 7) : __mul24(y, W) + x;
    }
    static __device__ __forceinline__ int flat(int i) {                 // row-major index -> cell index
        return KM ? cell(i / W, i % W) : i;
    }
    // validity of the halves of word k as a 0x0001-per-half mask
    static constexpr u32 vm1(int k) { return (ODD && k == WS - 1) ? 0x00000001u : 0x00010001u; }
};

template <int H, int W>
using RowWords = u32[Geom<H, W>::WS];

// ---- flat LDS image  <->  row-per-lane registers ------------------------------------------------
// Two 16-bit LDS reads per word (gfx950 has SRAM-ECC, so d16_hi loads do not preserve the other
// half and cannot merge in place).  The reads are volatile only to stop the compiler from fusing
// neighbouring cells into wide ds_read_b64 accesses that would be misaligned for odd row starts.
template <int H, int W>
__device__ __forceinline__ void read_row(const unsigned char *region, int gb, int r, RowWords<H, W> &b) {
    using Gm = Geom<H, W>;
    if (Gm::SWZ) {      // eight aligned 16-byte reads (chunk j sits at j ^ key), then one v_perm per word
        typedef const __attribute__((address_space(3))) u32x4 *lds_c128;
        lds_c128 row = (lds_c128)(region + Gm::PAD) + (gb * Gm::HW + r * W) / 8;
        const int key = Gm::key(r);
        u32 d[W / 2];
#pragma unroll
        for (int j = 0; j < W / 8; ++j) {
            const u32x4 v = row[j ^ key];
            d[4 * j + 0] = v.x;
            d[4 * j + 1] = v.y;
            d[4 * j + 2] = v.z;
            d[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < Gm::WS; ++k)        // (cell k, cell k + WS): both in the same half of their dwords
            b[k] = __builtin_amdgcn_perm(d[(k + Gm::WS) >> 1], d[k >> 1], (k & 1) ? 0x07060302u : 0x05040100u);
        return;
    }
    if (Gm::DW) {       // even rows: every row starts on a dword -- W / 2 aligned dword reads, one v_perm per word
        typedef const __attribute__((address_space(3), aligned(4))) u32 *lds_c32;
        lds_c32 row = (lds_c32)(region + Gm::PAD) + (gb * Gm::HW + r * W) / 2;
        u32 d[W / 2];
#pragma unroll
        for (int j = 0; j < W / 2; ++j) d[j] = row[j];
#pragma unroll
        for (int k = 0; k < Gm::WS; ++k) {      // cell c sits in dword c >> 1, half c & 1
            const u32 sel = ((k & 1) ? 0x0302u : 0x0100u) | ((((k + Gm::WS) & 1) ? 0x0706u : 0x0504u) << 16);
            b[k] = __builtin_amdgcn_perm(d[(k + Gm::WS) >> 1], d[k >> 1], sel);
        }
        return;
    }
    if (Gm::ODW) {      // odd rows (25, 15 cells): a row starts on a dword only every other row -- the aligned dwords
                        // around it (one more than the row has), moved down by the row's odd half with v_alignbit
        typedef const __attribute__((address_space(3), aligned(4))) u32 *lds_c32;
        const int first = gb * Gm::HW + r * W;                   // the row's first cell
        lds_c32 row = (lds_c32)(region + Gm::PAD) + (first >> 1);
        const u32 sh = (u32)(first & 1) << 4;
        u32 d[Gm::WS + 1], x[Gm::WS];
#pragma unroll
        for (int j = 0; j < Gm::WS + 1; ++j) d[j] = row[j];
#pragma unroll
        for (int j = 0; j < Gm::WS; ++j) x[j] = __builtin_amdgcn_alignbit(d[j + 1], d[j], sh);      // (cell 2j, cell 2j + 1)
#pragma unroll
        for (int k = 0; k < Gm::WS; ++k) {
            const u32 sel = ((k & 1) ? 0x0302u : 0x0100u) | ((((k + Gm::WS) & 1) ? 0x0706u : 0x0504u) << 16);
            b[k] = k == Gm::WS - 1 ? (x[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu : __builtin_amdgcn_perm(x[(k + Gm::WS) >> 1], x[k >> 1], sel);
        }
        return;
    }
    typedef const volatile __attribute__((address_space(3))) u16 *lds_cv16;
    lds_cv16 c = (lds_cv16)(region + Gm::PAD) + gb * Gm::HW + r * W;
    u32 lo[Gm::WS], hi[Gm::WS];
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) {
        lo[k] = c[k];
        hi[k] = (Gm::ODD && k == Gm::WS - 1) ? 0u : (u32)c[k + Gm::WS];
    }
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) b[k] = lo[k] | (hi[k] << 16);
}

template <int H, int W>
__device__ __forceinline__ void write_row(unsigned char *region, int gb, int r, const RowWords<H, W> &n) {
    using Gm = Geom<H, W>;
    if (Gm::SWZ) {
        u32x4 *row = (u32x4 *)(region + Gm::PAD) + (gb * Gm::HW + r * W) / 8;
        const int key = Gm::key(r);
#pragma unroll
        for (int j = 0; j < W / 8; ++j) {       // dword q = (cell 2q, cell 2q+1)
            u32 q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = 2 * (4 * j + i);
                q[i] = c0 < Gm::WS ? __builtin_amdgcn_perm(n[c0 + 1], n[c0], 0x05040100u)
                                   : __builtin_amdgcn_perm(n[c0 + 1 - Gm::WS], n[c0 - Gm::WS], 0x07060302u);
            }
            u32x4 v = {q[0], q[1], q[2], q[3]};
            row[j ^ key] = v;
        }
        return;
    }
    if (Gm::DW) {
        typedef __attribute__((address_space(3), aligned(4))) u32 *lds_32;
        lds_32 row = (lds_32)(region + Gm::PAD) + (gb * Gm::HW + r * W) / 2;
#pragma unroll
        for (int j = 0; j < W / 2; ++j) {       // dword j = (cell 2j, cell 2j + 1); cell c = low half of word c, or high of c - WS
            const int c0 = 2 * j, c1 = 2 * j + 1;
            const int w0 = c0 < Gm::WS ? c0 : c0 - Gm::WS, w1 = c1 < Gm::WS ? c1 : c1 - Gm::WS;
            const u32 sel = (c0 < Gm::WS ? 0x0100u : 0x0302u) | ((c1 < Gm::WS ? 0x0504u : 0x0706u) << 16);
            row[j] = __builtin_amdgcn_perm(n[w1], n[w0], sel);
        }
        return;
    }
#ifdef SL_EXP_WROW16
    volatile u16 *c = (volatile u16 *)(region + Gm::PAD) + gb * Gm::HW + r * W;
#else
    u16 *c = (u16 *)(region + Gm::PAD) + gb * Gm::HW + r * W;
#endif
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) {
        c[k] = (u16)n[k];
        if (!(Gm::ODD && k == Gm::WS - 1)) c[k + Gm::WS] = (u16)(n[k] >> 16);
    }
}

// A row straight from / to GLOBAL memory (round 5: the plain single-step kernels keep no goal image in LDS; a goal row
// is touched there only on the rare launches that find no cached goal words, or that evolve or reload a goal array).
// Rows that are whole 16-byte chunks (and therefore 16-byte aligned) go as chunks, the others cell by cell.
template <int H, int W>
__device__ __forceinline__ void read_row_global(const u16 *__restrict__ row, RowWords<H, W> &b) {
    using Gm = Geom<H, W>;
    if constexpr (W % 8 == 0) {
        u32 d[W / 2];
#pragma unroll
        for (int j = 0; j < W / 8; ++j) {
            const u32x4 v = ((const u32x4 *)row)[j];
            d[4 * j + 0] = v.x, d[4 * j + 1] = v.y, d[4 * j + 2] = v.z, d[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < Gm::WS; ++k)
            b[k] = __builtin_amdgcn_perm(d[(k + Gm::WS) >> 1], d[k >> 1], (k & 1) ? 0x07060302u : 0x05040100u);
    } else {
#pragma unroll
        for (int k = 0; k < Gm::WS; ++k) {
            const u32 lo = row[k];
            const u32 hi = (Gm::ODD && k == Gm::WS - 1) ? 0u : (u32)row[k + Gm::WS];
            b[k] = lo | (hi << 16);
        }
    }
}
template <int H, int W>
__device__ __forceinline__ void write_row_global(u16 *__restrict__ row, const RowWords<H, W> &n) {
    using Gm = Geom<H, W>;
    if constexpr (W % 8 == 0) {
#pragma unroll
        for (int j = 0; j < W / 8; ++j) {
            u32 q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = 2 * (4 * j + i);
                q[i] = c0 < Gm::WS ? __builtin_amdgcn_perm(n[c0 + 1], n[c0], 0x05040100u)
                                   : __builtin_amdgcn_perm(n[c0 + 1 - Gm::WS], n[c0 - Gm::WS], 0x07060302u);
            }
            ((u32x4 *)row)[j] = u32x4{q[0], q[1], q[2], q[3]};
        }
    } else {
#pragma unroll
        for (int k = 0; k < Gm::WS; ++k) {
            row[k] = (u16)n[k];
            if (!(Gm::ODD && k == Gm::WS - 1)) row[k + Gm::WS] = (u16)(n[k] >> 16);
        }
    }
}
// the split-halves words of a row held as consecutive cell pairs: pair j = (cell 2j, cell 2j + 1), an odd row's last cell
// alone in the low half of pair WS - 1
template <int H, int W>
__device__ __forceinline__ void words_from_pairs(const u32 (&tw)[Geom<H, W>::WS], RowWords<H, W> &b) {
    using Gm = Geom<H, W>;
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) {
        const u32 sel = ((k & 1) ? 0x0302u : 0x0100u) | ((((k + Gm::WS) & 1) ? 0x0706u : 0x0504u) << 16);
        b[k] = (Gm::ODD && k == Gm::WS - 1) ? (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu
                                            : __builtin_amdgcn_perm(tw[(k + Gm::WS) >> 1], tw[k >> 1], sel);
    }
}

// Left / right neighbour words of an array of per-cell words in the split layout.
template <int H, int W>
struct Seams {
    u32 left0;      // neighbours to the left of word 0
    u32 right_a;    // neighbours to the right of word WS-1
    u32 right_b;    // (W odd) neighbours to the right of word WS-2
};

template <int H, int W>
__device__ __forceinline__ Seams<H, W> make_seams(const RowWords<H, W> &q) {
    using Gm = Geom<H, W>;
    Seams<H, W> s;
    if (Gm::ODD) {
        // cells: word WS-2 = (WS-2, W-1), word WS-1 = (WS-1, -)
        s.left0 = pack_hi_lo(q[Gm::WS - 2], q[Gm::WS - 1]);      // (cell W-1, cell WS-1)
        s.right_b = pack_lo_lo(q[Gm::WS - 1], q[0]);             // right of (WS-2, W-1) = (WS-1, 0)
        s.right_a = rot16(q[0]);                                 // right of (WS-1, -)   = (WS, -)
    } else {
        s.left0 = rot16(q[Gm::WS - 1]);                          // (cell W-1, cell WS-1)
        s.right_a = rot16(q[0]);                                 // right of (WS-1, W-1) = (WS, 0)
        s.right_b = 0;
    }
    return s;
}

template <int H, int W>
__device__ __forceinline__ u32 left_of(const RowWords<H, W> &q, const Seams<H, W> &s, int k) {
    return k == 0 ? s.left0 : q[k - 1];
}
template <int H, int W>
__device__ __forceinline__ u32 right_of(const RowWords<H, W> &q, const Seams<H, W> &s, int k) {
    using Gm = Geom<H, W>;
    if (k == Gm::WS - 1) return s.right_a;
    if (Gm::ODD && k == Gm::WS - 2) return s.right_b;
    return q[k + 1];
}

// Which row of which board a lane holds.
template <int H, int W>
struct LaneMap {
    int g, r;           // board within the wave, row
    bool real;          // the lane owns row r (false: halo copy, V_SHIFT only)
    int up, dn;         // V_BPERM: ds_bpermute byte addresses of the lanes holding rows r-1 / r+1
    __device__ __forceinline__ LaneMap(int lane) {
        using Gm = Geom<H, W>;
        g = 0;
#pragma unroll
        for (int q = 1; q < Gm::G; ++q) g += (lane >= q * Gm::GL) ? 1 : 0;
        const int j = lane - g * Gm::GL;
        if (Gm::VERT == V_SHIFT) {
            real = j >= 1 && j <= H;
            r = j == 0 ? H - 1 : (j == H + 1 ? 0 : j - 1);
        } else {
            real = true;
            r = j;
        }
        const bool in = lane < Gm::NL;
        real = real && in;
        if (!in) r = 0;
        up = 4 * (in ? (r == 0 ? lane + H - 1 : lane - 1) : lane);
        dn = 4 * (in ? (r == H - 1 ? lane - (H - 1) : lane + 1) : lane);
    }
    // first / last lane of board q's group and the lane of its row 0
    static constexpr int first_lane(int q) { return q * Geom<H, W>::GL; }
    static constexpr int last_lane(int q) { return q * Geom<H, W>::GL + Geom<H, W>::GL - 1; }
};

template <int VERT>
__device__ __forceinline__ u32 from_above(int up, u32 v) {      // the value held by the lane of row r-1
    if (VERT == V_SHIFT) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);    // wave_shr:1
    if (VERT == V_ROTATE) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xF, 0xF, false);  // wave_ror:1
    return bperm(up, v);
}
template <int VERT>
__device__ __forceinline__ u32 from_below(int dn, u32 v) {      // the value held by the lane of row r+1
    if (VERT == V_SHIFT) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);    // wave_shl:1
    if (VERT == V_ROTATE) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x134, 0xF, 0xF, false);  // wave_rol:1
    return bperm(dn, v);
}

struct Elig {           // cells whose outcome needs a random draw: one bit per cell of the row
    u32 lo, hi;         // word k: low half -> lo bit WS-1-k, high half -> hi bit WS-1-k (WS > 16), or
                        // lo bit 16+WS-1-k with hi unused (WS <= 16)
    __device__ __forceinline__ bool any() const { return (lo | hi) != 0; }
    __device__ __forceinline__ void clear() { lo = hi = 0; }
};

struct Consts {         // masks held in VGPRs for the whole kernel
    u32 m1, once, anya, three, spawn, eight;
};
__device__ __forceinline__ Consts make_consts() {
    Consts c;
    c.m1 = vreg(0x00010001u);
    c.once = vreg(0x0E080E08u);     // bit 3 (destructible|exit) and the colour bits 9-11
    c.anya = vreg(0x00E100E1u);     // alive + preserving/inhibiting/spawning
    c.three = vreg(0x00030003u);
    c.spawn = vreg(0x00800080u);
    c.eight = vreg(0x00080008u);
    return c;
}

// ---- one CA step on registers ---------------------------------------------------------------------
// b: the row's cells (split layout).  n: new cells; halves whose outcome needs a random draw hold
// the spawned value tentatively and are flagged in `elig` (after the loop, word k's flags sit at bits
// WS-1-k (low half) and 16+WS-1-k (high half)).  up/dn: ds_bpermute byte addresses of rows r-1 / r+1.
//
// Per cell two 16-bit summaries are folded over the 3x3 block (advance_board.c:12-32 restated):
//   o: bit 0 alive | bit 3 (destructible|exit) if alive | bits 5-7 preserving/inhibiting/spawning |
//      bits 9-11 colour if alive          -> OR gives "seen once", majority gives "seen twice",
//      and plain integer addition of the words counts the alive bits in bits 0-2 (no field above
//      can carry across a 16-bit half: every half stays below 0x3000)
//   w: colour bits of spawner cells, plus, after each pass, the "seen twice" bits -- all at the
//      cells' native bit positions, so the new cell is assembled without shifts.
template <int H, int W, bool SPAWN, bool COLFIRST = false>
__device__ __forceinline__ void ca_rows(const RowWords<H, W> &b, RowWords<H, W> &n, Elig &elig, int up, int dn,
                                        const Consts &c) {
    using Gm = Geom<H, W>;
    constexpr int WS = Gm::WS;
    static_assert(!(SPAWN && COLFIRST), "column-first reduction: spawner-free variants only");
    RowWords<H, W> o, w;
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        const u32 bb = b[k];
        const u32 m = __umul24(bb & c.m1, 0x0E08u) + c.anya;        // once-bits where alive | always-bits
        o[k] = BO3_ORAND(bb, (bb >> 5) & c.eight, m);               // exit (bit 8) joins destructible on bit 3
        w[k] = SPAWN ? (bb & __umul24(bb & c.spawn, 0x1Cu)) : 0u;   // colours of spawners
    }
    // Spawner-free boards reduce the COLUMN first: the lanes above and below hand over the raw summary words
    // (two DPP moves per word instead of four: with the row pass first, its OR and its "twice" word both have to
    // travel), the per-column OR / majority / count stay in the lane, and the row pass runs on neighbouring
    // registers.  14 -> 10 vector instructions per word for the 3x3 reduction.  (Three arrays stay live between the
    // passes -- four with spawners: only the plain step's variant has the registers for it, COLFIRST.)
    RowWords<H, W> xc, mc, sc;
    Seams<H, W> sx = {0u, 0u, 0u}, sm = {0u, 0u, 0u}, ss = {0u, 0u, 0u};
    if (COLFIRST) {
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            const u32 U = from_above<Gm::VERT>(up, o[k]), D = from_below<Gm::VERT>(dn, o[k]);
            xc[k] = BO3_OR3(U, o[k], D);
            mc[k] = BO3_MAJ(U, o[k], D);
            sc[k] = U + o[k] + D;                                    // bits 0-1: alive cells in the column triple
        }
        sx = make_seams<H, W>(xc);
        sm = make_seams<H, W>(mc);
        ss = make_seams<H, W>(sc);
    }
    Seams<H, W> so = {0u, 0u, 0u}, sw = {0u, 0u, 0u};
    if (!COLFIRST) {
        so = make_seams<H, W>(o);
        if (SPAWN) sw = make_seams<H, W>(w);
    }
    elig.clear();
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        u32 X, tw, sum;
        if (COLFIRST) {
            const u32 xL = left_of<H, W>(xc, sx, k), xR = right_of<H, W>(xc, sx, k);
            X = BO3_OR3(xL, xc[k], xR);
            const u32 m2 = BO3_MAJ(xL, xc[k], xR);                   // at least two columns saw it
            const u32 mL = left_of<H, W>(mc, sm, k), mR = right_of<H, W>(mc, sm, k);
            const u32 t1 = BO3_OR3(mL, mc[k], mR);                   // a column saw it twice
            tw = BO3_OR_AND(t1, m2, c.once);                         // seen twice: bit 3 + colours
            const u32 sL = left_of<H, W>(sc, ss, k), sR = right_of<H, W>(sc, ss, k);
            sum = sL + sc[k] + sR;                                   // bits 0-2: alive count mod 8
        } else {
            const u32 oL = left_of<H, W>(o, so, k), oR = right_of<H, W>(o, so, k);
            const u32 wL = left_of<H, W>(w, sw, k), wR = right_of<H, W>(w, sw, k);
            // row pass
            const u32 xo = BO3_OR3(oL, o[k], oR);
            const u32 mj = BO3_MAJ(oL, o[k], oR);
            const u32 wr = SPAWN ? BO3_OR_AND(BO3_OR3(wL, w[k], wR), mj, c.once) : (mj & c.once);
            const u32 cs = oL + o[k] + oR;                           // bits 0-1: alive cells in the row triple
            const u32 ro = BO3_INSERT(xo, cs, c.three);
            // column pass
            const u32 Uo = from_above<Gm::VERT>(up, ro), Do = from_below<Gm::VERT>(dn, ro);
            const u32 Uw = from_above<Gm::VERT>(up, wr), Dw = from_below<Gm::VERT>(dn, wr);
            X = BO3_OR3(Uo, ro, Do);
            const u32 m2 = BO3_MAJ(Uo, ro, Do);
            const u32 xw2 = BO3_OR3(Uw, wr, Dw);
            tw = BO3_OR_AND(xw2, m2, c.once);                        // seen twice: bit 3 + colours
            sum = Uo + ro + Do;                                      // bits 0-2: alive count mod 8
        }
        const u32 c1 = sum >> 1, c2 = sum >> 2;
        const u32 s34 = SL_BO3((TA & TB & ~TC) | (~TA & ~TB & TC), sum, c1, c2);   // count in {3,4}
        const u32 is3 = SL_BO3(TA & TB & ~TC, sum, c1, c2);
        // rule (advance_board.c:94-124), evaluated at bit 0 of each half
        const u32 bb = b[k];
        const u32 fr = bb >> 4, pr = X >> 5, ih = X >> 6;
        const u32 keep_a = BO3_OR3(fr, pr, s34);
        const u32 keep_d = fr | ih;
        const u32 born = SL_BO3(TA & ~TB & ~TC, is3, keep_d, bb);
        u32 el = 0;                                                  // needs a random draw
        if (SPAWN) {
            const u32 e1 = SL_BO3(TA & ~TB & ~TC, X >> 7, keep_d, is3);
            const u32 vm = Gm::vm1(k) == 0x00010001u ? c.m1 : vreg(Gm::vm1(k));
            el = SL_BO3(TA & ~TB & TC, e1, bb, vm);
        }
        const u32 ne = SPAWN ? (born | el) : born;
        const u32 kp = SL_BO3((TA & TB) | (~TA & ~TC), bb, keep_a, ne);
        const u32 KM = __umul24(kp & c.m1, 0xFFFFu);
        const u32 NM = __umul24(ne & c.m1, 0xFFFFu);
        const u32 nv = BO3_AND_OR(tw, c.once, c.m1);                 // alive + inherited colours / destructible
        const u32 nm = SPAWN ? BO3_AND_OR(nv, NM, __umul24(el, 8u))  // spawned cells are always destructible
                             : (nv & NM);
        n[k] = BO3_AND_OR(bb, KM, nm);
        if (SPAWN) {
            if (WS <= 16) {
                elig.lo = elig.lo + elig.lo + el;       // both halves fit one word (hi flags at 16 + ...)
            } else {
                elig.lo = elig.lo + elig.lo + (el & 1u);
                elig.hi = elig.hi + elig.hi + (el >> 16);
            }
        }
    }
}

// Resolve the flagged halves with each board's PCG64 stream, row-major (wave-uniform call).
// rng_lds: this wave's G x {state_hi, state_lo, inc_hi, inc_lo}; advanced by the draws consumed.
//
// A lane's flagged cells, in row-major order, are the set bits of `ord` from the top down (low
// halves = cells 0..WS-1 in the upper 32 bits, high halves = cells WS..W-1 in the lower 32 bits, cell
// k of a part at bit WS-1-k).  Every lane jumps the board's generator to its own first draw (prefix
// sum of the flag counts over the board's lanes) and then steps through its cells; the wave iterates
// as often as its busiest lane has flags -- a handful -- rather than once per cell position.
template <int H, int W>
__device__ void resolve_draws(const RowWords<H, W> &b, RowWords<H, W> &n, const Elig &elig, u64 *rng_lds, int g,
                              double p, const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    constexpr int WS = Gm::WS;
    const u32 part_lo = WS <= 16 ? (elig.lo & 0xFFFFu) : elig.lo, part_hi = WS <= 16 ? (elig.lo >> 16) : elig.hi;
    const int mine = __popc(part_lo) + __popc(part_hi);
    const int incl = wave_scan(mine);
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < Gm::G; ++q) {
        const int lo = q ? __builtin_amdgcn_readlane(incl, LaneMap<H, W>::first_lane(q) - 1) : 0;
        const int hi = __builtin_amdgcn_readlane(incl, LaneMap<H, W>::last_lane(q));
        if (g == q) {
            before = lo;
            total = hi - lo;
        }
    }
    const int excl = incl - mine - before;
    const U128 st = {rng_lds[4 * g + 0], rng_lds[4 * g + 1]}, inc = {rng_lds[4 * g + 2], rng_lds[4 * g + 3]};
    wave_sync();     // every lane has read the old state before a leader replaces it
    u64 ord = ((u64)part_lo << 32) | part_hi, failed = 0;
    if (mine > 0) {
        U128 cur = pcg_jump(jump, excl, st, inc);
        while (ord) {
            const int pos = 63 - __clzll((long long)ord);
            cur = pcg_step(cur, inc);
            if (!(pcg_output_double(cur) < p)) failed |= 1ull << pos;      // advance_board.c:115
            ord &= ~(1ull << pos);
        }
        if (excl + mine == total) {     // the lane that made the board's last draw holds its new state
            rng_lds[4 * g + 0] = cur.hi;
            rng_lds[4 * g + 1] = cur.lo;
        }
    }
    if (__ballot(failed != 0)) {        // a failed draw keeps the old cell
        const u32 f_lo = (u32)(failed >> 32), f_hi = (u32)failed;
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            const u32 keep = ((f_lo >> (WS - 1 - k)) & 1u) * 0x0000FFFFu | ((f_hi >> (WS - 1 - k)) & 1u) * 0xFFFF0000u;
            n[k] = (n[k] & ~keep) | (b[k] & keep);
        }
    }
    wave_sync();
}

// The same for the bit-plane step (sl_planes.h): `elig` is a plane and so is the result, the cells whose draw
// succeeded.  One word per plane: bit 1+k = cell k, bit 17+k = cell WS+k -- row-major order = ascending bits of the
// low half, then of the high half.  Two words (64-cell rows): cell 32 i + k = bit k of word i.
// Multi-step kernels (life_occupancy: a thousand steps on one board) hand in a JumpCache: with the even deal a lane's
// first draw of a step is draw (lane << log2c) of the board's stream, and log2c -- the draws per lane, rounded up to a
// power of two -- hardly ever changes from one step to the next.  So the lane keeps ITS jump (the multiplier A^k and the
// increment's share C_k * inc, k = lane << log2c) instead of fetching the table entry from global memory and
// multiplying it out at every step: one 128-bit multiply and one L2 round trip less on the step's critical chain.
struct JumpCache {             // (in registers: kept in LDS instead -- 2 KB per wave, four 8-byte reads per step -- the
    int log2c = -1;             //  pass of C5 ran 13.9 instead of 12.6 ms, slower than without any cache, 13.4)
    U128 mult = {0, 0}, plus_inc = {0, 0};
    U128 plus64 = {0, 0};       // (the round-by-round deal: log2c == 64, mult / plus_inc the lane's jump by lane + 1)
    // (round 5, measured and dropped: the board's generator itself kept in registers from step to step, handed round by
    //  v_readlane from the lane of the last draw instead of through LDS -- 10.4-10.6 ms either way for the pass of C5)
};
__device__ __forceinline__ U128 pcg_jump_cached(const Jump *__restrict__ table, int k, int log2c, U128 state, U128 inc,
                                                JumpCache *jc) {
    if (!jc) return pcg_jump(table, k, state, inc);
    if (jc->log2c != log2c) {
        const Jump j = table[k];
        jc->mult = U128{j.mult_hi, j.mult_lo};
        jc->plus_inc = mul128(U128{j.plus_hi, j.plus_lo}, inc);
        jc->log2c = log2c;
    }
    return add128(mul128(jc->mult, state), jc->plus_inc);
}

// A lane's draw outcomes -> its flagged cells: bit k of R is the outcome of the lane's k-th flagged cell in row-major
// order (ascending bits, word 0 first).  Round 5: the words of a plane are dealt SIDE BY SIDE, two cells of each per trip
// (word 1's outcomes start popc(word 0) bits into R; a word that has run out of cells idles: its lowest bit is 0) -- the
// wave loops ceil(busiest word / 2) times instead of once per cell and word, and the trips' chains are independent.
// The loop is the longest serial stretch of a spawner board's step: 64x64 navigation, ~8 + ~8 flagged cells in the
// busiest row's two words: 16 trips -> 4.
#ifndef SL_DEAL_PAIRS
#define SL_DEAL_PAIRS 2         /* cells of each word per trip (0: the words one after the other, a cell per trip; 4: no faster) */
#endif
#ifndef SL_STRIDE_DEAL
#define SL_STRIDE_DEAL 1        /* A/B knob: 0 = the blocked deal for every step */
#endif
#ifndef SL_STRIDE_ROUNDS2
#define SL_STRIDE_ROUNDS2 4     /* two boards per wave: rounds (of 32 draws per board) the round-by-round deal takes; a step
                                   with more draws goes row by row (below) -- the blocked deal is not kept beside it: both
                                   together cost the 25x25 step kernels the registers of their scratch-free build */
#endif
template <int NW>
__device__ __forceinline__ void deal_outcomes(const pl::Pl<NW> &elig, u64 R, pl::Pl<NW> &ok) {
    if constexpr (SL_DEAL_PAIRS && NW <= 2) {
        u32 t[NW], g[NW], r[NW];
        t[0] = elig.w[0], g[0] = 0, r[0] = (u32)R;
        if constexpr (NW == 2) {
            t[1] = elig.w[1], g[1] = 0;
            r[1] = (u32)(R >> __popc(t[0]));    // (popc <= 32; the window holds the lane's <= 64 outcomes)
        }
        u32 any = t[0];
        if constexpr (NW == 2) any |= t[1];
        while (any) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
#pragma unroll
                for (int k = 0; k < SL_DEAL_PAIRS; ++k) {
                    const u32 bit = t[i] & (0u - t[i]);
                    g[i] |= (0u - (r[i] & 1u)) & bit;
                    r[i] >>= 1;
                    t[i] ^= bit;
                }
            }
            any = t[0];
            if constexpr (NW == 2) any |= t[1];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) ok.w[i] = g[i];
    } else {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            u32 todo = elig.w[i], got = 0;
            while (todo) {
                const u32 bit = todo & (0u - todo);
                got |= (R & 1ull) ? bit : 0u;
                R >>= 1;
                todo ^= bit;
            }
            ok.w[i] = got;
        }
    }
}

// (JC is a template parameter, not just a null pointer: the extra argument alone cost the fused step's spawner
//  variants five registers -- and the 25x25 one its scratch-free build, which the queue launcher relies on)
template <int H, int W, int NW, bool JC = false>
__device__ pl::Pl<NW> resolve_draws_planes(const pl::Pl<NW> &elig, u64 *rng_lds, int g, double p,
                                           const Jump *__restrict__ jump, JumpCache *jc_arg = nullptr) {
    JumpCache *const jc = JC ? jc_arg : nullptr;
    using Gm = Geom<H, W>;
    int mine = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) mine += __popc(elig.w[i]);
    const int incl = wave_scan(mine);
    int before = 0, total = 0;
    int tot_q[Gm::G];                   // (wave-uniform) draws of each of the wave's boards
#pragma unroll
    for (int q = 0; q < Gm::G; ++q) {
        const int lo = q ? __builtin_amdgcn_readlane(incl, LaneMap<H, W>::first_lane(q) - 1) : 0;
        const int hi = __builtin_amdgcn_readlane(incl, LaneMap<H, W>::last_lane(q));
        tot_q[q] = hi - lo;
        if (g == q) {
            before = lo;
            total = hi - lo;
        }
    }
    const int excl = incl - mine - before;
    constexpr int MAXR = 8;             // rounds of the round-by-round deal (below)
    const bool stride_now = Gm::G == 1 && SL_STRIDE_DEAL && total <= 64 * MAXR;        // (wave-uniform)
    const U128 st = {rng_lds[4 * g + 0], rng_lds[4 * g + 1]}, inc = {rng_lds[4 * g + 2], rng_lds[4 * g + 3]};
    // two boards per wave: lanes 0-31 make the draws of board 0, lanes 32-63 those of board 1 (below), whatever
    // rows they hold
    const int wq = Gm::G == 2 ? (int)(__lane_id() >> 5) : 0;
    U128 wst = st, winc = inc;
    if constexpr (Gm::G == 2) {
        wst = {rng_lds[4 * wq + 0], rng_lds[4 * wq + 1]};
        winc = {rng_lds[4 * wq + 2], rng_lds[4 * wq + 3]};
    }
    wave_sync();     // every lane has read the old state before a leader replaces it
    pl::Pl<NW> ok;
#pragma unroll
    for (int i = 0; i < NW; ++i) ok.w[i] = 0;
    const u64 thr = draw_threshold(p);
    if constexpr (Gm::G == 1) {
        // One board per wave: the board's draws are dealt out EVENLY over the 64 lanes instead of row by row (the
        // cells around a cluster of spawners all sit in a few rows: on the 64x64 navigation levels the busiest row
        // has ~15 of a board's ~250 draws, an even share is 4).  Lane j makes draws [j c, (j + 1) c), c a power
        // of two; 32 / c neighbouring lanes' outcome bits make one dword (xor swizzles); a row's lane then pulls
        // the dwords that hold ITS draws' outcomes -- consecutive bits from its prefix offset on -- across the
        // wave and deals them to its flagged cells in order.
        if (stride_now) {
            // Round 5: the deal goes ROUND by round -- lane j makes draws j, j + 64, j + 128 ... -- instead of in blocks
            // of 2^log2c per lane.  A lane's first draw is a jump by j + 1 from the board's state, the SAME jump at
            // every step (the cached multiplier never changes), every further one a step of 64 (the table's entry 64:
            // wave-uniform); the outcome bits of round r ARE the compare's lane mask -- draw 64 r + j is bit j of a
            // scalar pair -- so the outcome string needs no cross-lane assembly (was: five dependent LDS swizzles) and
            // a row's lane shifts its window out of the two pairs that hold it.  ceil(total / 64) multiplies per lane
            // instead of 1 + the next power of two.
            const int lane = (int)__lane_id();
            const int rounds = (total + 63) >> 6;
            const u32 bthr_hi = __builtin_amdgcn_readlane((u32)(thr >> 32), LaneMap<H, W>::first_lane(0) + 1);
            const u32 bthr_lo = __builtin_amdgcn_readlane((u32)thr, LaneMap<H, W>::first_lane(0) + 1);
            const u64 bthr = ((u64)bthr_hi << 32) | bthr_lo;
            u64 B[MAXR + 1];
#pragma unroll
            for (int r = 0; r <= MAXR; ++r) B[r] = 0;
            if (total > 0) {
                U128 cur = st, plus64 = {0, 0};
                const bool in0 = lane < total;
                if (jc) {
                    if (jc->log2c != 64) {      // (every lane, whether it draws at this step or not)
                        const Jump j = jump[lane + 1];
                        jc->mult = U128{j.mult_hi, j.mult_lo};
                        jc->plus_inc = mul128(U128{j.plus_hi, j.plus_lo}, inc);
                        jc->plus64 = mul128(U128{jump[64].plus_hi, jump[64].plus_lo}, inc);
                        jc->log2c = 64;
                    }
                    if (in0) cur = add128(mul128(jc->mult, st), jc->plus_inc);
                    plus64 = jc->plus64;
                } else {
                    if (in0) cur = pcg_jump(jump, lane + 1, st, inc);
                    if (rounds > 1) plus64 = mul128(U128{jump[64].plus_hi, jump[64].plus_lo}, inc);
                }
                B[0] = __ballot(in0 && pcg_output_u53(cur) < bthr);                 // advance_board.c:115
                const U128 mult64 = {jump[64].mult_hi, jump[64].mult_lo};
#pragma unroll
                for (int r = 1; r < MAXR; ++r) {
                    if (r < rounds) {
                        const bool in = lane + 64 * r < total;
                        if (in) cur = add128(mul128(mult64, cur), plus64);
                        B[r] = __ballot(in && pcg_output_u53(cur) < bthr);
                    }
                }
                if (lane == ((total - 1) & 63)) {       // the lane that made the board's last draw holds its new state
                    rng_lds[4 * g + 0] = cur.hi;
                    rng_lds[4 * g + 1] = cur.lo;
                }
            }
            // my draws' outcomes: bits [excl, excl + mine) of the string B[0] | B[1] << 64 | ...
            const int w0 = excl >> 6, sh = excl & 63;
            u64 lo = B[0], hi = B[1];
#pragma unroll
            for (int r = 1; r < MAXR; ++r) {
                if (r < rounds && w0 == r) {
                    lo = B[r];
                    hi = B[r + 1];
                }
            }
            u64 R = lo >> sh;
            if (sh) R |= hi << (64 - sh);
            deal_outcomes<NW>(elig, R, ok);
            wave_sync();
            return ok;
        }
        if (total <= 32 * 64) {
            const int lane = (int)__lane_id();
            const int need = (total + 63) >> 6;
            const int log2c = need <= 1 ? 0 : 32 - __clz(need - 1);            // c = 2^log2c >= total / 64
            const int first = lane << log2c;
            const int n_here = min(max(total - first, 0), 1 << log2c);
            // (the threshold is the BOARD's: a lane that holds none of its rows -- halo copies, idle lanes of boards
            //  of fewer than 64 rows -- makes draws too and may carry no probability of its own)
            const u32 bthr_hi = __builtin_amdgcn_readlane((u32)(thr >> 32), LaneMap<H, W>::first_lane(0) + 1);
            const u32 bthr_lo = __builtin_amdgcn_readlane((u32)thr, LaneMap<H, W>::first_lane(0) + 1);
            const u64 bthr = ((u64)bthr_hi << 32) | bthr_lo;
            u32 bits = 0;
            if (n_here > 0) {
                U128 cur = pcg_jump_cached(jump, first, log2c, st, inc, jc);
                for (int i = 0; i < n_here; ++i) {
                    cur = pcg_step(cur, inc);
                    bits |= (pcg_output_u53(cur) < bthr ? 1u : 0u) << i;        // advance_board.c:115
                }
                if (first + n_here == total) {      // the lane that made the board's last draw holds its new state
                    rng_lds[4 * g + 0] = cur.hi;
                    rng_lds[4 * g + 1] = cur.lo;
                }
            }
            const int per = 32 >> log2c;                                        // lanes per dword of outcomes
            u32 v = bits << ((lane & (per - 1)) << log2c);
            if (per > 1) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x041F);      // xor 1
            if (per > 2) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x081F);      // xor 2
            if (per > 4) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);      // xor 4
            if (per > 8) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);      // xor 8
            if (per > 16) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);     // xor 16
            // (round 5, measured and dropped: DPP quad permutes and row mirrors for the steps inside a row of sixteen
            //  lanes -- bit exact, and no faster: the pass of C5 12.0-12.1 ms either way, C4's step 8.04-8.14)
            // dword d of the outcome string sits in lanes [d per, (d + 1) per); mine start at bit `excl`
            const int w0 = excl >> 5, sh = excl & 31;
            const int log2per = 5 - log2c;
            const u32 d0 = bperm(4 * min(63, w0 << log2per), v), d1 = bperm(4 * min(63, (w0 + 1) << log2per), v);
            const u32 d2 = bperm(4 * min(63, (w0 + 2) << log2per), v);
            u64 R = (((u64)d1 << 32) | d0) >> sh;
            if (sh) R |= (u64)d2 << (64 - sh);
            deal_outcomes<NW>(elig, R, ok);
            wave_sync();
            return ok;
        }
    }
    if constexpr (Gm::G == 2) {
        // Two boards per wave (25x25, 26x26): the same even deal, each board's draws over a 32-lane half of the wave
        // -- the cells that draw sit around a few spawner clusters, i.e. in a few rows, and row by row the wave would
        // loop as often as its busiest row has such cells.  Lane j of half q makes draws [j c, (j + 1) c) of board
        // q (one c for both boards); the xor swizzles stay inside a half; a row's lane pulls its dwords from the
        // half of its OWN board.
        const int tmax = max(tot_q[0], tot_q[1]);
        constexpr int MAXR2 = SL_STRIDE_ROUNDS2;     // (more rounds cost the 25x25 step kernels registers they do not have)
        if (SL_STRIDE_DEAL && tmax <= 32 * MAXR2) {
            // the round-by-round deal on the two halves: lane j of half q makes draws j, j + 32, ... of board q -- a jump
            // by j + 1, then steps of 32 (the table's entry 32) -- and round r's outcomes are the compare's lane mask,
            // board 0's in its low dword, board 1's in its high one.  A 25x25 board with a spawner or two draws ~10
            // times a step: one round, i.e. the jump alone (was: jump + one step + five swizzles + three bpermutes).
            const int lane = (int)__lane_id(), j = lane & 31;
            const int rounds = (tmax + 31) >> 5;
            const int total_w = wq ? tot_q[1] : tot_q[0];
            const u32 tl0 = __builtin_amdgcn_readlane((u32)thr, LaneMap<H, W>::first_lane(0) + 1);
            const u32 th0 = __builtin_amdgcn_readlane((u32)(thr >> 32), LaneMap<H, W>::first_lane(0) + 1);
            const u32 tl1 = __builtin_amdgcn_readlane((u32)thr, LaneMap<H, W>::first_lane(1) + 1);
            const u32 th1 = __builtin_amdgcn_readlane((u32)(thr >> 32), LaneMap<H, W>::first_lane(1) + 1);
            const u64 wthr = wq ? (((u64)th1 << 32) | tl1) : (((u64)th0 << 32) | tl0);
            u64 B[MAXR2];
#pragma unroll
            for (int r = 0; r < MAXR2; ++r) B[r] = 0;
            if (tmax > 0) {
                U128 cur = wst, plus32 = {0, 0};
                const bool in0 = j < total_w;
                if (jc) {
                    if (jc->log2c != 32) {      // (every lane, whether it draws at this step or not)
                        const Jump jj = jump[j + 1];
                        jc->mult = U128{jj.mult_hi, jj.mult_lo};
                        jc->plus_inc = mul128(U128{jj.plus_hi, jj.plus_lo}, winc);
                        jc->plus64 = mul128(U128{jump[32].plus_hi, jump[32].plus_lo}, winc);
                        jc->log2c = 32;
                    }
                    if (in0) cur = add128(mul128(jc->mult, wst), jc->plus_inc);
                    plus32 = jc->plus64;
                } else {
                    if (in0) cur = pcg_jump(jump, j + 1, wst, winc);
                    if (rounds > 1) plus32 = mul128(U128{jump[32].plus_hi, jump[32].plus_lo}, winc);
                }
                B[0] = __ballot(in0 && pcg_output_u53(cur) < wthr);                 // advance_board.c:115
                const U128 mult32 = {jump[32].mult_hi, jump[32].mult_lo};
#pragma unroll
                for (int r = 1; r < MAXR2; ++r) {
                    if (r < rounds) {
                        const bool in = j + 32 * r < total_w;
                        if (in) cur = add128(mul128(mult32, cur), plus32);
                        B[r] = __ballot(in && pcg_output_u53(cur) < wthr);
                    }
                }
                if (total_w > 0 && j == ((total_w - 1) & 31)) {     // the lane that made its board's last draw
                    rng_lds[4 * wq + 0] = cur.hi;
                    rng_lds[4 * wq + 1] = cur.lo;
                }
            }
            // my draws' outcomes: bits [excl, excl + mine) of MY board's string, dword r = my board's half of B[r]
            const int w0 = excl >> 5, sh = excl & 31;
            u32 d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
            for (int r = 0; r < MAXR2; ++r) {
                if (r < rounds) {
                    const u32 v = g ? (u32)(B[r] >> 32) : (u32)B[r];
                    if (w0 == r) d0 = v;
                    if (w0 + 1 == r) d1 = v;
                    if (w0 + 2 == r) d2 = v;
                }
            }
            u64 R = (((u64)d1 << 32) | d0) >> sh;
            if (sh) R |= (u64)d2 << (64 - sh);
            deal_outcomes<NW>(elig, R, ok);
            wave_sync();
            return ok;
        }
#if !SL_STRIDE_DEAL
        if (tmax <= 32 * 32) {
            const int lane = (int)__lane_id(), j = lane & 31;
            const int need = (tmax + 31) >> 5;
            const int log2c = need <= 1 ? 0 : 32 - __clz(need - 1);            // c = 2^log2c >= tmax / 32
            const int total_w = wq ? tot_q[1] : tot_q[0];
            const int first = j << log2c;
            const int n_here = min(max(total_w - first, 0), 1 << log2c);
            // (the threshold is the BOARD's: taken from a lane that holds one of its rows)
            const u32 tl0 = __builtin_amdgcn_readlane((u32)thr, LaneMap<H, W>::first_lane(0) + 1);
            const u32 th0 = __builtin_amdgcn_readlane((u32)(thr >> 32), LaneMap<H, W>::first_lane(0) + 1);
            const u32 tl1 = __builtin_amdgcn_readlane((u32)thr, LaneMap<H, W>::first_lane(1) + 1);
            const u32 th1 = __builtin_amdgcn_readlane((u32)(thr >> 32), LaneMap<H, W>::first_lane(1) + 1);
            const u64 wthr = wq ? (((u64)th1 << 32) | tl1) : (((u64)th0 << 32) | tl0);
            u32 bits = 0;
            if (n_here > 0) {
                U128 cur = pcg_jump_cached(jump, first, log2c, wst, winc, jc);
                for (int i = 0; i < n_here; ++i) {
                    cur = pcg_step(cur, winc);
                    bits |= (pcg_output_u53(cur) < wthr ? 1u : 0u) << i;        // advance_board.c:115
                }
                if (first + n_here == total_w) {    // the lane that made the board's last draw holds its new state
                    rng_lds[4 * wq + 0] = cur.hi;
                    rng_lds[4 * wq + 1] = cur.lo;
                }
            }
            const int per = 32 >> log2c;                                        // lanes per dword of outcomes
            u32 v = bits << ((lane & (per - 1)) << log2c);
            if (per > 1) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x041F);      // xor 1
            if (per > 2) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x081F);      // xor 2
            if (per > 4) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);      // xor 4
            if (per > 8) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);      // xor 8
            if (per > 16) v |= (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);     // xor 16
            // (round 5, measured and dropped: DPP quad permutes and row mirrors for the steps inside a row of sixteen
            //  lanes -- bit exact, and no faster: the pass of C5 12.0-12.1 ms either way, C4's step 8.04-8.14)
            // dword d of board q's outcome string sits in lanes [32 q + d per, 32 q + (d + 1) per)
            const int w0 = excl >> 5, sh = excl & 31;
            const int log2per = 5 - log2c;
            const int base = 32 * g;
            const u32 d0 = bperm(4 * (base + min(31, w0 << log2per)), v);
            const u32 d1 = bperm(4 * (base + min(31, (w0 + 1) << log2per)), v);
            const u32 d2 = bperm(4 * (base + min(31, (w0 + 2) << log2per)), v);
            u64 R = (((u64)d1 << 32) | d0) >> sh;
            if (sh) R |= (u64)d2 << (64 - sh);
            deal_outcomes<NW>(elig, R, ok);
            wave_sync();
            return ok;
        }
#endif
    }
    if (mine > 0) {
        U128 cur = pcg_jump(jump, excl, st, inc);
        // the parts of the row in row-major order: (low half, high half) of the one word, or the two words
        u32 part[2] = {NW == 1 ? (elig.w[0] & 0xFFFFu) : elig.w[0], NW == 1 ? (elig.w[0] >> 16) : elig.w[NW - 1]};
        u32 okp[2] = {0u, 0u};
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            u32 todo = part[half], got = 0;
            while (todo) {
                const u32 bit = todo & (0u - todo);
                cur = pcg_step(cur, inc);
                if (pcg_output_u53(cur) < thr) got |= bit;                      // advance_board.c:115
                todo ^= bit;
            }
            okp[half] = got;
        }
        if (NW == 1) {
            ok.w[0] = okp[0] | (okp[1] << 16);
        } else {
            ok.w[0] = okp[0];
            ok.w[NW - 1] = okp[1];
        }
        if (excl + mine == total) {     // the lane that made the board's last draw holds its new state
            rng_lds[4 * g + 0] = cur.hi;
            rng_lds[4 * g + 1] = cur.lo;
        }
    }
    wave_sync();
    return ok;
}

// One CA step on a lane's row, in place.  Rows of up to 28 cells take the bit-plane form (sl_planes.h), wider
// ones the word form above.  `mine`: the lane owns a row of a board that is advancing.  Returns (wave-uniform)
// whether any cell of the wave changed -- the word form does not know and says yes.  `take`: the lane takes the
// new row even if it is not `mine` (the halo copies of the word form compute their own rows).
template <int H, int W>
#ifndef SL_SPLIT_PLANES_MULTI
#define SL_SPLIT_PLANES_MULTI 1 /* A/B knob: 0 = advance_board (n > 1) and life_occupancy keep rows of 32..48 cells in word form */
#endif
constexpr bool use_planes_multi() {                     // rows kept in plane form across steps
    return (W + 1) / 2 + 2 <= 16 || W == 64 || (SL_SPLIT_PLANES_MULTI && W >= 32 && W < 64 && (W & 1) == 0);
}
template <int H, int W>
constexpr bool use_planes() {       // single steps: also even rows of 32 to 60 cells (two words per plane, one per half;
                                    // at 30 cells the word form is the faster one: 8.8 against 9.3-9.8 us per step)
#ifndef SL_SPLIT_PLANES
#define SL_SPLIT_PLANES 1       /* A/B knob: 0 = rows of 30 to 48 cells keep the word form of the CA (rounds 1-3) */
#endif
    return use_planes_multi<H, W>() || (SL_SPLIT_PLANES && W >= 32 && W < 64 && (W & 1) == 0);
}

template <int H, int W, bool SPAWN, bool COLFIRST>
__device__ __forceinline__ bool ca_step(RowWords<H, W> &b, bool mine, bool take, int up, int dn, const Consts &c,
                                        const pl::PConsts &pc, u64 *rng_lds, int g, double p,
                                        const Jump *__restrict__ jump, u32 *row_changed = nullptr) {
    using Gm = Geom<H, W>;
    if constexpr (use_planes<H, W>()) {
        static_assert((int)pl::PV_BPERM == (int)V_BPERM && (int)pl::PV_SHIFT == (int)V_SHIFT && (int)pl::PV_ROTATE == (int)V_ROTATE, "");
        const pl::VCtx<Gm::VERT> vc = {up, dn};
        const u32 realm = mine ? vreg(pl::PG<W>::REAL) : 0u;
        constexpr int NW = pl::PG<W>::NW;
        return pl::ca_planes<W, Gm::VERT, SPAWN>(b, vc, realm, pc, [&](const pl::Pl<NW> &elig) {
            return resolve_draws_planes<H, W, NW>(elig, rng_lds, g, p, jump);
        }, row_changed);
    } else {
        Elig elig;
        RowWords<H, W> n;
        ca_rows<H, W, SPAWN, COLFIRST>(b, n, elig, up, dn, c);
        if (!mine) elig.clear();
        if (SPAWN && __ballot(elig.any())) resolve_draws<H, W>(b, n, elig, rng_lds, g, p, jump);
        if (row_changed) *row_changed = 1u;              // (the word form does not know)
        if (take) {
#pragma unroll
            for (int k = 0; k < Gm::WS; ++k) b[k] = n[k];
        }
        return true;
    }
}

// Sum of a per-lane value over the lanes of the caller's board (every lane gets its board's total).
template <int H, int W>
__device__ __forceinline__ int group_total(int v, int g) {
    constexpr int G = Geom<H, W>::G;
    const int incl = wave_scan(v);
    int total = 0;
#pragma unroll
    for (int q = 0; q < G; ++q) {
        const int lo = q ? __builtin_amdgcn_readlane(incl, LaneMap<H, W>::first_lane(q) - 1) : 0;
        const int hi = __builtin_amdgcn_readlane(incl, LaneMap<H, W>::last_lane(q));
        if (g == q) total = hi - lo;
    }
    return total;
}

// ---- score ----------------------------------------------------------------------------------------
// sum(points_table * alive_counts) as one byte gather per cell.  The table index is the cell itself
// masked to the bits that matter (alive 0, pushable 2, destructible 3, frozen 4, colour 9-11,
// pullable 15) with the goal colour dropped into the free bits 5-7, so forming it costs one bitop3
// per two cells.  Entries the filter of advance_board.c:201 excludes hold 0.  Two forms of the table
// (slhip_env_prepare builds both; points must fit int8):
//   * 16-bit index, 64 KiB per points table, gathered from global memory (any number of tables);
//   * 12-bit index (pullable folded onto bit 8), 4 KiB, copied to LDS when the batch uses ONE table.
constexpr u32 SCORE_CELL_MASK = 0x8E1D8E1Du;
constexpr int SCORE_LUT_BYTES = 4096 + 65536;     // per points table: compact form, then wide form
__device__ __forceinline__ u32 goal_shift(u32 g) { return (g >> 4) & 0x00E000E0u; }
// Five goal words to a dword (round 6): a goal word carries six bits -- the colours of its two cells at bits 5-7 and
// 21-23 -- so word 5 q + j of a row sits at bits 3 j .. 3 j + 2 of either half of packed dword q.  The form the
// goal-word cache keeps in memory, and the one its kernels keep in registers (64-cell rows: 7 registers instead of 32).
__device__ __forceinline__ u32 goal_unpack(u32 p, int j) {      // goal_shift() of word j of the five in p
    return (j == 0 ? p << 5 : j == 1 ? p << 2 : p >> (3 * j - 5)) & 0x00E000E0u;
}
template <int H, int W>
__device__ __forceinline__ void goal_pack_row(const RowWords<H, W> &g, u32 *pk) {       // g: the goal cells of a row
    constexpr int WS = Geom<H, W>::WS;
#pragma unroll
    for (int q = 0; q < (WS + 4) / 5; ++q) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (5 * q + j < WS) v |= ((g[5 * q + j] >> 9) & 0x00070007u) << (3 * j);
        pk[q] = v;
    }
}

template <int H, int W, bool LDS_LUT, bool PACKED = false>       // PACKED: gsh_lane holds the row's words five to a dword
__device__ __forceinline__ int row_score(const RowWords<H, W> &n, const u32 *gsh_lane,
                                         const int8_t *__restrict__ lut, u32 lut_base, const int8_t *lds_lut,
                                         u32 cell_mask, u32 c100) {
    using Gm = Geom<H, W>;
    int s = 0;
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) {
        u32 idx = BO3_AND_OR(n[k], cell_mask, PACKED ? goal_unpack(gsh_lane[k / 5], k % 5) : gsh_lane[k]);
        if (LDS_LUT) {
            idx = BO3_AND_OR(n[k] >> 7, c100, idx);       // pullable -> bit 8 (cell_mask leaves bit 15 out in this form)
            s += lds_lut[idx & 0xFFFFu];
            if (!(Gm::ODD && k == Gm::WS - 1)) s += lds_lut[idx >> 16];
        } else {
            s += lut[lut_base + (idx & 0xFFFFu)];
            if (!(Gm::ODD && k == Gm::WS - 1)) s += lut[lut_base + (idx >> 16)];
        }
    }
    return s;
}

// ---- SimpleSideEffectPenalty: cells of the lane's row that differ from the baseline row ------------
// (env_wrappers.py:186-208).  b: the board row; base: the baseline row, player bits already cleared;
// gsh: the row's goal colours on bits 5-7 of each half.  Exit cells are NOT excluded here: they all
// carry the same paint, so the leader subtracts them (see the kernel).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int H, int W>
__device__ __forceinline__ int row_side_effect(const RowWords<H, W> &b, const u32 *base_col, const u32 *gsh,
                                               bool ignore_reward_cells, u32 not_player, u32 m1) {
    using Gm = Geom<H, W>;
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) {
        const u32 b0 = base_col[k * 64];
        u32 x = SL_BO3((TA ^ TB) & TC, b[k], b0, not_player);
        if (Gm::ODD && k == Gm::WS - 1) x &= 0x0000FFFFu;
        const u16x2 one = {1, 1};
        u16x2 nzv = __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), one);      // v_pk_min_u16: half != 0
        u32 nz = __builtin_bit_cast(u32, nzv);
        if (ignore_reward_cells) {
            const u32 bb = b[k];
            // everything at bit 0 of each half; red_life = ALIVE | COLOR_R
            const u32 start_red = b0 & (b0 >> 9), end_red = bb & (bb >> 9);             // (x & red_life) == red_life
            const u32 end_alive = bb & ~(bb >> 9);                                      // (b & red_life) == ALIVE
            const u32 g = gsh[k];
            const u32 blue_goal = (g >> 7) & ~(g >> 6) & ~(g >> 5);                     // goal colour == blue
            const u32 excuse = (start_red & ~end_red) | (blue_goal & end_alive);
            nz &= ~excuse & m1;
        }
        acc += nz;
    }
    return (int)((acc & 0xFFFFu) + (acc >> 16));
}

// ---- execute_actions for one agent ------------------------------------------------------------------
__device__ __forceinline__ int wrap1(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }

// The four cells an action can touch: the agent's (0), one ahead (1), two ahead (2), one behind (3).
// img: indices into the LDS image (Geom::cell), gix: row-major indices (global memory); (y1, x1): the cell ahead.
template <int H, int W>
__device__ __forceinline__ void act_cells(int ly, int lx, int action, int (&img)[4], int (&gix)[4], int &y1, int &x1) {
    using Gm = Geom<H, W>;
    const int dir = (action - 1) & 3;
    if (Gm::KM) {
        const int dy = (dir & 1) ? 0 : dir - 1, dx = (dir & 1) ? 2 - dir : 0;
        y1 = wrap1(ly + dy, H);
        x1 = wrap1(lx + dx, W);
        const int y2 = wrap1(ly + 2 * dy, H), x2 = wrap1(lx + 2 * dx, W), y3 = wrap1(ly - dy, H), x3 = wrap1(lx - dx, W);
        img[0] = Gm::cell(ly, lx);
        img[1] = Gm::cell(y1, x1);
        img[2] = Gm::cell(y2, x2);
        img[3] = Gm::cell(y3, x3);
        gix[0] = __mul24(ly, W) + lx;
        gix[1] = __mul24(y1, W) + x1;
        gix[2] = __mul24(y2, W) + x2;
        gix[3] = __mul24(y3, W) + x3;
    } else {
        // the move runs along ONE axis: wrap three positions on that axis and scale them by the axis' pitch in the
        // row-major image, instead of wrapping three (row, column) pairs
        const bool horiz = (dir & 1) != 0;
        const int pos = horiz ? lx : ly, n = horiz ? W : H;
        const int s = horiz ? 2 - dir : dir - 1;                     // +1 / -1
        const int p1 = wrap1(pos + s, n), p2 = wrap1(p1 + s, n), p3 = wrap1(pos - s, n);
        const int pitch = horiz ? 1 : W;
        const int base = horiz ? __mul24(ly, W) : lx;                // the cell index minus the moving coordinate's share
        img[0] = __mul24(ly, W) + lx;
        img[1] = base + __mul24(p1, pitch);
        img[2] = base + __mul24(p2, pitch);
        img[3] = base + __mul24(p3, pitch);
#pragma unroll
        for (int k = 0; k < 4; ++k) gix[k] = img[k];
        y1 = horiz ? ly : p1;
        x1 = horiz ? p1 : lx;
    }
}

// execute_actions for one agent on the four cells gathered up front (advance_board.c:217-300; valid for
// H, W >= 4 where the four cells are distinct).  Returns whether the cells are to be written back.
__device__ __forceinline__ bool act_rule(u32 (&c)[4], int action, int &ly, int &lx, int y1, int x1) {
    u32 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
    if (action == 0 || !(c0 & AGENT)) return false;
    const int dir = (action - 1) & 3;
    c0 = (c0 & ~ORIENT_MASK) | ((u32)dir << ORIENT_SHIFT);
    const bool can_push = (~c0 & c1 & PUSHABLE) != 0;
    if (action >= 5) {
        if (c1 == 0) {
            c1 = ALIVE | DESTRUCTIBLE | (c0 & COLORS);
        } else if (c1 & DESTRUCTIBLE) {
            c1 = (c1 & AGENT) ? ((c1 ^ (AGENT | DESTRUCTIBLE)) | FROZEN) : 0u;
        } else if (can_push) {
            if (c2 == 0) {
                c2 = c1;
                c1 = 0;
            } else if (c2 & EXIT) {
                c1 = 0;
            }
        }
    } else {
        bool step_into = false, leave_only = false;
        if (can_push) {
            if (c2 == 0) {
                c2 = c1;
                step_into = true;
            } else if (c2 & EXIT) {
                step_into = true;
            }
        } else if (c1 == 0) {
            step_into = true;
        } else if ((c0 & c1 & EXIT) && !(c1 & AGENT)) {
            leave_only = true;
        }
        if (step_into || leave_only) {
            if (step_into) c1 = c0;
            ly = y1;
            lx = x1;
            if (~c0 & c3 & PULLABLE) {
                c0 = c3;
                c3 = 0;
            } else {
                c0 = 0;
            }
        }
    }
    c[0] = c0;
    c[1] = c1;
    c[2] = c2;
    c[3] = c3;
    return true;
}

template <int H, int W>
__device__ __forceinline__ void act_gather(u16 *board, int &ly, int &lx, int action) {
    int img[4], gix[4], y1, x1;
    act_cells<H, W>(ly, lx, action, img, gix, y1, x1);
    u32 c[4] = {board[img[0]], board[img[1]], board[img[2]], board[img[3]]};
    if (!act_rule(c, action, ly, lx, y1, x1)) return;
    board[img[0]] = (u16)c[0];
    board[img[1]] = (u16)c[1];
    board[img[2]] = (u16)c[2];
    board[img[3]] = (u16)c[3];
}

// update_exit_colors for the board of a leader lane, on the flat LDS image.  agent_cell (optional): receives the
// cell at the agent's location as it stands afterwards (an agent that has left through an exit stands ON an exit
// cell, which the paint then covers), so that the caller need not read it back.
template <int H, int W>
__device__ __forceinline__ bool recolor_exits_lds(u16 *board, int ly, int lx, const int32_t *exits,
                                                  int exit0, int E, int score, int initial, int required,
                                                  int exit_points, int *n_exits = nullptr, u32 *agent_cell = nullptr) {
    using Gm = Geom<H, W>;
    bool any_can = false;
    int at = -1;
    u32 mine = 0;
    if (ly >= 0) {
        at = Gm::cell(ly, lx);
        const u32 c = board[at];
        int earned = score - initial + (has_exited(c) ? exit_points : 0);
        if (earned < 0) earned = 0;
        const bool can = (c & AGENT) && earned >= required;
        mine = (c & ~EXIT) | (can ? EXIT : 0u);
        board[at] = (u16)mine;
        any_can = can;
    }
    const u16 paint = (u16)(FROZEN | EXIT | (any_can ? COLOR_R : 0u));
    int nx = 0;
    if (exit0 >= 0) {                             // slot 0 was prefetched; the usual level has one exit
        const int i = Gm::flat(exit0);
        board[i] = paint;
        if (i == at) mine = paint;
        nx = 1;
    }
    for (int k = 1; k < E; ++k) {
        const int ex = exits[k];
        if (ex >= 0) {
            const int i = Gm::flat(ex);
            board[i] = paint;
            if (i == at) mine = paint;
            ++nx;
        }
    }
    if (n_exits) *n_exits = nx;
    if (agent_cell) *agent_cell = mine;
    return any_can;
}

// ---- workgroup span <-> HBM -----------------------------------------------------------------------

typedef __attribute__((address_space(1))) const void *glds_src_t;
typedef __attribute__((address_space(3))) void *glds_dst_t;

// `bytes` of global memory -> LDS with the asynchronous global_load_lds DMA (16 bytes per lane,
// 1 KiB per wave instruction, no VGPR staging, no ds_write).  Chunk c of 1 KiB is moved by wave
// c % WAVES.  Completion: the vmcnt(0) the compiler places in front of the next __syncthreads().
// Perm: LDS slot s (16-byte chunk) receives global chunk Perm::swz_chunk(s) -- the board-image swizzle of Geom.
struct NoPerm {
    static __device__ __forceinline__ int swz_chunk(int s) { return s; }
};

// NW: number of waves that share the move (wave = 0..NW-1 among them).
template <int MAX_BYTES, bool SWZ = false, int NW = WAVES, class Perm = NoPerm>
__device__ __forceinline__ void dma_to_lds(const unsigned char *__restrict__ src, unsigned char *dst, int bytes,
                                           int lane, int wave) {
    static_assert(!SWZ || !std::is_same<Perm, NoPerm>::value, "a swizzled move names its geometry");
    const int nv = bytes >> 4;
    constexpr int NCH = (MAX_BYTES + 1023) / 1024;
#pragma unroll
    for (int j = 0; j < (NCH + NW - 1) / NW; ++j) {
        const int c = wave + NW * j;
        const int s = c * 64 + lane;
        if (s < nv)
            __builtin_amdgcn_global_load_lds((glds_src_t)(src + (SWZ ? Perm::swz_chunk(s) : s) * 16),
                                             (glds_dst_t)(dst + c * 1024), 16, 0, SL_LOAD_AUX);
    }
}

template <int H, int W, int NW = WAVES>
__device__ __forceinline__ void load_span(const u16 *__restrict__ src, unsigned char *region, int nbb, int lane, int wave) {
    using Gm = Geom<H, W>;
    const int bytes = nbb * Gm::HW * 2;
    dma_to_lds<Gm::SPAN, Gm::KM != 0, NW, Gm>((const unsigned char *)src, region + Gm::PAD, bytes, lane, wave);
    const int nv = bytes >> 4, rem = (bytes & 15) >> 1;          // leftover cells: tail workgroup only
    if (wave == 0 && lane < rem) ((u16 *)(region + Gm::PAD))[nv * 8 + lane] = src[nv * 8 + lane];
}

__device__ __forceinline__ void store16(u32x4 *p, u32x4 v) {
#if SL_STORE == 1
    __builtin_nontemporal_store(v, p);
#elif SL_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif SL_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

template <int H, int W>
__device__ __forceinline__ void store_span(u16 *__restrict__ dst, const unsigned char *region, int nbb, int tid) {
    using Gm = Geom<H, W>;
    const int bytes = nbb * Gm::HW * 2;
    const int nv = bytes >> 4;
    u32x4 *d = (u32x4 *)dst;
    const u32x4 *s = (const u32x4 *)(region + Gm::PAD);
    constexpr int NVI = (Gm::SPAN / 16 + 64 * WAVES - 1) / (64 * WAVES);
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
        const int slot = tid + 64 * WAVES * i;
        if (slot < nv) store16(d + Gm::swz_chunk(slot), s[slot]);
    }
    const int rem = (bytes & 15) >> 1;
    if (tid < rem) dst[nv * 8 + tid] = ((const u16 *)s)[nv * 8 + tid];
}

// The same for the fused step's boards, skipping the 16-byte chunks that lie entirely in boards whose mailbox says
// "nothing changed" (the few cells the leaders touch in such a board they store themselves).
template <int H, int W, class Box>
__device__ __forceinline__ void store_span_dirty(u16 *__restrict__ dst, const unsigned char *region, int nbb, int tid,
                                                 const Box *box) {
    using Gm = Geom<H, W>;
    static_assert(Gm::KM == 0, "row-major images only");
    const int bytes = nbb * Gm::HW * 2;
    const int nv = bytes >> 4;
    u32x4 *d = (u32x4 *)dst;
    const u32x4 *s = (const u32x4 *)(region + Gm::PAD);
    constexpr int NVI = (Gm::SPAN / 16 + 64 * WAVES - 1) / (64 * WAVES);
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
        const int slot = tid + 64 * WAVES * i;
        if (slot < nv) {
            const int q0 = (slot * 8) / Gm::HW, q1 = min((slot * 8 + 7) / Gm::HW, nbb - 1);
            if (box[q0].dirty | box[q1].dirty) store16(d + slot, s[slot]);
        }
    }
    const int rem = (bytes & 15) >> 1;
    if (tid < rem) dst[nv * 8 + tid] = ((const u16 *)s)[nv * 8 + tid];
}

// ---- advance_board --------------------------------------------------------------------------------

template <int H, int W>
__global__ __launch_bounds__(64 * WAVES, (Geom<H, W>::WAVES_PER_SIMD)) void k_advance_rowlane(const u16 *__restrict__ in, u16 *__restrict__ out,
                                                                int B, const float *__restrict__ spawn_prob,
                                                                int n_steps, const int32_t *__restrict__ n_each,
                                                                const int32_t *__restrict__ n_valid,
                                                                sl_pcg64 *rng, const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // n_valid (device): only the first *n_valid boards exist -- the side-effect pass sizes its launches by the
    // queue's capacity and lets the device decide how many entries there are
    if (n_valid) B = min(B, *n_valid);
    const int e0b = blockIdx.x * Gm::NB;
    if (e0b >= B) return;
    const int nbb = min(Gm::NB, B - e0b);
    const LaneMap<H, W> lm(lane);
    const int g = lm.g, r = lm.r, up = lm.up, dn = lm.dn;
    const int gb = wave * Gm::G + g;
    const bool rowl = lane < Gm::NL && gb < nbb;       // holds a row (its own or a halo copy)
    const bool live = rowl && lm.real;
    const unsigned e = e0b + (rowl ? gb : 0);
    unsigned char *board = smem + Gm::OFF_BOARD;
    u64 *rng_lds = (u64 *)(smem + Gm::OFF_RNG) + 4 * Gm::G * wave;

    load_span<H, W>(in + (size_t)e0b * Gm::HW, board, nbb, lane, wave);
    if (lane < 4 * Gm::G && wave * Gm::G + (lane >> 2) < nbb)
        rng_lds[lane] = ((const u64 *)(rng + e0b + wave * Gm::G))[lane];
    const double p = live ? (double)spawn_prob[e] : 0.0;
    // one step count per board (advance_board_each): the wave runs as long as its busiest board, lanes of
    // boards that are through keep their rows
    const int my_n = n_each ? (rowl ? max(0, n_each[e]) : 0) : n_steps;
    int wave_n = my_n;
    if (n_each) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wave_n = max(wave_n, __shfl_xor(wave_n, off));
    }
    __syncthreads();
    const Consts cst = make_consts();
    const pl::PConsts pcst = pl::make_pconsts();
    RowWords<H, W> b;
#pragma unroll
    for (int k = 0; k < Gm::WS; ++k) b[k] = 0;
    if (rowl) read_row<H, W>(board, gb, r, b);
    for (int s = 0; s < wave_n; ++s) {
        const bool going = s < my_n;
        const bool changed = ca_step<H, W, true, false>(b, live && going, going, up, dn, cst, pcst, rng_lds, live ? g : 0, p, jump);
        if (Gm::VERT == V_SHIFT && s + 1 < wave_n && changed) {      // refresh the halo copies through LDS
            if (live) write_row<H, W>(board, gb, r, b);
            wave_sync();
            if (rowl && !lm.real) read_row<H, W>(board, gb, r, b);
            wave_sync();
        }
    }
    if (live) write_row<H, W>(board, gb, r, b);
    __syncthreads();
    store_span<H, W>(out + (size_t)e0b * Gm::HW, board, nbb, tid);
    if (lane < 4 * Gm::G && wave * Gm::G + (lane >> 2) < nbb)
        ((u64 *)(rng + e0b + wave * Gm::G))[lane] = rng_lds[lane];
}

// Small batches (fewer workgroups of the kernel above than the chip has CUs: config C2's 1024 boards are 128): one
// wavefront per workgroup, the row straight from and to global memory -- four times the workgroups, no LDS image, no
// workgroup barrier -- and, for the shapes with a bit-plane form, the rows stay planes for all n steps.
template <int H, int W>
__global__ __launch_bounds__(64) void k_advance_small(const u16 *__restrict__ in, u16 *__restrict__ out, int B,
                                                      const float *__restrict__ spawn_prob, int n_steps,
                                                      const int32_t *__restrict__ n_each,
                                                      const int32_t *__restrict__ n_valid, sl_pcg64 *rng,
                                                      const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    constexpr int WS = Gm::WS;
    __shared__ u64 rng_lds[4 * Gm::G];
    const int lane = threadIdx.x;
    if (n_valid) B = min(B, *n_valid);
    const int e0b = blockIdx.x * Gm::G;
    if (e0b >= B) return;
    const int nbb = min(Gm::G, B - e0b);
    const LaneMap<H, W> lm(lane);
    const int g = lm.g, r = lm.r;
    const bool rowl = lane < Gm::NL && g < nbb;
    const bool live = rowl && lm.real;
    const unsigned e = e0b + (rowl ? g : 0);
    RowWords<H, W> b;
    {
        const u16 *row = in + ((size_t)e * H + r) * W;
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            const u32 lo = rowl ? row[k] : 0u;
            const u32 hi = (rowl && !(Gm::ODD && k == WS - 1)) ? row[k + WS] : 0u;
            b[k] = lo | (hi << 16);
        }
    }
    if (lane < 4 * nbb) rng_lds[lane] = ((const u64 *)(rng + e0b))[lane];
    const double p = live ? (double)spawn_prob[e] : 0.0;
    const int my_n = n_each ? (rowl ? max(0, n_each[e]) : 0) : n_steps;
    int wave_n = my_n;
    if (n_each) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wave_n = max(wave_n, __shfl_xor(wave_n, off));
    }
    const int partner = !rowl || lm.real ? lane : (r == 0 ? lane - H : lane + H);
    const Consts cst = make_consts();
    const pl::PConsts pcst = pl::make_pconsts();
    wave_sync();
    constexpr bool PLANES = use_planes_multi<H, W>();
    if constexpr (PLANES) {
        constexpr int NW = pl::PG<W>::NW;
        if (wave_n > 1) {               // many steps: one transposition at either end
            pl::PState<NW> st;
            const pl::VCtx<Gm::VERT> vctx = {lm.up, lm.dn, 4 * partner};
            pl::planes_load<W>(b, pcst, st);
            for (int s = 0; s < wave_n; ++s) {
                const u32 realm = live && s < my_n ? vreg(pl::PG<W>::REAL) : 0u;
                pl::planes_step<W, Gm::VERT, true>(st, vctx, realm, [&](const pl::Pl<NW> &elig) {
                    return resolve_draws_planes<H, W, NW>(elig, rng_lds, rowl ? g : 0, p, jump);
                });
            }
            pl::planes_store<W>(b, pcst, st);
        } else if (wave_n == 1) {
            ca_step<H, W, true, false>(b, live && my_n > 0, my_n > 0, lm.up, lm.dn, cst, pcst, rng_lds, rowl ? g : 0, p, jump);
        }
    } else {
        for (int s = 0; s < wave_n; ++s) {
            const bool going = s < my_n;
            const bool changed = ca_step<H, W, true, false>(b, live && going, going, lm.up, lm.dn, cst, pcst, rng_lds,
                                                            rowl ? g : 0, p, jump);
            if (Gm::VERT == V_SHIFT && changed && s + 1 < wave_n) {
#pragma unroll
                for (int k = 0; k < WS; ++k) b[k] = bperm(4 * partner, b[k]);
            }
        }
    }
    if (live) {
        u16 *row = out + ((size_t)e * H + r) * W;
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            row[k] = (u16)b[k];
            if (!(Gm::ODD && k == WS - 1)) row[k + WS] = (u16)(b[k] >> 16);
        }
    }
    wave_sync();
    if (lane < 4 * nbb) ((u64 *)(rng + e0b))[lane] = rng_lds[lane];
}

// ---- life_occupancy ---------------------------------------------------------------------------------
// (advance_board.c:153-189)  One wavefront per workgroup, G boards per wavefront, rows in registers for
// all n steps.  After every step each lane bumps, for each of its cells that is ALIVE and not
// AGENT|EXIT|FROZEN, a small counter (cell, colour) in LDS with a fire-and-forget ds_add_u32.  A lane's
// counters are one contiguous run with an odd dword pitch between lanes, so "same cell, every row"
// spreads over the banks.  CB = bits per counter:
//   16  two colours per dword; n_steps <= 65535 keeps a half from carrying into its neighbour; the
//       counters become the int32 [H,W,8] output at the end (boards up to 32 cells wide);
//   8   four colours per dword, drained into the zeroed output every 255 steps: half the LDS, twice the
//       wavefronts per CU -- what 64-wide boards need (a 16-bit set is 64 KiB per board).
// SLOTS = counters per cell.  A new cell's colour is made of colour bits of its live neighbours and of
// spawners (advance_board.c:12-32), so a board whose live and spawner cells use only two of the three colour
// bits can ever show four colours: SLOTS = 4 (colour -> slot by compressing the two bits) halves the LDS per
// wavefront, which is what bounds the wavefronts per CU of this kernel (64x64: 33 KiB -> 16.6 KiB, 1 -> 2 per
// SIMD; 25x25: 25.8 -> 13 KiB).  Boards that use all three bits take the SLOTS = 8 instantiation; both are
// launched over the whole batch and every wavefront decides from its boards which of the two it belongs to.
template <int H, int W, int SLOTS>
struct OccGeom {
    using Gm = Geom<H, W>;
    static constexpr int CB = W > 32 ? 8 : 16;
    static constexpr int PER_DWORD = 32 / CB;                           // counters per dword
    static constexpr int CELL_DWORDS = SLOTS / PER_DWORD;
    static_assert(CELL_DWORDS >= 1, "at least one dword of counters per cell");
    static constexpr int PITCH = W * CELL_DWORDS + 1;                   // dwords per lane
    // Round 5, measured and left off (SL_OCC_DIRECT): 64-cell rows WITHOUT counters in LDS -- the register counters
    // (4-bit, bit-sliced) flushed every 15 counted steps straight into the zeroed int32 output as fire-and-forget global
    // atomics, one per live cell, so that a wavefront needs 32 bytes of LDS instead of 16.6 KB and the kernel's 127
    // registers, not its LDS, bound the wavefronts per SIMD (2 -> 4).  Bit exact, and twice as slow: the episode-end pass of
    // C5 (2137 episodes) 23.3-23.9 ms against 12.5-12.7 -- 64 lanes x 64 addresses per atomic instruction are 64
    // transactions at the L2, and the pass issues ~1e8 of them (profiles/round5_d_se_pass_variants.txt).
#ifndef SL_OCC_DIRECT
#define SL_OCC_DIRECT 0
#endif
    static constexpr bool DIRECT = SL_OCC_DIRECT && CB == 8 && use_planes_multi<H, W>();
    // 64-cell rows (two plane words): the register counters are transposed into byte lanes and added four cells at a
    // time, so a lane's counters are laid out [slot][plane word i][q] with byte j of that dword = cell 32 i + q + 8 j
#ifndef SL_OCC_NIBBLE
#define SL_OCC_NIBBLE 1         /* A/B knob: 0 = 64-cell rows walk their non-zero cells like the other shapes */
#endif
    static constexpr bool NIBBLE = SL_OCC_NIBBLE && CB == 8 && W == 64 && !DIRECT;
    static constexpr int OFF_CNT = 0;
    static constexpr int OFF_RNG = DIRECT ? 0 : 64 * PITCH * 4;         // G x 4 u64
    static constexpr int LDS_BYTES = OFF_RNG + Gm::G * 32;
    static constexpr int FLUSH_EVERY = (CB == 8 && !DIRECT) ? 255 : 0x7FFFFFFF;
};

// counters of one lane's row -> its slice of the output (add: the output was zeroed), counters cleared.
// inv: colour of slot s at bits [3s, 3s+3).
template <int H, int W, int SLOTS>
__device__ __forceinline__ void occ_drain(u32 *cnt, int32_t *dst, u32 inv) {
    using Oc = OccGeom<H, W, SLOTS>;
    if constexpr (Oc::NIBBLE) {
        for (int d = 0; d < SLOTS * 16; ++d) {
            const u32 v = cnt[d];
            if (v) {
                const int slot = d >> 4, x0 = 32 * ((d >> 3) & 1) + (d & 7);
                int32_t *cell = dst + ((inv >> (3 * slot)) & 7u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (v >> (8 * j)) & 0xFFu;
                    if (c) cell[(x0 + 8 * j) * 8] += c;
                }
                cnt[d] = 0;
            }
        }
        return;
    }
    for (int x = 0; x < W; ++x) {
#pragma unroll
        for (int q = 0; q < Oc::CELL_DWORDS; ++q) {
            const u32 v = cnt[x * Oc::CELL_DWORDS + q];
            if (v) {
#pragma unroll
                for (int j = 0; j < Oc::PER_DWORD; ++j) {
                    const int c = (v >> (Oc::CB * j)) & ((1u << Oc::CB) - 1u);
                    const int slot = q * Oc::PER_DWORD + j;
                    if (c) dst[x * 8 + ((inv >> (3 * slot)) & 7u)] += c;
                }
                cnt[x * Oc::CELL_DWORDS + q] = 0;
            }
        }
    }
}

template <int H, int W, int SLOTS>
__global__ __launch_bounds__(64) void k_occupancy_rowlane(const u16 *__restrict__ in, int32_t *__restrict__ counts,
                                                          size_t counts_stride, int B, const int32_t *__restrict__ n_valid,
                                                          int valid_period, const int32_t *__restrict__ pre_steps,
                                                          const float *__restrict__ spawn_prob, int n_steps,
                                                          sl_pcg64 *rng, const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    using Oc = OccGeom<H, W, SLOTS>;
    constexpr int WS = Gm::WS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    // Boards come in runs of `valid_period` (0: one run of B): the side-effect pass lays out two runs, and the
    // number of entries that exist in each, *n_valid, is only known on the device.  Workgroups are numbered run
    // by run, so none straddles two runs.
    const int period = valid_period > 0 ? valid_period : B;
    const int blocks_per_run = (period + Gm::G - 1) / Gm::G;
    const int run = blockIdx.x / blocks_per_run, in_run = (blockIdx.x - run * blocks_per_run) * Gm::G;
    int run_len = min(period, B - run * period);
    if (n_valid) run_len = min(run_len, *n_valid);
    if (in_run >= run_len) return;
    const int e0b = run * period + in_run;
    const int nbb = min(Gm::G, run_len - in_run);
    const LaneMap<H, W> lm(lane);
    const int g = lm.g, r = lm.r;
    const bool rowl = lane < Gm::NL && g < nbb;
    const bool live = rowl && lm.real;
    const unsigned e = e0b + (rowl ? g : 0);
    RowWords<H, W> b;
    {   // the row straight from HBM (once per launch)
        const u16 *row = in + ((size_t)e * H + r) * W;
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            const u32 lo = rowl ? row[k] : 0u;
            const u32 hi = (rowl && !(Gm::ODD && k == WS - 1)) ? row[k + WS] : 0u;
            b[k] = lo | (hi << 16);
        }
    }
    // colour bits in use on each board: those of its live cells and of its spawners
    u32 used = 0, lacks = 0;            // (lacks: colour bits some source of the row does NOT have)
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        const u32 c = b[k];
        const u32 src = (c | (c >> 7)) & 0x00010001u;                   // ALIVE or SPAWNING, per half
        used |= c & (src * 0x0E00u);
        lacks |= ~c & (src * 0x0E00u);
    }
    used = (used | (used >> 16)) & 0x0E00u;
    lacks = (lacks | (lacks >> 16)) & 0x0E00u;
    // A colour bit that is set in EVERY source is set in every cell they ever produce as well (a birth takes the bits
    // seen in two of its three live neighbours, a spawned cell those plus its spawners' -- all of them sources), just
    // as a bit no source has stays clear: only the bits that VARY between a board's sources tell its colours apart.
    u32 bits = 0, fixed = 0;            // of the lane's own board: colour bits that vary / that every source has
    int wave_vary = Oc::CB == 8 ? 0 : 3;        // (only 64-wide boards are LDS-bound: see launch_occupancy_t)
#pragma unroll
    for (int q = 0; q < Gm::G; ++q) {
        const unsigned long long in_board = ((Gm::GL == 64 ? ~0ull : ((1ull << Gm::GL) - 1ull)) << (q * Gm::GL % 64));
        u32 bq = 0, aq = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (__ballot(rowl && (used & (0x0200u << i))) & in_board) bq |= 1u << i;
            if (!(__ballot(rowl && (lacks & (0x0200u << i))) & in_board)) aq |= 1u << i;
        }
        aq &= bq;
        if (q < nbb) wave_vary = max(wave_vary, __popc(bq & ~aq));
        if (g == q) bits = bq & ~aq, fixed = aq;
    }
    if ((wave_vary <= 2) != (SLOTS == 4)) return;   // the other instantiation takes this wavefront's boards
    // (wave-uniform) the slots the wave's boards can tick at all: the rest are skipped, counting and flushing
    const int n_active = SLOTS == 4 ? 1 << wave_vary : SLOTS;
    // colour -> slot (2 bits each) and slot -> colour (3 bits each) for the board's varying colour bits
    u32 lut = 0, inv = 0;
    if (SLOTS == 4) {
        int next = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if ((c & ~(bits | fixed)) == 0 && (c & fixed) == fixed) {       // a colour the board can show
                lut |= (u32)next << (2 * c);
                inv |= (u32)c << (3 * next);
                ++next;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) inv |= (u32)c << (3 * c);
    }
    u32 *cnt = (u32 *)(smem + Oc::OFF_CNT) + lane * Oc::PITCH;
    u64 *rng_lds = (u64 *)(smem + Oc::OFF_RNG);
    int32_t *dst = counts + (size_t)e * counts_stride + ((size_t)r * W) * 8;    // this lane's row of the output

    if (!Oc::DIRECT)
        for (int i = 0; i < Oc::PITCH; ++i) cnt[i] = 0;
    if (lane < 4 * nbb) rng_lds[lane] = ((const u64 *)(rng + e0b))[lane];
    const double p = rowl ? (double)spawn_prob[e] : 0.0;
    // V_SHIFT: after a step the halo lanes take the new first / last row from the lanes that own them
    const int partner = !rowl || lm.real ? lane : (r == 0 ? lane - H : lane + H);
    const Consts cst = make_consts();
    const pl::PConsts pcst = pl::make_pconsts();
    // pre_steps (optional): the board is first rolled forward that many steps without counting -- the side-effect
    // pass's advance_board(b0, p, num_steps) (side_effects.py:108) fused in front of its life_occupancy
    const int my_pre = pre_steps ? (rowl ? max(0, pre_steps[e]) : 0) : 0;
    const int my_end = rowl ? my_pre + n_steps : 0;
    int wave_end = my_end;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wave_end = max(wave_end, __shfl_xor(wave_end, off));
    wave_sync();
    int since_drain = 0;
    JumpCache jcache;
    // rows of up to 28 cells, or 64: the row stays in bit-plane form for all the steps (sl_planes.h) -- one
    // transposition at the start, the CA on whole rows, and the counting visits only the cells that ARE alive
    // (a handful per row) instead of every cell position
    constexpr bool PLANES = use_planes_multi<H, W>();
    constexpr int NW = pl::PG<PLANES ? W : 8>::NW;
    pl::PState<NW> st;
    const pl::VCtx<Gm::VERT> vctx = {lm.up, lm.dn, 4 * partner};
    if constexpr (PLANES) pl::planes_load<W>(b, pcst, st);
    constexpr int MINI_BITS = 4;
    u32 mini[PLANES ? SLOTS : 1][MINI_BITS][NW];
#pragma unroll
    for (int a = 0; a < (PLANES ? SLOTS : 1); ++a)
#pragma unroll
        for (int k = 0; k < MINI_BITS; ++k)
#pragma unroll
            for (int i = 0; i < NW; ++i) mini[a][k][i] = 0;
    int mini_steps = 0;
    // SLOTS == 4: which colour planes are the board's two varying colour bits (lowest first; 3 = none)
    const int sel_lo = (bits & 1u) ? 0 : ((bits & 2u) ? 1 : ((bits & 4u) ? 2 : 3));
    const int sel_hi = (bits & 1u) ? ((bits & 2u) ? 1 : ((bits & 4u) ? 2 : 3)) : (((bits & 2u) && (bits & 4u)) ? 2 : 3);
    auto flush_mini = [&]() {           // small counters -> LDS counters
        if constexpr (Oc::NIBBLE) {
            // the four bit planes of a word's 32 counters -> eight dwords of four byte lanes each (cells q + 8 j of the
            // word in dword q), added to the LDS counters with eight fire-and-forget adds: ~36 instructions per slot
            // and word whatever the number of live cells, no loop, no divergence (the walk below ran as often as the
            // wave's busiest row had live cells, ~25 instructions each time)
            const u32 M5 = 0x55555555u, M3 = 0x33333333u, MF = 0x0F0F0F0Fu;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                if (sl >= n_active) break;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const u32 p0 = mini[sl][0][i], p1 = mini[sl][1][i], p2 = mini[sl][2][i], p3 = mini[sl][3][i];
                    if (!__ballot((p0 | p1 | p2 | p3) != 0)) continue;
                    const u32 a = (p0 & M5) | ((p1 << 1) & ~M5), bb = ((p0 >> 1) & M5) | (p1 & ~M5);     // 2-bit fields:
                    const u32 c = (p2 & M5) | ((p3 << 1) & ~M5), d = ((p2 >> 1) & M5) | (p3 & ~M5);      // even / odd cells
                    const u32 e[4] = {(a & M3) | ((c << 2) & ~M3), (bb & M3) | ((d << 2) & ~M3),          // nibble n of e[r]:
                                      ((a >> 2) & M3) | (c & ~M3), ((bb >> 2) & M3) | (d & ~M3)};         // cell 4 n + r
                    u32 *base = cnt + (sl * NW + i) * 8;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const u32 lo = e[r] & MF, hi = (e[r] >> 4) & MF;        // cells 8 j + r, cells 8 j + 4 + r
                        if (lo) __hip_atomic_fetch_add(base + r, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (hi) __hip_atomic_fetch_add(base + 4 + r, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
#pragma unroll
                    for (int k = 0; k < MINI_BITS; ++k) mini[sl][k][i] = 0;
                }
            }
            return;
        }
#pragma unroll
        for (int sl = 0; sl < (PLANES ? SLOTS : 1); ++sl)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                u32 nz = 0;
#pragma unroll
                for (int k = 0; k < MINI_BITS; ++k) nz |= mini[sl][k][i];
                while (nz) {
                    const int pos = __ffs((int)nz) - 1;
                    nz &= nz - 1;
                    u32 val = 0;
#pragma unroll
                    for (int k = 0; k < MINI_BITS; ++k) val |= ((mini[sl][k][i] >> pos) & 1u) << k;
                    // the cell's column (split two-word planes: word i holds cells i WS .. from bit 1 on)
                    const int x = pl::PG<PLANES ? W : 8>::SPLIT ? i * WS + pos - 1
                                                                  : NW == 2 ? 32 * i + pos : (pos < 16 ? pos - 1 : pos - 17 + WS);
                    if (Oc::DIRECT)     // straight into the (zeroed) output: counts[y, x, colour of the slot]
                        __hip_atomic_fetch_add(dst + x * 8 + ((inv >> (3 * sl)) & 7u), (int32_t)val, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    else
                        __hip_atomic_fetch_add(cnt + x * Oc::CELL_DWORDS + sl / Oc::PER_DWORD,
                                               val << (Oc::CB * (sl % Oc::PER_DWORD)), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
#pragma unroll
                for (int k = 0; k < MINI_BITS; ++k) mini[sl][k][i] = 0;
            }
    };
    for (int s = 0; s < wave_end; ++s) {
        const bool going = s < my_end;
        if constexpr (PLANES) {
            const u32 realm = live && going ? vreg(pl::PG<W>::REAL) : 0u;
            pl::planes_step<W, Gm::VERT, true>(st, vctx, realm, [&](const pl::Pl<NW> &elig) {
#ifdef SL_OCC_NODRAW
                return elig;
#else
                return resolve_draws_planes<H, W, NW, true>(elig, rng_lds, rowl ? g : 0, p, jump, &jcache);
#endif
            });
        } else {
            const bool changed = ca_step<H, W, true, false>(b, live && going, going, lm.up, lm.dn, cst, pcst, rng_lds, rowl ? g : 0, p, jump);
            if (Gm::VERT == V_SHIFT && changed) {
#pragma unroll
                for (int k = 0; k < WS; ++k) b[k] = bperm(4 * partner, b[k]);
            }
        }
        if (!__ballot(going && s >= my_pre)) continue;          // nobody counts yet
#ifndef SL_OCC_NOCOUNT
        if constexpr (PLANES) {
            // Counting.  Every step: the cells that count -- alive, not agent / frozen / exit (advance_board.c:176-181)
            // -- bump small bit-sliced counters that live in registers, one 4-bit counter per cell and colour slot
            // (whole-row logic: a ripple of four half-adders per slot).  Every 15 counted steps the lanes walk the
            // cells whose small counter is not zero and add it to the LDS counters: the walk visits live cells only
            // (a handful per row), and it runs once per 15 steps instead of every step.
            const bool counting = live && going && s >= my_pre;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const u32 tick = counting ? (st.A.w[i] & ~st.G.w[i] & ~st.Z.w[i] & ~st.X.w[i] & pl::PG<W>::REAL) : 0u;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    if (sl >= n_active) break;          // (wave-uniform) a slot none of the wave's boards can tick
                    u32 carry;
                    if (SLOTS == 4) {       // slot = the colour's two varying bits, compressed
                        const u32 s0 = sel_lo == 0 ? st.C[0].w[i] : (sel_lo == 1 ? st.C[1].w[i] : (sel_lo == 2 ? st.C[2].w[i] : 0u));
                        const u32 s1 = sel_hi == 1 ? st.C[1].w[i] : (sel_hi == 2 ? st.C[2].w[i] : 0u);
                        carry = tick & ((sl & 1) ? s0 : ~s0) & ((sl & 2) ? s1 : ~s1);
                    } else {
                        carry = tick & ((sl & 1) ? st.C[0].w[i] : ~st.C[0].w[i]) & ((sl & 2) ? st.C[1].w[i] : ~st.C[1].w[i]) &
                                ((sl & 4) ? st.C[2].w[i] : ~st.C[2].w[i]);
                    }
#pragma unroll
                    for (int k = 0; k < MINI_BITS; ++k) {
                        const u32 t = mini[sl][k][i] & carry;
                        mini[sl][k][i] ^= carry;
                        carry = t;
                    }
                }
            }
            if (++mini_steps == (1 << MINI_BITS) - 1) {
                mini_steps = 0;
                flush_mini();
            }
        } else if (live && going && s >= my_pre) {
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                const u32 c = b[k];
                const u32 tick = c & ~(c >> 1) & ~(c >> 4) & ~(c >> 8) & 0x00010001u;    // alive, not agent/frozen/exit
                u32 lo_slot = (c >> 9) & 7u, hi_slot = (c >> 25) & 7u;
                if (SLOTS == 4) {
                    lo_slot = (lut >> (2 * lo_slot)) & 3u;
                    hi_slot = (lut >> (2 * hi_slot)) & 3u;
                }
                __hip_atomic_fetch_add(cnt + k * Oc::CELL_DWORDS + lo_slot / Oc::PER_DWORD,
                                       (tick & 1u) << (Oc::CB * (lo_slot % Oc::PER_DWORD)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (!(Gm::ODD && k == WS - 1))
                    __hip_atomic_fetch_add(cnt + (k + WS) * Oc::CELL_DWORDS + hi_slot / Oc::PER_DWORD,
                                           (tick >> 16) << (Oc::CB * (hi_slot % Oc::PER_DWORD)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
#endif
        if (++since_drain == Oc::FLUSH_EVERY) {         // (wave-uniform) 8-bit counters are about to wrap
            since_drain = 0;
            wave_sync();
            if (live) occ_drain<H, W, SLOTS>(cnt, dst, inv);
            wave_sync();
        }
    }
    if constexpr (PLANES) flush_mini();
    wave_sync();
    if (live && !Oc::DIRECT) {
        if (Oc::CB == 8) {
            occ_drain<H, W, SLOTS>(cnt, dst, inv);
        } else {            // nothing was drained on the way: plain stores, the output need not be zeroed
            for (int x = 0; x < W; ++x) {
                int32_t cell[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < Oc::CELL_DWORDS; ++q) {
                    const u32 v = cnt[x * Oc::CELL_DWORDS + q];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {               // (static register indices: no scratch array)
                        if (((inv >> (3 * (2 * q))) & 7u) == (u32)c && 2 * q < SLOTS) cell[c] += (int32_t)(v & 0xFFFFu);
                        if (((inv >> (3 * (2 * q + 1))) & 7u) == (u32)c && 2 * q + 1 < SLOTS) cell[c] += (int32_t)(v >> 16);
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) dst[x * 8 + c] = cell[c];
            }
        }
    }
    if (lane < 4 * nbb) ((u64 *)(rng + e0b))[lane] = rng_lds[lane];
}

// ---- observation ------------------------------------------------------------------------------------
constexpr int OBS_MAX_EXITS = 8;                       // exit slots handled by the fast obs path
constexpr int OBS_PAR_INTS = 2 + 2 * OBS_MAX_EXITS;

typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef u32 u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef u32 u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ int div_small(int i, int n, float inv_n) {     // i / n for 0 <= i < 2^20
    int q = (int)(((float)i + 0.5f) * inv_n);
    if (q * n > i) --q;
    if ((q + 1) * n <= i) ++q;
    return q;
}

// Position of a cell of the workgroup's cell stream (c = board_in_block * view_cells + view_cell),
// advanced cell by cell without divisions.
struct ObsCursor {
    int bq, v, vy, vx;
    __device__ __forceinline__ void init(int c, int nv, int vw, float inv_nv, float inv_vw) {
        bq = div_small(c, nv, inv_nv);
        v = c - bq * nv;
        vy = div_small(v, vw, inv_vw);
        vx = v - vy * vw;
    }
    // jump `step` cells ahead; (qy, qx) = divmod(step, vw) precomputed by the caller
    __device__ __forceinline__ void advance(int step, int qy, int qx, int nv, int vw, int vh) {
        v += step;
        vx += qx;
        vy += qy;
        if (vx >= vw) {
            vx -= vw;
            ++vy;
        }
        while (v >= nv) {          // (v - nv) / vw == vy - vh because nv == vh * vw
            v -= nv;
            vy -= vh;
            ++bq;
        }
    }
    __device__ __forceinline__ void next(int nv, int vw) {
        ++v;
        ++vx;
        if (vx == vw) {
            vx = 0;
            ++vy;
        }
        if (v == nv) {
            v = vy = vx = 0;
            ++bq;
        }
    }
};

// The cursor plus the board cell it looks at, kept incrementally: per-board view parameters (origin,
// first exit) are read from LDS when the board changes, the source cell is reduced modulo the board
// once per jump and then only stepped.
template <int H, int W>
struct ObsSource : ObsCursor {
    int oy, ox, tv0, xk0;       // board cell under the view's top-left corner; first exit: view cell, board cell
    int sy, sx;                 // board cell under the cursor
    __device__ __forceinline__ void load(const unsigned char *smem) {
        const int *pp = (const int *)(smem + Geom<H, W>::OFF_GSH) + bq * OBS_PAR_INTS;
        oy = pp[0];
        ox = pp[1];
        tv0 = pp[2];
        xk0 = pp[2 + OBS_MAX_EXITS];
    }
    __device__ __forceinline__ void locate() {
        sy = (int)((unsigned)(oy + vy) % (unsigned)H);
        sx = (int)((unsigned)(ox + vx) % (unsigned)W);
    }
    __device__ __forceinline__ void start(const unsigned char *smem, int c, int nv, int vw, float inv_nv, float inv_vw) {
        init(c, nv, vw, inv_nv, inv_vw);
        load(smem);
        locate();
    }
    __device__ __forceinline__ void jump(const unsigned char *smem, int step, int qy, int qx, int nv, int vw, int vh) {
        const int b0 = bq;
        advance(step, qy, qx, nv, vw, vh);
        if (bq != b0) load(smem);
        locate();
    }
    __device__ __forceinline__ void step(const unsigned char *smem, int nv, int vw) {
        ++v;
        ++vx;
        sx = sx + 1 == W ? 0 : sx + 1;
        if (vx == vw) {
            vx = 0;
            ++vy;
            sx = ox;
            sy = sy + 1 == H ? 0 : sy + 1;
        }
        if (v == nv) {
            v = vy = vx = 0;
            ++bq;
            load(smem);
            sy = oy;
            sx = ox;
        }
    }
};

// Observation word (board | goal colour << 16) at a cursor position.
template <int H, int W>
__device__ __forceinline__ u32 obs_fetch(const sl_env_batch &env, const unsigned char *smem, const ObsSource<H, W> &cu,
                                         int n_exits) {
    using Gm = Geom<H, W>;
    const u16 *b16 = (const u16 *)(smem + Gm::OFF_BOARD + Gm::PAD) + cu.bq * Gm::HW;
    const u16 *g16 = (const u16 *)(smem + Gm::OFF_GOALS + Gm::PAD) + cu.bq * Gm::HW;
    int cell = __mul24(cu.sy, W) + cu.sx;
    if (n_exits > 0 && cu.tv0 == cu.v) cell = cu.xk0;
    if (n_exits > 1) {                            // later exits overwrite earlier ones, as numpy does
        const int *pp = (const int *)(smem + Gm::OFF_GSH) + cu.bq * OBS_PAR_INTS;
        for (int k = 1; k < n_exits; ++k)
            if (pp[2 + k] == cu.v) cell = pp[2 + OBS_MAX_EXITS + k];
    }
    cell = Gm::flat(cell);
    u32 g = g16[cell] & COLORS;
    if (env.remove_white_goals && g == COLORS) g = 0;
    return (u32)b16[cell] | (g << 16);
}

// The four channel bytes [4g, 4g+4) of a cell as one dword (missing channels give 0).
template <int C>
__device__ __forceinline__ u32 obs_bytes(const sl_env_batch &env, u32 word, int g) {
    const int c0 = env.channels[4 * g];
    const int n = C - 4 * g >= 4 ? 4 : C - 4 * g;
    bool run = true;
    for (int t = 1; t < n; ++t) run = run && env.channels[4 * g + t] == c0 + t;
    if (run)      // consecutive bits: spread a nibble over four bytes with one multiply
        return (((word >> c0) & ((1u << n) - 1u)) * 0x00204081u) & 0x01010101u;
    u32 d = 0;
    for (int t = 0; t < n; ++t) d |= ((word >> env.channels[4 * g + t]) & 1u) << (8 * t);
    return d;
}

template <int C>
struct ObsSel {
    // byte i of the 4*C-byte group of four cells lives in padded dword (i / C) * P + ((i % C) >> 2), byte (i % C) & 3
    static constexpr int P = (C + 3) / 4;
    static constexpr int src(int i) { return (i / C) * P + ((i % C) >> 2); }
    static constexpr int first(int j) { return src(4 * j); }
    static constexpr int second(int j) {
        for (int t = 1; t < 4; ++t)
            if (src(4 * j + t) != src(4 * j)) return src(4 * j + t);
        return src(4 * j);
    }
    static constexpr bool two_sources(int j) {
        for (int t = 0; t < 4; ++t)
            if (src(4 * j + t) != first(j) && src(4 * j + t) != second(j)) return false;
        return true;
    }
    static constexpr bool ok() {
        for (int j = 0; j < C; ++j)
            if (!two_sources(j)) return false;
        return true;
    }
    static constexpr u32 selector(int j) {      // v_perm_b32 selector: bytes 0-3 = first source, 4-7 = second
        u32 sel = 0;
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * j + t;
            const u32 byte = (u32)((i % C) & 3) + (src(i) == first(j) ? 0u : 4u);
            sel |= byte << (8 * t);
        }
        return sel;
    }
};

// Observation of the workgroup's boards.  C = 15 / 19: uint8 channels, four cells (4*C bytes, a whole
// number of dwords) per lane, assembled in registers and written with wide stores; C = 0: any other
// channel list (byte stores) and the raw uint32 view.
template <int H, int W, int C>
__device__ __forceinline__ void write_obs_block(const sl_env_batch &env, const unsigned char *smem, int e0b,
                                                int nbb, int tid) {
    const int nv = env.view_h * env.view_w, vw = env.view_w;
    const float inv_nv = 1.0f / (float)nv, inv_vw = 1.0f / (float)vw;
    const int ncell = nbb * nv;
    const int n_exits = min(env.E, OBS_MAX_EXITS);
    ObsSource<H, W> cu;
    if (C == 0 && env.n_channels == 0) {
        // raw uint32 view: four consecutive cells per thread, one aligned 16-byte store each (the view
        // of a workgroup's first board starts on a 32-byte boundary: e0b is a multiple of 8)
        u32x4 *dst = (u32x4 *)((u32 *)env.obs + (size_t)e0b * nv);
        const int ngroup = ncell / 4;
        constexpr int STEP = 4 * 64 * WAVES - 4;
        const int qy = STEP / vw, qx = STEP - qy * vw;
        cu.start(smem, 4 * tid, nv, vw, inv_nv, inv_vw);
        for (int u = tid; u < ngroup; u += 64 * WAVES, cu.jump(smem, STEP, qy, qx, nv, vw, env.view_h)) {
            u32 w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w4[q] = obs_fetch<H, W>(env, smem, cu, n_exits);
                cu.step(smem, nv, vw);
            }
            dst[u] = u32x4{w4[0], w4[1], w4[2], w4[3]};
        }
        for (int c = 4 * ngroup + tid; c < ncell; c += 64 * WAVES) {      // tail workgroup leftovers
            cu.start(smem, c, nv, vw, inv_nv, inv_vw);
            ((u32 *)env.obs)[(size_t)e0b * nv + c] = obs_fetch<H, W>(env, smem, cu, n_exits);
        }
        return;
    }
    if constexpr (C == 0) {
        const int nc = env.n_channels;
        const int qy = (64 * WAVES) / vw, qx = (64 * WAVES) - qy * vw;
        cu.start(smem, tid, nv, vw, inv_nv, inv_vw);
        for (int c = tid; c < ncell; c += 64 * WAVES, cu.jump(smem, 64 * WAVES, qy, qx, nv, vw, env.view_h)) {
            const u32 word = obs_fetch<H, W>(env, smem, cu, n_exits);
            if (nc == 0) {
                ((u32 *)env.obs)[(size_t)e0b * nv + c] = word;
            } else {
                uint8_t *o = env.obs + ((size_t)e0b * nv + c) * nc;
                for (int k = 0; k < nc; ++k) o[k] = (word >> env.channels[k]) & 1u;
            }
        }
    } else {
        using Sel = ObsSel<C>;
        using Gm = Geom<H, W>;
        constexpr int P = Sel::P;
        // staging space: the goal-word and score-table regions are dead by now (their first bytes hold the
        // per-board view parameters)
        constexpr int STAGE_OFF = (Gm::NB * OBS_PAR_INTS * 4 + 15) & ~15;     // behind the per-board view parameters
        constexpr int STAGE_ROOM = Gm::GSH_BYTES + 4096 - STAGE_OFF;
        // when a whole wave's run does not fit, it goes out in SUB rounds of LPS lanes each
        constexpr int SUB = WAVES * 64 * 4 * C <= STAGE_ROOM ? 1 : (WAVES * 32 * 4 * C <= STAGE_ROOM ? 2 : 0);
        constexpr bool STAGED = SUB != 0;
        constexpr int LPS = STAGED ? 64 / SUB : 64;
        u32 *stage = (u32 *)(const_cast<unsigned char *>(smem) + Gm::OFF_GSH + STAGE_OFF) + (tid >> 6) * LPS * C;
        u32 *dst = (u32 *)(env.obs + (size_t)e0b * nv * C);      // e0b is a multiple of 8: dword aligned
        const int ngroup = ncell / 4;
        constexpr int STEP = 4 * 64 * WAVES - 4;              // after the 4 next() calls of a group
        const int qy = STEP / vw, qx = STEP - qy * vw;
        cu.start(smem, 4 * tid, nv, vw, inv_nv, inv_vw);
        // (whole waves iterate together -- the staged write-back below is a wave-wide job -- and lanes
        //  past the last group just skip the gather)
        for (int u = tid; u - (tid & 63) < ngroup; u += 64 * WAVES, cu.jump(smem, STEP, qy, qx, nv, vw, env.view_h)) {
            u32 pad[4 * P];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 word = u < ngroup ? obs_fetch<H, W>(env, smem, cu, n_exits) : 0u;
#pragma unroll
                for (int g = 0; g < P; ++g) pad[q * P + g] = obs_bytes<C>(env, word, g);
                cu.step(smem, nv, vw);
            }
            u32 out[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                static_assert(Sel::ok(), "obs packing needs at most two source dwords per output dword");
                out[j] = __builtin_amdgcn_perm(pad[Sel::second(j)], pad[Sel::first(j)], Sel::selector(j));
            }
            if (!STAGED && u >= ngroup) continue;
            if constexpr (STAGED) {
                // a thread's 4C bytes are contiguous, but a store instruction would write 16 of every 4C bytes
                // of the wave's run (partial lines from four instructions).  Each wave parks its 64 x 4C bytes
                // in LDS and writes them back lane-linear: every store instruction covers 1 KiB contiguously.
                const int ln = tid & 63;
#pragma unroll
                for (int sub = 0; sub < SUB; ++sub) {
                    if (ln / LPS == sub) {
                        u32 *mine = stage + (ln % LPS) * C;
#pragma unroll
                        for (int j = 0; j < C; ++j) mine[j] = out[j];
                    }
                    wave_sync();
                    const int first = u - ln + sub * LPS;                              // first group of this round
                    const int run_dwords = max(0, min(LPS, ngroup - first)) * C;       // dwords really produced
                    u32 *o = dst + (size_t)first * C;
#pragma unroll
                    for (int j = 0; j < (LPS * C + 255) / 256; ++j) {
                        const int d = 4 * (ln + 64 * j);
                        if (d + 4 <= run_dwords) {
                            *(u32x4_a4 *)(o + d) = *(const u32x4 *)(stage + d);
                        } else {
                            for (int q = d; q < run_dwords; ++q) o[q] = stage[q];
                        }
                    }
                    wave_sync();
                }
            } else {
                u32 *o = dst + (size_t)u * C;
                int j = 0;
#pragma unroll
                for (; j + 4 <= C; j += 4) *(u32x4_a4 *)(o + j) = u32x4_a4{out[j], out[j + 1], out[j + 2], out[j + 3]};
                if constexpr (C % 4 == 3) *(u32x3_a4 *)(o + C - 3) = u32x3_a4{out[C - 3], out[C - 2], out[C - 1]};
                if constexpr (C % 4 == 2) *(u32x2_a4 *)(o + C - 2) = u32x2_a4{out[C - 2], out[C - 1]};
                if constexpr (C % 4 == 1) o[C - 1] = out[C - 1];
            }
        }
        for (int c = 4 * ngroup + tid; c < ncell; c += 64 * WAVES) {      // tail workgroup leftovers
            cu.start(smem, c, nv, vw, inv_nv, inv_vw);
            const u32 word = obs_fetch<H, W>(env, smem, cu, n_exits);
            uint8_t *o = env.obs + ((size_t)e0b * nv + c) * C;
            for (int k = 0; k < C; ++k) o[k] = (word >> env.channels[k]) & 1u;
        }
    }
}

// uint8 planes for the two standard channel lists (training: bits 0-11, 25-27; default: bits 0-15, 25-27), where
// every channel is known at compile time.  No staging: a work item is sixteen consecutive cells of one board's
// TRANSPOSED view, i.e. a run down a column of the board image (wrapping into the next column), which it walks itself
// -- two 16-bit LDS reads per cell (board word, goal word), the cursor kept incrementally.  The sixteen values are
// byte-transposed four at a time (v_perm: low bytes, high bytes, the goal words' high bytes), after which a channel's
// four output bytes are one shift and one mask of a dword: ~13 vector instructions per cell and ~8 per 16-byte
// store, against ~35 per cell (fetch in view order, stage through LDS, two barriers per round) and ~13 per store
// for the 16 x 16 bit transposition this replaces.
template <int H, int W, int NCH>
__device__ __forceinline__ void policy_direct_u8_std(const sl_env_batch &env, const unsigned char *smem, int e0b, int nbb,
                                                     int tid) {
    using Gm = Geom<H, W>;
    typedef u32 u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    constexpr int C = NCH;
    const int vh = env.view_h, vw = env.view_w, nv = vh * vw;
    const int cpb = (nv + 15) >> 4;
    const float inv_cpb = 1.0f / (float)cpb, inv_vh = 1.0f / (float)vh, inv_vw = 1.0f / (float)vw;
    const int n_exits = min(env.E, OBS_MAX_EXITS);
    for (int it = tid; it < nbb * cpb; it += 64 * WAVES) {
        const int bq = div_small(it, cpb, inv_cpb), i0 = (it - bq * cpb) * 16;
        const int n_el = min(16, nv - i0);
        const int *pp = (const int *)(smem + Gm::OFF_GSH) + bq * OBS_PAR_INTS;
        const u16 *b16 = (const u16 *)(smem + Gm::OFF_BOARD + Gm::PAD) + bq * Gm::HW;
        const u16 *g16 = (const u16 *)(smem + Gm::OFF_GOALS + Gm::PAD) + bq * Gm::HW;
        const int oy = pp[0], ox = pp[1];
        int vx = div_small(i0, vh, inv_vh), vy = i0 - vx * vh;
        int sy = (int)((unsigned)(oy + vy) % (unsigned)H), sx = (int)((unsigned)(ox + vx) % (unsigned)W);
        // first exit: where it is painted, as an offset into this run (transposed index - i0), and its board cell
        int e_rel = -1, e_cell = 0;
        if (n_exits > 0 && pp[2] >= 0) {
            const int ty = div_small(pp[2], vw, inv_vw), tx = pp[2] - ty * vw;
            e_rel = tx * vh + ty - i0;
            e_cell = Gm::flat(pp[2 + OBS_MAX_EXITS]);
        }
        u32 bv[16], gv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int cell = Gm::cell(sy, sx);
            if (q == e_rel) cell = e_cell;
            bv[q] = b16[cell];
            gv[q] = g16[cell];
            ++vy;
            sy = sy + 1 == H ? 0 : sy + 1;
            if (vy == vh) {
                vy = 0;
                sy = oy;
                sx = sx + 1 == W ? 0 : sx + 1;
            }
        }
        if (n_exits > 1) {                          // later exits overwrite earlier ones, as numpy does
            for (int k = 1; k < n_exits; ++k) {
                const int tv = pp[2 + k];
                if (tv < 0) continue;
                const int ty = div_small(tv, vw, inv_vw), tx = tv - ty * vw;
                const int rel = tx * vh + ty - i0;
                if (rel < 0 || rel >= 16) continue;
                const int cell = Gm::flat(pp[2 + OBS_MAX_EXITS + k]);
                const u32 bk = b16[cell], gk = g16[cell];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    bv[q] = q == rel ? bk : bv[q];
                    gv[q] = q == rel ? gk : gv[q];
                }
            }
        }
        // byte transposition, four cells at a time: B0 / B1 = the board words' low / high bytes, G1 = the goal words'
        // high bytes (goal colour = bits 9-11 of the word, bits 1-3 of that byte; channels 25-27 of the view word)
        u32 B0[4], B1[4], G1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 t = __builtin_amdgcn_perm(bv[4 * k + 1], bv[4 * k], 0x05010400u);
            const u32 u = __builtin_amdgcn_perm(bv[4 * k + 3], bv[4 * k + 2], 0x05010400u);
            B0[k] = __builtin_amdgcn_perm(u, t, 0x05040100u);
            B1[k] = __builtin_amdgcn_perm(u, t, 0x07060302u);
            const u32 tg = __builtin_amdgcn_perm(gv[4 * k + 1], gv[4 * k], 0x0C0C0501u);
            const u32 ug = __builtin_amdgcn_perm(gv[4 * k + 3], gv[4 * k + 2], 0x0C0C0501u);
            u32 g = __builtin_amdgcn_perm(ug, tg, 0x05040100u);
            if (env.remove_white_goals) {           // all three colour bits set: no goal
                const u32 m = g & (g >> 1) & (g >> 2) & 0x02020202u;
                g &= ~(m | (m << 1) | (m << 2));
            }
            G1[k] = g;
        }
        uint8_t *dst = (uint8_t *)env.policy_obs + (size_t)(e0b + bq) * (size_t)C * nv + i0;
#pragma unroll
        for (int c = 0; c < C; ++c, dst += nv) {
            const int bit = c < C - 3 ? c : 25 + (c - (C - 3));                 // both lists end with 25, 26, 27
            const u32 *src = bit < 8 ? B0 : (bit < 16 ? B1 : G1);
            const int sh = bit < 16 ? (bit & 7) : bit - 24;
            u32 w4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w4[k] = (src[k] >> sh) & 0x01010101u;
            // (byte-aligned 16-byte vectors: planes are nv bytes apart.  Stores rounded down to 16 bytes -- wrong
            //  data, right count -- run no faster: the epilogue is bound by its instructions, not by split stores)
            if (n_el == 16) {
                *(u32x4_a1 *)dst = u32x4_a1{w4[0], w4[1], w4[2], w4[3]};
            } else {
                for (int q = 0; q < n_el; ++q) dst[q] = (uint8_t)((w4[q >> 2] >> (8 * (q & 3))) & 1u);
            }
        }
    }
}

// One round of write_policy_block: PER consecutive cells of a board's transposed view per work item (16 for
// uint8, 4 for float32: 16 output bytes per channel either way, one store).
template <int PER>
__device__ __forceinline__ void policy_planes(const sl_env_batch &env, const u32 *stage, int pnv, int nb, int nv, int C,
                                              size_t first_board, int tid) {
    typedef u32 u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    const int cpb = (nv + PER - 1) / PER;           // chunks per board
    const float inv_cpb = 1.0f / (float)cpb;
    for (int it = tid; it < nb * cpb; it += 64 * WAVES) {
        const int bq = div_small(it, cpb, inv_cpb), xy0 = (it - bq * cpb) * PER;
        const int n_el = min(PER, nv - xy0);
        const u32 *view = stage + bq * pnv;
        u32 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i0 = min(xy0 + q, nv - 1);
            v[q] = view[i0 + (i0 >> 4)];
        }
        const size_t o = (first_board + bq) * C * nv + xy0;
        for (int c = 0; c < C; ++c) {
            const u32 shift = (u32)env.channels[c];
            u32 b[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) b[q] = (v[q] >> shift) & 1u;
            if (PER == 16) {
                uint8_t *dst = (uint8_t *)env.policy_obs + o + (size_t)c * nv;
                if (n_el == PER) {
                    u32 w[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        w[q] = b[(4 * q) % PER] | (b[(4 * q + 1) % PER] << 8) | (b[(4 * q + 2) % PER] << 16) | (b[(4 * q + 3) % PER] << 24);
                    *(u32x4_a1 *)dst = u32x4_a1{w[0], w[1], w[2], w[3]};
                } else {
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        if (q < n_el) dst[q] = (uint8_t)b[q];
                }
            } else {
                u32 *dst = (u32 *)env.policy_obs + o + (size_t)c * nv;
                if (n_el == PER) {
                    *(u32x4_a4 *)dst = u32x4_a4{b[0] ? 0x3F800000u : 0u, b[1 % PER] ? 0x3F800000u : 0u,
                                                b[2 % PER] ? 0x3F800000u : 0u, b[3 % PER] ? 0x3F800000u : 0u};
                } else {
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        if (q < n_el) dst[q] = b[q] ? 0x3F800000u : 0u;
                }
            }
        }
    }
}

// The observation in the layout the policy network convolves (training/models.py:100-103, ppo.py:64): channel
// first, spatial axes swapped, out[b][c][x][y] = bit channels[c] of the view word at (y, x); uint8 or float32.
// The raw view words of a few boards at a time are parked in LDS (the goal-word / score-table regions are dead
// by now), then every thread produces four consecutive output elements -- contiguous along y -- per store.
template <int H, int W>
__device__ __forceinline__ void write_policy_block(const sl_env_batch &env, unsigned char *smem, int e0b, int nbb, int tid) {
    using Gm = Geom<H, W>;
    const int vh = env.view_h, vw = env.view_w, nv = vh * vw, C = env.n_channels;
    const float inv_nv = 1.0f / (float)nv, inv_vw = 1.0f / (float)vw, inv_vh = 1.0f / (float)vh;
    const int n_exits = min(env.E, OBS_MAX_EXITS);
    constexpr int STAGE_OFF = (Gm::NB * OBS_PAR_INTS * 4 + 15) & ~15;       // behind the per-board view parameters
    constexpr int ROOM = (Gm::GSH_BYTES + 4096 - STAGE_OFF) / 4;            // dwords
    u32 *stage = (u32 *)(smem + Gm::OFF_GSH + STAGE_OFF);
    // a board's words are padded by one per sixteen: lanes read 16 apart (one 16-element chunk each), which
    // would put a whole wave on two LDS banks
    const int pnv = nv + (nv >> 4) + 1;
    const int per_round = ROOM / pnv;           // boards per round (the launcher guarantees >= 1)
    // one of the two standard channel lists?  (wave-uniform: the channel list is a kernel argument) -- uint8 planes
    // of those are written without staging
    int std_list = (C == 15 || C == 19) ? C : 0;
    for (int c = 0; c < C && std_list; ++c)
        if (env.channels[c] != (c < C - 3 ? c : 25 + (c - (C - 3)))) std_list = 0;
    if (env.policy_dtype == 0 && std_list == 15) return policy_direct_u8_std<H, W, 15>(env, smem, e0b, nbb, tid);
    if (env.policy_dtype == 0 && std_list == 19) return policy_direct_u8_std<H, W, 19>(env, smem, e0b, nbb, tid);
    for (int b0 = 0; b0 < nbb; b0 += per_round) {
        const int nb = min(per_round, nbb - b0);
        __syncthreads();
        // the view words of the round's boards, stored TRANSPOSED (x-major): the output runs along y, so a
        // thread's 16 output elements then come from 16 consecutive words
        ObsSource<H, W> cu;
        {
            const int qy = (64 * WAVES) / vw, qx = (64 * WAVES) - qy * vw;
            cu.start(smem, b0 * nv + tid, nv, vw, inv_nv, inv_vw);
            for (int c = tid; c < nb * nv; c += 64 * WAVES, cu.jump(smem, 64 * WAVES, qy, qx, nv, vw, vh)) {
                const int i = cu.vx * vh + cu.vy;
                stage[(cu.bq - b0) * pnv + i + (i >> 4)] = obs_fetch<H, W>(env, smem, cu, n_exits);
            }
        }
        __syncthreads();
        // outputs of this round: [nb][C][vw][vh].  A work item is 16 consecutive cells of one board's transposed
        // view: its 16 words are read once and give a 16-element chunk of EVERY channel plane (one store per
        // channel, 16 uint8 or 4 float32 elements) -- the channel loop is wave-uniform, so the channel's bit
        // position is a scalar.  Planes are nv elements apart, i.e. the uint8 stores are only byte-aligned; a wave
        // still writes 1 KiB contiguously per channel.  (uint8, measured per C3 step: chunks cut from the flat
        // output run instead -- aligned stores, a division per chunk and per-element plane-crossing logic -- 55 us,
        // this form 38; a divergent slow path for the one chunk in 39 that crosses a plane 77; one 16-byte ALIGNED
        // chunk of one plane per work item, its words re-read per channel, 64; the store window moved up to the
        // plane's next dword boundary, 56.  The standard channel lists no longer come here: policy_direct_u8_std.)
        if (env.policy_dtype == 0) policy_planes<16>(env, stage, pnv, nb, nv, C, (size_t)(e0b + b0), tid);
        else policy_planes<4>(env, stage, pnv, nb, nv, C, (size_t)(e0b + b0), tid);
    }
}

// ---- fused env step / rollout ---------------------------------------------------------------------

// Per-board mailbox between a board's row lanes and its leader lane (LDS).
struct BoardBox {
    int score;          // rows -> leader: sum(points_table * alive_counts) of the board after the CA step
    int side;           // rows -> leader (WRAP): cells that differ from the side-effect baseline
    int gstat;          // rows -> leader: goals_static as the rows see it
    int reset_level;    // leader -> rows: pool level to load now, or -1
    int qslot;          // leader -> rows: slot of the finished-episode queue that takes the board, or -1
    int score0;         // rows -> leader: score of the freshly loaded level
    int any;            // (box 0 only) bit 0: some board of the workgroup resets, bit 1: some board is queued
    int dirty;          // rows -> everyone: the CA changed a cell of the board in this launch (or a level was loaded)
};
static_assert(sizeof(BoardBox) == 32, "mailbox stride");

// LEAN: the instantiation for batches without observation, policy-layout tensor, finished-episode queue and
// wrappers -- the plain step of the headline workload; every cold feature compiled into it costs the hot path
// scheduling freedom.
//
// Division of labour (round 3).  Everything that concerns a board as a whole -- the agent's move, the exit repaint,
// reward / done / episode accounting, the wrappers' arithmetic, the decision to reset -- is serial, branchy code
// for ONE lane per board.  Run by the board's first row lane (as rounds 1-2 did), every wavefront executes it for the
// sake of two lanes, and its latency chain (dependent LDS and global accesses) sits between the loads and the CA.
// Now wavefront 0 is the workgroup's LEADER wave: its lane q speaks for board q of the workgroup, and
//   * it issues none of the bulk DMA (the other three waves move the spans), so its memory counters are free for the
//     dependent loads of the agent's move: record + action, then the four cells the move can touch, straight from
//     global memory (the previous launch's board) -- the move is DECIDED while the spans are in flight and costs
//     four LDS stores once they have landed;
//   * row lanes and leader lanes talk through a mailbox in LDS and workgroup barriers (four per single step).
//
// ONE: the instantiation for single-step launches (T == 1, what step() issues).  Without a step loop there is nothing
// for the compiler to hoist: left as a loop, every loop-invariant address and constant of the RARE blocks (reset, exit
// tables, queue, the division by the pool size) is computed ahead of the loop by every wave of every launch --
// some 150 instructions between the load barrier and the first CA pass.
//
// LEADX (the LEAN variants of the narrow shapes): the leader wave is a FIFTH wavefront that holds no rows, so the
// move is in the image as soon as the spans have landed, and what the leaders do around the scores overlaps the rows'
// CA instead of following it.
template <int H, int W, bool LEAN>
constexpr bool leadx() { return LEAN && Geom<H, W>::LEADX_OK; }

// ---- the goal-word cache (round 5) ----------------------------------------------------------------------------------
// Goals are static in nearly every level (safelife_game.py:753-760 finds that out once per episode), yet every launch
// moved the workgroup's goal span into LDS, had every lane read its goal row back out of the image (14 LDS reads),
// re-assemble the split-halves words (26 permutes) and shift the colours into the score index (26 more) -- ~0.35 us of a
// 6.2 us step by the timing-only builds of round 4.  The LEAN kernels now KEEP those words: a workgroup whose boards all
// have static goals writes each lane's WS pre-shifted goal words (goal_shift) to a per-workgroup block of `goal_cache`, in
// the order the lanes read them back (16 bytes per lane and 1 KiB per wave instruction, then single dwords), and raises
// the workgroup's flag word; a launch that finds the flag raised moves no goal span at all and loads the words straight
// into registers, under the board's DMA.  A board that resets (or whose goals evolve) lowers the flag in the same
// launch, and the next launch takes the LDS route again and decides anew.  No LDS traffic; since round 6 a fifth of
// the goal span's bytes (the words are packed five to a dword, below) for two vector instructions per word.
//
// Layout: one block per workgroup of the BATCH (env index / NB: launches start at multiples of NB -- the launcher passes
// no cache otherwise), whatever slice, queue or stream steps it: a 256-byte header whose first dword is the flag, then
// WAVES x PW x 64 dwords (PW = ceil(WS / 5): five goal words per dword, below).  Zeroing the whole cache lowers every flag.
template <int H, int W>
struct GoalCache {
    using Gm = Geom<H, W>;
    // (round 6) the cache keeps the words FIVE to a dword (goal_unpack above): a fifth of the bytes (25x25: 3 dwords per
    // row lane and step instead of 13 -- 940 of the 3750 bytes an env-step moves; 64x64: 7 instead of 32), for a shift
    // and a mask per word where the score reads them
    static constexpr int PW = (Gm::WS + 4) / 5;                     // packed dwords per lane
    static constexpr int X4 = PW / 4, TAIL = PW % 4;               // 16-byte loads per lane, then single dwords
    static constexpr int WAVE_DWORDS = PW * 64;
    static constexpr int HEAD_DWORDS = 64;
    static constexpr int BLOCK_DWORDS = HEAD_DWORDS + WAVES * WAVE_DWORDS;
    static __host__ __device__ constexpr size_t bytes(int B) { return 4 * (size_t)((B + Gm::NB - 1) / Gm::NB) * BLOCK_DWORDS; }
    // the lane's words of wave `wave` of block `blk`: words 4c..4c+3 at x4(c), word 4 X4 + j at tail(j)
    static __device__ __forceinline__ u32x4 *x4(u32 *blk, int wave, int lane, int c) {
        return (u32x4 *)(blk + HEAD_DWORDS + wave * WAVE_DWORDS + c * 256 + lane * 4);
    }
    static __device__ __forceinline__ u32 *tail(u32 *blk, int wave, int lane, int j) {
        return blk + HEAD_DWORDS + wave * WAVE_DWORDS + X4 * 256 + j * 64 + lane;
    }
};

#ifndef SL_SPAWN_GSH_REG
#define SL_SPAWN_GSH_REG 1      /* A/B knob: the LEAN single-step spawner variant keeps the goal words in registers (1) or
                                   takes the move box (0) -- both together tip it into scratch */
#endif
// the goal colours of a lane's row live in registers (else in the OFF_GSH region of LDS)
template <int H, int W>
constexpr bool gsh_in_registers(bool spawn, bool lean, bool one) {
    return !spawn || Geom<H, W>::WAVES_PER_SIMD < 4 || (SL_SPAWN_GSH_REG && one && lean && W <= 25);
}
#ifndef SL_LEAN_LDS
#define SL_LEAN_LDS 1           /* A/B knob: 0 = every variant asks for the full LDS layout (rounds 1-3) */
#endif
// LEAN variants whose goal words live in registers use nothing of the OFF_GSH region (no observation parks its
// parameters there, no wrapper its baseline rows): the score table and the move box move down into it and the
// workgroup asks for 13 KB less (25x25: 37.7 -> 24.4 KB).  The four workgroups a CU holds of the four-queue step then
// leave 62 instead of 9 KB of its LDS free -- room for the workgroups of OTHER kernels: RCCL's exchange kernel (40
// workgroups of 19.5 KB) otherwise waits tens of microseconds for a CU on which two step workgroups happen to retire
// together, and while it waits the hardware pipe it is being dispatched from serves nobody else -- the slice whose
// queue shares that pipe stands still (profiles/round4_g_*).
template <int H, int W>
constexpr bool lean_lds(bool spawn, bool lean, bool one) {
    return SL_LEAN_LDS && lean && gsh_in_registers<H, W>(spawn, lean, one);
}
template <int H, int W>
constexpr int lean_lds_bytes() { return Geom<H, W>::OFF_GSH + 4096 + Geom<H, W>::NB * 16; }
// the plain single-step kernels whose goal words live in registers: goal-word cache, no goal image in LDS
template <int H, int W>
constexpr bool nogoals_lds(bool spawn, bool lean, bool one) {
    return lean && one && !Geom<H, W>::LEADX_OK && gsh_in_registers<H, W>(spawn, lean, one);
}
template <int H, int W>
constexpr int nogoals_shift() { return Geom<H, W>::REGION - 16; }

#ifndef SL_WIDE_WAVES
#define SL_WIDE_WAVES 4         /* A/B knob: waves per SIMD the plain single-step kernels of the wide shapes are compiled for (2: rounds 1-4;
                                   3: round 5; 4 since the goal words stay packed in registers -- 64x64 without spawners 141 -> 123 registers,
                                   so that all four workgroups a CU gets of a four-queue step are resident at once; the spawner variants
                                   of 48- and 64-cell rows do not fit 128 and stay at 3) */
#endif
template <int H, int W>
constexpr int wide_waves(bool spawn) { return spawn && Geom<H, W>::WS > 20 ? (SL_WIDE_WAVES < 3 ? SL_WIDE_WAVES : 3) : SL_WIDE_WAVES; }
template <int H, int W, bool LDS_LUT, bool SPAWN, bool WRAP, bool LEAN, bool ONE>
__global__ __launch_bounds__(64 * (WAVES + (leadx<H, W, LEAN>() ? 1 : 0)),
                             (leadx<H, W, LEAN>() ? 5
                              : (Geom<H, W>::WAVES_PER_SIMD < 4 && nogoals_lds<H, W>(SPAWN, LEAN, ONE)) ? wide_waves<H, W>(SPAWN)
                                                                                                       : Geom<H, W>::WAVES_PER_SIMD)) void k_env_rollout_rowlane(
    // the eight arguments the prologue needs before anything else come first: with
    // -amdgpu-kernarg-preload-count=8 they arrive in SGPRs with the wave instead of behind an s_load
    const u16 *__restrict__ hot_board, const u16 *__restrict__ hot_goals, const sl_pcg64 *__restrict__ hot_rng,
    sl_env_scalars *__restrict__ hot_scalars,
    // the goal-word cache of the batch (GoalCache above), or null: its flag word is the kernel's first fetch
    u32 *__restrict__ hot_gcache,
    const int32_t *__restrict__ actions, int hot_first, int hot_end,
    // (the preload ends here -- 14 SGPRs next to the kernarg pointer; the score table's DMA is the last to be issued)
    const int8_t *__restrict__ hot_lut,
    // The batch constants travel by value.  (Measured alternative: the struct resident in device memory behind
    // a pointer -- ~100 bytes of arguments instead of ~700 -- is SLOWER, 12.1 vs 11.5 us per C3 step and 16.6 vs
    // 13.8 in the first steps after a reset: loads through a global pointer are not invariant for the compiler,
    // which re-fetches fields after every store with a scalar-cache round trip each time, on the leader's
    // critical path; kernel-argument loads are.)
    sl_env_batch env, int hot_E, int tstride, int T_arg, sl_step_out *__restrict__ out_rec,
    float *__restrict__ reward_t, uint8_t *__restrict__ done_t, double *__restrict__ shaped_t,
    const Jump *__restrict__ jump,
    // queue stepping without a release fence between steps (sl_aql.hip, opt-in): workgroup i of this launch must run on
    // XCD (xcd_base + i) mod 8 -- where the launches before it left this slice's state -- and raises the host-visible
    // word xcd_flag if it finds itself anywhere else; xcd_flag is null on every other launch
    int xcd_base, u32 *__restrict__ xcd_flag,
    // T_arg == -1: SafeLifeEnv.reset() for the envs with reset_mask[e] != 0 (null: all of them) instead of steps
    const uint8_t *__restrict__ reset_mask) {
    using Gm = Geom<H, W>;
    constexpr int WS = Gm::WS, HW = Gm::HW;
    const int T = ONE ? 1 : T_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifndef SL_SCALAR_WAVE
#define SL_SCALAR_WAVE 1        /* A/B knob: the wave's index as a scalar (uniform branches and scalar DMA addresses in the prologue) */
#endif
#ifndef SL_PRIO_PROLOGUE
#define SL_PRIO_PROLOGUE 0      /* A/B knob: s_setprio of every wave until its loads have been issued */
#endif
    if (SL_PRIO_PROLOGUE) __builtin_amdgcn_s_setprio(SL_PRIO_PROLOGUE);
    const int tid = threadIdx.x, lane = tid & 63;
    // (the plain single-step kernels only: the others sit at their register limit, and a scalar there tips them into scratch)
    const int wave = (SL_SCALAR_WAVE && ONE && LEAN) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    // envs [hot_first, hot_end) of the batch: one slice (slhip_env_step_slices) or all of it
    const unsigned B = tstride;                        // row pitch of the [T, B] per-step arrays
    const int E = hot_E;
    const int e0b = hot_first + blockIdx.x * Gm::NB;
    const int nbb = min(Gm::NB, hot_end - e0b);
    const LaneMap<H, W> lm(lane);
    const int g = lm.g, r = lm.r, up = lm.up, dn = lm.dn;
    const int gb = wave * Gm::G + g;
    const bool rowl = lane < Gm::NL && gb < nbb;       // holds a row: its own, or a halo copy (V_SHIFT)
    const bool live = rowl && lm.real;                 // owns row r of board gb
    const bool rlead = live && r == 0;                 // the row lane that writes the board's mailbox
    constexpr bool LEADX = leadx<H, W, LEAN>();
    // The plain single-step kernels keep the lanes' goal words across launches (GoalCache) and NO goal image in LDS
    // (NOGOALS): a launch on cached words needs none, and the rare launch that does need goal rows -- no cache yet, an
    // evolving goal array, a level being loaded -- moves them between registers and global memory row by row.
    // Everything behind the board image then sits one image lower (smem_hi), and the workgroup asks for that much less
    // LDS (25x25: 24.4 -> 14.4 KB; 64x64: 70 -> 37 KB: four workgroups fit a CU instead of two).
    constexpr bool NOGOALS = nogoals_lds<H, W>(SPAWN, LEAN, ONE);
    static_assert(!NOGOALS || (LEAN && !WRAP), "the goal-image-free layout is the plain kernels'");
    unsigned char *const smem_hi = smem - (NOGOALS ? Gm::REGION - 16 : 0);
    const bool lwave = wave == (LEADX ? WAVES : 0);    // the leader wave ...
#ifndef SL_LEADER_PRIO
#define SL_LEADER_PRIO 0        /* A/B knob: s_setprio of the leader wave (it holds rows AND leads: the wave its workgroup waits for) */
#endif
    if (SL_LEADER_PRIO && __builtin_amdgcn_readfirstlane(tid >> 6) == (LEADX ? WAVES : 0)) __builtin_amdgcn_s_setprio(SL_LEADER_PRIO);
    const bool rwave = !(LEADX && lwave);              // waves that hold rows
    // SL_SPARSE_STORE (measured, off): single-step launches store only the boards the CA changed -- in a level of
    // still lifes 57 % of the boards of a step are untouched apart from the agent's own cells, which the leaders then
    // store themselves -- i.e. half the write traffic and half the dirty lines at the kernel boundary.  Bit-exact
    // (suite + soak), and no faster: 8.01-8.09 vs 7.91 us per two-slice C3 step in one session (the per-chunk test
    // and 14 more registers cost what the bytes save).
#ifndef SL_SPARSE_STORE
#define SL_SPARSE_STORE 0
#endif
    constexpr bool SPARSE_STORE = SL_SPARSE_STORE && ONE && use_planes<H, W>() && !Gm::SWZ;
    const bool lead = lwave && lane < nbb;             // ... whose lane q is the leader of board q
    const int lq = lead ? lane : 0;
    const unsigned e = e0b + (rowl ? gb : 0);          // the row lane's env
    const unsigned el = e0b + lq;                      // the leader lane's env
    unsigned char *board = smem + Gm::OFF_BOARD, *goals = smem + Gm::OFF_GOALS;
    u16 *board16 = (u16 *)(board + Gm::PAD) + (live ? gb : 0) * HW;
    u16 *lboard16 = (u16 *)(board + Gm::PAD) + lq * HW;
    u64 *rng_lds = (u64 *)(smem_hi + Gm::OFF_RNG) + 4 * Gm::G * wave;
    u64 *lrng = (u64 *)(smem_hi + Gm::OFF_RNG) + 4 * lq;
    BoardBox *box = (BoardBox *)(smem_hi + Gm::OFF_BOX);
    // goal colours of the lane's row, pre-shifted for the score index: in registers where the
    // budget allows (spawner-free variants; 64-wide boards run 2 waves/SIMD), else in LDS
    // (and the LEAN single-step instantiation of the spawner variant where it fits: 114 -> 125 VGPRs at 25x25, C4's
    //  share 9.6 -> 9.25 us per two-slice step on top of the even deal of the draws; the non-LEAN one and 26x26 tip
    //  into 8 bytes of scratch with it and keep the words in LDS)
    constexpr bool GSH_REG = gsh_in_registers<H, W>(SPAWN, LEAN, ONE);
    constexpr bool SHRINK = lean_lds<H, W>(SPAWN, LEAN, ONE);
    constexpr int OFF_LUT_V = SHRINK ? Gm::OFF_GSH : Gm::OFF_LUT, OFF_MOVE_V = SHRINK ? Gm::OFF_GSH + 4096 : Gm::OFF_MOVE;
    // (round 6) the kernels that keep the goal-word cache hold the words as the cache does, five to a dword (goal_unpack):
    // what lives from the prologue's loads to the score is 3 registers at 25x25 and 7 at 64x64 instead of 13 and 32
    constexpr bool GPK = LEAN && GSH_REG && ONE && !LEADX;
    u32 gsh_reg[GSH_REG ? (GPK ? (WS + 4) / 5 : WS) : 1];
    u32 *gsh_lane = GSH_REG ? gsh_reg : (u32 *)(smem_hi + Gm::OFF_GSH) + (wave * 64 + lane) * WS;
    auto set_goal_words = [&](const RowWords<H, W> &g) {        // g: the goal cells of the lane's row
        if constexpr (GPK) {
            goal_pack_row<H, W>(g, gsh_reg);
        } else {
#pragma unroll
            for (int k = 0; k < WS; ++k) gsh_lane[k] = goal_shift(g[k]);
        }
    };
    // the goal-word cache (GoalCache above): LEAN kernels whose goal words live in registers
    // (single-step launches: the T-step instantiations sit at their register limit -- a T-step launch of a batch that
    //  has a cache lowers every flag first, launch_rollout_t)
    constexpr bool GCACHE = LEAN && GSH_REG && ONE && !LEADX;
    static_assert(GCACHE == NOGOALS, "the kernels that keep the cache are the ones without a goal image");
    static_assert(GCACHE == GPK, "the kernels that keep the cache hold the words in its packed form");
    using Gc = GoalCache<H, W>;
    u32 *const gc_block = GCACHE && hot_gcache ? hot_gcache + (size_t)((unsigned)hot_first / Gm::NB + blockIdx.x) * Gc::BLOCK_DWORDS
                                               : nullptr;
    u32 *const gc_flag = gc_block;
    // (wave-uniform) this launch runs without the goal span: every board of the workgroup has static goals and its words
    // are in the cache.  Step launches only: a reset launch rewrites goals, an observation launch reads the image.
    // The flag is FETCHED here, first thing, and LOOKED AT where the goal span would be issued (gc_look below) -- behind
    // the record loads and the board's DMA instructions, so that its round trip runs under them.
    u32 gc_word = 0;
#ifndef SL_GOALS_FIRST
#define SL_GOALS_FIRST 1        /* A/B knob: 0 = the goal words are asked for behind the flag's round trip (round 5) */
#endif
    // (both knobs: the narrow shapes only -- at 64x64 either costs C5's step 1 %, 24.2 against 23.9 us)
    constexpr bool GOALS_FIRST = SL_GOALS_FIRST && Gm::WAVES_PER_SIMD == 4;
    if (GCACHE && gc_flag && T > 0) {
        gc_word = *(const u32 *)gc_flag;
        // (round 6) ... and the lane's goal words with it, BEFORE the flag says whether they are any good (a block has
        // words for all 64 lanes of its four waves; a launch that finds the flag lowered overwrites them below): with
        // the leaders' move out of the way (MOVE_LDS) the load barrier waits for the loading waves' youngest loads, and
        // behind the flag's round trip these were the youngest (own loads landed 1.05 -> 0.85 us after the wave's start)
        if (GOALS_FIRST) {
#pragma unroll
            for (int c = 0; c < Gc::X4; ++c) {
                const u32x4 v = *Gc::x4(gc_block, wave, lane, c);
                gsh_reg[4 * c + 0] = v.x;
                gsh_reg[4 * c + 1] = v.y;
                gsh_reg[4 * c + 2] = v.z;
                gsh_reg[4 * c + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < Gc::TAIL; ++j) gsh_reg[4 * Gc::X4 + j] = *Gc::tail(gc_block, wave, lane, j);
        }
    }
    bool goals_free = false;
    auto gc_look = [&]() {      // (through a VGPR: an SGPR constraint here has tripped "illegal VGPR to SGPR copy" in the backend)
        if constexpr (GCACHE) {
            u32 seen = gc_word;
            asm volatile("" : "+v"(seen));
            goals_free = __builtin_amdgcn_readfirstlane(seen) == 1u;
        }
    };
    const int8_t *lds_lut = (const int8_t *)(smem_hi + OFF_LUT_V);
    // wrappers: this wave's baseline rows, word k of lane l at [k * 64 + l] (the layout the b32 DMA writes)
    constexpr bool BASE_IN_GSH = GSH_REG && Gm::GSH_BYTES >= WAVES * 64 * WS * 4;
    unsigned char *base_rows = smem_hi + (BASE_IN_GSH ? Gm::OFF_GSH : Gm::OFF_BASE) + wave * 64 * WS * 4;
    // SimpleSideEffectPenalty's "inaction" baseline (env_wrappers.py:179-180), folded into this kernel in round 4: the
    // baseline boards of the workgroup's envs in a third LDS image, advanced by one more pass of the CA loop below
    const bool inaction = WRAP && (env.wrap.flags & SL_WRAP_INACTION) != 0;
    constexpr int OFF_INB = BASE_IN_GSH ? Gm::LDS_WRAP_GSHREG : Gm::LDS_WRAP_GSHLDS;
    unsigned char *inb = smem_hi + OFF_INB;
    u64 *irng_lds = (u64 *)(smem_hi + OFF_INB + Gm::REGION) + 4 * Gm::G * wave;
    sl_wrap_state *wst = (sl_wrap_state *)(smem_hi + Gm::OFF_WST);
    const double *mvt = (const double *)(smem_hi + Gm::OFF_MVT);
    const int8_t *__restrict__ lut = env.score_lut + 4096;        // wide form of table t at + t * SCORE_LUT_BYTES
    const Consts cst = make_consts();
    const pl::PConsts pcst = pl::make_pconsts();
    const u32 cell_mask = vreg(LDS_LUT ? (SCORE_CELL_MASK & 0x7FFF7FFFu) : SCORE_CELL_MASK), c100 = vreg(0x01000100u);

    SL_STAMP(0);
    // Prologue.  The kernel arguments the loads need are fetched in one batch (the compiler otherwise sinks each
    // s_load next to its first use: dependent scalar-cache round trips in a row).
    const u16 *k_board = hot_board, *k_goals = hot_goals;
    const sl_pcg64 *k_rng = hot_rng;
    const int8_t *k_lut = hot_lut;                      // (not preloaded: its DMA goes last, behind the look at the flag)
    asm volatile("" ::"s"(k_board), "s"(k_goals), "s"(k_rng));
    // what the rows need of their board's record (every lane loads, unconditionally: a load inside a branch is
    // waited for at the end of that branch, in front of the DMA issue)
    const sl_env_scalars *const sc = hot_scalars + e;
    int gstatic = sc->goals_static, level = sc->level_idx;
    double p = (double)sc->spawn_prob;
    u32 lut_base = (u32)sc->table_idx * (u32)SCORE_LUT_BYTES;
    // The leader lanes' view of their boards: the env records are copied to LDS by the DMA and worked on there (a
    // dozen per-board values held in registers for the whole launch cost every wave of the kernel those registers);
    // only the agent's location, which the move of step 0 needs before the copies have landed, is loaded directly.
    static_assert(sizeof(sl_env_scalars) == 64, "record stride");
    sl_env_scalars *const lrec = (sl_env_scalars *)(smem_hi + Gm::OFF_REC) + lq;
    int ly = -1, lx = 0, exit0 = -1, action = 0;
    bool pool_exits = false;                            // the board's exit table: its own row of env.exit_locs, or
                                                        // (after a reset in this launch) the pool level's
    int pre_i[4] = {0, 0, 0, 0}, pre_y1 = 0, pre_x1 = 0;   // the move of step 0, on cells taken from global memory
    u32 pre_c[4] = {0u, 0u, 0u, 0u};
    bool pre_write = false;
#ifndef SL_ARGS_EARLY
#define SL_ARGS_EARLY 1         /* A/B knob: 0 = the leader wave fetches the kernel arguments where the loading waves do (round 5) */
#endif
    // Every kernel argument the rest of the kernel needs that did not arrive preloaded: fetched in ONE batch, in the
    // shadow of the bulk loads -- by the loading waves behind their DMA instructions, by the leader wave (round 6) behind
    // the loads of its first round trip: behind the look at the flag it stood, 0.15 us, on the leaders' way to the barrier.
    constexpr bool ARGS_EARLY = SL_ARGS_EARLY && Gm::WAVES_PER_SIMD == 4;
    auto fetch_args = [&]() {
        const int a0 = env.time_limit, a1 = env.exit_points, a2 = env.auto_reset, a3 = env.L, a4 = env.level_stride;
        const int a5 = env.B, a6 = env.stream_salt, a7 = env.out_compact;
        const void *p0 = out_rec, *p1 = env.pool_board, *p2 = env.pool_goals, *p3 = env.pool_exit_locs;
        const void *p4 = env.pool_rng, *p5 = env.pool_scalars, *p6 = env.exit_locs;
        asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(T), "s"(p0), "s"(p1), "s"(p2), "s"(p3),
                     "s"(p4), "s"(p5), "s"(p6), "s"(reward_t), "s"(done_t), "s"(xcd_base), "s"(xcd_flag), "s"(a5), "s"(a6), "s"(a7));
    };
#ifndef SL_MOVE_BOX
#define SL_MOVE_BOX 1           /* A/B knob: 0 = the round-3 form of the move (leader writes the image, a barrier of its own) */
#endif
#ifndef SL_MOVE_LDS
#define SL_MOVE_LDS 1           /* A/B knob: 0 = the move's cells come from global memory, in front of the load barrier (round 4) */
#endif
    // (round 6) single-step launches: the leaders take the four cells of their move from the IMAGE, right behind the load
    // barrier, and the rows wait at a second barrier for the move to be in it.  Fetched from global memory ahead of the
    // barrier (round 4) the cells are a second round trip behind the record's -- 0.80 -> 1.4 us after the wave's start,
    // while the loading waves have the whole board in LDS at 1.05 (profiles/round6_t4_leader_path_trace.txt).
    // (not the wrappers' kernels: the "inaction" baseline copies the board as it stands BEFORE the move, inside the step
    //  loop; not the wide shapes: at 64x64 the second barrier costs more than the round trip, 23.7 against 23.45 us per C5 step)
    constexpr bool MOVE_LDS = SL_MOVE_LDS && SL_MOVE_BOX && ONE && !WRAP && Gm::WAVES_PER_SIMD == 4;
    if (lwave) {
        ly = hot_scalars[el].agent_row;
        lx = hot_scalars[el].agent_col;
        action = actions[el];
        exit0 = env.exit_locs[(size_t)el * E];
        if (ARGS_EARLY) fetch_args();
        if (MOVE_LDS && T > 0 && lead && ly >= 0) {
            int gi[4];
            act_cells<H, W>(ly, lx, action, pre_i, gi, pre_y1, pre_x1);
        }
        if (!MOVE_LDS && T > 0 && lead && ly >= 0) {
            // safelife_env.py:151 for the first step of the launch: the four cells the move can touch are
            // fetched from global memory now (the previous launch's board); they are not waited for here --
            // the leader wave goes through the load barrier and its own goal rows first
            int gi[4];
            act_cells<H, W>(ly, lx, action, pre_i, gi, pre_y1, pre_x1);
            const u16 *src = k_board + (size_t)el * HW;
            pre_c[0] = src[gi[0]];
            pre_c[1] = src[gi[1]];
            pre_c[2] = src[gi[2]];
            pre_c[3] = src[gi[3]];
        }
        SL_STAMP(14);       // (trace builds: the leaders' first round trip is back, the move's cells are asked for)
        gc_look();
        SL_STAMP(13);       // (trace builds, leader wave: the goal-word flag is in)
    } else {
        // everything bulky goes through the LDS DMA, issued by the waves that are not the leader
        constexpr int DW = LEADX ? WAVES : WAVES - 1;
        const int dw = LEADX ? wave : wave - 1;
        dma_to_lds<Gm::NB * 32, false, DW>((const unsigned char *)(k_rng + e0b), smem_hi + Gm::OFF_RNG, nbb * 32, lane, dw);
        dma_to_lds<Gm::NB * 64, false, DW>((const unsigned char *)(hot_scalars + e0b), smem_hi + Gm::OFF_REC, nbb * 64, lane, dw);
        // (the score table's pointer is not preloaded: where the flag is looked at anyway, its DMA goes behind that wait)
        if (LDS_LUT && !GCACHE) dma_to_lds<4096, false, DW>((const unsigned char *)k_lut, smem_hi + OFF_LUT_V, 4096, lane, dw);
        load_span<H, W, DW>(k_board + (size_t)e0b * HW, board, nbb, lane, dw);
        gc_look();
        if (!NOGOALS) load_span<H, W, DW>(k_goals + (size_t)e0b * HW, goals, nbb, lane, dw);
        if (LDS_LUT && GCACHE) dma_to_lds<4096, false, DW>((const unsigned char *)k_lut, smem_hi + OFF_LUT_V, 4096, lane, dw);
        if (WRAP) {
            dma_to_lds<Gm::NB * (int)sizeof(sl_wrap_state), false, DW>((const unsigned char *)(env.wrap.state + e0b),
                                                                       smem_hi + Gm::OFF_WST,
                                                                       nbb * (int)sizeof(sl_wrap_state), lane, dw);
            if (env.wrap.flags & SL_WRAP_MOVEMENT)
                dma_to_lds<Gm::MVT_N * 8, false, DW>((const unsigned char *)env.wrap.move_table, smem_hi + Gm::OFF_MVT,
                                                     min(env.wrap.move_table_len & ~1, Gm::MVT_N) * 8, lane, dw);
            if (inaction) {
                load_span<H, W, DW>(env.wrap.inaction_board + (size_t)e0b * HW, inb, nbb, lane, dw);
                dma_to_lds<Gm::NB * 32, false, DW>((const unsigned char *)(env.wrap.inaction_rng + e0b),
                                                   smem_hi + OFF_INB + Gm::REGION, nbb * 32, lane, dw);
            }
        }
    }
    if (!GOALS_FIRST && GCACHE && goals_free && live) {
        // the lane's goal words, as an earlier launch left them (GoalCache): straight into the registers the score reads
#pragma unroll
        for (int c = 0; c < Gc::X4; ++c) {
            const u32x4 v = *Gc::x4(gc_block, wave, lane, c);
            gsh_reg[4 * c + 0] = v.x;
            gsh_reg[4 * c + 1] = v.y;
            gsh_reg[4 * c + 2] = v.z;
            gsh_reg[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < Gc::TAIL; ++j) gsh_reg[4 * Gc::X4 + j] = *Gc::tail(gc_block, wave, lane, j);
    }
    constexpr bool MOVE_BOX = !MOVE_LDS && SL_MOVE_BOX && ONE && !(SPAWN && GSH_REG && Gm::WAVES_PER_SIMD == 4);
    // (Round 4, measured and dropped -- as round 3's variant of it was: every wave sending its OWN boards to global
    //  memory right behind its CA pass, under the score phase and the leaders' work, the leaders storing the agent's
    //  and the exits' cells themselves behind the end barrier.  Same-box A/B, K = 400: 6.47-6.52 us per step without,
    //  6.70-6.75 with it through release-free queues; 7.61 / 8.09 with agent fences; 8.34 / 8.79 through streams.)
    // (Round 4, measured and dropped: every lane fetching its OWN goal row from global memory at the start -- thirteen
    //  2-byte aligned dwords, permuted into the split layout behind the load barrier -- instead of the pass over the
    //  LDS image, the "goal rows" phase of the trace.  Same-box A/B, K = 400: 6.46 us per step without it, 6.60-6.65
    //  with it through the queues, 8.03 against 8.72 through stream slices: the extra vector-memory instructions in
    //  front of the DMA cost more than the LDS pass they replace.)
    if (!lwave || !ARGS_EARLY) fetch_args();
    if (lwave) SL_STAMP(15);    // (trace builds, leader wave: goal words asked for, the kernel arguments are in)
    typedef __attribute__((address_space(3))) int *lds_int;       // (a generic volatile pointer would go through FLAT)
    lds_int dirty_flag = (lds_int)(smem + Gm::OFF_GOALS);              // in the region's leading pad
    if (tid == 0) *dirty_flag = 0;
    // (the finished-episode queue is also compiled into the LEAN kernels of the wide shapes -- two waves per SIMD, registers
    //  to spare: C5's envs then step on the LEAN kernel, goal-word cache and all, with half the LDS per workgroup, which is
    //  what lets the episode-end pass run beside them)
    constexpr bool QUEUE_OK = !LEAN || Gm::WAVES_PER_SIMD < 4;
    const bool has_queue = QUEUE_OK && env.finished.capacity > 0;
    const bool hand_over = env.auto_reset || has_queue;           // (uniform) leaders may ask the rows for something
    // Round 4, single-step launches: the leaders decide the move HERE, in front of the load barrier (their four cells
    // came from global memory, two dependent round trips that run under the bulk loads), and leave what is to be
    // written -- (image index, cell) x 4 -- in LDS.  The board's own wave puts it into the image right behind the
    // barrier, in front of its row reads (LDS operations of one wave stay in order), so no workgroup barrier stands
    // between the loads and the CA any more, and the leaders' serial section is off the rows' path.
    typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t *const move_box = (u32x4_t *)(smem_hi + OFF_MOVE_V);
    if (MOVE_BOX && lwave) {
        u32x4_t mv = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
        if (lead && ly >= 0) pre_write = act_rule(pre_c, action, ly, lx, pre_y1, pre_x1);
        if (pre_write)
            mv = u32x4_t{(u32)pre_i[0] | (pre_c[0] << 16), (u32)pre_i[1] | (pre_c[1] << 16), (u32)pre_i[2] | (pre_c[2] << 16),
                         (u32)pre_i[3] | (pre_c[3] << 16)};
        if (lead) move_box[lq] = mv;
    }
    SL_STAMP(1);
    if (SL_PRIO_PROLOGUE) __builtin_amdgcn_s_setprio(0);
    // the load barrier: the DMA waves wait for their loads, the leader wave only for its LDS stores (it moved none
    // of the spans, and what it has in flight is its own business)
    if (lwave) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SL_TRACE
    else {              // (trace builds: when this wave's own loads have landed, apart from when the barrier lets it go)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        SL_STAMP(13);       // (slots 13 / 14 are the reload block's as well: it overwrites them in the workgroups that reload)
        __syncthreads();
    }
#else
    else __syncthreads();
#endif
    SL_STAMP(2);
    // Release-free queue stepping: the state this workgroup has just loaded was left in the L2 of the XCD that
    // workgroup i of this slice's queue always runs on.  Scalar code, no memory access unless it fails, and behind the
    // bulk loads: the two arguments arrive with the batch above (a returning atomic on a per-workgroup record, looked
    // at here, cost 0.25 us per step -- the record sits behind the fabric while the state it guards sits in the L2; and
    // the same comparison in FRONT of the loads cost a scalar-cache round trip before the first DMA instruction).
    if (xcd_flag && wave == 1 && (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u) != (((u32)xcd_base + blockIdx.x) & 7u) &&
        lane == 0)
        __hip_atomic_store(xcd_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    if (MOVE_LDS && T > 0) {
        if (lwave && lead && ly >= 0) {
            pre_c[0] = lboard16[pre_i[0]];
            pre_c[1] = lboard16[pre_i[1]];
            pre_c[2] = lboard16[pre_i[2]];
            pre_c[3] = lboard16[pre_i[3]];
            pre_write = act_rule(pre_c, action, ly, lx, pre_y1, pre_x1);
            if (pre_write) {
                lboard16[pre_i[0]] = (u16)pre_c[0];
                lboard16[pre_i[1]] = (u16)pre_c[1];
                lboard16[pre_i[2]] = (u16)pre_c[2];
                lboard16[pre_i[3]] = (u16)pre_c[3];
            }
        }
        // (LDS only: nobody's vector loads -- the goal words are still on their way -- are waited for here)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    RowWords<H, W> b;
    Elig elig;
    // (lanes without a row compute on whatever their registers hold -- nothing of theirs is ever stored or summed;
    //  "defined, value irrelevant" costs no instruction, thirteen zeroing moves would)
#pragma unroll
    for (int k = 0; k < WS; ++k) asm volatile("" : "=v"(b[k]));
    if (GCACHE && goals_free) gstatic = 1;           // (what the flag vouches for; the goal image does not exist in this launch)
    if (rwave) {
    if (live && !goals_free) {
        if constexpr (NOGOALS) read_row_global<H, W>(k_goals + (size_t)e * HW + r * W, b);
        else read_row<H, W>(goals, gb, r, b);
        set_goal_words(b);
    }
    // Goals still undecided (the first step after a reset; the reference finds out by advancing them once,
    // safelife_game.py:753-760): a goal array without a single ALIVE or SPAWNING cell cannot change and draws
    // nothing, so it IS static and the second CA pass of that step is skipped -- the usual case for every level a
    // reset loads.  Decided here, once per launch, for the step this launch is about to take (an env that
    // resets in the middle of a T-step launch takes the two-pass route for one step).
    if (T > 0 && __ballot(rowl && gstatic == 0)) {
        u32 goal_bits = 0;              // (b still holds the goal row; halves beyond an odd width are zero)
        if (live) {
#pragma unroll
            for (int k = 0; k < WS; ++k) goal_bits |= b[k];
        }
        const int restless = group_total<H, W>(live && (goal_bits & 0x00810081u) ? 1 : 0, rowl ? g : 0);
        if (rowl && gstatic == 0 && restless == 0) gstatic = 1;
    }
    if (rlead) box[gb].gstat = gstatic;
    }
    SL_STAMP(3);

    // What the leaders ask of the rows at the end of a step -- queue the finished episode's board, load the next
    // level -- is carried out behind the NEXT workgroup barrier: the one at the top of the following step, or the
    // one in front of the final stores.
    bool any_reset = false;             // (wave-uniform) some board of the workgroup loaded a level in this launch
    auto hand_over_block = [&]() {
        const int any = box[0].any;
        if (any & 1) any_reset = true;
        if ((!LEAN || Gm::WAVES_PER_SIMD < 4) && (any & 2)) {
            // the step that ended an episode queued it for the side-effect pass (include/safelife_hip.h): the board
            // as the agent left it, from the LDS image, before any reset reloads the slot
            const int slot = rowl ? box[gb].qslot : -1;
            if (live && slot >= 0) {
                u16 *dst = env.finished.boards + (size_t)slot * HW + r * W;
#pragma unroll 1
                for (int x = 0; x < W; ++x) dst[x] = board16[Gm::cell(r, x)];      // (rare: kept out of the register budget)
            }
        }
        if (!(any & 1)) return;
        SL_STAMP(13);       // (trace builds: the reload block's own phases -- entered / rows fetched and placed / leaders done)
        // on-device auto-reset (training/base_algo.py:231-236 calls env.reset() after a done step)
        // (the leaders' own fetches -- the level's constants and its first exit -- go out FIRST, beside the rows' below,
        //  not behind them: one memory round trip less on the chain a reloading workgroup holds its launch up with)
        const bool ready_pool = env.pool_ready != nullptr;          // (uniform)
        const bool l_reset = lead && box[lq].reset_level >= 0;
        sl_level_scalars lv_pre = {};
        int exit0_pre = -1;
        if (l_reset) {
            lv_pre = env.pool_scalars[box[lq].reset_level];
            exit0_pre = env.pool_exit_locs[(size_t)box[lq].reset_level * E];
        }
        const int new_level = rowl ? box[gb].reset_level : -1;
        const bool mine = live && new_level >= 0;
        if (rowl && new_level >= 0) gstatic = 0;
        if (mine) {
            // (opaque copy of the row index: addresses of this rare block are formed here, not hoisted above the
            //  step loop as loop invariants and spilled)
            int r2 = r;
            asm volatile("" : "+v"(r2));
            level = new_level;
            // (round 6: where the batch keeps sl_env_batch.pool_ready the level comes as an episode starts on it --
            //  exits painted, its value in pool_scalars[l].ready -- and is neither scored nor repainted below)
            const u16 *pb = (ready_pool ? env.pool_ready : env.pool_board) + (size_t)level * HW, *pg = env.pool_goals + (size_t)level * HW;
            u16 *gdst = (u16 *)(goals + Gm::PAD) + gb * HW;
            // each lane copies its OWN row of the new level: W cells = one contiguous run, fetched as 4-byte
            // pairs (2-byte aligned: the hardware splits what it must) all issued before the first is used --
            // one memory round trip for the board and one for the goals, not one per cell
            typedef u32 u32_a2 __attribute__((aligned(2)));
            const u16 *rows[2] = {pb + r2 * W, pg + r2 * W};
            u16 *imgs[2] = {board16, gdst};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                u32 tw[WS];
#pragma unroll
                for (int j = 0; j < W / 2; ++j) tw[j] = *(const u32_a2 *)(rows[a] + 2 * j);
                if (Gm::ODD) tw[WS - 1] = rows[a][W - 1];
                if (NOGOALS && a == 1) {
                    // no goal image: the row goes to the env's goal array as it came, and into the registers the score reads
                    u16 *grow = env.goals + (size_t)e * HW + r2 * W;
#pragma unroll
                    for (int j = 0; j < W / 2; ++j) *(u32_a2 *)(grow + 2 * j) = tw[j];
                    if (Gm::ODD) grow[W - 1] = (u16)tw[WS - 1];
                    RowWords<H, W> gw;
                    words_from_pairs<H, W>(tw, gw);
                    set_goal_words(gw);
                    continue;
                }
#pragma unroll
                for (int j = 0; j < W / 2; ++j) {
                    imgs[a][Gm::cell(r, 2 * j)] = (u16)tw[j];
                    imgs[a][Gm::cell(r, 2 * j + 1)] = (u16)(tw[j] >> 16);
                }
                if (Gm::ODD) imgs[a][Gm::cell(r, W - 1)] = (u16)tw[WS - 1];
            }
            lut_base = (u32)env.pool_scalars[level].table_idx * (u32)SCORE_LUT_BYTES;
            p = (double)env.pool_scalars[level].spawn_prob;
            if (r2 < 4) rng_lds[4 * g + r2] = ((const u64 *)(env.pool_rng + level))[r2];
            for (int k = r2; k < E; k += H)
                env.exit_locs[(size_t)e * E + k] = env.pool_exit_locs[(size_t)level * E + k];
            if (!NOGOALS) *dirty_flag = 1;               // (any wave that changes its goals raises the flag)
            if (r2 == 0) box[gb].dirty = 1;              // (and the whole board is new)
        }
        wave_sync();
        if (mine) {
            if constexpr (!NOGOALS) {
                read_row<H, W>(goals, gb, r, b);
#pragma unroll
                for (int k = 0; k < WS; ++k) gsh_lane[k] = goal_shift(b[k]);
            }
            if (!ready_pool) read_row<H, W>(board, gb, r, b);
        }
        if (!ready_pool) {
            const int s0 = group_total<H, W>(
                mine ? row_score<H, W, LDS_LUT, GPK>(b, gsh_lane, lut, lut_base, lds_lut, cell_mask, c100) : 0, live ? g : 0);
            if (rlead) box[gb].score0 = s0;
        }
        if (rlead) box[gb].gstat = gstatic;
        SL_STAMP(14);
        wg_sync();
        SL_STAMP(15);
        if (lead && box[lq].reset_level >= 0) {
            const int l_level = box[lq].reset_level;
            const int episodes = lrec->episode_idx + 1;
            if (env.stream_salt) {          // the new episode's own stream (the rows wrote the level's state above)
                u64 hi = lrng[0], lo = lrng[1];
                sl_episode_stream(hi, lo, env.stream_salt + (int)el, episodes);
                lrng[0] = hi;
                lrng[1] = lo;
            }
            pool_exits = true;                                  // (the pool's table: never written by this launch)
            const int32_t *exits = env.pool_exit_locs + (size_t)l_level * E;
            exit0 = exit0_pre;
            const sl_level_scalars lv = lv_pre;
            ly = lv.agent_row;
            lx = lv.agent_col;
            int fresh, open0, exited;
            if (ready_pool) {           // the level came painted; its value at the start of an episode with it
                open0 = lv.ready & 1;
                fresh = lv.ready >> 1;
                exited = 0;
            } else {
                fresh = box[lq].score0;
                open0 = recolor_exits_lds<H, W>(lboard16, ly, lx, exits, exit0, E, fresh, lv.initial_points,
                                                lv.required_reset, env.exit_points) ? 1 : 0;
                exited = ly >= 0 ? (has_exited(lboard16[Gm::cell(ly, lx)]) ? 1 : 0) : 0;
            }
            if (WRAP) wrap_reset(wst[lq], ly, lx);
            sl_env_scalars rec;
            rec.agent_row = ly;
            rec.agent_col = lx;
            rec.num_steps = 0;
            rec.old_value = fresh + env.exit_points * exited;
            rec.required_points = lv.required_step;
            rec.initial_points = lv.initial_points;
            rec.table_idx = lv.table_idx;
            rec.level_idx = l_level;
            rec.episode_idx = episodes;
            rec.episode_length = 0;
            rec.episode_reward = 0.0f;
            rec.spawn_prob = lv.spawn_prob;
            rec.goals_static = 0;
            rec.is_active = 1;
            rec.exit_open_at_reset = open0;
            rec.loaded = 1;
            *lrec = rec;
        }
        wg_sync();
    };

    for (int t = 0;; ++t) {
        if (t > 0) {
            wg_sync();                             // the leaders' requests of the step before; after the last
            if (hand_over) hand_over_block();            // step also the barrier in front of the stores
        }
        if (t >= T) break;
        if (WRAP && (env.wrap.flags & SL_WRAP_SIDE_EFFECT) && !inaction) {
            // baseline row (level, r) of every lane -> LDS, asynchronously; read after the CA pass
            // ("inaction": the env's own baseline instead, advanced and laid out by this kernel's third CA pass)
            const u32 *src = env.wrap.pool_baseline + ((size_t)level * H + r) * WS;
#pragma unroll
            for (int k = 0; k < WS; ++k)
                __builtin_amdgcn_global_load_lds((glds_src_t)(src + k), (glds_dst_t)(base_rows + k * 256), 4, 0, 0);
        }
        if (WRAP && inaction) {
            // An env in the first step of an episode (num_steps == 0) takes its board as it stands now -- after the
            // reset, before this step's action -- as its baseline: what the wrapper's reset() copies.
            const bool fresh = rowl && ((const sl_env_scalars *)(smem_hi + Gm::OFF_REC))[gb].num_steps == 0;
            if (rwave && __ballot(fresh)) {
                if (fresh && live) {
                    read_row<H, W>(board, gb, r, b);
                    write_row<H, W>(inb, gb, r, b);
                }
            }
            if (!MOVE_BOX && !MOVE_LDS) wg_sync();           // (the leaders write the move into the image: behind the rows' reads)
        }
        if (MOVE_BOX && rwave && rlead) {
            // the move the board's leader decided: into the image, by the board's own wave, ahead of its row reads
            // (one 16-byte read: four dependent read -> write round trips in a loop cost 0.3 us on the step's chain)
            const u32x4_t mv = move_box[gb];
            if ((mv.x & 0xFFFFu) != 0xFFFFu) {
                board16[mv.x & 0xFFFFu] = (u16)(mv.x >> 16);
                board16[mv.y & 0xFFFFu] = (u16)(mv.y >> 16);
                board16[mv.z & 0xFFFFu] = (u16)(mv.z >> 16);
                board16[mv.w & 0xFFFFu] = (u16)(mv.w >> 16);
            }
        }
        // safelife_env.py:151
        if (!MOVE_BOX && !MOVE_LDS) {
            if (lwave) {
                if (t == 0) {
                    if (lead && ly >= 0) pre_write = act_rule(pre_c, action, ly, lx, pre_y1, pre_x1);
                    if (pre_write) {
                        lboard16[pre_i[0]] = (u16)pre_c[0];
                        lboard16[pre_i[1]] = (u16)pre_c[1];
                        lboard16[pre_i[2]] = (u16)pre_c[2];
                        lboard16[pre_i[3]] = (u16)pre_c[3];
                    }
                } else if (lead && ly >= 0) {
                    action = actions[(size_t)t * B + el];
                    act_gather<H, W>(lboard16, ly, lx, action);
                }
            }
            wg_sync();                             // the move is in the image
        }                                          // (single-step launches: it went in behind the load barrier)
        SL_STAMP(4);
        // safelife_env.py:152 : board first, then goals unless they are static (safelife_game.py:746-761)
        if (rwave) {
        const bool dyn = rowl && gstatic != 1;
        const int goal_pass = __ballot(dyn) ? 1 : -1;
        // (WRAP, "inaction": one more pass, over the baseline boards with the baselines' own generators)
        const int base_pass = (WRAP && inaction) ? (goal_pass > 0 ? 2 : 1) : -1;
        const int passes = 1 + (goal_pass > 0 ? 1 : 0) + (base_pass > 0 ? 1 : 0);
        bool board_dirty = !SPARSE_STORE;
#pragma nounroll
        for (int pass = 0; pass < passes; ++pass) {
            const bool base = WRAP && pass == base_pass;
            const bool has = rowl && (pass == 0 || base || dyn);     // row to advance (halo copies included)
            const bool mine = has && lm.real;
            unsigned char *img = pass == 0 ? board : (base ? inb : goals);
            u64 *const pass_rng = base ? irng_lds : rng_lds;
            u16 *const grow = NOGOALS ? env.goals + (size_t)e * HW + r * W : nullptr;      // the lane's goal row in global memory
            const bool global_row = NOGOALS && pass == goal_pass;
            if (has) {
                if (global_row) read_row_global<H, W>(grow, b);
                else read_row<H, W>(img, gb, r, b);
            }
            bool changed = true;                             // (wave-uniform) some cell of the wave's rows changed
            if constexpr (use_planes<H, W>()) {
                u32 row_changed = 0;
#ifndef SL_TIMING_SKIP
#define SL_TIMING_SKIP 0        /* TIMING-ONLY builds (wrong results; tools/exp/timing_only.sh): 1 = no CA, 2 = no row scores, 4 = no leader work */
#endif
                if (SL_TIMING_SKIP & 1) changed = false;
                else
                changed = ca_step<H, W, SPAWN, false>(b, mine, mine, up, dn, cst, pcst, pass_rng, live ? g : 0, p, jump,
                                                      &row_changed);
                if (SPARSE_STORE && pass == 0) {             // which of the wave's boards changed
                    const unsigned long long rows = __ballot(mine && row_changed != 0);
#pragma unroll
                    for (int q = 0; q < Gm::G; ++q) {
                        const unsigned long long of_q = ((Gm::GL == 64 ? ~0ull : ((1ull << Gm::GL) - 1ull)) << (q * Gm::GL % 64));
                        if (g == q) board_dirty = (rows & of_q) != 0;
                    }
                }
            } else {
                // (column-first reduction where its three arrays fit beside the row: not at 15-16 words with 128 registers)
                constexpr bool COLFIRST = LEAN && !SPAWN && (WS <= 13 || Gm::WAVES_PER_SIMD < 4);
                ca_rows<H, W, SPAWN, COLFIRST>(b, b, elig, up, dn, cst);   // in place: b now holds the new cells
                if (!mine) elig.clear();
                if (SPAWN && __ballot(elig.any())) {
                    RowWords<H, W> old;                          // failed draws keep the old cell: re-read it
#pragma unroll
                    for (int k = 0; k < WS; ++k) old[k] = 0;
                    if (mine) read_row<H, W>(img, gb, r, old);
                    resolve_draws<H, W>(old, b, elig, pass_rng, live ? g : 0, p, jump);
                }
            }
            if (base && mine) {
                // the advanced baseline row as the side-effect count reads it: player bits cleared, word k of lane l
                // at [k * 64 + l] (the layout the starting-state baseline's DMA writes)
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    ((u32 *)base_rows)[k * 64 + lane] =
                        b[k] & ~(PLAYER | (PLAYER << 16)) & ((Gm::ODD && k == WS - 1) ? 0xFFFFu : 0xFFFFFFFFu);
            }
            if (pass == goal_pass) {
                u32 diff = 0;
                if (mine) {
                    RowWords<H, W> old;
                    if (changed) {
                        if (global_row) read_row_global<H, W>(grow, old);
                        else read_row<H, W>(img, gb, r, old);
                    } else {
#pragma unroll
                        for (int k = 0; k < WS; ++k) old[k] = b[k];
                    }
#pragma unroll
                    for (int k = 0; k < WS; ++k)
                        diff |= ((b[k] ^ old[k]) | (b[k] & 0x00800080u)) & (Gm::vm1(k) * 0xFFFFu);
                }
                const int moved = group_total<H, W>(mine && diff ? 1 : 0, rowl ? g : 0);
                if (has && gstatic == 0) gstatic = moved ? 2 : 1;
                if (mine) {
                    set_goal_words(b);
                    if (!NOGOALS) *dirty_flag = 1;
                }
            }
            if (mine && changed) {
                if (global_row) write_row_global<H, W>(grow, b);
                else write_row<H, W>(img, gb, r, b);
            }
        }
        if (passes > 1) {                // board rows back into registers for scoring
            wave_sync();
            if (live) read_row<H, W>(board, gb, r, b);
        }
        SL_STAMP(5);
        // safelife_env.py:153-160
        const int score_rows = (SL_TIMING_SKIP & 2) ? (int)(b[0] & 1u) : group_total<H, W>(
            live ? row_score<H, W, LDS_LUT, GPK>(b, gsh_lane, lut, lut_base, lds_lut, cell_mask, c100) : 0, live ? g : 0);
        if (rlead) {
            box[gb].score = score_rows;
            box[gb].gstat = gstatic;
            box[gb].dirty = board_dirty ? 1 : 0;
        }
        }
        // the leaders: what does not depend on the scores is read before the barrier (in the LEADX kernels, under
        // the rows' CA pass)
        int b_initial = 0, b_required = 0, b_old_value = 0, b_steps = 0, b_ep_len = 0;
        float b_ep_rew = 0.0f;
        bool b_active = false;
        if (lead) {
            b_initial = lrec->initial_points;
            b_required = lrec->required_points;
            b_old_value = lrec->old_value;
            b_steps = lrec->num_steps + 1;
            b_ep_len = lrec->episode_length;
            b_ep_rew = lrec->episode_reward;
            b_active = lrec->is_active != 0;
        }
        wg_sync();                                 // scores in the mailbox, new boards in the images
        SL_STAMP(6);
        bool done = false;
        float w_reward = 0.0f, w_ep_rew = 0.0f;     // WRAP: this step's outputs, kept for the wrapper stage below
        bool w_times_up = false, w_open = false;
        int w_exits = 0;
        if (lwave) {
            if (lane == 0) box[0].any = 0;
            if (lead && !(SL_TIMING_SKIP & 4)) {
                const int score = box[lq].score;
                const int32_t *exits = pool_exits ? env.pool_exit_locs + (size_t)lrec->level_idx * E
                                                  : env.exit_locs + (size_t)el * E;
                const bool active = b_active;
                u32 cell = 0;
                w_open = recolor_exits_lds<H, W>(lboard16, ly, lx, exits, exit0, E, score, b_initial, b_required,
                                                 env.exit_points, WRAP ? &w_exits : nullptr, &cell);
                const int steps = b_steps;
                const bool times_up = steps >= env.time_limit;
                float reward = 0.0f;
                bool success = false;
                done = true;
                if (ly >= 0) {
                    success = has_exited(cell);
                    const int value = score + (success ? env.exit_points : 0);
                    reward = (float)((value - b_old_value) * (active ? 1 : 0));
                    lrec->old_value = value;
                    done = !(cell & AGENT) || times_up;
                }
                const float ep_rew = b_ep_rew + reward;
                const int ep_len = b_ep_len + (active ? 1 : 0);
                const bool ended = done && active;      // this step ends the episode
                lrec->agent_row = ly;
                lrec->agent_col = lx;
                lrec->num_steps = steps;
                lrec->episode_reward = ep_rew;
                lrec->episode_length = ep_len;
                lrec->is_active = active && !done ? 1 : 0;
                w_reward = reward;
                w_times_up = times_up;
                w_ep_rew = ep_rew;
                sl_step_out o;
                o.reward = reward;
                o.done = done;
                o.success = success;
                o.times_up = times_up;
                o.reserved = 0;
                o.episode_reward = ep_rew;
                o.episode_length = ep_len;
                unsigned e3 = el;
                asm volatile("" : "+v"(e3));
                if (env.out_compact)        // (the record's first half, assembled in registers: reward, then the three flag bytes)
                    ((unsigned long long *)out_rec)[e3] =
                        (unsigned long long)__float_as_uint(reward) |
                        ((unsigned long long)((done ? 1u : 0u) | ((success ? 1u : 0u) << 8) | ((times_up ? 1u : 0u) << 16)) << 32);
                else out_rec[e3] = o;
#ifndef SL_TRACE
                if (reward_t) reward_t[(size_t)t * B + el] = reward;
                if (done_t) done_t[(size_t)t * B + el] = done;
#endif
                if (SPARSE_STORE && env.auto_reset && done) {
                    box[lq].dirty = 1;      // a level is about to be loaded over this board: the span store takes it all
                } else if (SPARSE_STORE && !box[lq].dirty) {
                    // the rows changed nothing on this board: what the move and the repaint touched goes to global
                    // memory cell by cell (the values are in registers: with the CA idle the move's cells are as
                    // the move left them)
                    u16 *gdst = env.board + (size_t)el * HW;
                    if (pre_write) {
                        gdst[pre_i[0]] = (u16)pre_c[0];
                        gdst[pre_i[1]] = (u16)pre_c[1];
                        gdst[pre_i[2]] = (u16)pre_c[2];
                        gdst[pre_i[3]] = (u16)pre_c[3];
                    }
                    if (ly >= 0) gdst[Gm::cell(ly, lx)] = (u16)cell;
                    const u16 paint = (u16)(FROZEN | EXIT | (w_open ? COLOR_R : 0u));
                    if (exit0 >= 0) gdst[exit0] = paint;
                    for (int k = 1; k < E; ++k) {
                        const int ex = exits[k];
                        if (ex >= 0) gdst[ex] = paint;
                    }
                }
                if (hand_over) {
                    int slot = -1, next_level = -1, any = 0;
                    if (has_queue && ended) {
                        slot = atomicAdd(env.finished.count, 1);
                        if (slot < env.finished.capacity) {
                            // (the 32-byte record as two vector stores assembled in registers: a struct temporary
                            //  ends up in scratch memory here)
                            const u32 flags = (success ? 1u : 0u) | ((times_up ? 1u : 0u) << 8);
                            u32x4 *qdst = (u32x4 *)(env.finished.records + slot);
                            qdst[0] = u32x4{(u32)((int)el + env.finished.env_base), (u32)lrec->level_idx, (u32)steps,
                                            (u32)lrec->episode_idx};
                            qdst[1] = u32x4{__float_as_uint(lrec->spawn_prob), __float_as_uint(ep_rew), (u32)ep_len, flags};
                            any |= 2;
                        } else {
                            slot = -1;
                        }
                    }
                    if (env.auto_reset && done) {
                        next_level = env.pool_next ? env.pool_next[lrec->level_idx] : (lrec->level_idx + env.level_stride) % env.L;
                        any |= 1;
                    }
                    box[lq].qslot = slot;
                    box[lq].reset_level = next_level;
                    if (any) atomicOr(&box[0].any, any);
                }
            }
        }
        if (WRAP) {     // env_wrappers.py: movement bonus, exit bonus, side-effect penalty (float64)
            if (env.wrap.flags & SL_WRAP_SIDE_EFFECT) {
                wg_sync();                                 // the leaders' exit repaint is in the images
                int mine = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // baseline rows have landed
                if (live) {
                    read_row<H, W>(board, gb, r, b);
                    mine = row_side_effect<H, W>(b, (const u32 *)base_rows + lane, gsh_lane,
                                                 (env.wrap.flags & SL_WRAP_IGNORE_REWARD_CELLS) != 0,
                                                 vreg(~(PLAYER | (PLAYER << 16))), cst.m1);
                }
                const int side_rows = group_total<H, W>(mine, rowl ? g : 0);
                if (rlead) box[gb].side = side_rows;
                wg_sync();
            }
            if (lead) {
                int side = (env.wrap.flags & SL_WRAP_SIDE_EFFECT) ? box[lq].side : 0;
                // exit cells are ignored by the reference; here they all differ from the baseline (its
                // exits carry the reset's paint) or none does
                if (w_open != (lrec->exit_open_at_reset != 0)) side -= w_exits;
                const double shaped = wrap_step(env.wrap, wst[lq], mvt, w_reward, done, w_times_up, w_ep_rew, ly, lx, side);
                unsigned e4 = el;
                asm volatile("" : "+v"(e4));
                env.wrap.shaped_reward[e4] = shaped;
                if (shaped_t) shaped_t[(size_t)t * B + e4] = shaped;
            }
        }
    }

    if (T < 0) {
        // slhip_env_reset on the row kernels (safelife_env.py:203-218): the leaders ask for what the auto-reset asks for
        // at the end of an episode -- an env that holds a level moves on to its next one and counts an episode, one that
        // has never been loaded takes level_idx as it stands -- and the block that serves the auto-reset does the rest
        if (lwave) {
            if (lane == 0) box[0].any = 0;
            if (lead) {
                int next = -1;
                if (!reset_mask || reset_mask[el]) {
                    const bool was = lrec->loaded != 0;
                    next = !was ? lrec->level_idx
                                : env.pool_next ? env.pool_next[lrec->level_idx] : (lrec->level_idx + env.level_stride) % env.L;
                    if (!was) lrec->episode_idx -= 1;       // (the block below counts one)
                    atomicOr(&box[0].any, 1);
                }
                box[lq].reset_level = next;
                box[lq].qslot = -1;
            }
        }
        wg_sync();
        hand_over_block();
    } else if (T == 0) {
        wg_sync();
    }
    SL_STAMP(7);
    // write-back of the records: the leaders complete their LDS copies, the leader wave stores them
    if (lwave) {
        if (lead) {
            lrec->agent_row = ly;
            lrec->agent_col = lx;
            lrec->goals_static = box[lq].gstat;
            if (T >= 0) lrec->loaded = 1;           // (a reset launch marks the envs it loads, above)
        }
        wave_sync();
        int lane3 = lane;
        asm volatile("" : "+v"(lane3));
        for (int i = lane3; i < nbb * 4; i += 64)
            ((u32x4 *)(env.scalars + e0b))[i] = ((const u32x4 *)(smem_hi + Gm::OFF_REC))[i];
    }
    const int dirty = *dirty_flag;
    SL_STAMP(8);
    // (an opaque copy of the thread index: the stores' per-lane addresses are formed HERE -- left alone the compiler
    //  hoists them above the step loop, where a handful of live 64-bit pointers tips the variants that sit at the
    //  register limit into scratch; a kernel that uses scratch at all costs ~3 us more per launch)
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, wave2 = tid2 >> 6;
    if (rwave) {
        if constexpr (SPARSE_STORE) store_span_dirty<H, W>(env.board + (size_t)e0b * HW, board, nbb, tid2, box);
        else store_span<H, W>(env.board + (size_t)e0b * HW, board, nbb, tid2);
        if (dirty && !NOGOALS) store_span<H, W>(env.goals + (size_t)e0b * HW, goals, nbb, tid2);
    }
    if (GCACHE && gc_flag && T != 0) {
        // GoalCache: a launch that had the goal span keeps the lanes' words if every board of the workgroup ends it with
        // static goals (the words in the registers are current: a reset and an evolving goal both rewrite them); a
        // launch that ran on cached words lowers the flag if one of its boards took a new level.
        bool all_static = !any_reset;
        if (!goals_free) {
            all_static = true;
#pragma unroll
            for (int q = 0; q < Gm::NB; ++q)
                if (q < nbb && box[q].gstat != 1) all_static = false;
            if (all_static && rwave && live && (nbb == Gm::NB || e0b + nbb >= env.B)) {
#pragma unroll
                for (int c = 0; c < Gc::X4; ++c)
                    *Gc::x4(gc_block, wave2, lane2, c) = u32x4{gsh_reg[4 * c], gsh_reg[4 * c + 1], gsh_reg[4 * c + 2], gsh_reg[4 * c + 3]};
#pragma unroll
                for (int j = 0; j < Gc::TAIL; ++j) *Gc::tail(gc_block, wave2, lane2, j) = gsh_reg[4 * Gc::X4 + j];
            }
        }
        // (a workgroup at the ragged end of a SLICE holds fewer boards than the block has words for: it may use and
        //  lower the flag, never raise it)
        const bool gc_full = nbb == Gm::NB || e0b + nbb >= env.B;
        if (tid2 == 0) {
            if (!all_static) *gc_flag = 0u;
            else if (!goals_free && gc_full) *gc_flag = 1u;
        }
    }
    if (lane2 < 4 * Gm::G && wave2 * Gm::G + (lane2 >> 2) < nbb)
        ((u64 *)(env.rng + e0b + wave2 * Gm::G))[lane2] = ((const u64 *)(smem_hi + Gm::OFF_RNG) + 4 * Gm::G * wave2)[lane2];
    if (WRAP)       // (10-row boards: 24 per workgroup, more state words than threads)
        for (int i = tid2; i < nbb * (int)(sizeof(sl_wrap_state) / 4); i += 64 * WAVES)
            ((u32 *)(env.wrap.state + e0b))[i] = ((const u32 *)wst)[i];
    if (WRAP && inaction && T > 0) {
        if (rwave) store_span<H, W>(env.wrap.inaction_board + (size_t)e0b * HW, inb, nbb, tid2);
        if (lane2 < 4 * Gm::G && wave2 * Gm::G + (lane2 >> 2) < nbb)
            ((u64 *)(env.wrap.inaction_rng + e0b + wave2 * Gm::G))[lane2] =
                ((const u64 *)(smem_hi + OFF_INB + Gm::REGION) + 4 * Gm::G * wave2)[lane2];
    }

    SL_STAMP(9);
#ifdef SL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SL_STAMP(10);       // this wave's stores acknowledged
    if ((threadIdx.x & 63) == 0 && reward_t) {
        long long *tr = (long long *)reward_t + (blockIdx.x * WAVES + (threadIdx.x >> 6)) * 16;
        tr[11] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
        tr[12] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID
    }
#endif
    // observation (safelife_env.py:105-146) from the LDS images of the final state
    if (!LEAN && (env.obs || env.policy_obs)) {
        // per board: view centre and, per exit slot, the view cell it is painted on + its board cell
        int *par = (int *)(smem_hi + Gm::OFF_GSH);                 // goal words are dead by now
        if (lead) {
            int *pp = par + lq * OBS_PAR_INTS;
            const int32_t *obs_exits = pool_exits ? env.pool_exit_locs + (size_t)lrec->level_idx * E
                                                  : env.exit_locs + (size_t)el * E;
            const int y0 = ly >= 0 ? ly : 0, x0 = ly >= 0 ? lx : 0;
            const int vh = env.view_h, vw = env.view_w;
            pp[0] = pos_mod(y0 - vh / 2, H);
            pp[1] = pos_mod(x0 - vw / 2, W);
            for (int k = 0; k < OBS_MAX_EXITS; ++k) {
                int tv = -1, xk = -1;
                if (k < E) xk = k == 0 ? exit0 : obs_exits[k];
                if (xk >= 0) {      // helper_utils.py:64-74: offset wrapped into [-H/2, H/2), clipped to the view
                    const int iy = xk / W, ix = xk - iy * W;
                    int jy = pos_mod(iy - y0 + H / 2, H) - H / 2 + vh / 2;
                    int jx = pos_mod(ix - x0 + W / 2, W) - W / 2 + vw / 2;
                    jy = min(max(jy, 0), vh - 1);
                    jx = min(max(jx, 0), vw - 1);
                    tv = jy * vw + jx;
                }
                pp[2 + k] = tv;
                pp[2 + OBS_MAX_EXITS + k] = xk;
            }
        }
        wg_sync();
        if (env.obs) {
            if (env.n_channels == 15) write_obs_block<H, W, 15>(env, smem, e0b, nbb, tid);
            else if (env.n_channels == 19) write_obs_block<H, W, 19>(env, smem, e0b, nbb, tid);
            else write_obs_block<H, W, 0>(env, smem, e0b, nbb, tid);
        }
        if (env.policy_obs) write_policy_block<H, W>(env, smem, e0b, nbb, tid);
    }
}

#ifndef SL_ROWLANE_PART
// ---- score table ------------------------------------------------------------------------------------

__global__ void k_build_score_lut(const int32_t *__restrict__ points_table, int n_tables, int8_t *__restrict__ lut) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tables * SCORE_LUT_BYTES) return;
    const int t = i / SCORE_LUT_BYTES, j = i - t * SCORE_LUT_BYTES;
    int idx;                       // decoded to the wide index: pullable on bit 15
    bool valid;
    if (j < 4096) {                // compact form: pullable on bit 8
        idx = (j & 0x0EFF) | ((j & 0x100) << 7);
        idx &= ~0x100;
        valid = (j & ~0x0FFD) == 0;
    } else {
        idx = j - 4096;
        valid = (idx & ~0x8EFD) == 0;
    }
    int v = 0;
    if (valid) {
        const bool alive = idx & 1, pushable = idx & 4, destr = idx & 8, frozen = idx & 16, pullable = idx & 0x8000;
        const int gc = (idx >> 5) & 7, col = (idx >> 9) & 7;
        const bool excluded = frozen && !(pullable || pushable || destr);
        v = excluded ? 0 : points_table[t * 72 + gc * 9 + (alive ? col : 8)];
    }
    lut[i] = (int8_t)v;
}

// ---- side-effect baseline of every pool level, in the row kernels' register layout ------------------
// One thread per (level, row).  Cell values: sl_device.h baseline_cell().
__global__ void k_build_baseline(sl_env_batch env) {
    const int H = env.H, W = env.W, WS = (W + 1) / 2, E = env.E;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= env.L * H) return;
    const int l = i / H, r = i - l * H;
    const sl_level_scalars lv = env.pool_scalars[l];
    const u16 *pb = env.pool_board + (size_t)l * H * W;
    const int32_t *px = env.pool_exit_locs + (size_t)l * E;
    // can_exit() during reset: nothing earned yet (safelife_game.py:716-719)
    const bool open = lv.agent_row >= 0 && (pb[lv.agent_row * W + lv.agent_col] & AGENT) && 0 >= lv.required_reset;
    u32 *dst = env.wrap.pool_baseline + (size_t)i * WS;
    for (int k = 0; k < WS; ++k) {
        u32 word = 0;
        for (int half = 0; half < 2; ++half) {
            const int c = k + half * WS;
            if (c >= W) continue;
            const int idx = r * W + c;
            bool is_exit = false;
            for (int q = 0; q < E; ++q) is_exit |= px[q] == idx;
            word |= baseline_cell(pb[idx], is_exit, open) << (16 * half);
        }
        dst[k] = word;
    }
}

#endif  // !SL_ROWLANE_PART

template <int H, int W>
hipError_t launch_advance_t(const u16 *in, u16 *out, int B, const float *spawn_prob, int n_steps,
                                   const int32_t *n_each, const int32_t *n_valid, sl_pcg64 *rng, const Jump *jump,
                                   hipStream_t stream) {
    using Gm = Geom<H, W>;
    // batches that would leave CUs without a workgroup of the four-wave kernel: one wave per workgroup
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    if ((B + Gm::NB - 1) / Gm::NB < 2 * n_cu) {
        hipLaunchKernelGGL((k_advance_small<H, W>), dim3((B + Gm::G - 1) / Gm::G), dim3(64), 0, stream, in, out, B, spawn_prob,
                           n_steps, n_each, n_valid, rng, jump);
        return hipGetLastError();
    }
    auto fn = k_advance_rowlane<H, W>;
    hipError_t err = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, Gm::LDS_ADVANCE);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(fn, dim3((B + Gm::NB - 1) / Gm::NB), dim3(64 * WAVES), Gm::LDS_ADVANCE, stream, in, out, B,
                       spawn_prob, n_steps, n_each, n_valid, rng, jump);
    return hipGetLastError();
}

template <int H, int W>
hipError_t launch_occupancy_t(const u16 *in, int32_t *counts, size_t counts_stride, int B, const int32_t *n_valid,
                                     int valid_period, const int32_t *pre_steps, const float *spawn_prob, int n_steps,
                                     sl_pcg64 *rng, const Jump *jump, hipStream_t stream) {
    using Gm = Geom<H, W>;
    // four-slot counters only where the LDS bounds the wavefronts per CU (64-wide boards: +42 %); boards up to 32
    // wide are issue-bound and the slot lookup only costs them (measured 10.3 vs 9.3 ms on 8192 25x25 boards)
    constexpr bool COMPACT = OccGeom<H, W, 8>::CB == 8;
    constexpr int lds4 = OccGeom<H, W, 4>::LDS_BYTES, lds8 = OccGeom<H, W, 8>::LDS_BYTES;
    auto fn8 = k_occupancy_rowlane<H, W, 8>;
    hipError_t err = hipFuncSetAttribute((const void *)fn8, hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
    if (err != hipSuccess) return err;
    if (OccGeom<H, W, 8>::CB == 8) {      // the drained counters are added into the output
        if (counts_stride == (size_t)H * W * 8) {
            err = hipMemsetAsync(counts, 0, (size_t)B * H * W * 8 * sizeof(int32_t), stream);
        } else {
            err = hipMemset2DAsync(counts, counts_stride * sizeof(int32_t), 0, (size_t)H * W * 8 * sizeof(int32_t),
                                   (size_t)B, stream);
        }
        if (err != hipSuccess) return err;
    }
    // both instantiations cover the batch; a wavefront whose boards belong to the other one exits at once
    const int period = valid_period > 0 ? valid_period : B;
    const int runs = (B + period - 1) / period;
    const dim3 grid(runs * ((period + Gm::G - 1) / Gm::G));
    if constexpr (COMPACT) {
        auto fn4 = k_occupancy_rowlane<H, W, 4>;
        err = hipFuncSetAttribute((const void *)fn4, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(fn4, grid, dim3(64), lds4, stream, in, counts, counts_stride, B, n_valid, valid_period,
                           pre_steps, spawn_prob, n_steps, rng, jump);
    }
    hipLaunchKernelGGL(fn8, grid, dim3(64), lds8, stream, in, counts, counts_stride, B, n_valid, valid_period, pre_steps,
                       spawn_prob, n_steps, rng, jump);
    return hipGetLastError();
}

// The fused kernel's argument block, packed once per launch: the runtime copies ONE buffer instead of walking eighteen
// arguments (a step is two ~2.5 us launches through streams; the host must keep ahead of a ~8 us device step), and the
// library's own queues write it into their argument ring as it is.  Mirrors the kernel's parameter list (natural
// alignment == the kernel-argument layout).
struct RolloutArgs {
    const u16 *board, *goals;
    const sl_pcg64 *rng;
    sl_env_scalars *scalars;
    u32 *gcache;
    const int32_t *actions;
    int first, end;
    const int8_t *lut;
    sl_env_batch env;
    int E, tstride, T;
    sl_step_out *out;
    float *reward_t;
    uint8_t *done_t;
    double *shaped_t;
    const Jump *jump;
    int xcd_base;
    u32 *xcd_flag;
    const uint8_t *reset_mask;
};
static_assert(sizeof(RolloutArgs) <= sizeof(PreparedStep::args), "argument block of a prepared step");

// variant selection, LDS size and the module-level handle of the kernel that steps `env` (T steps per launch)
template <int H, int W>
hipError_t pick_rollout_t(const sl_env_batch &env, int T, void **kernel, hipFunction_t *f_out, unsigned *threads_out, int *lds_out,
                          bool *gcache_ok) {
    using Gm = Geom<H, W>;
    const bool lean = !env.wrap.flags && !env.obs && !env.policy_obs && (env.finished.capacity == 0 || Gm::WAVES_PER_SIMD < 4);
    const int variant = (env.n_tables == 1 ? 1 : 0) | (env.spawner_free ? 2 : 0) | (env.wrap.flags ? 4 : (lean ? 8 : 0));
    typedef void (*kernel_t)(const u16 *, const u16 *, const sl_pcg64 *, sl_env_scalars *, u32 *,
                             const int32_t *, int, int, const int8_t *, sl_env_batch, int, int, int, sl_step_out *, float *,
                             uint8_t *, double *, const Jump *, int, u32 *, const uint8_t *);
#define SL_VARIANTS(ONE)                                                                                               \
    k_env_rollout_rowlane<H, W, false, true, false, false, ONE>, k_env_rollout_rowlane<H, W, true, true, false, false, ONE>,   \
    k_env_rollout_rowlane<H, W, false, false, false, false, ONE>, k_env_rollout_rowlane<H, W, true, false, false, false, ONE>, \
    k_env_rollout_rowlane<H, W, false, true, true, false, ONE>, k_env_rollout_rowlane<H, W, true, true, true, false, ONE>,     \
    k_env_rollout_rowlane<H, W, false, false, true, false, ONE>, k_env_rollout_rowlane<H, W, true, false, true, false, ONE>,   \
    k_env_rollout_rowlane<H, W, false, true, false, true, ONE>, k_env_rollout_rowlane<H, W, true, true, false, true, ONE>,     \
    k_env_rollout_rowlane<H, W, false, false, false, true, ONE>, k_env_rollout_rowlane<H, W, true, false, false, true, ONE>
    static const kernel_t table[24] = {SL_VARIANTS(false), SL_VARIANTS(true)};
#undef SL_VARIANTS
    const int slot = variant + (T == 1 ? 12 : 0);
    *threads_out = 64 * (WAVES + ((variant & 8) && Gm::LEADX_OK ? 1 : 0));     // LEAN: a fifth, leader wave
    const kernel_t fn = table[slot];
    const bool spawn = !(variant & 2), base_in_gsh = !spawn && Gm::WAVES_PER_SIMD == 4;
    // (the kernel's GCACHE: LEAN instantiations whose goal words live in registers)
    *gcache_ok = (variant & 8) && T == 1 && gsh_in_registers<H, W>(spawn, true, true) && !Gm::LEADX_OK;
    const int lds_wrap = base_in_gsh ? Gm::LDS_WRAP_GSHREG : Gm::LDS_WRAP_GSHLDS;
    // (the plain single-step kernels keep no goal image: nogoals_lds)
    const bool nogoals = (variant & 8) && T == 1 && nogoals_lds<H, W>(spawn, true, true);
    const int lds_plain = (lean_lds<H, W>(spawn, (variant & 8) != 0, T == 1) ? lean_lds_bytes<H, W>() : Gm::LDS_BYTES) -
                          (nogoals ? nogoals_shift<H, W>() : 0);
    const int lds = !(variant & 4) ? lds_plain : lds_wrap + ((env.wrap.flags & SL_WRAP_INACTION) ? Gm::INACTION_BYTES : 0);
    const int lds_limit = !(variant & 4) ? Gm::LDS_BYTES : lds_wrap + Gm::INACTION_BYTES;     // (set once per variant)
    // per (device, variant), once: raise the dynamic LDS limit and look up the module-level handle of the kernel
    struct Entry {
        std::atomic<hipFunction_t> fn{nullptr};
        std::atomic<bool> ready{false};
    };
    static Entry cache[24][64];                     // (64 = the device limit of sl_abi.hip)
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    Entry &ce = cache[slot][dev];
    if (!ce.ready.load(std::memory_order_acquire)) {
        err = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit);
        if (err != hipSuccess) return err;
        hipFunction_t f = nullptr;
        if (hipGetFuncBySymbol(&f, (const void *)fn) != hipSuccess) f = nullptr;
        (void)hipGetLastError();
        ce.fn.store(f, std::memory_order_relaxed);
        ce.ready.store(true, std::memory_order_release);
    }
    *kernel = (void *)fn;
    *f_out = ce.fn.load(std::memory_order_relaxed);
    // SAFELIFE_STEP_LDS_MIN (bytes, read once; an experiment's knob): ask for at least that much LDS per step workgroup,
    // up to the kernel's limit -- i.e. FEWER step workgroups per CU than the kernel's own needs would allow, to leave
    // the wavefronts of the episode-end pass of C5 (side_effects_flush(overlap=True)) room beside the steps.  Measured
    // (profiles/round5_p_c5_overlap.txt): two step workgroups per CU instead of three cost the steps 24.3 -> 27.7 us
    // and the pass under them gives nothing back -- 46.7 against 44.4 us per step; left at 0.
    static const int lds_min = [] {
        const char *e = getenv("SAFELIFE_STEP_LDS_MIN");
        return e ? atoi(e) : 0;
    }();
    *lds_out = lds_min > lds ? (lds_min < lds_limit ? lds_min : lds_limit) : lds;
    return hipSuccess;
}

template <int H, int W>
hipError_t launch_rollout_t(const sl_env_batch &env, int e_first, int e_count, const int32_t *actions, int T,
                                   int tstride, float *reward_t, uint8_t *done_t, const Jump *jump,
                                   hipStream_t stream, PreparedStep *prepared, const uint8_t *reset_mask) {
    using Gm = Geom<H, W>;
    void *kernel = nullptr;
    hipFunction_t f = nullptr;
    unsigned threads = 0;
    int lds = 0;
    bool gcache_ok = false;
    hipError_t err = pick_rollout_t<H, W>(env, T, &kernel, &f, &threads, &lds, &gcache_ok);
    if (err != hipSuccess) return err;
    const unsigned grid = (unsigned)((e_count + Gm::NB - 1) / Gm::NB);
    // the goal-word cache (GoalCache): handed to launches whose workgroups coincide with its blocks.  Any other launch of
    // a batch that has one may load levels without lowering the flags -- it lowers them all first.
    u32 *gcache = nullptr;
    if (env.goal_cache) {
        if (gcache_ok && e_first % Gm::NB == 0) {
            gcache = env.goal_cache;
        } else if (T != 0 && !prepared) {
            err = hipMemsetAsync(env.goal_cache, 0, GoalCache<H, W>::bytes(env.B), stream);
            if (err != hipSuccess) return err;
        } else if (T != 0 && gcache_ok) {
            return hipErrorNotSupported;        // (a prepared launch off the blocks' grid: the queues' slices never are)
        }
    }
    RolloutArgs args = {env.board, env.goals, env.rng, env.scalars, gcache, actions, e_first, e_first + e_count, env.score_lut, env,
                        env.E, tstride, T, env.out, reward_t, done_t, env.wrap.shaped_reward_t, jump, 0, nullptr, reset_mask};
    if (prepared) {
        // not launched: the argument block and the launch geometry, for the library's own queues (sl_aql.hip) to
        // dispatch any number of times with the per-step fields patched in
        if (!f || T != 1) return hipErrorNotSupported;
        prepared->f = f;
        prepared->grid = grid;
        prepared->threads = threads;
        prepared->lds = (unsigned)lds;
        prepared->arg_bytes = sizeof(args);
        memcpy(prepared->args, &args, sizeof(args));
        prepared->off_actions = offsetof(RolloutArgs, actions);
        prepared->off_out = offsetof(RolloutArgs, out);
        prepared->off_base = offsetof(RolloutArgs, xcd_base);
        prepared->off_flag = offsetof(RolloutArgs, xcd_flag);
        prepared->off_trace = offsetof(RolloutArgs, reward_t);
        prepared->off_next = offsetof(RolloutArgs, env) + offsetof(sl_env_batch, pool_next);
        return hipSuccess;
    }
    if (f) {
        size_t size = sizeof(args);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(f, grid, 1, 1, threads, 1, 1, (unsigned)lds, stream, nullptr, extra);
    }
    void *params[] = {&args.board, &args.goals, &args.rng, &args.scalars, &args.gcache, &args.actions, &args.first, &args.end,
                      &args.lut, &args.env, &args.E, &args.tstride, &args.T, &args.out, &args.reward_t, &args.done_t, &args.shaped_t,
                      &args.jump, &args.xcd_base, &args.xcd_flag, &args.reset_mask};
    return hipLaunchKernel(kernel, dim3(grid), dim3(threads), params, (size_t)lds, stream);
}

}  // namespace rl

// The shapes are compiled in three translation units (this file, and sl_rowlane_b.hip / sl_rowlane_c.hip, which
// include it with SL_ROWLANE_PART defined): the launcher templates of a part's shapes are explicitly instantiated
// there -- their kernels with them -- and only declared here.
#ifdef SL_DEV_SHAPES            /* development builds: the shapes of BASELINE.json only (a quarter of the compile time) */
#define SL_ROWLANE_SHAPES_A(X) X(25, 25) X(64, 64)
#define SL_ROWLANE_SHAPES_B(X)
#define SL_ROWLANE_SHAPES_C(X)
#else
#define SL_ROWLANE_SHAPES_A(X) X(25, 25) X(26, 26) X(64, 64) X(24, 24)
#define SL_ROWLANE_SHAPES_B(X) X(15, 15) X(20, 20) X(10, 10) X(8, 8) X(12, 12) X(16, 16)
#define SL_ROWLANE_SHAPES_C(X) X(30, 30) X(32, 32) X(40, 40) X(48, 48)
#endif
#define SL_ROWLANE_SHAPES(X) SL_ROWLANE_SHAPES_A(X) SL_ROWLANE_SHAPES_B(X) SL_ROWLANE_SHAPES_C(X)

#define SL_ROWLANE_LAUNCHERS(PREFIX, h, w)                                                                                 \
    PREFIX template hipError_t rl::launch_advance_t<h, w>(const u16 *, u16 *, int, const float *, int, const int32_t *,   \
                                                          const int32_t *, sl_pcg64 *, const Jump *, hipStream_t);         \
    PREFIX template hipError_t rl::launch_occupancy_t<h, w>(const u16 *, int32_t *, size_t, int, const int32_t *, int,    \
                                                            const int32_t *, const float *, int, sl_pcg64 *, const Jump *, \
                                                            hipStream_t);                                                  \
    PREFIX template hipError_t rl::launch_rollout_t<h, w>(const sl_env_batch &, int, int, const int32_t *, int, int,      \
                                                          float *, uint8_t *, const Jump *, hipStream_t, PreparedStep *,     \
                                                          const uint8_t *);
#ifdef SL_ROWLANE_PART
#define X(h, w) SL_ROWLANE_LAUNCHERS(, h, w)
#if SL_ROWLANE_PART == 1
SL_ROWLANE_SHAPES_B(X)
#elif SL_ROWLANE_PART == 2
SL_ROWLANE_SHAPES_C(X)
#endif      /* any other part number: no shape at all -- tools/isa_one.sh instantiates single kernels behind it */
#undef X
#else
#define X(h, w) SL_ROWLANE_LAUNCHERS(extern, h, w)
SL_ROWLANE_SHAPES_B(X)
SL_ROWLANE_SHAPES_C(X)
#undef X

// view cells the fused policy-layout epilogue can stage per board (its LDS room), 0 for unsupported shapes
int rowlane_policy_room(int H, int W) {
#define X(h, w)                                                                                              \
    if (H == h && W == w)                                                                                    \
        return ((rl::Geom<h, w>::GSH_BYTES + 4096 - ((rl::Geom<h, w>::NB * rl::OBS_PAR_INTS * 4 + 15) & ~15)) / 4 - 1) * 16 / 17;
    SL_ROWLANE_SHAPES(X)
#undef X
    return 0;
}

bool rowlane_lean_takes_queue(int H, int W) {
#define X(h, w) if (H == h && W == w) return rl::Geom<h, w>::WAVES_PER_SIMD < 4;
    SL_ROWLANE_SHAPES(X)
#undef X
    return false;
}

size_t rowlane_goal_cache_bytes(int H, int W, int B, bool spawn, int *boards_per_block) {
    // (slices and queue slices start at multiples of 64 envs: shapes whose workgroups hold 12, 20 or 24 boards would have
    //  launches off the blocks' grid -- they go without a cache; and so do batches whose plain single-step kernel keeps
    //  none -- the kernel's GCACHE: goal words in registers, no fifth leader wave)
#define X(h, w)                                                         \
    if (H == h && W == w) {                                             \
        if (64 % rl::Geom<h, w>::NB != 0) break;                        \
        if (!rl::gsh_in_registers<h, w>(spawn, true, true) || rl::Geom<h, w>::LEADX_OK) break; \
        if (boards_per_block) *boards_per_block = rl::Geom<h, w>::NB;   \
        return rl::GoalCache<h, w>::bytes(B);                           \
    }
    do {
        SL_ROWLANE_SHAPES(X)
    } while (false);
#undef X
    if (boards_per_block) *boards_per_block = 0;
    return 0;
}

bool rowlane_supports(int H, int W) {
#define X(h, w) if (H == h && W == w) return true;
    SL_ROWLANE_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_build_score_lut(const int32_t *points_table, int n_tables, int8_t *lut, hipStream_t stream) {
    const int n = n_tables * rl::SCORE_LUT_BYTES;
    hipLaunchKernelGGL(rl::k_build_score_lut, dim3((n + 255) / 256), dim3(256), 0, stream, points_table, n_tables, lut);
    return hipGetLastError();
}

hipError_t launch_build_baseline(const sl_env_batch &env, hipStream_t stream) {
    const int n = env.L * env.H;
    hipLaunchKernelGGL(rl::k_build_baseline, dim3((n + 255) / 256), dim3(256), 0, stream, env);
    return hipGetLastError();
}

hipError_t launch_advance_rowlane(const u16 *in, u16 *out, int B, int H, int W, const float *spawn_prob,
                                  int n_steps, const int32_t *n_each, const int32_t *n_valid, sl_pcg64 *rng,
                                  const Jump *jump, hipStream_t stream) {
#define X(h, w) if (H == h && W == w) return rl::launch_advance_t<h, w>(in, out, B, spawn_prob, n_steps, n_each, n_valid, rng, jump, stream);
    SL_ROWLANE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_occupancy_rowlane(const u16 *in, int32_t *counts, size_t counts_stride, int B, const int32_t *n_valid,
                                    int valid_period, const int32_t *pre_steps, int H, int W, const float *spawn_prob,
                                    int n_steps, sl_pcg64 *rng, const Jump *jump, hipStream_t stream) {
#define X(h, w) if (H == h && W == w) return rl::launch_occupancy_t<h, w>(in, counts, counts_stride, B, n_valid, valid_period, pre_steps, spawn_prob, n_steps, rng, jump, stream);
    SL_ROWLANE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipFunction_t rowlane_probe_function() {
    hipFunction_t f = nullptr;
    if (hipGetFuncBySymbol(&f, (const void *)rl::k_build_score_lut) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return f;
}

hipError_t launch_env_rollout_rowlane(const sl_env_batch &env, int e_first, int e_count, const int32_t *actions,
                                      int T, int tstride, float *reward_t, uint8_t *done_t, const Jump *jump,
                                      hipStream_t stream, PreparedStep *prepared, const uint8_t *reset_mask) {
#define X(h, w) if (env.H == h && env.W == w) return rl::launch_rollout_t<h, w>(env, e_first, e_count, actions, T, tstride, reward_t, done_t, jump, stream, prepared, reset_mask);
    SL_ROWLANE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

#endif  // !SL_ROWLANE_PART

}  // namespace sl
