// sl_rowlane.hip -- the fast gfx950 path: one board ROW per lane, two cells per 32-bit register.
//
// Mapping (template on the board shape H x W, H <= 64):
//   * a wavefront owns G = floor(64/H) consecutive boards; lane l = g*H + r holds row r of board g
//     entirely in VGPRs as WP = (W+3)/2 words of two uint16 cells each, with one halo cell on either
//     side ("ext" index e = x+1; word k holds e = 2k (low half) and e = 2k+1 (high half));
//   * HBM <-> LDS traffic is lane-linear dwords over the wave's contiguous span of boards (fully
//     coalesced, no per-row alignment constraints); LDS <-> register traffic converts between the
//     flat layout and the row-per-lane layout (funnel shifts for odd row starts);
//   * the 3x3 neighbourhood reduction is the commutative merge of sl_device.h in SWAR form:
//     horizontal neighbours are funnel shifts of adjacent registers, vertical neighbours come from
//     the lanes above/below through ds_bpermute (wrap inside the board's lane group);
//   * random draws (spawners) are rare: eligible cells are flagged in the fast pass and resolved in
//     a wave-uniform slow path with a segmented prefix count + PCG64 jump, preserving the
//     reference's row-major draw order (advance_board.c:115);
//   * waves never synchronise with each other: no __syncthreads anywhere in this file.
//
// Reference behaviour restated: advance_board.c:34-125 (CA step), :217-300 (actions, via act_one of
// sl_generic), safelife_env.py:148-218 + safelife_game.py:505-552,684-719,746-761 (step / reset glue),
// safelife_env.py:105-146 + helper_utils.py:42-75 (observation).
#include "sl_device.h"
#include "sl_kernels.h"

namespace sl {

namespace rl {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr u32 M1 = 0x00010001u;

__device__ __forceinline__ u32 pk_add(u32 x, u32 y) {
    us2 r = __builtin_bit_cast(us2, x) + __builtin_bit_cast(us2, y);
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_shr(u32 x, u32 sh) {
    us2 r = __builtin_bit_cast(us2, x) >> __builtin_bit_cast(us2, sh);
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 maj3(u32 x, u32 y, u32 z) {
    u32 d = x ^ y;
    return (d & z) | (~d & x);          // v_bfi_b32
}
__device__ __forceinline__ u32 funnel(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ u32 bperm(int byte_addr, u32 v) {
    return (u32)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int H, int W>
struct Geom {
    static constexpr int HW = H * W;
    static constexpr int WP = (W + 3) / 2;
    static constexpr int G0 = 64 / H;
    // the wave's span of boards must be a whole number of dwords: G*HW even
    static constexpr int G = (HW % 2 == 1 && G0 % 2 == 1) ? G0 - 1 : G0;
    static_assert(G >= 1, "shape not supported by the row-per-lane path");
    static constexpr int NL = G * H;                       // active lanes
    static constexpr int ND = G * HW / 2;                  // dwords per wave span
    static constexpr int NDI = (ND + 63) / 64;
    static constexpr int PAD = 16;
    static constexpr int REGION = ((G * HW * 2 + 15) / 16) * 16 + 2 * PAD;
    static constexpr int OFF_BOARD = 0;
    static constexpr int OFF_GOALS = REGION;
    static constexpr int OFF_TABLE = 2 * REGION;           // G x 72 int32
    static constexpr int OFF_RNG = OFF_TABLE + G * 72 * 4; // G x 4 u64
    static constexpr int WAVE_BYTES = OFF_RNG + G * 32;
    static constexpr int LAST_WORD = W >> 1, LAST_HI = W & 1;          // cell x = W-1 (e = W)
    static constexpr int RH_WORD = (W + 1) >> 1, RH_HI = (W + 1) & 1;  // right halo (e = W+1)
    static constexpr u32 vmask(int k) {
        return ((2 * k >= 1 && 2 * k <= W) ? 0x0000FFFFu : 0u) | ((2 * k + 1 <= W) ? 0xFFFF0000u : 0u);
    }
};

constexpr int WAVES = 4;   // waves per workgroup (independent of each other)

// ---- LDS flat board  <->  row-per-lane registers ------------------------------------------------

// Row r of board g from the flat uint16 image at `region` (+PAD) into ext words; halos fixed up.
template <int H, int W>
__device__ __forceinline__ void read_row(const unsigned char *region, int g, int r, u32 (&b)[Geom<H, W>::WP]) {
    using Gm = Geom<H, W>;
    const int byte = Gm::PAD + 2 * (g * Gm::HW + r * W - 1);
    const u32 *p = (const u32 *)(region + (byte & ~3));
    const u32 sh = (byte & 2) * 8;
    u32 raw[Gm::WP + 1];
#pragma unroll
    for (int j = 0; j <= Gm::WP; ++j) raw[j] = p[j];
#pragma unroll
    for (int k = 0; k < Gm::WP; ++k) b[k] = funnel(raw[k + 1], raw[k], sh);
    // halos: e = 0 <- cell W-1, e = W+1 <- cell 0; anything beyond is cleared
    const u32 last = Gm::LAST_HI ? (b[Gm::LAST_WORD] >> 16) : (b[Gm::LAST_WORD] & 0xFFFFu);
    const u32 first = b[0] >> 16;
    b[0] = (b[0] & 0xFFFF0000u) | last;
    if (Gm::RH_HI) b[Gm::RH_WORD] = (b[Gm::RH_WORD] & 0xFFFFu) | (first << 16);
    else b[Gm::RH_WORD] = first;
}

// Inverse of read_row for the W real cells of the row.
template <int H, int W>
__device__ __forceinline__ void write_row(unsigned char *region, int g, int r, const u32 (&n)[Geom<H, W>::WP]) {
    using Gm = Geom<H, W>;
    const int i1 = g * Gm::HW + r * W;            // uint16 index of cell x = 0
    u16 *c16 = (u16 *)(region + Gm::PAD);
    const bool odd = i1 & 1;
    // dword j (from the first aligned one inside the row) holds e = (2j+1, 2j+2) if the row starts
    // aligned, e = (2j+2, 2j+3) otherwise
    u32 *d = (u32 *)(region + Gm::PAD + 2 * (i1 + (odd ? 1 : 0)));
    constexpr int NF_EVEN = W / 2, NF_ODD = (W - 1) / 2;
#pragma unroll
    for (int j = 0; j < NF_EVEN; ++j) {
        u32 even_word = funnel(n[j + 1], n[j], 16);
        u32 odd_word = n[j + 1];
        if (j < NF_ODD) d[j] = odd ? odd_word : even_word;
        else if (!odd) d[j] = even_word;
    }
    if (odd) {
        c16[i1] = (u16)(n[0] >> 16);                               // e = 1
        if ((W - 1) & 1) c16[i1 + W - 1] = Gm::LAST_HI ? (u16)(n[Gm::LAST_WORD] >> 16) : (u16)n[Gm::LAST_WORD];
    } else if (W & 1) {
        c16[i1 + W - 1] = Gm::LAST_HI ? (u16)(n[Gm::LAST_WORD] >> 16) : (u16)n[Gm::LAST_WORD];
    }
}

// ---- one CA step on registers ---------------------------------------------------------------------
// b: ext words with halos.  n: new cells; halves whose outcome needs a random draw hold the spawned
// value tentatively and are flagged in elig (bit k = low half of word k, bit 16+k = high half;
// word index k/16).  up/dn: ds_bpermute byte addresses of the lanes holding rows r-1 / r+1.
template <int H, int W>
__device__ __forceinline__ u32 ca_rows(const u32 (&b)[Geom<H, W>::WP], u32 (&n)[Geom<H, W>::WP],
                                       u32 (&elig)[(Geom<H, W>::WP + 15) / 16], int up, int dn) {
    using Gm = Geom<H, W>;
    constexpr int WP = Gm::WP;
    u32 s[WP], a[WP];
#pragma unroll
    for (int k = 0; k < WP; ++k) {
        u32 t = b[k] | ((b[k] & 0x00080008u) << 5);           // destructible -> bit 8
        a[k] = t & M1;
        u32 am = a[k] * 0x0F00u;                              // once-mask where alive
        u32 spm = ((t >> 7) & M1) * 0xE000u;                  // twice-colour mask where spawning
        s[k] = (t & 0x00E000E0u) | (t & am) | ((t << 4) & spm);
    }
#pragma unroll
    for (int j = 0; j < (WP + 15) / 16; ++j) elig[j] = 0;
    u32 any = 0;
#pragma unroll
    for (int k = 0; k < WP; ++k) {
        if (Gm::vmask(k) == 0) {
            n[k] = 0;
            continue;
        }
        const u32 sl = k > 0 ? s[k - 1] : 0u, sr = k + 1 < WP ? s[k + 1] : 0u;
        const u32 al = k > 0 ? a[k - 1] : 0u, ar = k + 1 < WP ? a[k + 1] : 0u;
        const u32 L = funnel(s[k], sl, 16), R = funnel(sr, s[k], 16);
        const u32 cl = funnel(a[k], al, 16), cr = funnel(ar, a[k], 16);
        const u32 rc = (L | s[k] | R) | ((maj3(L, s[k], R) & 0x0F000F00u) << 4) | (cl + a[k] + cr);
        const u32 U = bperm(up, rc), D = bperm(dn, rc);
        const u32 X = U | rc | D;
        const u32 F = (X & 0xFFE0FFE0u) | ((maj3(U, rc, D) & 0x0F000F00u) << 4);
        const u32 cnt = pk_add(pk_add(U, rc), D) & 0x000F000Fu;
        const u32 s34 = pk_shr(0x00180018u, cnt) & M1;
        const u32 is3 = pk_shr(0x00080008u, cnt) & M1;
        const u32 bb = b[k];
        const u32 alive = bb & M1;
        const u32 frozen = (bb >> 4) & M1;
        const u32 keep_a = frozen | ((X >> 5) & M1) | s34;
        const u32 keep_d = frozen | ((X >> 6) & M1);
        const u32 born = is3 & ~keep_d & ~alive;
        const u32 el = ((X >> 7) & M1) & ~(keep_d | is3 | alive) & (Gm::vmask(k) & M1);
        const u32 keep = ((alive & keep_a) | (~alive & ~born & ~el)) & M1;
        const u32 newcol = (F >> 4) & 0x0E000E00u;
        const u32 newborn = M1 | newcol | ((F >> 9) & 0x00080008u);
        const u32 newspawn = M1 | 0x00080008u | newcol;
        n[k] = (bb & (keep * 0xFFFFu)) | (newborn & (born * 0xFFFFu)) | (newspawn & (el * 0xFFFFu));
        elig[k / 16] |= el << (k % 16);
        any |= el;
    }
    return any;
}

// Resolve the flagged halves with the board's PCG64 stream, row-major (wave-uniform call).
// rng_lds: G x {state_hi, state_lo, inc_hi, inc_lo} in LDS; advanced by the draws consumed.
template <int H, int W>
__device__ void resolve_draws(const u32 (&b)[Geom<H, W>::WP], u32 (&n)[Geom<H, W>::WP],
                              const u32 (&elig)[(Geom<H, W>::WP + 15) / 16], u64 *rng_lds, int g, int r,
                              int lane, double p, const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    constexpr int WP = Gm::WP;
    int mine = 0;
#pragma unroll
    for (int j = 0; j < (WP + 15) / 16; ++j) mine += __popc(elig[j]);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const int first_lane = g * H;
    int before_group = (int)bperm(4 * (first_lane > 0 ? first_lane - 1 : 0), (u32)incl);
    if (first_lane == 0) before_group = 0;
    const int total = (int)bperm(4 * (first_lane + H - 1), (u32)incl) - before_group;
    const int excl = incl - mine - before_group;
    const U128 st = {rng_lds[4 * g + 0], rng_lds[4 * g + 1]}, inc = {rng_lds[4 * g + 2], rng_lds[4 * g + 3]};
    wave_sync();     // everyone has read the old state before the leader replaces it
    if (mine > 0) {
        U128 cur = pcg_jump(jump, excl, st, inc);
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            if ((elig[k / 16] >> (k % 16)) & 1u) {
                cur = pcg_step(cur, inc);
                if (!(pcg_output_double(cur) < p)) n[k] = (n[k] & 0xFFFF0000u) | (b[k] & 0x0000FFFFu);
            }
            if ((elig[k / 16] >> (16 + k % 16)) & 1u) {
                cur = pcg_step(cur, inc);
                if (!(pcg_output_double(cur) < p)) n[k] = (n[k] & 0x0000FFFFu) | (b[k] & 0xFFFF0000u);
            }
        }
    }
    if (r == 0 && total > 0) {
        U128 s2 = pcg_jump(jump, total, st, inc);
        rng_lds[4 * g + 0] = s2.hi;
        rng_lds[4 * g + 1] = s2.lo;
    }
    wave_sync();
}

// sum(points_table * alive_counts) contribution of one row (valid halves only); table in LDS.
template <int H, int W>
__device__ __forceinline__ int row_score(const u32 (&n)[Geom<H, W>::WP], const u32 (&gl)[Geom<H, W>::WP],
                                         const int *table) {
    using Gm = Geom<H, W>;
    int s = 0;
#pragma unroll
    for (int k = 0; k < Gm::WP; ++k) {
        if (Gm::vmask(k) == 0) continue;
        const u32 c = n[k];
        const u32 gc = (gl[k] >> 9) & 0x00070007u;
        const u32 alive = c & M1;
        const u32 col = (((c >> 9) & 0x00070007u) & (alive * 0xFu)) | (0x00080008u & ~(alive * 0xFu));
        const u32 bin = gc * 9u + col;                       // < 72 per half, no carry across halves
        const u32 movable = ((c >> 2) | (c >> 3) | (c >> 15)) & M1;
        const u32 counted = (~(c >> 4) | movable) & M1;      // !(frozen && !movable)
        if (Gm::vmask(k) & 0xFFFFu) s += (counted & 1u) ? table[bin & 0xFFu] : 0;
        if (Gm::vmask(k) >> 16) s += (counted >> 16) ? table[bin >> 16] : 0;
    }
    return s;
}

// Segmented (per board) sum of a per-lane value; every lane of the group gets its board's total.
template <int H>
__device__ __forceinline__ int group_sum(int v, int g, int lane) {
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const int first_lane = g * H;
    int before = (int)bperm(4 * (first_lane > 0 ? first_lane - 1 : 0), (u32)incl);
    if (first_lane == 0) before = 0;
    return (int)bperm(4 * (first_lane + H - 1), (u32)incl) - before;
}

__device__ __forceinline__ u32 obs_word(u32 b, u32 g, int remove_white) {
    u32 gc = g & COLORS;
    if (remove_white && gc == COLORS) gc = 0;
    return b | (gc << 16);
}

// ---- wave span <-> HBM ----------------------------------------------------------------------------

template <int H, int W>
__device__ __forceinline__ void load_span(const u16 *__restrict__ src, unsigned char *region, int nb, int lane) {
    using Gm = Geom<H, W>;
    const u32 *s32 = (const u32 *)src;
    u32 *d32 = (u32 *)(region + Gm::PAD);
    const int nd = nb * Gm::HW / 2;
    u32 v[Gm::NDI];
#pragma unroll
    for (int i = 0; i < Gm::NDI; ++i) v[i] = (lane + 64 * i < nd) ? s32[lane + 64 * i] : 0u;
#pragma unroll
    for (int i = 0; i < Gm::NDI; ++i)
        if (lane + 64 * i < nd) d32[lane + 64 * i] = v[i];
    if (((nb * Gm::HW) & 1) && lane == 0) ((u16 *)d32)[nb * Gm::HW - 1] = src[nb * Gm::HW - 1];
}

template <int H, int W>
__device__ __forceinline__ void store_span(u16 *__restrict__ dst, const unsigned char *region, int nb, int lane) {
    using Gm = Geom<H, W>;
    u32 *d32 = (u32 *)dst;
    const u32 *s32 = (const u32 *)(region + Gm::PAD);
    const int nd = nb * Gm::HW / 2;
#pragma unroll
    for (int i = 0; i < Gm::NDI; ++i)
        if (lane + 64 * i < nd) d32[lane + 64 * i] = s32[lane + 64 * i];
    if (((nb * Gm::HW) & 1) && lane == 0) dst[nb * Gm::HW - 1] = ((const u16 *)s32)[nb * Gm::HW - 1];
}

// ---- advance_board --------------------------------------------------------------------------------

template <int H, int W>
__global__ __launch_bounds__(64 * WAVES) void k_advance_rowlane(const u16 *__restrict__ in, u16 *__restrict__ out,
                                                                int B, const float *__restrict__ spawn_prob,
                                                                int n_steps, sl_pcg64 *rng,
                                                                const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *lds = smem + wave * Gm::WAVE_BYTES;
    const int e0 = (blockIdx.x * WAVES + wave) * Gm::G;
    if (e0 >= B) return;
    const int nb = min(Gm::G, B - e0);
    int g = 0;
#pragma unroll
    for (int q = 1; q < Gm::G; ++q) g += (lane >= q * H) ? 1 : 0;
    const int r = lane - g * H;
    const bool live = lane < Gm::NL && g < nb;
    const int up = 4 * (live ? (r == 0 ? lane + H - 1 : lane - 1) : lane);
    const int dn = 4 * (live ? (r == H - 1 ? lane - (H - 1) : lane + 1) : lane);
    u64 *rng_lds = (u64 *)(lds + Gm::OFF_RNG);
    unsigned char *board = lds + Gm::OFF_BOARD;

    load_span<H, W>(in + (size_t)e0 * Gm::HW, board, nb, lane);
    if (lane < 4 * nb) rng_lds[lane] = ((const u64 *)(rng + e0))[lane];
    const double p = live ? (double)spawn_prob[e0 + g] : 0.0;
    wave_sync();
    u32 b[Gm::WP], n[Gm::WP], elig[(Gm::WP + 15) / 16];
    if (live) read_row<H, W>(board, g, r, b);
    else
#pragma unroll
        for (int k = 0; k < Gm::WP; ++k) b[k] = 0;
    for (int s = 0; s < n_steps; ++s) {
        u32 any = ca_rows<H, W>(b, n, elig, up, dn);
        if (!live) any = 0;
        if (__ballot(any != 0)) {
            if (!live)
#pragma unroll
                for (int j = 0; j < (Gm::WP + 15) / 16; ++j) elig[j] = 0;
            resolve_draws<H, W>(b, n, elig, rng_lds, live ? g : 0, live ? r : 1, lane, p, jump);
        }
        // next step's input: new cells + refreshed halos
#pragma unroll
        for (int k = 0; k < Gm::WP; ++k) b[k] = n[k];
        const u32 last = Gm::LAST_HI ? (b[Gm::LAST_WORD] >> 16) : (b[Gm::LAST_WORD] & 0xFFFFu);
        const u32 first = b[0] >> 16;
        b[0] = (b[0] & 0xFFFF0000u) | last;
        if (Gm::RH_HI) b[Gm::RH_WORD] = (b[Gm::RH_WORD] & 0xFFFFu) | (first << 16);
        else b[Gm::RH_WORD] = first;
    }
    if (live) write_row<H, W>(board, g, r, b);
    wave_sync();
    store_span<H, W>(out + (size_t)e0 * Gm::HW, board, nb, lane);
    if (lane < 4 * nb) ((u64 *)(rng + e0))[lane] = rng_lds[lane];
}

// ---- fused env step / rollout ---------------------------------------------------------------------

// update_exit_colors for the board of a leader lane, on the flat LDS image.
__device__ __forceinline__ void recolor_exits_lds(u16 *board, int W, int ly, int lx, const int32_t *exits, int E,
                                                  int score, int initial, int required, int exit_points) {
    bool any_can = false;
    if (ly >= 0) {
        u16 *cell = board + ly * W + lx;
        int earned = score - initial + exit_points * (has_exited(*cell) ? 1 : 0);
        if (earned < 0) earned = 0;
        bool can = (*cell & AGENT) && earned >= required;
        *cell = (u16)((*cell & ~EXIT) | (can ? EXIT : 0u));
        any_can = can;
    }
    const u16 paint = (u16)(FROZEN | EXIT | (any_can ? COLOR_R : 0u));
    for (int k = 0; k < E; ++k) {
        int ex = exits[k];
        if (ex >= 0) board[ex] = paint;
    }
}

template <int H, int W>
__global__ __launch_bounds__(64 * WAVES) void k_env_rollout_rowlane(sl_env_batch env,
                                                                    const int32_t *__restrict__ actions, int T,
                                                                    float *__restrict__ reward_t,
                                                                    uint8_t *__restrict__ done_t,
                                                                    const Jump *__restrict__ jump) {
    using Gm = Geom<H, W>;
    constexpr int WP = Gm::WP, HW = Gm::HW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *lds = smem + wave * Gm::WAVE_BYTES;
    const int B = env.B, E = env.E;
    const int e0 = (blockIdx.x * WAVES + wave) * Gm::G;
    if (e0 >= B) return;
    const int nb = min(Gm::G, B - e0);
    int g = 0;
#pragma unroll
    for (int q = 1; q < Gm::G; ++q) g += (lane >= q * H) ? 1 : 0;
    const int r = lane - g * H;
    const bool live = lane < Gm::NL && g < nb;
    const bool leader = live && r == 0;
    const int e = e0 + (live ? g : 0);
    const int up = 4 * (live ? (r == 0 ? lane + H - 1 : lane - 1) : lane);
    const int dn = 4 * (live ? (r == H - 1 ? lane - (H - 1) : lane + 1) : lane);
    unsigned char *board = lds + Gm::OFF_BOARD, *goals = lds + Gm::OFF_GOALS;
    u16 *board16 = (u16 *)(board + Gm::PAD) + (live ? g : 0) * HW;
    int *table = (int *)(lds + Gm::OFF_TABLE) + (live ? g : 0) * 72;
    u64 *rng_lds = (u64 *)(lds + Gm::OFF_RNG);

    load_span<H, W>(env.board + (size_t)e0 * HW, board, nb, lane);
    load_span<H, W>(env.goals + (size_t)e0 * HW, goals, nb, lane);
    if (lane < 4 * nb) rng_lds[lane] = ((const u64 *)(env.rng + e0))[lane];
    // per-board scalars live in the leader lane's registers for the whole launch
    int ly = -1, lx = -1, steps = 0, old_value = 0, required = 0, initial = 0, ep_len = 0, gstatic = 1;
    float ep_rew = 0.0f;
    bool active = false;
    double p = 0.0;
    int tidx = 0, level = 0, episodes = 0;
    if (live) {
        p = (double)env.spawn_prob[e];
        gstatic = env.goals_static[e];
        tidx = env.table_idx[e];
        level = env.level_idx[e];
    }
    if (leader) {
        ly = env.agent_loc[2 * e];
        lx = env.agent_loc[2 * e + 1];
        steps = env.num_steps[e];
        old_value = env.old_value[e];
        required = env.required_points[e];
        initial = env.initial_points[e];
        ep_len = env.episode_length[e];
        ep_rew = env.episode_reward[e];
        active = env.is_active[e] != 0;
        episodes = env.episode_idx[e];
    }
    if (live) {
        for (int i = r; i < 72; i += H) table[i] = env.points_table[72 * tidx + i];
    }
    wave_sync();

    u32 b[WP], n[WP], gl[WP], elig[(WP + 15) / 16];
#pragma unroll
    for (int k = 0; k < WP; ++k) b[k] = n[k] = gl[k] = 0;
    if (live) read_row<H, W>(goals, g, r, gl);
    bool goals_dirty = false;
    const int32_t *exits = env.exit_locs + (size_t)e * E;

    for (int t = 0; t < T; ++t) {
        // safelife_env.py:151
        if (leader && ly >= 0) {
            int loc[2] = {ly, lx};
            act_one<int>(board16, H, W, loc, actions[(size_t)t * B + e]);
            ly = loc[0];
            lx = loc[1];
        }
        wave_sync();
        if (live) read_row<H, W>(board, g, r, b);
        // safelife_env.py:152 : board, then goals unless static
        u32 any = ca_rows<H, W>(b, n, elig, up, dn);
        if (!live) any = 0;
        if (__ballot(any != 0)) {
            if (!live)
#pragma unroll
                for (int j = 0; j < (WP + 15) / 16; ++j) elig[j] = 0;
            resolve_draws<H, W>(b, n, elig, rng_lds, live ? g : 0, live ? r : 1, lane, p, jump);
        }
        if (__ballot(live && gstatic != 1)) {
            u32 gn[WP], gel[(WP + 15) / 16];
            u32 gany = ca_rows<H, W>(gl, gn, gel, up, dn);
            const bool dyn = live && gstatic != 1;
            if (!dyn) gany = 0;
            if (__ballot(gany != 0)) {
                if (!dyn)
#pragma unroll
                    for (int j = 0; j < (WP + 15) / 16; ++j) gel[j] = 0;
                resolve_draws<H, W>(gl, gn, gel, rng_lds, live ? g : 0, live ? r : 1, lane, p, jump);
            }
            u32 diff = 0;
#pragma unroll
            for (int k = 0; k < WP; ++k) {
                u32 vm = Gm::vmask(k);
                diff |= ((gn[k] ^ gl[k]) | (gn[k] & 0x00800080u)) & vm;
            }
            const int changed = group_sum<H>(dyn && diff ? 1 : 0, live ? g : 0, lane);
            if (dyn) {
                if (gstatic == 0) gstatic = changed ? 2 : 1;
#pragma unroll
                for (int k = 0; k < WP; ++k) gl[k] = gn[k];
                const u32 last = Gm::LAST_HI ? (gl[Gm::LAST_WORD] >> 16) : (gl[Gm::LAST_WORD] & 0xFFFFu);
                const u32 first = gl[0] >> 16;
                gl[0] = (gl[0] & 0xFFFF0000u) | last;
                if (Gm::RH_HI) gl[Gm::RH_WORD] = (gl[Gm::RH_WORD] & 0xFFFFu) | (first << 16);
                else gl[Gm::RH_WORD] = first;
                write_row<H, W>(goals, g, r, gl);
                goals_dirty = true;
            }
        }
        // safelife_env.py:153-160
        const int score = group_sum<H>(live ? row_score<H, W>(n, gl, table) : 0, live ? g : 0, lane);
        if (live) write_row<H, W>(board, g, r, n);
        wave_sync();
        bool done = false;
        if (leader) {
            recolor_exits_lds(board16, W, ly, lx, exits, E, score, initial, required, env.exit_points);
            steps += 1;
            const bool times_up = steps >= env.time_limit;
            float reward = 0.0f;
            bool success = false;
            done = true;
            if (ly >= 0) {
                u32 cell = board16[ly * W + lx];
                success = has_exited(cell);
                int value = score + env.exit_points * (success ? 1 : 0);
                reward = (float)((value - old_value) * (active ? 1 : 0));
                old_value = value;
                done = !(cell & AGENT) || times_up;
            }
            ep_rew += reward;
            ep_len += active ? 1 : 0;
            active = active && !done;
            env.reward[e] = reward;
            env.done[e] = done;
            env.success[e] = success;
            env.times_up[e] = times_up;
            if (env.info_episode_reward) env.info_episode_reward[e] = ep_rew;
            if (env.info_episode_length) env.info_episode_length[e] = ep_len;
            if (reward_t) reward_t[(size_t)t * B + e] = reward;
            if (done_t) done_t[(size_t)t * B + e] = done;
        }
        // on-device auto-reset (training/base_algo.py:231-236 calls env.reset() after a done step)
        if (env.auto_reset && __ballot(leader && done)) {
            const int flag = group_sum<H>(leader && done ? 1 : 0, live ? g : 0, lane);
            const bool mine = live && flag != 0;
            int lvl = 0;
            if (mine) {
                lvl = level = (level + env.level_stride) % env.L;
                const u16 *pb = env.pool_board + (size_t)lvl * HW, *pg = env.pool_goals + (size_t)lvl * HW;
                u16 *gdst = (u16 *)(goals + Gm::PAD) + g * HW;
                for (int i = r; i < HW; i += H) {
                    board16[i] = pb[i];
                    gdst[i] = pg[i];
                }
                tidx = env.pool_table_idx[lvl];
                for (int i = r; i < 72; i += H) table[i] = env.points_table[72 * tidx + i];
                p = (double)env.pool_spawn_prob[lvl];
                gstatic = 0;
                if (r < 4) rng_lds[4 * g + r] = ((const u64 *)(env.pool_rng + lvl))[r];
                for (int k = r; k < E; k += H) env.exit_locs[(size_t)e * E + k] = env.pool_exit_locs[(size_t)lvl * E + k];
                goals_dirty = true;
            }
            wave_sync();
            if (mine) {
                read_row<H, W>(board, g, r, n);
                read_row<H, W>(goals, g, r, gl);
            }
            const int s0 = group_sum<H>(mine ? row_score<H, W>(n, gl, table) : 0, live ? g : 0, lane);
            if (mine && r == 0) {
                episodes += 1;
                exits = env.pool_exit_locs + (size_t)lvl * E;   // never written by this launch
                ly = env.pool_agent_loc[2 * lvl];
                lx = env.pool_agent_loc[2 * lvl + 1];
                initial = env.pool_initial_points[lvl];
                recolor_exits_lds(board16, W, ly, lx, env.pool_exit_locs + (size_t)lvl * E, E, s0, initial,
                                  env.pool_required_reset[lvl], env.exit_points);
                int exited = ly >= 0 ? (has_exited(board16[ly * W + lx]) ? 1 : 0) : 0;
                old_value = s0 + env.exit_points * exited;
                required = env.pool_required_step[lvl];
                steps = 0;
                active = true;
                ep_rew = 0.0f;
                ep_len = 0;
            }
            wave_sync();
        }
    }

    // write-back
    wave_sync();
    store_span<H, W>(env.board + (size_t)e0 * HW, board, nb, lane);
    if (__ballot(goals_dirty)) store_span<H, W>(env.goals + (size_t)e0 * HW, goals, nb, lane);
    if (lane < 4 * nb) ((u64 *)(env.rng + e0))[lane] = rng_lds[lane];
    if (live) {
        if (r == 0) {
            env.agent_loc[2 * e] = ly;
            env.agent_loc[2 * e + 1] = lx;
            env.num_steps[e] = steps;
            env.old_value[e] = old_value;
            env.required_points[e] = required;
            env.initial_points[e] = initial;
            env.episode_length[e] = ep_len;
            env.episode_reward[e] = ep_rew;
            env.is_active[e] = active ? 1 : 0;
            env.goals_static[e] = (uint8_t)gstatic;
            env.table_idx[e] = tidx;
            env.spawn_prob[e] = (float)p;
            env.level_idx[e] = level;
            env.episode_idx[e] = episodes;
        }
    }

    // observation (safelife_env.py:105-146) from the LDS images
    if (env.obs) {
        const int vh = env.view_h, vw = env.view_w, C = env.n_channels, nv = vh * vw;
        const u16 *b16 = (const u16 *)(board + Gm::PAD), *g16 = (const u16 *)(goals + Gm::PAD);
        for (int q = 0; q < nb; ++q) {
            const int eq = e0 + q;
            const int qy = (int)bperm(4 * (q * H), (u32)ly), qx = (int)bperm(4 * (q * H), (u32)lx);
            const int y0 = qy >= 0 ? qy : 0, x0 = qy >= 0 ? qx : 0;
            const u64 exq = (u64)(uintptr_t)exits;
            const int32_t *ex = (const int32_t *)(uintptr_t)(((u64)bperm(4 * (q * H), (u32)(exq >> 32)) << 32) |
                                                             (u64)bperm(4 * (q * H), (u32)exq));
            const u16 *bq = b16 + q * HW, *gq = g16 + q * HW;
            for (int v = lane; v < nv; v += 64) {
                const int vy = v / vw, vx = v - vy * vw;
                const int sy = pos_mod(y0 - vh / 2 + vy, H), sx = pos_mod(x0 - vw / 2 + vx, W);
                u32 word = obs_word(bq[sy * W + sx], gq[sy * W + sx], env.remove_white_goals);
                for (int k = 0; k < E; ++k) {
                    const int xk = ex[k];
                    if (xk < 0) continue;
                    const int iy = xk / W, ix = xk - iy * W;
                    int jy = pos_mod(iy - y0 + H / 2, H) - H / 2 + vh / 2;
                    int jx = pos_mod(ix - x0 + W / 2, W) - W / 2 + vw / 2;
                    jy = min(max(jy, 0), vh - 1);
                    jx = min(max(jx, 0), vw - 1);
                    if (jy == vy && jx == vx) word = obs_word(bq[xk], gq[xk], env.remove_white_goals);
                }
                if (C == 0) {
                    ((u32 *)env.obs)[(size_t)eq * nv + v] = word;
                } else {
                    uint8_t *o = env.obs + ((size_t)eq * nv + v) * C;
                    for (int c = 0; c < C; ++c) o[c] = (word >> env.channels[c]) & 1u;
                }
            }
        }
    }
}

template <int H, int W>
static hipError_t launch_advance_t(const u16 *in, u16 *out, int B, const float *spawn_prob, int n_steps,
                                   sl_pcg64 *rng, const Jump *jump, hipStream_t stream) {
    using Gm = Geom<H, W>;
    const size_t lds = (size_t)WAVES * Gm::WAVE_BYTES;
    auto fn = k_advance_rowlane<H, W>;
    hipError_t err = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    const int per_block = WAVES * Gm::G;
    hipLaunchKernelGGL(fn, dim3((B + per_block - 1) / per_block), dim3(64 * WAVES), lds, stream, in, out, B,
                       spawn_prob, n_steps, rng, jump);
    return hipGetLastError();
}

template <int H, int W>
static hipError_t launch_rollout_t(const sl_env_batch &env, const int32_t *actions, int T, float *reward_t,
                                   uint8_t *done_t, const Jump *jump, hipStream_t stream) {
    using Gm = Geom<H, W>;
    const size_t lds = (size_t)WAVES * Gm::WAVE_BYTES;
    auto fn = k_env_rollout_rowlane<H, W>;
    hipError_t err = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    const int per_block = WAVES * Gm::G;
    hipLaunchKernelGGL(fn, dim3((env.B + per_block - 1) / per_block), dim3(64 * WAVES), lds, stream, env, actions,
                       T, reward_t, done_t, jump);
    return hipGetLastError();
}

}  // namespace rl

#define SL_ROWLANE_SHAPES(X) X(25, 25) X(26, 26)

bool rowlane_supports(int H, int W) {
#define X(h, w) if (H == h && W == w) return true;
    SL_ROWLANE_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_advance_rowlane(const u16 *in, u16 *out, int B, int H, int W, const float *spawn_prob,
                                  int n_steps, sl_pcg64 *rng, const Jump *jump, hipStream_t stream) {
#define X(h, w) if (H == h && W == w) return rl::launch_advance_t<h, w>(in, out, B, spawn_prob, n_steps, rng, jump, stream);
    SL_ROWLANE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_env_rollout_rowlane(const sl_env_batch &env, const int32_t *actions, int T, float *reward_t,
                                      uint8_t *done_t, const Jump *jump, hipStream_t stream) {
#define X(h, w) if (env.H == h && env.W == w) return rl::launch_rollout_t<h, w>(env, actions, T, reward_t, done_t, jump, stream);
    SL_ROWLANE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace sl
