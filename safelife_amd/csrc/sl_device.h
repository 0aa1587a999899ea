// sl_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Cell bit layout: reference safelife/speedups_src/constants.h:4-33.
// Neighbourhood algebra: the reference folds neighbours with two helper routines
// (advance_board.c:12-32); both are the same commutative/associative merge of 16-bit
// "summaries", which is what lets the kernels reduce the 3x3 block in any order and in SWAR form.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/safelife_hip.h"

typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

namespace sl {

enum : u32 {
    ALIVE = 1u << 0,
    AGENT = 1u << 1,
    PUSHABLE = 1u << 2,
    DESTRUCTIBLE = 1u << 3,
    FROZEN = 1u << 4,
    PRESERVING = 1u << 5,
    INHIBITING = 1u << 6,
    SPAWNING = 1u << 7,
    EXIT = 1u << 8,
    COLOR_R = 1u << 9,
    COLOR_B = 1u << 11,
    COLORS = 7u << 9,
    ORIENT_SHIFT = 12,
    ORIENT_MASK = 3u << 12,
    PULLABLE = 1u << 15,
    MOVABLE_OR_DESTRUCTIBLE = DESTRUCTIBLE | PUSHABLE | PULLABLE,
    PLAYER = AGENT | DESTRUCTIBLE | FROZEN | PRESERVING | INHIBITING,   // CellTypes.player (122)
};

// summary / accumulator fields (advance_board.c:6-9)
enum : u32 {
    S_COUNT = 0x000Fu,  // alive cells folded in
    S_ANY = 0x00E0u,    // preserving | inhibiting | spawning seen anywhere
    S_ONCE = 0x0F00u,   // bit8 (exit|destructible) + colours seen in >= 1 alive cell
    S_TWICE = 0xF000u,  // ... in >= 2 alive cells (spawner colours land here directly)
};

// ---- scalar (one cell per lane) forms, used by the generic kernels -------------------------

__device__ __forceinline__ u32 cell_summary(u32 b) {
    u32 t = b | ((b & DESTRUCTIBLE) << 5);          // advance_board.c:45-47
    u32 s = t & S_ANY;
    if (t & ALIVE) s |= (t & S_ONCE) | 1u;
    if (t & SPAWNING) s |= (t & COLORS) << 4;       // advance_board.c:19
    return s;
}

__device__ __forceinline__ u32 merge2(u32 x, u32 y) {
    u32 both = x & y & S_ONCE;
    return (((x | y) & (S_ANY | S_ONCE | S_TWICE)) | (both << 4)) + (x & S_COUNT) + (y & S_COUNT);
}

__device__ __forceinline__ u32 merge3(u32 x, u32 y, u32 z) { return merge2(merge2(x, y), z); }

// Deterministic part of the rule (advance_board.c:94-124).  Returns the new cell; sets
// `eligible` when the outcome depends on a random draw (dead, not frozen, not inhibited,
// count != 3, spawner in the 3x3 block) -- the draw itself is made by the caller in row-major
// order.  For an eligible cell the returned value is the old cell.
__device__ __forceinline__ u32 apply_rule(u32 b, u32 acc, bool &eligible) {
    u32 cnt = acc & S_COUNT;
    eligible = false;
    if (b & ALIVE) {
        bool keep = (b & FROZEN) || (acc & PRESERVING) || cnt == 3 || cnt == 4;
        return keep ? b : 0u;
    }
    if ((b & FROZEN) || (acc & INHIBITING)) return b;
    if (cnt == 3) return ALIVE | ((acc >> 4) & COLORS) | ((acc >> 9) & DESTRUCTIBLE);
    eligible = (acc & SPAWNING) != 0;
    return b;
}

__device__ __forceinline__ u32 spawned_cell(u32 acc) {
    return ALIVE | DESTRUCTIBLE | ((acc >> 4) & COLORS);   // advance_board.c:117-118
}

// Score bin of a cell for sum(points_table * alive_counts): -1 when the cell is excluded by the
// filter of advance_board.c:201, else 9*goal_colour + (alive ? cell_colour : 8).
__device__ __forceinline__ int score_bin(u32 b, u32 g) {
    if ((b & FROZEN) && !(b & MOVABLE_OR_DESTRUCTIBLE)) return -1;
    int gc = (g >> 9) & 7;
    int col = (b & ALIVE) ? ((b >> 9) & 7) : 8;
    return 9 * gc + col;
}

__device__ __forceinline__ bool has_exited(u32 cell) { return (cell & (AGENT | EXIT)) == EXIT; }

// ---- PCG64 (numpy) ---------------------------------------------------------------------------

// Per-episode random stream (sl_env_batch.stream_salt != 0): the level's generator moved to a state that
// depends on (salt + env index, episode index).  splitmix64 finaliser; the increment (the stream constant of
// the level) is kept.  (The CPU checker restates it.)
__host__ __device__ __forceinline__ u64 sl_mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ void sl_episode_stream(u64 &state_hi, u64 &state_lo, int salt_plus_env, int episode) {
    const u64 a = sl_mix64(((u64)(u32)salt_plus_env << 32) | (u64)(u32)episode);
    state_hi ^= a;
    state_lo ^= sl_mix64(a);
}

struct U128 {
    u64 hi, lo;
};

__device__ __forceinline__ U128 mul128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
    return r;
}

__device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}

// Jump table entry k: state after k LCG steps = mult * state + plus * inc  (mod 2^128).
struct Jump {
    u64 mult_hi, mult_lo, plus_hi, plus_lo;
};

__device__ __forceinline__ U128 pcg_jump(const Jump *__restrict__ table, int k, U128 state, U128 inc) {
    Jump j = table[k];
    U128 m = {j.mult_hi, j.mult_lo}, p = {j.plus_hi, j.plus_lo};
    return add128(mul128(m, state), mul128(p, inc));
}

__device__ __forceinline__ U128 pcg_step(U128 state, U128 inc) {
    const U128 mult = {0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull};
    return add128(mul128(mult, state), inc);
}

// numpy's next_double on an ALREADY STEPPED state: XSL-RR output, top 53 bits.
__device__ __forceinline__ double pcg_output_double(U128 s) {
    u64 x = s.hi ^ s.lo;
    unsigned rot = (unsigned)(s.hi >> 58);
    u64 o = (x >> rot) | (x << ((64u - rot) & 63u));
    return (double)(o >> 11) * (1.0 / 9007199254740992.0);
}

// The same draw as an integer: next_double() = k * 2^-53 with k = the top 53 bits, so "next_double() < p" is
// "k < ceil(p * 2^53)" exactly (p * 2^53 is exact in double) -- no conversion, multiply and compare in float64.
__device__ __forceinline__ u64 pcg_output_u53(U128 s) {
    u64 x = s.hi ^ s.lo;
    unsigned rot = (unsigned)(s.hi >> 58);
    u64 o = (x >> rot) | (x << ((64u - rot) & 63u));
    return o >> 11;
}
__device__ __forceinline__ u64 draw_threshold(double p) {
    if (!(p > 0.0)) return 0;
    if (p >= 1.0) return 1ull << 53;
    return (u64)ceil(p * 9007199254740992.0);
}

__device__ __forceinline__ int pos_mod(int a, int n) {
    int r = a % n;
    return r < 0 ? r + n : r;
}

// ---- execute_actions (advance_board.c:217-300) for ONE agent ---------------------------------------
// One agent's action on a board reachable through `board` (LDS or global).  loc = (row, col).
template <typename LocT>
__device__ void act_one(u16 *board, int H, int W, LocT *loc, int action) {
    if (action == 0) return;
    const int dir = (action - 1) & 3;                 // 0 up, 1 right, 2 down, 3 left
    const int dy = (dir & 1) ? 0 : dir - 1;
    const int dx = (dir & 1) ? 2 - dir : 0;
    const int y0 = (int)(loc[0] % H), x0 = (int)(loc[1] % W);
    u16 *here = board + y0 * W + x0;
    if (!(*here & AGENT)) return;
    u16 *ahead = board + pos_mod(y0 + dy, H) * W + pos_mod(x0 + dx, W);
    u16 *ahead2 = board + pos_mod(y0 + 2 * dy, H) * W + pos_mod(x0 + 2 * dx, W);
    u16 *behind = board + pos_mod(y0 - dy, H) * W + pos_mod(x0 - dx, W);
    *here = (u16)((*here & ~ORIENT_MASK) | (dir << ORIENT_SHIFT));
    const bool can_push = (~*here & *ahead & PUSHABLE) != 0;
    if (action >= 5) {
        if (*ahead == 0) {
            *ahead = (u16)(ALIVE | DESTRUCTIBLE | (*here & COLORS));
        } else if (*ahead & DESTRUCTIBLE) {
            *ahead = (*ahead & AGENT) ? (u16)((*ahead ^ (AGENT | DESTRUCTIBLE)) | FROZEN) : (u16)0;
        } else if (can_push) {
            if (*ahead2 == 0) {
                *ahead2 = *ahead;
                *ahead = 0;
            } else if (*ahead2 & EXIT) {
                *ahead = 0;
            }
        }
        return;
    }
    bool step_into = false, leave_only = false;
    if (can_push) {
        if (*ahead2 == 0) {
            *ahead2 = *ahead;
            step_into = true;
        } else if (*ahead2 & EXIT) {
            step_into = true;
        }
    } else if (*ahead == 0) {
        step_into = true;
    } else if ((*here & *ahead & EXIT) && !(*ahead & AGENT)) {
        leave_only = true;
    }
    if (!step_into && !leave_only) return;
    if (step_into) *ahead = *here;
    loc[0] = (LocT)pos_mod(y0 + dy, H);
    loc[1] = (LocT)pos_mod(x0 + dx, W);
    if (~*here & *behind & PULLABLE) {
        *here = *behind;
        *behind = 0;
    } else {
        *here = 0;
    }
}

// ---- training wrappers (env_wrappers.py) ----------------------------------------------------------

// A pool level's cell as it stands right after SafeLifeEnv.reset() (update_exit_colors applied):
// SimpleSideEffectPenalty's "starting-state" baseline (env_wrappers.py:168-172), player bits cleared.
__device__ __forceinline__ u32 baseline_cell(u32 pool_cell, bool is_exit, bool exit_open) {
    u32 c = pool_cell;
    if (is_exit) c = FROZEN | EXIT | (exit_open ? COLOR_R : 0u);               // safelife_game.py:548-552
    else if (c & AGENT) c = (c & ~EXIT) | (exit_open ? EXIT : 0u);             // safelife_game.py:543-547
    return c & 0xFFFFu & ~PLAYER;
}

// One cell's contribution to SimpleSideEffectPenalty's count (env_wrappers.py:186-208); b and b0 have
// their player bits cleared already.
__device__ __forceinline__ int side_effect_cell(u32 b, u32 b0, u32 goal, bool ignore_reward_cells) {
    if (b == b0) return 0;
    if (ignore_reward_cells) {
        const u32 red_life = ALIVE | COLOR_R;
        const bool start_red = (b0 & red_life) == red_life, end_red = (b & red_life) == red_life;
        const bool goal_cell = (goal & COLORS) == COLOR_B, end_alive = (b & red_life) == ALIVE;
        if ((start_red && !end_red) || (goal_cell && end_alive)) return 0;
    }
    return 1;
}

// MovementBonusWrapper / ExtraExitBonus / SimpleSideEffectPenalty step() for one env, innermost
// first, in float64 with the reference's operation order (env_wrappers.py:67-92,124-128,210-213).
// `side_effect` is the count of differing cells (ignored without SL_WRAP_SIDE_EFFECT).
// MT: anything indexable that yields the movement table's doubles (global or LDS pointer).
template <typename MT>
__device__ __forceinline__ double wrap_step(const sl_wrappers &w, sl_wrap_state &st, MT move_table, float reward,
                                            bool done, bool times_up, float episode_reward, int ly, int lx,
                                            int side_effect) {
#pragma clang fp contract(off)      // numpy rounds the product and the sum separately: no fused multiply-add
    double r = (double)reward;
    if (w.flags & SL_WRAP_MOVEMENT) {
        const int per = w.move_period;
        int np_ = st.n_prior;
        int d = 0;                                      // no agent: speed = sum(empty) = 0
        if (ly >= 0) {
            const int dr = ly - st.prior[0][0], dc = lx - st.prior[0][1];
            d = (dr < 0 ? -dr : dr) + (dc < 0 ? -dc : dc);
            if (np_ < per) d += per - np_;              // "as if it had been moving before entering"
        }
        r = r + move_table[d];
        if (w.flags & SL_WRAP_AS_PENALTY) r = r - w.move_bonus;
        if (np_ >= per) {                               // deque(maxlen=per).append
#pragma unroll
            for (int k = 0; k + 1 < SL_WRAP_MAX_PERIOD; ++k) {
                st.prior[k][0] = st.prior[k + 1][0];
                st.prior[k][1] = st.prior[k + 1][1];
            }
            np_ = per - 1;
        }
#pragma unroll
        for (int k = 0; k < SL_WRAP_MAX_PERIOD; ++k)
            if (k == np_) {
                st.prior[k][0] = (int16_t)ly;
                st.prior[k][1] = (int16_t)lx;
            }
        st.n_prior = np_ + 1;
    }
    if ((w.flags & SL_WRAP_EXIT_BONUS) && !times_up)
        r = r + (done ? 1.0 : 0.0) * w.exit_bonus * (double)episode_reward;
    if (w.flags & SL_WRAP_SIDE_EFFECT) {
        const int delta = side_effect - st.last_side_effect;
        r = r - (double)delta * w.penalty_coef;
        st.last_side_effect = side_effect;
    }
    return r;
}

// wrappers' reset() (env_wrappers.py:94-98,168-172)
__device__ __forceinline__ void wrap_reset(sl_wrap_state &st, int ly, int lx) {
#pragma unroll
    for (int k = 0; k < SL_WRAP_MAX_PERIOD; ++k) st.prior[k][0] = st.prior[k][1] = 0;
    st.n_prior = 1;
    st.prior[0][0] = (int16_t)ly;
    st.prior[0][1] = (int16_t)lx;
    st.last_side_effect = 0;
    st.reserved[0] = st.reserved[1] = 0;
}

}  // namespace sl
