// sl_generic.hip -- size-generic gfx950 kernels: one workgroup (64 to 256 threads) per board, the board
// staged in LDS, one cell per lane per pass.  Correct for every 3 <= H, W with H*W <= SL_MAX_CELLS;
// this is the path for board shapes the row-per-lane kernels (sl_rowlane.hip) do not cover, and
// the implementation of the non-fused primitives (alive_counts, execute_actions, life_occupancy).
//
// Reference behaviour restated here:
//   advance_board        safelife/speedups_src/advance_board.c:34-149
//   life_occupancy       safelife/speedups_src/advance_board.c:153-189
//   alive_counts         safelife/speedups_src/advance_board.c:192-207
//   execute_actions      safelife/speedups_src/advance_board.c:217-300
//   SafeLifeEnv.step     safelife/safelife_env.py:148-201 (+ safelife_game.py:505-552,684-719,746-761)
//   SafeLifeEnv.reset    safelife/safelife_env.py:203-218
//   SafeLifeEnv.get_obs  safelife/safelife_env.py:105-146, helper_utils.py:42-75
#include "sl_device.h"
#include "sl_kernels.h"

namespace sl {

constexpr int GB_MAX = 256;      // threads per workgroup, at most
constexpr int GW = GB_MAX / 64;  // waves per workgroup, at most
// Threads per workgroup (= per board): one wavefront for boards of up to 256 cells, two up to 1024, else four -- a
// 13x13 board left more than half of a 256-thread workgroup's lanes idle and paid four waves' barriers for every
// phase (round 4: 185 -> see profiles/round4_i_board_shapes.txt).  Device code reads the size it was launched with.
#define GB ((int)blockDim.x)
static inline unsigned generic_threads(int HW) { return HW <= 256 ? 64u : HW <= 1024 ? 128u : 256u; }

__device__ __forceinline__ int row_of(int i, int W, float inv_w) {
    int y = (int)(((float)i + 0.5f) * inv_w);
    if (y * W > i) --y;
    if ((y + 1) * W <= i) ++y;
    return y;
}

// ---------------------------------------------------------------------------------------------
// One CA step of the board held in LDS `cur` into LDS `nxt` (`rows` is scratch), all GB threads.
// rng: LDS {state_hi, state_lo, inc_hi, inc_lo}; advanced by the number of draws.
// Ends with every write to nxt/rng visible to the whole workgroup.
// ---------------------------------------------------------------------------------------------
__device__ void ca_step_block(const u16 *cur, u16 *rows, u16 *nxt, int H, int W, float inv_w,
                              double p, u64 *rng, const Jump *__restrict__ jump, int *wave_tot) {
    const int HW = H * W;
    const int tid = threadIdx.x;

    for (int i = tid; i < HW; i += GB) {
        int y = row_of(i, W, inv_w), x = i - y * W;
        const u16 *r = cur + y * W;
        int xl = x ? x - 1 : W - 1, xr = (x + 1 < W) ? x + 1 : 0;
        rows[i] = (u16)merge3(cell_summary(r[xl]), cell_summary(r[x]), cell_summary(r[xr]));
    }
    __syncthreads();

    u64 mask = 0;
    int it = 0;
    for (int i = tid; i < HW; i += GB, ++it) {
        int up = i - W, dn = i + W;
        if (up < 0) up += HW;
        if (dn >= HW) dn -= HW;
        u32 acc = merge3(rows[up], rows[i], rows[dn]);
        bool el;
        u32 r = apply_rule(cur[i], acc, el);
        nxt[i] = (u16)(el ? acc : r);          // eligible cells park their accumulator in nxt
        mask |= (u64)el << it;
    }
    if (!__syncthreads_or(mask != 0)) return;

    // Random draws in row-major order (advance_board.c:115): exclusive prefix count of the
    // eligibility flags, then a table jump of the board's PCG64 stream per eligible cell.
    U128 st = {rng[0], rng[1]}, inc = {rng[2], rng[3]};
    const int lane = tid & 63, wave = tid >> 6;
    int base = 0;
    it = 0;
    for (int i0 = 0; i0 < HW; i0 += GB, ++it) {
        bool el = (mask >> it) & 1;
        u64 bal = __ballot(el);
        int lane_pref = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < GW; ++w) {
            if (w * 64 >= GB) break;
            int t = wave_tot[w];
            woff += (w < wave) ? t : 0;
            tot += t;
        }
        if (el) {
            int i = i0 + tid;
            U128 s = pcg_jump(jump, base + woff + lane_pref + 1, st, inc);
            u32 acc = nxt[i];
            nxt[i] = (pcg_output_double(s) < p) ? (u16)spawned_cell(acc) : cur[i];
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) {
        U128 s = pcg_jump(jump, base, st, inc);
        rng[0] = s.hi;
        rng[1] = s.lo;
    }
    __syncthreads();
}

struct GenericLds {
    u64 *rng;        // [4]
    int *wave_tot;   // [GW]
    int *ivar;       // [16] misc workgroup-shared ints
    u16 *buf[4];
};

__device__ __forceinline__ GenericLds carve(unsigned char *smem, int HW, int nbuf) {
    GenericLds l;
    l.rng = (u64 *)smem;
    l.wave_tot = (int *)(smem + 32);
    l.ivar = (int *)(smem + 64);
    int hwp = (HW + 7) & ~7;
    u16 *b = (u16 *)(smem + 128);
    for (int k = 0; k < 4; ++k) l.buf[k] = (k < nbuf) ? b + (size_t)k * hwp : nullptr;
    return l;
}

size_t generic_lds_bytes(int HW, int nbuf) { return 128 + (size_t)nbuf * ((HW + 7) & ~7) * sizeof(u16); }

// ------------------------------------------------------------------ advance_board / occupancy

__global__ __launch_bounds__(GB_MAX) void k_advance_generic(const u16 *__restrict__ in, u16 *__restrict__ out,
                                                        int H, int W, const float *__restrict__ spawn_prob,
                                                        int n_steps, sl_pcg64 *rng,
                                                        const Jump *__restrict__ jump,
                                                        int32_t *__restrict__ occupancy,
                                                        const int32_t *__restrict__ n_each) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W, b = blockIdx.x, tid = threadIdx.x;
    if (n_each) n_steps = n_each[b];            // per-board step counts (workgroup-uniform)
    GenericLds l = carve(smem, HW, 3);
    const u16 *src = in + (size_t)b * HW;
    for (int i = tid; i < HW; i += GB) l.buf[0][i] = src[i];
    if (tid < 4) l.rng[tid] = ((const u64 *)(rng + b))[tid];
    __syncthreads();
    const float inv_w = 1.0f / (float)W;
    const double p = (double)spawn_prob[b];
    u16 *cur = l.buf[0], *nxt = l.buf[2];
    int32_t *occ = occupancy ? occupancy + (size_t)b * HW * 8 : nullptr;
    if (occ)
        for (int i = tid; i < HW * 8; i += GB) occ[i] = 0;
    for (int s = 0; s < n_steps; ++s) {
        ca_step_block(cur, l.buf[1], nxt, H, W, inv_w, p, l.rng, jump, l.wave_tot);
        if (occ) {
            // each cell is owned by one thread for the whole launch: plain read-modify-write
            for (int i = tid; i < HW; i += GB) {
                u32 c = nxt[i];
                if ((c & ALIVE) && !(c & (AGENT | EXIT | FROZEN))) occ[8 * i + ((c >> 9) & 7)] += 1;
            }
        }
        u16 *t = cur;
        cur = nxt;
        nxt = t;
    }
    if (out) {
        u16 *dst = out + (size_t)b * HW;
        for (int i = tid; i < HW; i += GB) dst[i] = cur[i];
    }
    if (tid < 4) ((u64 *)(rng + b))[tid] = l.rng[tid];
}

// SimpleSideEffectPenalty's "inaction" baseline (env_wrappers.py:179-180): every env's baseline board one CA step on,
// with the baseline's own generator.  An env that has not stepped since its reset (num_steps == 0) starts from its
// current board: the wrapper's reset() copies it (:168-172).
__global__ __launch_bounds__(GB_MAX) void k_inaction_generic(sl_env_batch env, const Jump *__restrict__ jump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = env.H, W = env.W, HW = H * W, b = blockIdx.x, tid = threadIdx.x;
    GenericLds l = carve(smem, HW, 3);
    const sl_env_scalars *sc = env.scalars + b;
    u16 *base = env.wrap.inaction_board + (size_t)b * HW;
    const u16 *src = sc->num_steps == 0 ? env.board + (size_t)b * HW : base;
    for (int i = tid; i < HW; i += GB) l.buf[0][i] = src[i];
    if (tid < 4) l.rng[tid] = ((const u64 *)(env.wrap.inaction_rng + b))[tid];
    __syncthreads();
    ca_step_block(l.buf[0], l.buf[1], l.buf[2], H, W, 1.0f / (float)W, (double)sc->spawn_prob, l.rng, jump, l.wave_tot);
    for (int i = tid; i < HW; i += GB) base[i] = l.buf[2][i];
    if (tid < 4) ((u64 *)(env.wrap.inaction_rng + b))[tid] = l.rng[tid];
}

// ------------------------------------------------------------------------------ alive_counts

__global__ __launch_bounds__(GB_MAX) void k_alive_counts(const u16 *__restrict__ board,
                                                     const u16 *__restrict__ goals, int HW,
                                                     int64_t *__restrict__ out) {
    __shared__ int hist[72];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 72; i += GB) hist[i] = 0;
    __syncthreads();
    const u16 *bd = board + (size_t)b * HW, *gl = goals + (size_t)b * HW;
    for (int i = tid; i < HW; i += GB) {
        int bin = score_bin(bd[i], gl[i]);
        if (bin >= 0) atomicAdd(&hist[bin], 1);
    }
    __syncthreads();
    for (int i = tid; i < 72; i += GB) out[(size_t)b * 72 + i] = hist[i];
}

// --------------------------------------------------------------------------- execute_actions

__global__ void k_execute_actions(u16 *board, int B, int H, int W, int64_t *locs,
                                  const int64_t *__restrict__ actions, int A, int action_stride,
                                  int action_batch_stride) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    u16 *bd = board + (size_t)b * H * W;
    for (int k = 0; k < A; ++k) {
        int a = (int)actions[(size_t)b * action_batch_stride + (size_t)k * action_stride];
        act_one<int64_t>(bd, H, W, locs + ((size_t)b * A + k) * 2, a);
    }
}

// ------------------------------------------------------------------------------ env helpers

__device__ __forceinline__ u32 obs_word(u32 b, u32 g, int remove_white) {
    u32 gc = g & COLORS;
    if (remove_white && gc == COLORS) gc = 0;
    return b | (gc << 16);
}

// Sum over the workgroup of an int held by every thread; result broadcast to all threads.
__device__ int block_sum(int v, int *wave_tot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < GW && w * 64 < GB; ++w) s += wave_tot[w];
    return s;
}

// sum(points_table * alive_counts) of the board in LDS against goals (LDS or global).
__device__ int board_score(const u16 *board, const u16 *goals, int HW,
                           const int32_t *__restrict__ table, int *wave_tot) {
    int s = 0;
    for (int i = threadIdx.x; i < HW; i += GB) {
        int bin = score_bin(board[i], goals[i]);
        if (bin >= 0) s += table[bin];
    }
    return block_sum(s, wave_tot);
}

// GameState.update_exit_colors (safelife_game.py:537-552), single agent, by thread 0.
__device__ bool recolor_exits(u16 *board, int W, int ly, int lx, const int32_t *exits, int E,
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
    u16 paint = (u16)(FROZEN | EXIT | (any_can ? COLOR_R : 0u));
    for (int k = 0; k < E; ++k) {
        int ex = exits[k];
        if (ex >= 0) board[ex] = paint;
    }
    return any_can;
}

// SafeLifeEnv.get_obs for env e from the board in LDS; goals through `goals` (LDS or global).
__device__ void write_obs(const sl_env_batch &env, int e, const u16 *board, const u16 *goals,
                          int ly, int lx, const int32_t *exits, uint8_t *obs_base = nullptr, int slot = -1,
                          uint8_t *stage = nullptr) {
    // (obs_base / slot: the multi-agent kernels write agent a of env e to slot e * A + a of their own tensor; stage: room
    //  in LDS for one observation's view_h x view_w x C bytes -- the channel bytes are parked there and leave as dwords)
    uint8_t *const obs_out = obs_base ? obs_base : env.obs;
    if (slot >= 0) e = slot;
    if (!obs_out && !env.policy_obs) return;
    const int H = env.H, W = env.W, vh = env.view_h, vw = env.view_w, C = env.n_channels;
    const int y0 = ly >= 0 ? ly : 0, x0 = ly >= 0 ? lx : 0;
    const int nv = vh * vw;
    const float inv_vw = 1.0f / (float)vw;
    const float inv_w = 1.0f / (float)W;
    for (int v = threadIdx.x; v < nv; v += GB) {
        int vy = row_of(v, vw, inv_vw), vx = v - vy * vw;
        int sy = pos_mod(y0 - vh / 2 + vy, H), sx = pos_mod(x0 - vw / 2 + vx, W);
        int s = sy * W + sx;
        u32 word = obs_word(board[s], goals[s], env.remove_white_goals);
        // exits that fall outside the view are painted on its perimeter (helper_utils.py:64-74);
        // later exits overwrite earlier ones, as numpy's fancy assignment does
        for (int k = 0; k < env.E; ++k) {
            int ex = exits[k];
            if (ex < 0) continue;
            int iy = row_of(ex, W, inv_w), ix = ex - iy * W;
            int jy = pos_mod(iy - y0 + H / 2, H) - H / 2 + vh / 2;
            int jx = pos_mod(ix - x0 + W / 2, W) - W / 2 + vw / 2;
            jy = min(max(jy, 0), vh - 1);
            jx = min(max(jx, 0), vw - 1);
            if (jy == vy && jx == vx) word = obs_word(board[ex], goals[ex], env.remove_white_goals);
        }
        if (obs_out) {
            if (C == 0) {
                ((u32 *)obs_out)[(size_t)e * nv + v] = word;
            } else {
                uint8_t *o = stage ? stage + (size_t)v * C : obs_out + ((size_t)e * nv + v) * C;
                for (int c = 0; c < C; ++c) o[c] = (word >> env.channels[c]) & 1u;
            }
        }
        if (env.policy_obs && slot < 0) {        // channel-first, spatial axes swapped: [C, vw, vh] (training/models.py:100-103)
            const size_t base = (size_t)e * C * nv + (size_t)vx * vh + vy;
            for (int c = 0; c < C; ++c) {
                const u32 bit = (word >> env.channels[c]) & 1u;
                if (env.policy_dtype == 0) ((uint8_t *)env.policy_obs)[base + (size_t)c * nv] = (uint8_t)bit;
                else ((float *)env.policy_obs)[base + (size_t)c * nv] = (float)bit;
            }
        }
    }
    if (stage && obs_out && C > 0) {
        // the parked observation -> global memory: the bytes up to the first 4-byte boundary of the destination one by
        // one, then whole dwords (assembled from four LDS bytes: the stage is not aligned with the destination), then
        // the tail.  15 byte stores per cell become 15/4 dword stores.
        __syncthreads();
        const int total = nv * C;
        uint8_t *dst = obs_out + (size_t)e * total;
        const int head = min(total, (int)((4 - ((size_t)dst & 3)) & 3));
        const int nd = (total - head) >> 2;
        for (int i = threadIdx.x; i < head; i += GB) dst[i] = stage[i];
        u32 *d4 = (u32 *)(dst + head);
        const uint8_t *s1 = stage + head;
        for (int k = threadIdx.x; k < nd; k += GB)
            d4[k] = (u32)s1[4 * k] | ((u32)s1[4 * k + 1] << 8) | ((u32)s1[4 * k + 2] << 16) | ((u32)s1[4 * k + 3] << 24);
        for (int i = head + 4 * nd + threadIdx.x; i < total; i += GB) dst[i] = stage[i];
        __syncthreads();            // (the next agent's observation reuses the stage)
    }
}

// SafeLifeEnv.reset() body for env e: pool level -> LDS board (`brd`) and global per-env state.
// ivar[0..1] receives the agent location.  All threads participate.
__device__ void reset_block(const sl_env_batch &env, int e, u16 *brd, int *ivar, int *wave_tot) {
    const int HW = env.H * env.W, tid = threadIdx.x, E = env.E;
    sl_env_scalars *sc = env.scalars + e;
    const int l = sc->level_idx;
    const sl_level_scalars lv = env.pool_scalars[l];
    const u16 *pb = env.pool_board + (size_t)l * HW, *pg = env.pool_goals + (size_t)l * HW;
    u16 *gdst = env.goals + (size_t)e * HW;
    for (int i = tid; i < HW; i += GB) {
        brd[i] = pb[i];
        gdst[i] = pg[i];
    }
    for (int k = tid; k < E; k += GB) env.exit_locs[(size_t)e * E + k] = env.pool_exit_locs[(size_t)l * E + k];
    if (tid == 0) {
        ivar[0] = lv.agent_row;
        ivar[1] = lv.agent_col;
        sl_pcg64 gen = env.pool_rng[l];
        if (env.stream_salt) sl_episode_stream(gen.state_hi, gen.state_lo, env.stream_salt + e, sc->episode_idx);
        env.rng[e] = gen;
    }
    __syncthreads();
    const int32_t *table = env.points_table + 72 * lv.table_idx;
    int score = board_score(brd, pg, HW, table, wave_tot);
    if (tid == 0) {
        const bool open = recolor_exits(brd, env.W, ivar[0], ivar[1], env.pool_exit_locs + (size_t)l * E, E,
                                        score, lv.initial_points, lv.required_reset, env.exit_points);
        int exited = ivar[0] >= 0 ? (has_exited(brd[ivar[0] * env.W + ivar[1]]) ? 1 : 0) : 0;
        sl_env_scalars n;
        n.agent_row = ivar[0];
        n.agent_col = ivar[1];
        n.num_steps = 0;
        n.old_value = score + env.exit_points * exited;
        n.required_points = lv.required_step;
        n.initial_points = lv.initial_points;
        n.table_idx = lv.table_idx;
        n.level_idx = l;
        n.episode_idx = sc->episode_idx;
        n.episode_length = 0;
        n.episode_reward = 0.0f;
        n.spawn_prob = lv.spawn_prob;
        n.goals_static = 0;
        n.is_active = 1;
        n.exit_open_at_reset = open ? 1 : 0;
        n.loaded = 1;
        *sc = n;
        if (env.wrap.flags) wrap_reset(env.wrap.state[e], ivar[0], ivar[1]);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------- env step / reset

__global__ __launch_bounds__(GB_MAX) void k_env_rollout_generic(sl_env_batch env,
                                                            const int32_t *__restrict__ actions, int T,
                                                            float *__restrict__ reward_t,
                                                            uint8_t *__restrict__ done_t,
                                                            const Jump *__restrict__ jump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = env.H, W = env.W, HW = H * W, E = env.E;
    const int e = blockIdx.x, tid = threadIdx.x;
    GenericLds l = carve(smem, HW, 4);
    u16 *cur = l.buf[0], *rows = l.buf[1], *nxt = l.buf[2], *aux = l.buf[3];
    int *ivar = l.ivar;   // [0],[1] agent loc; [2] done flag
    u16 *gboard = env.board + (size_t)e * HW;
    u16 *ggoals = env.goals + (size_t)e * HW;
    sl_env_scalars *sc = env.scalars + e;
    const int32_t *exits = env.exit_locs + (size_t)e * E;
    const float inv_w = 1.0f / (float)W;

    for (int i = tid; i < HW; i += GB) cur[i] = gboard[i];
    if (tid < 4) l.rng[tid] = ((const u64 *)(env.rng + e))[tid];
    if (tid == 0) {
        ivar[0] = sc->agent_row;
        ivar[1] = sc->agent_col;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        // safelife_env.py:151  game.execute_actions
        if (tid == 0 && ivar[0] >= 0) act_one<int>(cur, H, W, ivar, actions[(size_t)t * env.B + e]);
        __syncthreads();
        // safelife_env.py:152  game.advance_board (board, then goals unless static)
        const double p = (double)sc->spawn_prob;
        ca_step_block(cur, rows, nxt, H, W, inv_w, p, l.rng, jump, l.wave_tot);
        const u16 *goals = ggoals;
        const int gstatic = sc->goals_static;
        if (gstatic != 1) {
            for (int i = tid; i < HW; i += GB) cur[i] = ggoals[i];
            __syncthreads();
            ca_step_block(cur, rows, aux, H, W, inv_w, p, l.rng, jump, l.wave_tot);
            int changed = 0;
            for (int i = tid; i < HW; i += GB) {
                u16 g = aux[i];
                changed |= (g != cur[i]) || (g & SPAWNING);
                ggoals[i] = g;
            }
            changed = __syncthreads_or(changed);
            if (tid == 0 && gstatic == 0) sc->goals_static = changed ? 2 : 1;
            goals = aux;
        }
        // safelife_env.py:153-160
        const int32_t *table = env.points_table + 72 * sc->table_idx;
        int score = board_score(nxt, goals, HW, table, l.wave_tot);
        if (tid == 0) {
            recolor_exits(nxt, W, ivar[0], ivar[1], exits, E, score, sc->initial_points, sc->required_points,
                          env.exit_points);
            int steps = sc->num_steps + 1;
            sc->num_steps = steps;
            bool times_up = steps >= env.time_limit;
            float reward = 0.0f;
            bool done = true, success = false;
            bool active = sc->is_active != 0;
            if (ivar[0] >= 0) {
                u32 cell = nxt[ivar[0] * W + ivar[1]];
                success = has_exited(cell);
                int value = score + env.exit_points * (success ? 1 : 0);
                reward = (float)((value - sc->old_value) * (active ? 1 : 0));
                sc->old_value = value;
                done = !(cell & AGENT) || times_up;
            }
            // safelife_env.py:172-175
            float ep_r = sc->episode_reward + reward;
            int ep_l = sc->episode_length + (active ? 1 : 0);
            sc->episode_reward = ep_r;
            sc->episode_length = ep_l;
            sc->is_active = (active && !done) ? 1 : 0;
            sl_step_out o;
            o.reward = reward;
            o.done = done;
            o.success = success;
            o.times_up = times_up;
            o.reserved = 0;
            o.episode_reward = ep_r;
            o.episode_length = ep_l;
            if (env.out_compact)
                ((unsigned long long *)env.out)[e] =
                    (unsigned long long)__float_as_uint(reward) |
                    ((unsigned long long)((done ? 1u : 0u) | ((success ? 1u : 0u) << 8) | ((times_up ? 1u : 0u) << 16)) << 32);
            else env.out[e] = o;
            if (reward_t) reward_t[(size_t)t * env.B + e] = reward;
            if (done_t) done_t[(size_t)t * env.B + e] = done;
            ivar[2] = done;
            ivar[3] = __float_as_int(reward);
            ivar[4] = times_up;
            ivar[5] = __float_as_int(ep_r);
            // the step that ends the episode queues it for the side-effect pass (include/safelife_hip.h)
            int slot = -1;
            if (env.finished.capacity > 0 && done && active) {
                slot = atomicAdd(env.finished.count, 1);
                if (slot < env.finished.capacity) {
                    sl_episode_record rec;
                    rec.env = e + env.finished.env_base;
                    rec.level = sc->level_idx;
                    rec.num_steps = steps;
                    rec.episode_idx = sc->episode_idx;
                    rec.spawn_prob = sc->spawn_prob;
                    rec.episode_reward = ep_r;
                    rec.episode_length = ep_l;
                    rec.success = success;
                    rec.times_up = times_up;
                    rec.n_cell_types = rec.reserved = 0;
                    env.finished.records[slot] = rec;
                } else {
                    slot = -1;
                }
            }
            ivar[6] = slot;
        }
        __syncthreads();
        if (ivar[6] >= 0) {
            u16 *dst = env.finished.boards + (size_t)ivar[6] * HW;
            for (int i = tid; i < HW; i += GB) dst[i] = nxt[i];
        }
        if (env.wrap.flags) {       // env_wrappers.py: movement bonus, exit bonus, side-effect penalty
            int side = 0;
            if (env.wrap.flags & SL_WRAP_SIDE_EFFECT) {
                // exit cells never count (env_wrappers.py:191-193): flag them in the free `rows` buffer
                for (int i = tid; i < HW; i += GB) rows[i] = 0;
                __syncthreads();
                for (int k = tid; k < E; k += GB)
                    if (exits[k] >= 0) rows[exits[k]] = 1;
                __syncthreads();
                const u16 *pb = env.pool_board + (size_t)sc->level_idx * HW;
                // "inaction": the baseline board k_inaction_generic advanced just before this launch
                const u16 *ib = (env.wrap.flags & SL_WRAP_INACTION) ? env.wrap.inaction_board + (size_t)e * HW : nullptr;
                const bool open = sc->exit_open_at_reset != 0;
                const bool ignore = (env.wrap.flags & SL_WRAP_IGNORE_REWARD_CELLS) != 0;
                int mine = 0;
                for (int i = tid; i < HW; i += GB)
                    if (!rows[i])
                        mine += side_effect_cell(nxt[i] & 0xFFFFu & ~PLAYER,
                                                 ib ? (u32)ib[i] & ~PLAYER : baseline_cell(pb[i], false, open),
                                                 goals[i], ignore);
                side = block_sum(mine, l.wave_tot);
            }
            if (tid == 0) {
                const double shaped = wrap_step(env.wrap, env.wrap.state[e], env.wrap.move_table,
                                                __int_as_float(ivar[3]), ivar[2] != 0, ivar[4] != 0,
                                                __int_as_float(ivar[5]), ivar[0], ivar[1], side);
                env.wrap.shaped_reward[e] = shaped;
                if (env.wrap.shaped_reward_t) env.wrap.shaped_reward_t[(size_t)t * env.B + e] = shaped;
            }
            __syncthreads();
        }
        u16 *sw = cur;
        cur = nxt;
        nxt = sw;
        if (env.auto_reset && ivar[2]) {
            if (tid == 0) {
                sc->level_idx = env.pool_next ? env.pool_next[sc->level_idx] : (sc->level_idx + env.level_stride) % env.L;
                sc->episode_idx += 1;
            }
            __syncthreads();
            reset_block(env, e, cur, ivar, l.wave_tot);
            if (tid < 4) l.rng[tid] = ((const u64 *)(env.rng + e))[tid];
            __syncthreads();
        }
    }

    for (int i = tid; i < HW; i += GB) gboard[i] = cur[i];
    if (tid < 4) ((u64 *)(env.rng + e))[tid] = l.rng[tid];
    if (tid == 0) {
        sc->agent_row = ivar[0];
        sc->agent_col = ivar[1];
    }
    __syncthreads();   // goals written by this workgroup are read back below
    write_obs(env, e, cur, ggoals, ivar[0], ivar[1], exits);
}

// ------------------------------------------------------------------------- multi-agent env step / reset
// SafeLifeEnv(single_agent=False): include/safelife_hip.h, sl_multi_agent.  One workgroup per board; what concerns a
// single agent is serial work of thread 0, in agent order (advance_board.c:217-220; safelife_game.py:537-552).

// per-agent words shared by the workgroup, behind the four board buffers of the dynamic LDS
struct MultiLds {
    int *loc;       // [2 A]
    int *score;     // [A]  sum(points_table[a] * alive_counts)
    int *flag;      // [0] every agent is done
    uint8_t *stage; // one observation's channel bytes (write_obs), or null
};
__device__ __forceinline__ MultiLds carve_multi(unsigned char *smem, int HW, bool staged) {
    int *base = (int *)(smem + 128 + (size_t)4 * ((HW + 7) & ~7) * sizeof(u16));        // (= generic_lds_bytes(HW, 4))
    return MultiLds{base, base + 2 * SL_MAX_AGENTS, base + 3 * SL_MAX_AGENTS,
                    staged ? (uint8_t *)(base + 4 * SL_MAX_AGENTS) : nullptr};
}
static size_t multi_stage_bytes(const sl_env_batch &env, const sl_multi_agent &m) {
    return (m.obs && env.n_channels > 0) ? (((size_t)env.view_h * env.view_w * env.n_channels + 15) & ~(size_t)15) : 0;
}
static size_t multi_lds_bytes(const sl_env_batch &env, const sl_multi_agent &m) {
    return generic_lds_bytes(env.H * env.W, 4) + 4 * SL_MAX_AGENTS * sizeof(int) + multi_stage_bytes(env, m);
}

// GameState.update_exit_colors for A agents (safelife_game.py:537-552), by thread 0: every agent's cell gets the EXIT bit
// iff THAT agent may leave (its own points against its own requirement); the exits turn red when any agent may.
__device__ bool recolor_exits_multi(u16 *board, int W, const int *loc, const int *score, int A, const sl_agent_state *ag,
                                    const int *required, const int32_t *exits, int E, int exit_points) {
    bool can[SL_MAX_AGENTS];
    bool any_can = false;
    for (int a = 0; a < A; ++a) {           // can_exit() is evaluated on the board as it stands, for all agents, first
        const u32 cell = board[loc[2 * a] * W + loc[2 * a + 1]];
        int earned = score[a] - ag[a].initial_points + exit_points * (has_exited(cell) ? 1 : 0);
        if (earned < 0) earned = 0;
        can[a] = (cell & AGENT) && earned >= required[a];
        any_can = any_can || can[a];
    }
    for (int a = 0; a < A; ++a) {
        u16 *cell = board + loc[2 * a] * W + loc[2 * a + 1];
        *cell = (u16)((*cell & ~EXIT) | (can[a] ? EXIT : 0u));
    }
    const u16 paint = (u16)(FROZEN | EXIT | (any_can ? COLOR_R : 0u));
    for (int k = 0; k < E; ++k) {
        const int ex = exits[k];
        if (ex >= 0) board[ex] = paint;
    }
    return any_can;
}

// SafeLifeEnv.reset() body of a multi-agent env: pool level -> LDS board and the per-env / per-agent state.
__device__ void reset_block_multi(const sl_env_batch &env, const sl_multi_agent &m, int e, u16 *brd, MultiLds ml, int *wave_tot) {
    const int HW = env.H * env.W, tid = threadIdx.x, E = env.E, A = m.n_agents;
    sl_env_scalars *sc = env.scalars + e;
    const int l = sc->level_idx;
    const sl_level_scalars lv = env.pool_scalars[l];
    const sl_level_agent *pa = m.pool_agents + (size_t)l * A;
    sl_agent_state *ag = m.agents + (size_t)e * A;
    const u16 *pb = env.pool_board + (size_t)l * HW, *pg = env.pool_goals + (size_t)l * HW;
    u16 *gdst = env.goals + (size_t)e * HW;
    for (int i = tid; i < HW; i += GB) {
        brd[i] = pb[i];
        gdst[i] = pg[i];
    }
    for (int k = tid; k < E; k += GB) env.exit_locs[(size_t)e * E + k] = env.pool_exit_locs[(size_t)l * E + k];
    if (tid < A) {
        ml.loc[2 * tid] = pa[tid].row;
        ml.loc[2 * tid + 1] = pa[tid].col;
    }
    if (tid == 0) {
        sl_pcg64 gen = env.pool_rng[l];
        if (env.stream_salt) sl_episode_stream(gen.state_hi, gen.state_lo, env.stream_salt + e, sc->episode_idx);
        env.rng[e] = gen;
    }
    __syncthreads();
    for (int a = 0; a < A; ++a) {
        const int s = board_score(brd, pg, HW, env.points_table + 72 * pa[a].table_idx, wave_tot);
        if (tid == 0) ml.score[a] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int required[SL_MAX_AGENTS];
        for (int a = 0; a < A; ++a) {
            ag[a].initial_points = pa[a].initial_points;
            required[a] = pa[a].required_reset;
        }
        recolor_exits_multi(brd, env.W, ml.loc, ml.score, A, ag, required, env.pool_exit_locs + (size_t)l * E, E, env.exit_points);
        for (int a = 0; a < A; ++a) {
            const int exited = has_exited(brd[ml.loc[2 * a] * env.W + ml.loc[2 * a + 1]]) ? 1 : 0;
            sl_agent_state n;
            n.row = ml.loc[2 * a];
            n.col = ml.loc[2 * a + 1];
            n.old_value = ml.score[a] + env.exit_points * exited;
            n.required_points = pa[a].required_step;
            n.initial_points = pa[a].initial_points;
            n.table_idx = pa[a].table_idx;
            n.episode_length = 0;
            n.episode_reward = 0.0f;
            n.is_active = 1;
            n.reserved[0] = n.reserved[1] = n.reserved[2] = 0;
            ag[a] = n;
        }
        sl_env_scalars n0 = *sc;
        n0.agent_row = ml.loc[0];
        n0.agent_col = ml.loc[1];
        n0.num_steps = 0;
        n0.old_value = ag[0].old_value;
        n0.required_points = ag[0].required_points;
        n0.initial_points = ag[0].initial_points;
        n0.table_idx = ag[0].table_idx;
        n0.episode_length = 0;
        n0.episode_reward = 0.0f;
        n0.spawn_prob = lv.spawn_prob;
        n0.goals_static = 0;
        n0.is_active = 1;
        n0.exit_open_at_reset = 0;
        n0.loaded = 1;
        *sc = n0;
    }
    __syncthreads();
}

__global__ __launch_bounds__(GB_MAX) void k_env_step_multi(sl_env_batch env, sl_multi_agent m,
                                                       const int32_t *__restrict__ actions,
                                                       const Jump *__restrict__ jump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = env.H, W = env.W, HW = H * W, E = env.E, A = m.n_agents;
    const int e = blockIdx.x, tid = threadIdx.x;
    GenericLds l = carve(smem, HW, 4);
    MultiLds ml = carve_multi(smem, HW, m.obs && env.n_channels > 0);
    u16 *cur = l.buf[0], *rows = l.buf[1], *nxt = l.buf[2], *aux = l.buf[3];
    u16 *gboard = env.board + (size_t)e * HW;
    u16 *ggoals = env.goals + (size_t)e * HW;
    sl_env_scalars *sc = env.scalars + e;
    sl_agent_state *ag = m.agents + (size_t)e * A;
    const int32_t *exits = env.exit_locs + (size_t)e * E;
    const float inv_w = 1.0f / (float)W;

    for (int i = tid; i < HW; i += GB) cur[i] = gboard[i];
    if (tid < 4) l.rng[tid] = ((const u64 *)(env.rng + e))[tid];
    if (tid < A) {
        ml.loc[2 * tid] = ag[tid].row;
        ml.loc[2 * tid + 1] = ag[tid].col;
    }
    __syncthreads();
    // safelife_env.py:151  game.execute_actions: every agent's action, in index order (advance_board.c:217-220)
    if (tid == 0)
        for (int a = 0; a < A; ++a) act_one<int>(cur, H, W, ml.loc + 2 * a, actions[(size_t)e * A + a]);
    __syncthreads();
    // safelife_env.py:152  game.advance_board (board, then goals unless static)
    const double p = (double)sc->spawn_prob;
    ca_step_block(cur, rows, nxt, H, W, inv_w, p, l.rng, jump, l.wave_tot);
    const u16 *goals = ggoals;
    const int gstatic = sc->goals_static;
    if (gstatic != 1) {
        for (int i = tid; i < HW; i += GB) cur[i] = ggoals[i];
        __syncthreads();
        ca_step_block(cur, rows, aux, H, W, inv_w, p, l.rng, jump, l.wave_tot);
        int changed = 0;
        for (int i = tid; i < HW; i += GB) {
            u16 g = aux[i];
            changed |= (g != cur[i]) || (g & SPAWNING);
            ggoals[i] = g;
        }
        changed = __syncthreads_or(changed);
        if (tid == 0 && gstatic == 0) sc->goals_static = changed ? 2 : 1;
        goals = aux;
    }
    // safelife_env.py:153-160, per agent: its own table against the shared counts (safelife_game.py:684-687)
    for (int a = 0; a < A; ++a) {
        const int s = board_score(nxt, goals, HW, env.points_table + 72 * ag[a].table_idx, l.wave_tot);
        if (tid == 0) ml.score[a] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int required[SL_MAX_AGENTS];
        for (int a = 0; a < A; ++a) required[a] = ag[a].required_points;
        recolor_exits_multi(nxt, W, ml.loc, ml.score, A, ag, required, exits, E, env.exit_points);
        const int steps = sc->num_steps + 1;
        sc->num_steps = steps;
        const bool times_up = steps >= env.time_limit;
        bool all_done = true;
        for (int a = 0; a < A; ++a) {
            const u32 cell = nxt[ml.loc[2 * a] * W + ml.loc[2 * a + 1]];
            const bool success = has_exited(cell);
            const bool active = ag[a].is_active != 0;
            const int value = ml.score[a] + env.exit_points * (success ? 1 : 0);
            const float reward = (float)((value - ag[a].old_value) * (active ? 1 : 0));
            const bool done = !(cell & AGENT) || times_up;
            sl_step_out o;
            o.reward = reward;
            o.done = done;
            o.success = success;
            o.times_up = times_up;
            o.reserved = 0;
            o.episode_reward = ag[a].episode_reward + reward;
            o.episode_length = ag[a].episode_length + (active ? 1 : 0);
            m.out[(size_t)e * A + a] = o;
            ag[a].row = ml.loc[2 * a];
            ag[a].col = ml.loc[2 * a + 1];
            ag[a].old_value = value;
            ag[a].episode_reward = o.episode_reward;
            ag[a].episode_length = o.episode_length;
            ag[a].is_active = (active && !done) ? 1 : 0;
            all_done = all_done && done;
        }
        sc->agent_row = ml.loc[0];
        sc->agent_col = ml.loc[1];
        ml.flag[0] = all_done ? 1 : 0;
    }
    __syncthreads();
    u16 *sw = cur;
    cur = nxt;
    nxt = sw;
    if (env.auto_reset && ml.flag[0]) {     // training/base_algo.py:231-236: a fresh level once every agent is done
        if (tid == 0) {
            sc->level_idx = env.pool_next ? env.pool_next[sc->level_idx] : (sc->level_idx + env.level_stride) % env.L;
            sc->episode_idx += 1;
        }
        __syncthreads();
        reset_block_multi(env, m, e, cur, ml, l.wave_tot);
        if (tid < 4) l.rng[tid] = ((const u64 *)(env.rng + e))[tid];
        __syncthreads();
    }
    for (int i = tid; i < HW; i += GB) gboard[i] = cur[i];
    if (tid < 4) ((u64 *)(env.rng + e))[tid] = l.rng[tid];
    __syncthreads();   // goals written by this workgroup are read back below
    if (m.obs)
        for (int a = 0; a < A; ++a)
            write_obs(env, e, cur, ggoals, ml.loc[2 * a], ml.loc[2 * a + 1], exits, m.obs, e * A + a, ml.stage);
}

__global__ __launch_bounds__(GB_MAX) void k_env_reset_multi(sl_env_batch env, sl_multi_agent m,
                                                        const uint8_t *__restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = env.H * env.W, e = blockIdx.x, tid = threadIdx.x, A = m.n_agents;
    if (mask && !mask[e]) return;
    GenericLds l = carve(smem, HW, 4);
    MultiLds ml = carve_multi(smem, HW, m.obs && env.n_channels > 0);
    if (tid == 0) {         // a slot that already holds a level moves on to its next one (safelife_env.py:204)
        sl_env_scalars *sc = env.scalars + e;
        if (sc->loaded) {
            sc->level_idx = env.pool_next ? env.pool_next[sc->level_idx] : pos_mod(sc->level_idx + env.level_stride, env.L);
            sc->episode_idx += 1;
        }
    }
    __syncthreads();
    reset_block_multi(env, m, e, l.buf[0], ml, l.wave_tot);
    u16 *gboard = env.board + (size_t)e * HW;
    for (int i = tid; i < HW; i += GB) gboard[i] = l.buf[0][i];
    __syncthreads();
    if (m.obs)
        for (int a = 0; a < A; ++a)
            write_obs(env, e, l.buf[0], env.goals + (size_t)e * HW, ml.loc[2 * a], ml.loc[2 * a + 1],
                      env.exit_locs + (size_t)e * env.E, m.obs, e * A + a, ml.stage);
}

__global__ __launch_bounds__(GB_MAX) void k_env_reset_generic(sl_env_batch env,
                                                          const uint8_t *__restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = env.H * env.W, e = blockIdx.x, tid = threadIdx.x;
    if (mask && !mask[e]) return;
    GenericLds l = carve(smem, HW, 1);
    if (tid == 0) {         // a slot that already holds a level moves on to its next one (safelife_env.py:204)
        sl_env_scalars *sc = env.scalars + e;
        if (sc->loaded) {
            sc->level_idx = env.pool_next ? env.pool_next[sc->level_idx] : pos_mod(sc->level_idx + env.level_stride, env.L);
            sc->episode_idx += 1;
        }
    }
    __syncthreads();
    reset_block(env, e, l.buf[0], l.ivar, l.wave_tot);
    u16 *gboard = env.board + (size_t)e * HW;
    for (int i = tid; i < HW; i += GB) gboard[i] = l.buf[0][i];
    write_obs(env, e, l.buf[0], env.goals + (size_t)e * HW, l.ivar[0], l.ivar[1],
              env.exit_locs + (size_t)e * env.E);
}

__global__ __launch_bounds__(GB_MAX) void k_env_obs_generic(sl_env_batch env) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = env.H * env.W, e = blockIdx.x, tid = threadIdx.x;
    GenericLds l = carve(smem, HW, 1);
    const u16 *gboard = env.board + (size_t)e * HW;
    for (int i = tid; i < HW; i += GB) l.buf[0][i] = gboard[i];
    __syncthreads();
    write_obs(env, e, l.buf[0], env.goals + (size_t)e * HW, env.scalars[e].agent_row,
              env.scalars[e].agent_col, env.exit_locs + (size_t)e * env.E);
}

// One thread per output element quad along y (contiguous in the output); view reads hit L2.
template <typename T>
__global__ __launch_bounds__(256) void k_obs_to_policy(const u32 *__restrict__ view, int vh, int vw, int C,
                                                       sl_channel_list ch, T *__restrict__ out, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // index into out [B,C,vw,vh]
    if (i >= total) return;
    const int y = (int)(i % vh);
    long long q = i / vh;
    const int x = (int)(q % vw);
    q /= vw;
    const int c = (int)(q % C);
    const long long b = q / C;
    const u32 word = view[(b * vh + y) * vw + x];
    out[i] = (T)((word >> ch.c[c]) & 1u);
}

// ------------------------------------------------------------------------------ launchers

// One categorical draw per env from the policy's probabilities (training/ppo.py:66-69 draws on the host with numpy):
// u = a 24-bit uniform from splitmix64(seed, counter, env), the action = the first k with u < p_0 + ... + p_k (the last
// action takes what rounding leaves).  One thread per env; the result goes straight into the int32 buffer the step reads.
__global__ __launch_bounds__(256) void k_sample_actions(const float *__restrict__ probs, int B, int A, unsigned long long seed,
                                                        unsigned long long counter, int32_t *__restrict__ actions) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (counter * 0x100000001B3ull + (unsigned long long)e + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
    const float *p = probs + (size_t)e * A;
    float cum = 0.0f;
    int a = A - 1;
    for (int k = 0; k < A - 1; ++k) {
        cum += p[k];
        if (u < cum) {
            a = k;
            break;
        }
    }
    actions[e] = a;
}

hipError_t launch_sample_actions(const float *probs, int B, int A, unsigned long long seed, unsigned long long counter,
                                 int32_t *actions, hipStream_t stream) {
    hipLaunchKernelGGL(k_sample_actions, dim3((B + 255) / 256), dim3(256), 0, stream, probs, B, A, seed, counter, actions);
    return hipGetLastError();
}

hipError_t launch_obs_to_policy(const u32 *view, int B, int vh, int vw, const sl_channel_list &ch, int C, void *out,
                                int dtype, hipStream_t stream) {
    const long long total = (long long)B * C * vw * vh;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_obs_to_policy<uint8_t>, grid, dim3(256), 0, stream, view, vh, vw, C, ch, (uint8_t *)out, total);
    else
        hipLaunchKernelGGL(k_obs_to_policy<float>, grid, dim3(256), 0, stream, view, vh, vw, C, ch, (float *)out, total);
    return hipGetLastError();
}


static hipError_t set_lds(const void *fn, size_t bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_advance_generic(const u16 *in, u16 *out, int B, int H, int W, const float *spawn_prob,
                                  int n_steps, sl_pcg64 *rng, const Jump *jump, int32_t *occupancy,
                                  hipStream_t stream, const int32_t *n_each) {
    size_t lds = generic_lds_bytes(H * W, 3);
    hipError_t err = set_lds((const void *)k_advance_generic, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_advance_generic, dim3(B), dim3(generic_threads(H * W)), lds, stream, in, out, H, W, spawn_prob,
                       n_steps, rng, jump, occupancy, n_each);
    return hipGetLastError();
}

hipError_t launch_alive_counts(const u16 *board, const u16 *goals, int B, int HW, int64_t *out,
                               hipStream_t stream) {
    hipLaunchKernelGGL(k_alive_counts, dim3(B), dim3(generic_threads(HW)), 0, stream, board, goals, HW, out);
    return hipGetLastError();
}

hipError_t launch_execute_actions(u16 *board, int B, int H, int W, int64_t *locs, const int64_t *actions,
                                  int A, int action_stride, int action_batch_stride, hipStream_t stream) {
    hipLaunchKernelGGL(k_execute_actions, dim3((B + 63) / 64), dim3(64), 0, stream, board, B, H, W, locs,
                       actions, A, action_stride, action_batch_stride);
    return hipGetLastError();
}

hipError_t launch_env_rollout_generic(const sl_env_batch &env, const int32_t *actions, int T,
                                      float *reward_t, uint8_t *done_t, const Jump *jump,
                                      hipStream_t stream) {
    size_t lds = generic_lds_bytes(env.H * env.W, 4);
    hipError_t err = set_lds((const void *)k_env_rollout_generic, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_env_rollout_generic, dim3(env.B), dim3(generic_threads(env.H * env.W)), lds, stream, env, actions, T,
                       reward_t, done_t, jump);
    return hipGetLastError();
}

hipError_t launch_inaction_generic(const sl_env_batch &env, const Jump *jump, hipStream_t stream) {
    size_t lds = generic_lds_bytes(env.H * env.W, 3);
    hipError_t err = set_lds((const void *)k_inaction_generic, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_inaction_generic, dim3(env.B), dim3(generic_threads(env.H * env.W)), lds, stream, env, jump);
    return hipGetLastError();
}

hipError_t launch_env_reset_generic(const sl_env_batch &env, const uint8_t *mask, hipStream_t stream) {
    size_t lds = generic_lds_bytes(env.H * env.W, 1);
    hipError_t err = set_lds((const void *)k_env_reset_generic, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_env_reset_generic, dim3(env.B), dim3(generic_threads(env.H * env.W)), lds, stream, env, mask);
    return hipGetLastError();
}

hipError_t launch_env_step_multi(const sl_env_batch &env, const sl_multi_agent &m, const int32_t *actions, const Jump *jump,
                                 hipStream_t stream) {
    const size_t lds = multi_lds_bytes(env, m);
    hipError_t err = set_lds((const void *)k_env_step_multi, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_env_step_multi, dim3(env.B), dim3(generic_threads(env.H * env.W)), lds, stream, env, m, actions, jump);
    return hipGetLastError();
}

hipError_t launch_env_reset_multi(const sl_env_batch &env, const sl_multi_agent &m, const uint8_t *mask, hipStream_t stream) {
    const size_t lds = multi_lds_bytes(env, m);
    hipError_t err = set_lds((const void *)k_env_reset_multi, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_env_reset_multi, dim3(env.B), dim3(generic_threads(env.H * env.W)), lds, stream, env, m, mask);
    return hipGetLastError();
}

hipError_t launch_env_obs_generic(const sl_env_batch &env, hipStream_t stream) {
    size_t lds = generic_lds_bytes(env.H * env.W, 1);
    hipError_t err = set_lds((const void *)k_env_obs_generic, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_env_obs_generic, dim3(env.B), dim3(generic_threads(env.H * env.W)), lds, stream, env);
    return hipGetLastError();
}

// ---- do two streams run concurrently? -----------------------------------------------------------------
// One wavefront that does nothing for `ticks` of the 100 MHz s_memrealtime counter.
__global__ void k_idle(long long ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

hipError_t launch_idle(long long ticks, hipStream_t stream) {
    hipLaunchKernelGGL(k_idle, dim3(1), dim3(64), 0, stream, ticks);
    return hipGetLastError();
}

}  // namespace sl
