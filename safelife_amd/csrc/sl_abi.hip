// sl_abi.hip -- the extern "C" boundary declared in include/safelife_hip.h: argument validation,
// the per-device PCG64 jump table, and dispatch to the gfx950 kernels.
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <dlfcn.h>
#include <string.h>

#include "sl_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t err, const char *what) {
    return fail(SL_E_HIP, std::string(what) + ": " + hipGetErrorString(err));
}

// ---- PCG64 jump table: entry k maps a state k LCG steps ahead (mult_k * s + plus_k * inc) -----
constexpr int kMaxDevices = 64;
std::mutex g_jump_mutex;
sl::Jump *g_jump[kMaxDevices] = {nullptr};
std::atomic<const sl::Jump *> g_jump_ready[kMaxDevices];

typedef unsigned __int128 u128;

int jump_table(const sl::Jump **out) {
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return hip_fail(err, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDevices) return fail(SL_E_UNSUPPORTED, "device index out of range");
    if (const sl::Jump *ready = g_jump_ready[dev].load(std::memory_order_acquire)) {      // (every launch comes through here)
        *out = ready;
        return SL_OK;
    }
    std::lock_guard<std::mutex> lock(g_jump_mutex);
    if (!g_jump[dev]) {
        const u128 mult = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull;
        std::vector<sl::Jump> host(SL_MAX_CELLS + 1);
        u128 m = 1, p = 0;   // k = 0: identity
        for (int k = 0; k <= SL_MAX_CELLS; ++k) {
            host[k].mult_hi = (uint64_t)(m >> 64);
            host[k].mult_lo = (uint64_t)m;
            host[k].plus_hi = (uint64_t)(p >> 64);
            host[k].plus_lo = (uint64_t)p;
            p = p * mult + 1;   // plus_{k+1} = sum_{i<=k} mult^i
            m = m * mult;
        }
        sl::Jump *d = nullptr;
        err = hipMalloc(&d, host.size() * sizeof(sl::Jump));
        if (err != hipSuccess) return hip_fail(err, "hipMalloc(jump table)");
        err = hipMemcpy(d, host.data(), host.size() * sizeof(sl::Jump), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            (void)hipFree(d);
            return hip_fail(err, "hipMemcpy(jump table)");
        }
        g_jump[dev] = d;
        g_jump_ready[dev].store(d, std::memory_order_release);
    }
    *out = g_jump[dev];
    return SL_OK;
}

// SAFELIFE_HIP_FORCE_GENERIC=1 routes every shape through the size-generic kernels (A/B testing of
// the two device implementations against each other; both are HIP, neither is a fallback to CPU).
bool force_generic() {
    static const bool v = [] {
        const char *e = getenv("SAFELIFE_HIP_FORCE_GENERIC");
        return e && e[0] == '1';
    }();
    return v;
}

int check_board_shape(int B, int H, int W) {
    if (B < 0) return fail(SL_E_ARG, "negative batch size");
    if (H < 3 || W < 3) return fail(SL_E_SHAPE, "Board must be at least 3x3.");
    if ((long long)H * W > SL_MAX_CELLS) return fail(SL_E_SHAPE, "board larger than SL_MAX_CELLS cells");
    return SL_OK;
}

int check_env(const sl_env_batch *env) {
    if (!env) return fail(SL_E_ARG, "null env");
    int rc = check_board_shape(env->B, env->H, env->W);
    if (rc) return rc;
    if (env->E < 1) return fail(SL_E_ARG, "E must be >= 1");
    if (env->n_channels < 0 || env->n_channels > SL_MAX_CHANNELS) return fail(SL_E_ARG, "bad n_channels");
    if (env->view_h < 1 || env->view_w < 1) return fail(SL_E_ARG, "bad view shape");
    if (env->L < 1 || env->n_tables < 1) return fail(SL_E_ARG, "empty level pool or points table");
    if (env->level_stride < 0) return fail(SL_E_ARG, "level_stride must be >= 0");
    for (int k = 0; k < env->n_channels; ++k)
        if (env->channels[k] < 0 || env->channels[k] > 31) return fail(SL_E_ARG, "output channel outside 0..31");
    const void *need[] = {env->board, env->goals, env->exit_locs, env->rng, env->scalars, env->points_table,
                          env->pool_board, env->pool_goals, env->pool_exit_locs, env->pool_rng,
                          env->pool_scalars, env->out};
    for (const void *p : need)
        if (!p) return fail(SL_E_ARG, "null pointer in sl_env_batch");
    if (env->policy_obs && (env->n_channels < 1 || (env->policy_dtype != 0 && env->policy_dtype != 1)))
        return fail(SL_E_ARG, "policy_obs needs n_channels >= 1 and policy_dtype 0 (uint8) or 1 (float32)");
    const sl_episode_queue &q = env->finished;
    if (q.capacity < 0 || (q.capacity > 0 && (!q.count || !q.records || !q.boards)))
        return fail(SL_E_ARG, "finished-episode queue: negative capacity or null buffers");
    const sl_wrappers &w = env->wrap;
    if (w.flags) {
        if (w.flags & ~(SL_WRAP_MOVEMENT | SL_WRAP_AS_PENALTY | SL_WRAP_EXIT_BONUS | SL_WRAP_SIDE_EFFECT |
                        SL_WRAP_IGNORE_REWARD_CELLS | SL_WRAP_INACTION))
            return fail(SL_E_ARG, "unknown bit in wrap.flags");
        if (w.flags & SL_WRAP_INACTION) {
            if (!(w.flags & SL_WRAP_SIDE_EFFECT)) return fail(SL_E_ARG, "SL_WRAP_INACTION without SL_WRAP_SIDE_EFFECT");
            if (!w.inaction_board || !w.inaction_rng) return fail(SL_E_ARG, "wrap.inaction_board / wrap.inaction_rng is null");
        }
        if (!w.state || !w.shaped_reward) return fail(SL_E_ARG, "wrap.state / wrap.shaped_reward is null");
        if (w.flags & SL_WRAP_MOVEMENT) {
            if (w.move_period < 1 || w.move_period > SL_WRAP_MAX_PERIOD)
                return fail(SL_E_ARG, "wrap.move_period outside 1..SL_WRAP_MAX_PERIOD");
            if (!w.move_table || w.move_table_len < env->H + env->W + w.move_period)
                return fail(SL_E_ARG, "wrap.move_table missing or shorter than H + W + move_period");
        }
    }
    return SL_OK;
}

// A contiguous sub-range of the batch as a batch of its own (pointer arithmetic only; the level pool, tables
// and workspaces are shared): what the size-generic kernels take for a slice.
sl_env_batch env_slice(const sl_env_batch &env, int e0, int n) {
    sl_env_batch s = env;
    const size_t hw = (size_t)env.H * env.W;
    s.B = n;
    s.board = env.board + (size_t)e0 * hw;
    s.goals = env.goals + (size_t)e0 * hw;
    s.exit_locs = env.exit_locs + (size_t)e0 * env.E;
    s.rng = env.rng + e0;
    s.scalars = env.scalars + e0;
    s.out = env.out + e0;
    if (env.stream_salt) s.stream_salt = env.stream_salt + e0;
    s.finished.env_base = env.finished.env_base + e0;
    if (env.obs) {
        const size_t cell = env.n_channels > 0 ? (size_t)env.n_channels : 4;     // uint8 channels, or the raw uint32 view
        s.obs = env.obs + (size_t)e0 * env.view_h * env.view_w * cell;
    }
    if (env.policy_obs)
        s.policy_obs = (char *)env.policy_obs + (size_t)e0 * env.n_channels * env.view_h * env.view_w *
                                                    (env.policy_dtype ? sizeof(float) : 1);
    if (env.wrap.flags) {
        s.wrap.state = env.wrap.state + e0;
        s.wrap.shaped_reward = env.wrap.shaped_reward + e0;
        if (env.wrap.flags & SL_WRAP_INACTION) {
            s.wrap.inaction_board = env.wrap.inaction_board + (size_t)e0 * hw;
            s.wrap.inaction_rng = env.wrap.inaction_rng + e0;
        }
    }
    return s;
}

}  // namespace

// ---- the records of every rank's envs -> rank 0, through RCCL point-to-point calls ----------------------------------
// RCCL's C API, resolved at first use (librccl.so.1: the copy already in the process -- torch's -- if there is one).
namespace {
struct Rccl {
    typedef struct { char internal[SL_GATHER_ID_BYTES]; } unique_id;      // ncclUniqueId
    int (*GetUniqueId)(unique_id *) = nullptr;
    int (*CommInitRank)(void **, int, unique_id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            r.why = std::string("dlopen(librccl): ") + dlerror();
            return;
        }
        bool all = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(h, n);
            if (!p) {
                all = false;
                r.why = std::string("librccl lacks ") + n;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.ok = all;
    });
    return r;
}
int rccl_fail(const Rccl &r, int code, const char *what) {
    return fail(SL_E_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error"));
}
struct GatherRequest {
    const void *send;
    void *recv;
    size_t bytes;
    hipStream_t writers[8];
    int n_writers;
    hipStream_t stream;
    long long ticket;
    long long marker;       // >= 0: the window was written from the library's AQL queues; the worker waits for this
                            // marker of theirs (slhip_queues_marker) instead of ordering streams
    const volatile uint32_t *placement;     // release-free queues: their placement word -- a window whose steps ran on the
                                            // wrong XCD is not handed to RCCL (null: stream's fences, nothing to check)
};
struct GatherComm {
    void *comm;
    int world, rank;
    // asynchronous hand-off (slhip_gather_window_async): a worker thread of the library's own issues the stream
    // ordering and the RCCL group, so the stepping thread pays for a queue push instead of ~20-80 us of runtime calls
    int device = 0;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::deque<GatherRequest> queue;
    bool stop = false;
    long long submitted = 0;                 // tickets handed out
    std::atomic<long long> pushed{0};        // == submitted, readable without the lock (the worker polls it)
    bool poked = false;                      // slhip_gather_poke: a window is about to close -- wake up and poll
    std::atomic<long long> issued{0};        // tickets whose RCCL group has been enqueued (and done event recorded)
    int error = 0;
    static constexpr int RING = 8;
    hipEvent_t order_ev[8] = {};             // writer stream -> gather stream
    hipEvent_t done_ev[RING] = {};           // recorded behind ticket t's group: slot t % RING
};
}  // namespace


extern "C" {

int slhip_abi_version(void) { return SL_ABI_VERSION; }

const char *slhip_last_error(void) { return g_last_error.c_str(); }

int slhip_device_count(void) {
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) return hip_fail(err, "hipGetDeviceCount");
    return n;
}

int slhip_advance_board(const uint16_t *in, uint16_t *out, int B, int H, int W, const float *spawn_prob,
                        int n_steps, sl_pcg64 *rng, void *stream) {
    int rc = check_board_shape(B, H, W);
    if (rc) return rc;
    if (n_steps < 0) return fail(SL_E_ARG, "negative n_steps");
    if (B == 0) return SL_OK;            // (an empty tensor has no storage: checked before the pointers)
    if (!in || !out || !spawn_prob || !rng) return fail(SL_E_ARG, "null pointer");
    const sl::Jump *jump;
    if ((rc = jump_table(&jump))) return rc;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    hipError_t err = (sl::rowlane_supports(H, W) && aligned && !force_generic())
                         ? sl::launch_advance_rowlane(in, out, B, H, W, spawn_prob, n_steps, nullptr, nullptr, rng, jump,
                                                      (hipStream_t)stream)
                         : sl::launch_advance_generic(in, out, B, H, W, spawn_prob, n_steps, rng, jump, nullptr,
                                                      (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "advance_board launch");
}

int slhip_advance_board_each(const uint16_t *in, uint16_t *out, int B, int H, int W, const float *spawn_prob,
                             const int32_t *n_steps, sl_pcg64 *rng, void *stream) {
    int rc = check_board_shape(B, H, W);
    if (rc) return rc;
    if (B == 0) return SL_OK;
    if (!in || !out || !spawn_prob || !rng || !n_steps) return fail(SL_E_ARG, "null pointer");
    const sl::Jump *jump;
    if ((rc = jump_table(&jump))) return rc;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    hipError_t err = (sl::rowlane_supports(H, W) && aligned && !force_generic())
                         ? sl::launch_advance_rowlane(in, out, B, H, W, spawn_prob, 0, n_steps, nullptr, rng, jump,
                                                      (hipStream_t)stream)
                         : sl::launch_advance_generic(in, out, B, H, W, spawn_prob, 0, rng, jump, nullptr,
                                                      (hipStream_t)stream, n_steps);
    return err == hipSuccess ? SL_OK : hip_fail(err, "advance_board_each launch");
}

int slhip_life_occupancy(const uint16_t *in, int32_t *counts, int B, int H, int W, const float *spawn_prob,
                         int n_steps, sl_pcg64 *rng, void *stream) {
    int rc = check_board_shape(B, H, W);
    if (rc) return rc;
    if (n_steps < 0) return fail(SL_E_ARG, "negative n_steps");
    if (B == 0) return SL_OK;
    if (!in || !counts || !spawn_prob || !rng) return fail(SL_E_ARG, "null pointer");
    const sl::Jump *jump;
    if ((rc = jump_table(&jump))) return rc;
    // 16-bit (or drained 8-bit) per-colour counters in LDS: the row kernel covers every step count the
    // reference is called with
    hipError_t err = (sl::rowlane_supports(H, W) && n_steps <= 65535 && !force_generic())
                         ? sl::launch_occupancy_rowlane(in, counts, (size_t)H * W * 8, B, nullptr, 0, nullptr, H, W, spawn_prob,
                                                        n_steps, rng, jump, (hipStream_t)stream)
                         : sl::launch_advance_generic(in, nullptr, B, H, W, spawn_prob, n_steps, rng, jump, counts,
                                                      (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "life_occupancy launch");
}

int slhip_alive_counts(const uint16_t *board, const uint16_t *goals, int B, int HW, int64_t *out,
                       void *stream) {
    if (B < 0 || HW < 0) return fail(SL_E_ARG, "negative size");
    if (B == 0) return SL_OK;
    if (!board || !goals || !out) return fail(SL_E_ARG, "null pointer");
    hipError_t err = sl::launch_alive_counts(board, goals, B, HW, out, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "alive_counts launch");
}

int slhip_execute_actions(uint16_t *board, int B, int H, int W, int64_t *locs, const int64_t *actions,
                          int A, int action_stride, int action_batch_stride, void *stream) {
    int rc = check_board_shape(B, H, W);
    if (rc) return rc;
    if (A < 0) return fail(SL_E_ARG, "negative agent count");
    if (B == 0 || A == 0) return SL_OK;
    if (!board || !locs || !actions) return fail(SL_E_ARG, "null pointer");
    hipError_t err = sl::launch_execute_actions(board, B, H, W, locs, actions, A, action_stride,
                                                action_batch_stride, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "execute_actions launch");
}

// sl_env_batch.pool_ready: level `s` of the pool as an episode starts on it (SafeLifeEnv.reset(), safelife_env.py:203-218
// on a level of its own: safelife_game.py:537-552 update_exit_colors, :684-687 current_points), by one 256-thread
// workgroup -- the generic reset's arithmetic (sl_generic.hip: reset_block), whose result depends on the level alone.
__device__ void pool_ready_level(const sl_env_batch &env, size_t s) {
    __shared__ int wave_tot[4];
    const int t = threadIdx.x, cells = env.H * env.W;
    const uint16_t *board = env.pool_board + s * cells, *goals = env.pool_goals + s * cells;
    uint16_t *ready = env.pool_ready + s * cells;
    sl_level_scalars *lvp = const_cast<sl_level_scalars *>(env.pool_scalars + s);
    const sl_level_scalars lv = *lvp;
    const int32_t *table = env.points_table + 72 * lv.table_idx;
    int sum = 0;
    for (int k = t; k < cells; k += 256) {
        const uint16_t b = board[k];
        const int bin = sl::score_bin(b, goals[k]);
        if (bin >= 0) sum += table[bin];
        ready[k] = b;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((t & 63) == 0) wave_tot[t >> 6] = sum;
    __syncthreads();                // (also: the copy above is visible to thread 0 below)
    if (t == 0) {
        const int score = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        const int ly = lv.agent_row, lx = lv.agent_col;
        bool open = false;
        if (ly >= 0) {
            uint16_t *cell = ready + ly * env.W + lx;
            int earned = score - lv.initial_points + env.exit_points * (sl::has_exited(*cell) ? 1 : 0);
            if (earned < 0) earned = 0;
            open = (*cell & sl::AGENT) && earned >= lv.required_reset;
            *cell = (uint16_t)((*cell & ~sl::EXIT) | (open ? sl::EXIT : 0u));
        }
        const uint16_t paint = (uint16_t)(sl::FROZEN | sl::EXIT | (open ? sl::COLOR_R : 0u));
        for (int k = 0; k < env.E; ++k) {
            const int ex = env.pool_exit_locs[s * env.E + k];
            if (ex >= 0) ready[ex] = paint;
        }
        const int exited = ly >= 0 ? (sl::has_exited(ready[ly * env.W + lx]) ? 1 : 0) : 0;
        lvp->ready = (int32_t)(((uint32_t)(score + env.exit_points * exited) << 1) | (open ? 1u : 0u));
    }
}

__global__ void __launch_bounds__(256) k_pool_ready(sl_env_batch env) { pool_ready_level(env, blockIdx.x); }

int slhip_env_prepare(const sl_env_batch *env, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!env->score_lut) return fail(SL_E_ARG, "score_lut workspace is null");
    std::vector<int32_t> host((size_t)env->n_tables * 72);
    hipError_t err = hipMemcpyAsync(host.data(), env->points_table, host.size() * sizeof(int32_t),
                                    hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
    if (err != hipSuccess) return hip_fail(err, "env_prepare copy");
    for (int32_t v : host)
        if (v < -128 || v > 127) return fail(SL_E_UNSUPPORTED, "points_table entry outside int8 range");
    err = sl::launch_build_score_lut(env->points_table, env->n_tables, env->score_lut, (hipStream_t)stream);
    if (err == hipSuccess && env->wrap.pool_baseline)
        err = sl::launch_build_baseline(*env, (hipStream_t)stream);
    if (err == hipSuccess && env->pool_ready && env->L > 0) {
        hipLaunchKernelGGL(k_pool_ready, dim3(env->L), dim3(256), 0, (hipStream_t)stream, *env);
        err = hipGetLastError();
    }
    return err == hipSuccess ? SL_OK : hip_fail(err, "env_prepare launch");
}

int slhip_pool_baseline(const sl_env_batch *env, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!(env->wrap.flags & SL_WRAP_SIDE_EFFECT) || !env->wrap.pool_baseline) return SL_OK;
    if (!sl::rowlane_supports(env->H, env->W)) return SL_OK;       // (the size-generic kernels read the pool itself)
    hipError_t err = sl::launch_build_baseline(*env, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "pool_baseline launch");
}

// One workgroup per new level (plus one for the successor table): 2 x H*W cells, E exits, 32 B of generator, 32 B of
// scalars -- a few KB each, fetched from wherever the rows lie (pinned host memory: no staging copy, no second launch).
__global__ void __launch_bounds__(256) k_pool_write(sl_env_batch env, sl_pool_rows r) {
    const int i = blockIdx.x, t = threadIdx.x;
    if (i == r.n) {
        for (int k = t; k < env.L; k += 256) r.next_dst[k] = r.next[k];
        return;
    }
    const size_t cells = (size_t)env.H * env.W, s = (size_t)r.slot[i];
    uint16_t *board = const_cast<uint16_t *>(env.pool_board) + s * cells;
    uint16_t *goals = const_cast<uint16_t *>(env.pool_goals) + s * cells;
    for (size_t k = t; k < cells; k += 256) {
        board[k] = r.board[i * cells + k];
        goals[k] = r.goals[i * cells + k];
    }
    for (int k = t; k < env.E; k += 256) const_cast<int32_t *>(env.pool_exit_locs)[s * env.E + k] = r.exit_locs[(size_t)i * env.E + k];
    constexpr int RNG_WORDS = sizeof(sl_pcg64) / 4, SCALAR_WORDS = sizeof(sl_level_scalars) / 4;
    if (t < RNG_WORDS)
        ((uint32_t *)const_cast<sl_pcg64 *>(env.pool_rng + s))[t] = ((const uint32_t *)(r.rng + i))[t];
    else if (t >= 64 && t < 64 + SCALAR_WORDS)
        ((uint32_t *)const_cast<sl_level_scalars *>(env.pool_scalars + s))[t - 64] = ((const uint32_t *)(r.scalars + i))[t - 64];
    if (env.pool_ready) {           // the slot as an episode starts on it, by the workgroup that has just written it
        __syncthreads();
        pool_ready_level(env, s);
    }
}

int slhip_pool_write(const sl_env_batch *env, const sl_pool_rows *rows, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!rows || rows->n < 0 || rows->n > env->L) return fail(SL_E_ARG, "pool_write: n outside 0..L");
    if (rows->n && (!rows->slot || !rows->board || !rows->goals || !rows->exit_locs || !rows->rng || !rows->scalars))
        return fail(SL_E_ARG, "pool_write: a row array is missing");
    if ((rows->next != nullptr) != (rows->next_dst != nullptr)) return fail(SL_E_ARG, "pool_write: next and next_dst go together");
    const int blocks = rows->n + (rows->next ? 1 : 0);
    if (!blocks) return SL_OK;
    hipLaunchKernelGGL(k_pool_write, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *env, *rows);
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? SL_OK : hip_fail(err, "pool_write launch");
}

size_t slhip_goal_cache_bytes(const sl_env_batch *env, int *boards_per_block) {
    if (boards_per_block) *boards_per_block = 0;
    if (!env || env->B <= 0 || force_generic() || !env->score_lut) return 0;
    // only the plain step kernels keep the cache: no observation, no wrappers, and a finished-episode queue only where
    // the shape's plain kernels serve one (the 64-cell rows of C5)
    if (env->wrap.flags || env->obs || env->policy_obs) return 0;
    if (env->finished.capacity > 0 && !sl::rowlane_lean_takes_queue(env->H, env->W)) return 0;
    return sl::rowlane_goal_cache_bytes(env->H, env->W, env->B, !env->spawner_free, boards_per_block);
}

// A launch of the size-generic kernels on a batch that carries a goal-word cache: they load levels without keeping its
// flags, so every flag goes down first (the row kernels' launcher does the same for its own odd launches).
static hipError_t drop_goal_cache(const sl_env_batch *env, hipStream_t st) {
    if (!env->goal_cache) return hipSuccess;
    const size_t bytes = sl::rowlane_goal_cache_bytes(env->H, env->W, env->B, !env->spawner_free);
    return bytes ? hipMemsetAsync(env->goal_cache, 0, bytes, st) : hipSuccess;
}

static bool use_rowlane(const sl_env_batch *env, int e_first);
int slhip_env_reset(const sl_env_batch *env, const uint8_t *mask, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (env->B == 0) return SL_OK;
    hipError_t err;
    if (use_rowlane(env, 0)) {
        // the fused row kernel in its reset mode: the block that serves the in-kernel auto-reset, for the masked envs
        // (eight boards per workgroup; the size-generic kernel below takes a workgroup per board: 31 us for 8192 envs)
        const sl::Jump *jump;
        if ((rc = jump_table(&jump))) return rc;
        err = sl::launch_env_rollout_rowlane(*env, 0, env->B, (const int32_t *)env->scalars, -1, env->B, nullptr, nullptr, jump,
                                             (hipStream_t)stream, nullptr, mask);
    } else {
        err = drop_goal_cache(env, (hipStream_t)stream);
        if (err == hipSuccess) err = sl::launch_env_reset_generic(*env, mask, (hipStream_t)stream);
    }
    return err == hipSuccess ? SL_OK : hip_fail(err, "env_reset launch");
}

// Row kernels or size-generic kernels for this batch?  (The generic family is a complete, slower HIP
// implementation: one workgroup per board.  Falling to it is legitimate but ~5x slower at C3, so the first
// time it happens for a reason other than an explicit request the library says so once on stderr.)
static bool use_rowlane(const sl_env_batch *env, int e_first) {
    if (force_generic()) return false;
    const char *why = nullptr;
    const size_t slice_bytes = (size_t)e_first * env->H * env->W * sizeof(uint16_t);
    if (!sl::rowlane_supports(env->H, env->W)) why = "no row-lane kernel for this board shape";
    else if (!env->score_lut) why = "points_table entries outside int8";
    else if ((((uintptr_t)env->board | (uintptr_t)env->goals | (uintptr_t)env->rng | (uintptr_t)env->score_lut) & 15) ||
             (slice_bytes & 15))
        why = "board / goals / rng / score_lut (or the slice start) not 16-byte aligned";
    else if (env->E > 8) why = "more than 8 exit slots";
    else if (env->policy_obs && env->view_h * env->view_w > sl::rowlane_policy_room(env->H, env->W))
        why = "view too large for the fused policy-layout observation";
    else if (env->wrap.flags && (((env->wrap.flags & SL_WRAP_SIDE_EFFECT) && !env->wrap.pool_baseline) ||
                                 (((uintptr_t)env->wrap.state | (uintptr_t)env->wrap.move_table) & 15)))
        why = "wrapper workspace missing or unaligned";
    else if ((env->wrap.flags & SL_WRAP_INACTION) &&
             ((((uintptr_t)env->wrap.inaction_board | (uintptr_t)env->wrap.inaction_rng) & 15) || (slice_bytes & 15)))
        why = "inaction-baseline state unaligned";
    if (!why) return true;
    static std::atomic<bool> warned{false};
    if (!warned.exchange(true))
        fprintf(stderr, "libsafelife_hip: %dx%d batch runs on the size-generic kernels (%s)\n", env->H, env->W, why);
    return false;
}

// One launch over envs [e_first, e_first + e_count).  actions / reward_t / done_t are indexed
// [t * tstride + (env index in the whole batch)].
static int rollout_range(const sl_env_batch *env, int e_first, int e_count, const int32_t *actions, int T, int tstride,
                         float *reward_t, uint8_t *done_t, void *stream) {
    const sl::Jump *jump;
    int rc;
    if ((rc = jump_table(&jump))) return rc;
    hipError_t err;
    const bool inaction = (env->wrap.flags & SL_WRAP_INACTION) != 0;
    if (use_rowlane(env, e_first)) {
        // (the "inaction" baseline of SimpleSideEffectPenalty is part of the WRAP variants of the row kernel: a third
        //  pass of its CA loop, every step of a T-step launch)
        err = sl::launch_env_rollout_rowlane(*env, e_first, e_count, actions, T, tstride, reward_t, done_t, jump,
                                             (hipStream_t)stream);
    } else {
        if (inaction && T > 1)
            return fail(SL_E_UNSUPPORTED, "the inaction baseline advances between steps: T must be 1 on the generic kernels");
        if (T > 1 && e_count != env->B) return fail(SL_E_UNSUPPORTED, "T-step launches of a slice need the row kernels");
        const sl_env_batch s = env_slice(*env, e_first, e_count);
        if ((err = drop_goal_cache(env, (hipStream_t)stream)) != hipSuccess) return hip_fail(err, "goal cache");
        if (inaction && (err = sl::launch_inaction_generic(s, jump, (hipStream_t)stream)) != hipSuccess)
            return hip_fail(err, "inaction baseline launch");
        err = sl::launch_env_rollout_generic(s, actions + e_first, T, reward_t ? reward_t + e_first : nullptr,
                                             done_t ? done_t + e_first : nullptr, jump, (hipStream_t)stream);
    }
    return err == hipSuccess ? SL_OK : hip_fail(err, "env_step launch");
}

int slhip_env_rollout(const sl_env_batch *env, const int32_t *actions, int T, float *reward_t,
                      uint8_t *done_t, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!actions) return fail(SL_E_ARG, "null actions");
    if (T < 0) return fail(SL_E_ARG, "negative T");
    if (env->B == 0 || T == 0) return SL_OK;
    return rollout_range(env, 0, env->B, actions, T, env->B, reward_t, done_t, stream);
}

int slhip_env_step(const sl_env_batch *env, const int32_t *actions, void *stream) {
    return slhip_env_rollout(env, actions, 1, nullptr, nullptr, stream);
}

int slhip_env_step_range(const sl_env_batch *env, int first, int count, const int32_t *actions, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!actions) return fail(SL_E_ARG, "null pointer");
    if (first < 0 || count < 0 || first + count > env->B) return fail(SL_E_ARG, "env range outside the batch");
    if (count == 0) return SL_OK;
    return rollout_range(env, first, count, actions, 1, env->B, nullptr, nullptr, stream);
}

int slhip_streams_order(void *const *before, int n_before, void *const *after, int n_after) {
    if (n_before < 0 || n_after < 0 || (n_before && !before) || (n_after && !after)) return fail(SL_E_ARG, "bad stream lists");
    // ordering events: a ring of timing-less events, created on first use (an event may be re-recorded once the
    // waits that named it have been enqueued, which they have by the time the ring comes round)
    constexpr int RING = 64;
    struct Ring {
        hipEvent_t ev[RING];
        std::atomic<unsigned> next{0};
        std::once_flag once;
        hipError_t made = hipSuccess;
    };
    static Ring rings[kMaxDevices];                 // one per device: an event belongs to the device it was made on
    int dev = 0;
    hipError_t derr = hipGetDevice(&dev);
    if (derr != hipSuccess) return hip_fail(derr, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDevices) return fail(SL_E_UNSUPPORTED, "device index out of range");
    Ring &rg = rings[dev];
    std::call_once(rg.once, [&rg] {
        for (int i = 0; i < RING && rg.made == hipSuccess; ++i) rg.made = hipEventCreateWithFlags(&rg.ev[i], hipEventDisableTiming);
    });
    if (rg.made != hipSuccess) return hip_fail(rg.made, "streams_order events");
    for (int i = 0; i < n_before; ++i) {
        bool needed = false;
        for (int j = 0; j < n_after; ++j) needed |= after[j] != before[i];
        if (!needed) continue;
        hipEvent_t ev = rg.ev[rg.next.fetch_add(1, std::memory_order_relaxed) % RING];
        hipError_t err = hipEventRecord(ev, (hipStream_t)before[i]);
        for (int j = 0; j < n_after && err == hipSuccess; ++j)
            if (after[j] != before[i]) err = hipStreamWaitEvent((hipStream_t)after[j], ev, 0);
        if (err != hipSuccess) return hip_fail(err, "streams_order");
    }
    return SL_OK;
}

int slhip_streams_concurrent(void *stream_a, void *stream_b, int *concurrent) {
    if (!concurrent) return fail(SL_E_ARG, "null pointer");
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    const long long ticks = 10000;                 // 100 us of the 100 MHz counter
    hipError_t err = hipStreamSynchronize(a);
    if (err == hipSuccess) err = hipStreamSynchronize(b);
    double best = 1e30;
    for (int rep = 0; rep < 2 && err == hipSuccess; ++rep) {        // (first pass also warms the kernel up)
        const auto t0 = std::chrono::steady_clock::now();
        err = sl::launch_idle(ticks, a);
        if (err == hipSuccess) err = sl::launch_idle(ticks, b);
        if (err == hipSuccess) err = hipStreamSynchronize(a);
        if (err == hipSuccess) err = hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    if (err != hipSuccess) return hip_fail(err, "streams_concurrent");
    *concurrent = best < 160.0 ? 1 : 0;            // one after the other: >= 200 us
    return SL_OK;
}

#ifdef SL_TRACE
// profiling builds: every launch of slhip_env_step_slices writes its waves' phase stamps to the next slot of this buffer
static char *g_trace_base = nullptr;
static long long g_trace_bytes = 0;
static int g_trace_slots = 0, g_trace_next = 0;
extern "C" int slhip_trace_set(void *base, long long bytes_per_launch, int slots) {
    g_trace_base = (char *)base;
    g_trace_bytes = bytes_per_launch;
    g_trace_slots = slots;
    g_trace_next = 0;
    return SL_OK;
}
#endif

int slhip_env_step_slices(const sl_env_batch *env, int n_slices, const int32_t *bounds, const int32_t *actions,
                          void *const *streams) {
    int rc = check_env(env);
    if (rc) return rc;
    if (n_slices < 1 || !bounds || !streams || !actions) return fail(SL_E_ARG, "bad slice arguments");
    if (bounds[0] != 0 || bounds[n_slices] != env->B) return fail(SL_E_ARG, "slice bounds must run from 0 to B");
    for (int i = 0; i < n_slices; ++i) {
        if (bounds[i + 1] < bounds[i]) return fail(SL_E_ARG, "slice bounds must not decrease");
        const int n = bounds[i + 1] - bounds[i];
        if (n == 0) continue;
        float *trace = nullptr;
#ifdef SL_TRACE
        if (g_trace_base && g_trace_next < g_trace_slots) trace = (float *)(g_trace_base + g_trace_bytes * g_trace_next++);
#endif
        rc = rollout_range(env, bounds[i], n, actions, 1, env->B, trace, nullptr, streams[i]);
        if (rc) return rc;
    }
    return SL_OK;
}

// ---- sliced stepping on the library's own AQL queues (sl_aql.hip) ----------------------------------------------------
namespace sl {
// The placement probe of release-free stepping: where does workgroup i of a dispatch run?
__global__ void k_xcd_probe(u32 *__restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;      // XCC_ID
}
}  // namespace sl

namespace {
struct StepQueues {
    int n_slices = 0;
    int qmap[8] = {0, 1, 2, 3, 4, 5, 6, 7};     // slice i is dispatched on the library's queue qmap[i] (slhip_queues_open_on)
    int n_phys = 0;                             // queues 0 .. n_phys - 1 exist
    int32_t bounds[9] = {};
    int H = 0, W = 0, B = 0;
    // SL_QUEUES_RELEASE_FREE (opt-in): no release fence between the steps of a queue; the placement this rests on is
    // probed when the queues are opened and verified by every step (sl_rowlane.hip: xcd_base / xcd_flag).
    bool release_free = false;
    int base[8] = {};               // queue i runs workgroup w of a dispatch on XCD (base[i] + w) mod 8 (probed at open)
    uint32_t *flag = nullptr;       // host memory: raised by a step kernel that finds itself on another XCD
    bool pending = false;           // steps dispatched since this handle last waited for a marker of its own
    bool swap = false;              // self-test: slice i goes to queue (i + step) mod n, behind a release-less drain
    // what every slice's argument block looked like the last time round (the per-step pointers aside): as long as it
    // stays the same, the blocks already in the queues' argument rings are patched instead of rewritten
    uint32_t serial = 0, version[8] = {};
    unsigned char last_args[8][sizeof(sl::PreparedStep::args)] = {};
    size_t last_bytes[8] = {};
    long long steps = 0;
    std::string downgraded;         // why release-free stepping was asked for and not granted
    // the prepared launches of the last call, valid while the caller's batch description stays the same byte for byte
    // (preparing four slices -- variant lookup, argument blocks, comparison with the rings' contents -- cost ~13 us at
    // the head of every call: 0.7 us per step of a 20-step region)
    bool have_prepared = false;
    sl_env_batch prepared_env;
    sl::PreparedStep prepared[8];
};

// Release-free stepping is only sound where workgroup w of a dispatch of queue i always runs on the same XCD (slice i is
// always stepped from queue i).  MI355X runs it on XCD (q_i + w) mod 8, q_i a constant of the queue, whatever the grid,
// the kernel, the queue's history and the other queues are doing (tools/ubench/xcd_place.hip,
// profiles/round4_a_xcd_placement.txt).  Probed here: three grids on every queue -- the slice's, one workgroup, the
// slice's again; every run must follow that formula with the queue's q_i, which the step kernels then check themselves
// against at every step.
bool probe_placement(int n_queues, int grid, int *base, std::string *why) {
    hipFunction_t f = nullptr;
    if (hipGetFuncBySymbol(&f, (const void *)sl::k_xcd_probe) != hipSuccess || sl::aql_probe(f)) {
        (void)hipGetLastError();
        *why = "the placement probe kernel was not found";
        return false;
    }
    const int runs = 3 * n_queues;                  // queue = r % n_queues; a queue's second run is the one-workgroup grid
    uint32_t *out = nullptr;
    if (hipMalloc((void **)&out, sizeof(uint32_t) * (size_t)grid * runs) != hipSuccess) {
        (void)hipGetLastError();
        *why = "no memory for the placement probe";
        return false;
    }
    bool ok = hipMemset(out, 0xFF, sizeof(uint32_t) * (size_t)grid * runs) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    for (int r = 0; r < runs && ok; ++r) {
        struct {
            uint32_t *out;
        } args = {out + (size_t)r * grid};
        const sl::AqlLaunch a{r % n_queues, true, false};
        ok = sl::aql_dispatch(a, f, (unsigned)(r / n_queues == 1 ? 1 : grid), 256, 0, &args, sizeof(args)) == hipSuccess;
    }
    ok = ok && sl::aql_fence(n_queues) == hipSuccess;
    std::vector<uint32_t> host((size_t)grid * runs);
    ok = ok && hipMemcpy(host.data(), out, sizeof(uint32_t) * host.size(), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(out);
    if (!ok) {
        (void)hipGetLastError();
        *why = "the placement probe could not be run";
        return false;
    }
    for (int r = 0; r < runs; ++r) {
        const int q = r % n_queues;
        const uint32_t *mine = host.data() + (size_t)r * grid;
        if (r < n_queues) base[q] = (int)mine[0];
        for (int w = 0; w < (r / n_queues == 1 ? 1 : grid); ++w)
            if (mine[w] > 7u || mine[w] != (((uint32_t)base[q] + (uint32_t)w) & 7u)) {
                *why = "workgroup " + std::to_string(w) + " of a dispatch of queue " + std::to_string(q) + " ran on XCD " +
                       std::to_string(mine[w]) + ", not on XCD (" + std::to_string(base[q]) + " + " + std::to_string(w) + ") mod 8";
                return false;
            }
    }
    return true;
}
}  // namespace

// Which of the library's queues does a HIP stream share a hardware pipe with?  MI355X has four compute pipes; the queues
// of a process -- HIP's own hardware queues behind its streams, and these -- are spread over them, and two queues of one
// pipe take turns: a kernel of the stream (RCCL's exchange, ~40 us per window; a policy network's) holds up the slice
// whose queue sits on its pipe while the other slices run on (the kernels' own clocks show one queue's next launch 40 us
// late behind every exchange, profiles/round4_g_*), and it also moves that queue's workgroups to other XCDs.  Measured
// here, per queue: a 200 us one-wavefront kernel on the stream, then -- once it runs -- a one-workgroup dispatch on the
// queue; a dispatch that comes back only when the long kernel ends shares its pipe.
int slhip_queues_stream_shares(int n_queues, void *stream, int *mask) {
    if (!mask || n_queues < 1 || n_queues > 8) return fail(SL_E_ARG, "bad arguments (1 to 8 queues)");
    *mask = 0;
    if (const char *why = sl::aql_open(n_queues)) return fail(SL_E_UNSUPPORTED, std::string("AQL queues unavailable: ") + why);
    hipFunction_t f = nullptr;
    if (hipGetFuncBySymbol(&f, (const void *)sl::k_xcd_probe) != hipSuccess || sl::aql_probe(f)) {
        (void)hipGetLastError();
        return fail(SL_E_UNSUPPORTED, "the probe kernel was not found");
    }
    uint32_t *out = nullptr;
    hipError_t err = hipMalloc((void **)&out, 64);
    if (err != hipSuccess) return hip_fail(err, "hipMalloc");
    const hipStream_t st = (hipStream_t)stream;
    const long long ticks = 20000;                  // 200 us of the 100 MHz counter
    struct {
        uint32_t *out;
    } args = {out};
    for (int q = 0; q < n_queues && err == hipSuccess; ++q) {
        // (a host-timer measurement on a shared box: one warm-up pass, then the MEDIAN of five -- a single fast or slow
        //  outlier flags nothing)
        double us_of[6];
        int n_us = 0;
        for (int rep = 0; rep < 6 && err == hipSuccess; ++rep) {
            err = hipStreamSynchronize(st);
            if (err == hipSuccess) err = sl::aql_fence(n_queues);
            if (err == hipSuccess) err = sl::launch_idle(ticks, st);
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 40.0) {}
            const auto t1 = std::chrono::steady_clock::now();
            if (err == hipSuccess) err = sl::aql_dispatch(sl::AqlLaunch{q, false, false}, f, 1, 256, 0, &args, sizeof(args));
            if (err == hipSuccess) err = sl::aql_fence(n_queues);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
            if (rep > 0) us_of[n_us++] = us;
        }
        std::sort(us_of, us_of + n_us);
        if (n_us && us_of[n_us / 2] > 80.0) *mask |= 1 << q;   // (alone: a few microseconds; behind the long kernel: >= 150)
    }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    (void)hipFree(out);
    if (err != hipSuccess) return hip_fail(err, "queues_stream_shares");
    return SL_OK;
}

int slhip_queues_open(const sl_env_batch *env, int n_slices, const int32_t *bounds, int flags, void **handle) {
    return slhip_queues_open_on(env, n_slices, bounds, nullptr, flags, handle);
}

int slhip_queues_open_on(const sl_env_batch *env, int n_slices, const int32_t *bounds, const int32_t *queue_ids, int flags,
                         void **handle) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!handle || !bounds || n_slices < 1 || n_slices > 8) return fail(SL_E_ARG, "bad queue arguments (1 to 8 slices)");
    int n_phys = n_slices;
    if (queue_ids) {
        unsigned seen = 0;
        for (int i = 0; i < n_slices; ++i) {
            if (queue_ids[i] < 0 || queue_ids[i] > 7 || (seen >> queue_ids[i] & 1u)) return fail(SL_E_ARG, "queue ids: distinct, 0 to 7");
            seen |= 1u << queue_ids[i];
            n_phys = std::max(n_phys, queue_ids[i] + 1);
        }
    }
    if (flags & ~SL_QUEUES_RELEASE_FREE) return fail(SL_E_ARG, "unknown queue flags");
    if (bounds[0] != 0 || bounds[n_slices] != env->B) return fail(SL_E_ARG, "slice bounds must run from 0 to B");
    for (int i = 0; i < n_slices; ++i) {
        if (bounds[i + 1] < bounds[i]) return fail(SL_E_ARG, "slice bounds must not decrease");
        if (!use_rowlane(env, bounds[i])) return fail(SL_E_UNSUPPORTED, "queue stepping needs the row kernels");
    }
    if (const char *why = sl::aql_open(n_phys)) return fail(SL_E_UNSUPPORTED, std::string("AQL queues unavailable: ") + why);
    if (sl::aql_poisoned()) return fail(SL_E_HIP, "AQL queues: an earlier wait timed out on this device");
    if (const char *why = sl::aql_probe(sl::rowlane_probe_function()))
        return fail(SL_E_UNSUPPORTED, std::string("AQL queues unavailable: ") + why);
    StepQueues *c = new StepQueues;
    static std::atomic<uint32_t> serials{0};
    c->serial = ++serials;
    c->n_slices = n_slices;
    c->n_phys = n_phys;
    if (queue_ids)
        for (int i = 0; i < n_slices; ++i) c->qmap[i] = queue_ids[i];
    c->H = env->H;
    c->W = env->W;
    c->B = env->B;
    memcpy(c->bounds, bounds, sizeof(int32_t) * (n_slices + 1));
    if (flags & SL_QUEUES_RELEASE_FREE) {
        int grid = 8;
        for (int i = 0; i < n_slices; ++i) grid = std::max(grid, bounds[i + 1] - bounds[i]);    // (>= workgroups of a slice)
        grid = std::min(grid, 4096);
        std::string why;
        int phys_base[8] = {};
        if (!probe_placement(n_phys, grid, phys_base, &why)) {
            c->downgraded = why;
        } else if (hipHostMalloc((void **)&c->flag, 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->flag = nullptr;
            c->downgraded = "no host memory for the placement flag";
        } else {
            *c->flag = 0;
            c->release_free = true;
            for (int i = 0; i < n_slices; ++i) c->base[i] = phys_base[c->qmap[i]];
        }
    }
    *handle = c;
    return SL_OK;
}

int slhip_queues_mode(void *handle, const char **why_not) {
    StepQueues *c = (StepQueues *)handle;
    if (!c) return fail(SL_E_ARG, "null pointer");
    if (why_not) *why_not = c->downgraded.empty() ? nullptr : c->downgraded.c_str();
    return c->release_free ? SL_QUEUES_RELEASE_FREE : 0;
}

int slhip_queues_selftest(void *handle, int what, int arg) {
    StepQueues *c = (StepQueues *)handle;
    if (!c) return fail(SL_E_ARG, "null pointer");
    if (what == SL_QUEUES_SELFTEST_PLANT) {
        // slice 0 is told to expect its workgroups one XCD further on than they run: the next step must raise the flag
        if (!c->release_free) return fail(SL_E_UNSUPPORTED, "no placement check in this mode");
        c->base[0] = (c->base[0] + 1) & 7;
        c->have_prepared = false;
        return SL_OK;
    }
    if (what == SL_QUEUES_SELFTEST_SWAP) {
        // from now on step t dispatches slice i on queue (i + t) mod n, behind a drain of all queues that carries NO
        // release: the envs of a slice are then stepped, in order, by the same workgroup indices of ANOTHER queue --
        // i.e., where placement is (queue constant + index) mod 8, on another XCD -- than the step before.  With a
        // stream's fences that changes nothing; without a release it must trip the placement check.
        if (arg < 0 || arg > 1 || (arg && c->n_slices < 2)) return fail(SL_E_ARG, "swap needs two slices");
        c->swap = arg != 0;
        c->have_prepared = false;
        return SL_OK;
    }
    return fail(SL_E_ARG, "unknown self-test");
}

static int queues_steps(void *handle, const sl_env_batch *env, const int32_t *actions, long long action_stride,
                        long long out_stride, int n_steps, int head, bool stage);

int slhip_queues_steps(void *handle, const sl_env_batch *env, const int32_t *actions, long long action_stride,
                       long long out_stride, int n_steps, int head) {
    return queues_steps(handle, env, actions, action_stride, out_stride, n_steps, head, false);
}

// A region of steps written ahead of time: argument blocks and packets of all n_steps are laid down, nothing is handed to
// the device; slhip_queues_go() then makes the packets valid and rings one doorbell per queue -- what a hipGraph's
// instantiate / launch split does for a stream.  The action BUFFERS must exist when the region is staged; their contents
// only when it goes.  At most SL_QUEUES_STAGE_MAX steps (the queues' argument rings), nothing else may be dispatched on
// the handle in between (a marker or a steps call hands the staged packets over early: harmless, just not deferred).
int slhip_queues_stage(void *handle, const sl_env_batch *env, const int32_t *actions, long long action_stride,
                       long long out_stride, int n_steps, int head) {
    if (n_steps > SL_QUEUES_STAGE_MAX) return fail(SL_E_ARG, "queues_stage: more steps than the argument rings hold");
    return queues_steps(handle, env, actions, action_stride, out_stride, n_steps, head, true);
}

int slhip_queues_go(void *handle) {
    if (!handle) return fail(SL_E_ARG, "null pointer");
    sl::aql_commit();
    return SL_OK;
}

static int queues_steps(void *handle, const sl_env_batch *env, const int32_t *actions, long long action_stride,
                        long long out_stride, int n_steps, int head, bool stage) {
    StepQueues *c = (StepQueues *)handle;
    if (!c || !env || !actions) return fail(SL_E_ARG, "null pointer");
    if (n_steps < 0) return fail(SL_E_ARG, "n_steps < 0");
    if (env->H != c->H || env->W != c->W || env->B != c->B) return fail(SL_E_ARG, "the queues were opened for another batch");
    if (n_steps == 0) return SL_OK;
    sl::aql_timeline_mark("slhip_queues_steps entered");
    const sl::Jump *jump;
    int rc;
    if ((rc = jump_table(&jump))) return rc;
    // one prepared launch per slice; per step only the action and output pointers are patched into its argument block
    sl::PreparedStep *const ps = c->prepared;
    // (SL_AQL_NO_CHECK=1, A/B timing only: release-free steps WITHOUT their placement check)
    static const bool no_check = getenv("SL_AQL_NO_CHECK") != nullptr;
    uint32_t *const flag = no_check ? nullptr : c->flag;
    // (the output records' pointer is patched into every dispatch -- the row kernels take it from their own argument,
    //  not from the batch description -- so windows that rotate it do not count as a change)
    sl_env_batch key;
    memcpy(&key, env, sizeof(key));                 // (bytes, padding included: the caller passes the same buffer)
    key.out = nullptr;
    key.pool_next = nullptr;                        // (patched into every dispatch as well: a refreshed pool alternates two tables)
    const bool reuse = c->have_prepared && !c->swap && memcmp(&c->prepared_env, &key, sizeof(sl_env_batch)) == 0;
    for (int i = 0; i < c->n_slices && !reuse; ++i) {
        const int lo = c->bounds[i], hi = c->bounds[i + 1];
        ps[i].grid = 0;
        if (hi <= lo) continue;
        const hipError_t err = sl::launch_env_rollout_rowlane(*env, lo, hi - lo, actions, 1, env->B, nullptr, nullptr, jump,
                                                              nullptr, &ps[i]);
        if (err != hipSuccess) return hip_fail(err, "AQL dispatch (prepare)");
        memcpy(ps[i].args + ps[i].off_base, &c->base[i], sizeof(int));
        memcpy(ps[i].args + ps[i].off_flag, &flag, sizeof(void *));
        // (compared with the per-step fields blanked: they are patched into every dispatch anyway)
        const void *none = nullptr;
        memcpy(ps[i].args + ps[i].off_actions, &none, sizeof(void *));
        memcpy(ps[i].args + ps[i].off_out, &none, sizeof(void *));
        memcpy(ps[i].args + ps[i].off_next, &none, sizeof(void *));
        if (c->last_bytes[i] != ps[i].arg_bytes || memcmp(c->last_args[i], ps[i].args, ps[i].arg_bytes)) {
            memcpy(c->last_args[i], ps[i].args, ps[i].arg_bytes);
            c->last_bytes[i] = ps[i].arg_bytes;
            ++c->version[i];
#ifndef SL_TRACE
            // a new batch (or a changed one): its block goes into every idle slot of the slice's argument ring now,
            // once (~40 us per slice), instead of ~0.6 us per dispatch for the ring's first lap
            if (!c->swap)
                sl::aql_warm(c->qmap[i], ps[i].f, ps[i].args, ps[i].arg_bytes,
                             sl::AqlPatch{c->serial * 8u + (uint32_t)i, c->version[i], 0, {}});
#endif
        }
    }
    if (!reuse) {
        memcpy(&c->prepared_env, &key, sizeof(sl_env_batch));
        c->have_prepared = true;
    }
    struct Batch {
        const bool keep;
        explicit Batch(bool keep_) : keep(keep_) { sl::aql_begin(); }
        ~Batch() {
            if (!keep) sl::aql_commit();        // (a staged region stays pending until slhip_queues_go)
        }
    } batch(stage);
    c->pending = true;
    for (int t = 0; t < n_steps; ++t) {
        const int32_t *a_t = actions + (long long)t * action_stride;
        sl_step_out *o_t = (sl_step_out *)((char *)env->out + (long long)t * out_stride * (env->out_compact ? 8 : (long long)sizeof(sl_step_out)));
        if (c->swap) {
            const hipError_t err = sl::aql_drain(c->n_phys);
            if (err != hipSuccess) return hip_fail(err, "AQL drain (self-test)");
        }
        for (int i = 0; i < c->n_slices; ++i) {
            sl::PreparedStep &p = ps[i];
            if (!p.grid) continue;
            memcpy(p.args + p.off_actions, &a_t, sizeof(void *));
            memcpy(p.args + p.off_out, &o_t, sizeof(void *));
            memcpy(p.args + p.off_next, &env->pool_next, sizeof(void *));
            const int queue = c->qmap[c->swap ? (int)((i + c->steps) % c->n_slices) : i];
            const sl::AqlLaunch a{queue, head != 0 && t == 0, c->release_free};
            sl::AqlPatch patch{c->serial * 8u + (uint32_t)i, c->version[i], 3, {p.off_actions, p.off_out, p.off_next}};
#ifdef SL_TRACE
            if (g_trace_base && g_trace_next < g_trace_slots) {
                float *trace = (float *)(g_trace_base + g_trace_bytes * g_trace_next++);
                memcpy(p.args + p.off_trace, &trace, sizeof(void *));
            }
            patch.owner = 0;                        // (profiling build: every block is written in full)
#endif
            const hipError_t err = sl::aql_dispatch(a, p.f, p.grid, p.threads, p.lds, p.args, p.arg_bytes, &patch);
            if (err != hipSuccess) return hip_fail(err, "AQL dispatch");
        }
        ++c->steps;
        // A flush of the argument ring (sfence + read-back) and the doorbells cost ~1.5 us; a step's four dispatches
        // ~1-3 us of host time, and the device needs the next step ~6 us after the last.  So the first steps go out one
        // by one (the device starts at once and is never left waiting while a batch is being written: with eight steps
        // per flush from the start the kernels' own clocks showed it idle for 7 us behind step 0), later ones -- the
        // host is ahead by then -- in fours.
        if (!stage && (t < 3 || (t & 3) == 3 || c->swap)) sl::aql_flush();
    }
    return SL_OK;
}

int slhip_queues_step(void *handle, const sl_env_batch *env, const int32_t *actions, int head) {
    return slhip_queues_steps(handle, env, actions, 0, 0, 1, head);
}

static int queues_flag(StepQueues *c) {
    if (c->release_free && *(volatile uint32_t *)c->flag)
        return fail(SL_E_HIP, "queue stepping: envs were stepped by a workgroup on another XCD than the step before, so a step "
                              "without a release fence may have read stale state -- the envs' state since the queues were "
                              "opened is not valid; open the queues without SL_QUEUES_RELEASE_FREE");
    return SL_OK;
}

int slhip_queues_marker(void *handle, long long *ticket) {
    StepQueues *c = (StepQueues *)handle;
    if (!c || !ticket) return fail(SL_E_ARG, "null pointer");
    const hipError_t err = sl::aql_marker(c->n_phys, c->pending, ticket);
    if (err != hipSuccess) return hip_fail(err, "AQL marker");
    return SL_OK;
}

int slhip_queues_wait(void *handle, long long ticket) {
    StepQueues *c = (StepQueues *)handle;
    if (!c) return fail(SL_E_ARG, "null pointer");
    const hipError_t err = sl::aql_wait(ticket);
    if (err != hipSuccess) return hip_fail(err, "AQL wait");
    return queues_flag(c);
}

int slhip_queues_sync(void *handle) {
    StepQueues *c = (StepQueues *)handle;
    if (!c) return fail(SL_E_ARG, "null pointer");
    long long ticket = -1;
    hipError_t err = sl::aql_marker(c->n_phys, c->pending, &ticket);
    if (err == hipSuccess) err = sl::aql_wait(ticket);
    if (err != hipSuccess) return hip_fail(err, "AQL fence");
    c->pending = false;
    sl::aql_timeline_mark("slhip_queues_sync returns");
    sl::aql_timeline_dump();
    return queues_flag(c);
}

int slhip_queues_close(void *handle) {
    StepQueues *c = (StepQueues *)handle;
    if (!c) return SL_OK;
    long long ticket = -1;
    hipError_t err = sl::aql_marker(c->n_phys, c->pending, &ticket);
    if (err == hipSuccess) err = sl::aql_wait(ticket);
    // (a fence that failed or timed out leaves step kernels in flight that may still write the flag: it is leaked
    //  rather than freed under them)
    if (err == hipSuccess && !sl::aql_poisoned()) {
        if (c->flag) (void)hipHostFree(c->flag);
    }
    delete c;
    return err == hipSuccess ? SL_OK : hip_fail(err, "AQL fence (close)");
}

// ---- the records of every rank's envs -> rank 0 (RCCL point-to-point calls; helpers above) ----------------------
int slhip_gather_unique_id(void *id_out) {
    if (!id_out) return fail(SL_E_ARG, "null pointer");
    const Rccl &r = rccl();
    if (!r.ok) return fail(SL_E_UNSUPPORTED, r.why);
    Rccl::unique_id id;
    const int rc = r.GetUniqueId(&id);
    if (rc) return rccl_fail(r, rc, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return SL_OK;
}

int slhip_gather_init(const void *id, int world, int rank, void **comm) {
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return fail(SL_E_ARG, "bad gather arguments");
    const Rccl &r = rccl();
    if (!r.ok) return fail(SL_E_UNSUPPORTED, r.why);
    Rccl::unique_id uid;
    memcpy(&uid, id, sizeof(uid));
    void *c = nullptr;
    const int rc = r.CommInitRank(&c, world, uid, rank);
    if (rc) return rccl_fail(r, rc, "ncclCommInitRank");
    *comm = new GatherComm{c, world, rank};
    return SL_OK;
}

static int gather_issue(GatherComm *g, const void *send, void *recv, size_t bytes, hipStream_t st);
int slhip_gather_window(void *comm, const void *send, void *recv, size_t bytes, void *stream) {
    GatherComm *g = (GatherComm *)comm;
    if (!g || !send || (g->rank == 0 && !recv)) return fail(SL_E_ARG, "bad gather arguments");
    if (bytes == 0) return SL_OK;
    return gather_issue(g, send, recv, bytes, (hipStream_t)stream);
}

static int gather_issue(GatherComm *g, const void *send, void *recv, size_t bytes, hipStream_t st) {
    const Rccl &r = rccl();
    int rc = r.GroupStart();
    if (rc) return rccl_fail(r, rc, "ncclGroupStart");
    const int kChar = 0;                                            // ncclInt8
    rc = r.Send(send, bytes, kChar, 0, g->comm, st);
    if (g->rank == 0)
        for (int peer = 0; peer < g->world && !rc; ++peer) rc = r.Recv((char *)recv + (size_t)peer * bytes, bytes, kChar, peer, g->comm, st);
    const int rc_end = r.GroupEnd();
    if (rc) return rccl_fail(r, rc, "ncclSend / ncclRecv");
    if (rc_end) return rccl_fail(r, rc_end, "ncclGroupEnd");
    return SL_OK;
}

// Which of the library's queues 0 .. n_queues - 1 does the EXCHANGE hold up when it runs on `stream`?  Measured with the
// real thing: an 8 MiB window through RCCL on the stream (every rank must make this call: the exchange is collective),
// and, as soon as the group has been issued, a one-workgroup dispatch on the queue -- alone it is back within a few
// microseconds; on a queue that takes turns with the stream's hardware queue only when RCCL's kernel has ended.
int slhip_gather_stream_shares(void *comm, int n_queues, void *stream, int *mask) {
    GatherComm *g = (GatherComm *)comm;
    if (!g || !mask || n_queues < 1 || n_queues > 8) return fail(SL_E_ARG, "bad arguments (1 to 8 queues)");
    *mask = 0;
    // COLLECTIVE: every rank issues the same 5 * n_queues exchanges whatever happens to it locally -- a rank whose queues
    // or probe kernel are unavailable still takes part (its peers would otherwise block inside RCCL's group for ever)
    // and reports its error afterwards.  Only a rank that cannot even allocate the exchange's buffers leaves early.
    std::string local_why;
    if (const char *why = sl::aql_open(n_queues)) local_why = std::string("AQL queues unavailable: ") + why;
    hipFunction_t f = nullptr;
    if (local_why.empty() && (hipGetFuncBySymbol(&f, (const void *)sl::k_xcd_probe) != hipSuccess || sl::aql_probe(f))) {
        (void)hipGetLastError();
        local_why = "the probe kernel was not found";
    }
    const size_t bytes = 8u << 20;
    char *send = nullptr, *recv = nullptr;
    uint32_t *out = nullptr;
    hipError_t err = hipMalloc((void **)&send, bytes);
    if (err == hipSuccess && g->rank == 0) err = hipMalloc((void **)&recv, bytes * (size_t)g->world);
    if (err != hipSuccess) {
        if (send) (void)hipFree(send);
        return hip_fail(err, "gather_stream_shares (exchange buffers)");
    }
    if (local_why.empty() && hipMalloc((void **)&out, 64) != hipSuccess) {
        (void)hipGetLastError();
        local_why = "no memory for the probe's output";
    }
    const hipStream_t st = (hipStream_t)stream;
    struct {
        uint32_t *out;
    } args = {out};
    int rc = SL_OK;
    hipError_t probe_err = hipSuccess;              // local: the one-workgroup dispatches
    auto tiny = [&](int q, double *us) {            // one workgroup on queue q, host time until it is back
        *us = 0;
        if (!local_why.empty() || probe_err != hipSuccess) return;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = sl::aql_dispatch(sl::AqlLaunch{q, false, false}, f, 1, 256, 0, &args, sizeof(args));
        if (e == hipSuccess) e = sl::aql_fence(n_queues);
        *us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        probe_err = e;
    };
    constexpr int REPS = 6;                         // one that sets things up, then five that count
    double alone[8], behind[8];
    for (int q = 0; q < n_queues; ++q) {
        double a_us[REPS], b_us[REPS];
        int n = 0;
        for (int rep = 0; rep < REPS; ++rep) {
            double ua = 0, ub = 0;
            (void)hipStreamSynchronize(st);
            if (local_why.empty() && probe_err == hipSuccess) probe_err = sl::aql_fence(n_queues);
            tiny(q, &ua);
            if (rc == SL_OK) rc = gather_issue(g, send, recv, bytes, st);      // (an RCCL error is every rank's error)
            tiny(q, &ub);
            if (rep > 0) {                          // (the first exchange of a stream sets things up)
                a_us[n] = ua;
                b_us[n++] = ub;
            }
        }
        // (measured: 9 us alone and 9-10 us behind the exchange on a queue it does not touch; 15 and 22 us on the one
        //  that shares a pipe with the stream's hardware queue.  MEDIANS of the five repetitions: one quiet or one noisy
        //  repetition decides nothing, on either side of the comparison.)
        std::sort(a_us, a_us + n);
        std::sort(b_us, b_us + n);
        alone[q] = a_us[n / 2];
        behind[q] = b_us[n / 2];
        if (local_why.empty() && probe_err == hipSuccess && rc == SL_OK && behind[q] > alone[q] + 3.0) *mask |= 1 << q;
    }
    err = hipStreamSynchronize(st);
    if (getenv("SL_GATHER_DEBUG"))
        for (int q = 0; q < n_queues; ++q)
            fprintf(stderr, "gather_stream_shares: queue %d alone %.1f us, behind the exchange %.1f us\n", q, alone[q], behind[q]);
    (void)hipFree(send);
    if (recv) (void)hipFree(recv);
    if (out) (void)hipFree(out);
    if (rc != SL_OK) return rc;
    if (err != hipSuccess) return hip_fail(err, "gather_stream_shares");
    if (!local_why.empty()) {
        *mask = 0;
        return fail(SL_E_UNSUPPORTED, local_why);
    }
    if (probe_err != hipSuccess) {
        *mask = 0;
        return hip_fail(probe_err, "gather_stream_shares (probe dispatch)");
    }
    return SL_OK;
}

static void gather_worker(GatherComm *g) {
    (void)hipSetDevice(g->device);
    for (;;) {
        GatherRequest rq;
        {
            // A worker that has just handed a window over polls for the next one for a while before it goes to sleep:
            // a wake-up through the condition variable costs tens of microseconds on a quiet host and milliseconds on
            // a busy one, and a window of a running step loop follows the last within a few hundred microseconds.
            // (slhip_gather_poke: the stepping thread says a window is coming -- the sleeping worker wakes up NOW and polls)
            const long long seen = g->issued.load(std::memory_order_relaxed);
            for (;;) {
                const auto t0 = std::chrono::steady_clock::now();
                while (g->pushed.load(std::memory_order_acquire) <= seen &&
                       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 3000.0)
                    __builtin_ia32_pause();
                std::unique_lock<std::mutex> lock(g->m);
                g->cv.wait(lock, [g] { return g->stop || !g->queue.empty() || g->poked; });
                g->poked = false;
                if (!g->queue.empty()) {
                    rq = g->queue.front();
                    g->queue.pop_front();
                    break;
                }
                if (g->stop) return;
            }
        }
        int rc = SL_OK;
        // a window written from the AQL queues: their marker (system-scope release behind the window's last step) is
        // waited for HERE, on this thread -- the stepping thread keeps dispatching the next window's steps meanwhile
        if (rq.marker >= 0 && sl::aql_wait(rq.marker) != hipSuccess) rc = SL_E_HIP;
        // (release-free stepping: the same verdict slhip_queues_wait would give -- records of envs that may have read
        //  stale state never reach rank 0; the ticket fails and every later call on the communicator says so)
        if (rc == SL_OK && rq.placement && *rq.placement) rc = SL_E_HIP;
        for (int i = 0; i < rq.n_writers && rc == SL_OK; ++i) {     // the exchange's stream waits for the window's writers
            if (rq.writers[i] == rq.stream) continue;
            hipError_t err = hipEventRecord(g->order_ev[i], rq.writers[i]);
            if (err == hipSuccess) err = hipStreamWaitEvent(rq.stream, g->order_ev[i], 0);
            if (err != hipSuccess) rc = SL_E_HIP;
        }
        if (rc == SL_OK) rc = gather_issue(g, rq.send, rq.recv, rq.bytes, rq.stream);
        if (rc == SL_OK && hipEventRecord(g->done_ev[rq.ticket % GatherComm::RING], rq.stream) != hipSuccess) rc = SL_E_HIP;
        {
            std::lock_guard<std::mutex> lock(g->m);
            if (rc != SL_OK && g->error == 0) g->error = rc;
            g->issued.store(rq.ticket + 1, std::memory_order_release);
        }
        g->cv_done.notify_all();
    }
}

static int gather_submit(void *comm, const void *send, void *recv, size_t bytes, void *const *writers, int n_writers,
                         void *stream, long long marker, long long *ticket, const volatile uint32_t *placement = nullptr);
int slhip_gather_window_async(void *comm, const void *send, void *recv, size_t bytes, void *const *writers, int n_writers,
                              void *stream, long long *ticket) {
    return gather_submit(comm, send, recv, bytes, writers, n_writers, stream, -1, ticket);
}

int slhip_gather_window_queued(void *comm, const void *send, void *recv, size_t bytes, void *queues, void *stream,
                               long long *ticket) {
    // the marker is put behind the window's steps now, by the stepping thread (a few packets); the wait is the worker's
    long long marker = -1;
    if (!queues) return fail(SL_E_ARG, "null pointer");
    const int rc = slhip_queues_marker(queues, &marker);
    if (rc) return rc;
    const StepQueues *c = (const StepQueues *)queues;
    return gather_submit(comm, send, recv, bytes, nullptr, 0, stream, marker, ticket, c->release_free ? c->flag : nullptr);
}

static int gather_submit(void *comm, const void *send, void *recv, size_t bytes, void *const *writers, int n_writers,
                         void *stream, long long marker, long long *ticket, const volatile uint32_t *placement) {
    GatherComm *g = (GatherComm *)comm;
    if (!g || !send || (g->rank == 0 && !recv) || n_writers < 0 || n_writers > 8 || (n_writers && !writers) || !ticket)
        return fail(SL_E_ARG, "bad gather arguments");
    std::lock_guard<std::mutex> lock(g->m);
    if (g->error) return fail(g->error, "an earlier asynchronous gather failed");
    if (!g->worker.joinable()) {                        // first use: events on this device, then the thread
        hipError_t err = hipGetDevice(&g->device);
        for (int i = 0; i < 8 && err == hipSuccess; ++i) err = hipEventCreateWithFlags(&g->order_ev[i], hipEventDisableTiming);
        for (int i = 0; i < GatherComm::RING && err == hipSuccess; ++i) err = hipEventCreateWithFlags(&g->done_ev[i], hipEventDisableTiming);
        if (err != hipSuccess) return hip_fail(err, "gather events");
        g->worker = std::thread(gather_worker, g);
    }
    if (g->submitted - g->issued.load(std::memory_order_acquire) >= GatherComm::RING)
        return fail(SL_E_ARG, "too many gather windows in flight");
    GatherRequest rq;
    rq.send = send, rq.recv = recv, rq.bytes = bytes, rq.n_writers = n_writers, rq.stream = (hipStream_t)stream;
    for (int i = 0; i < n_writers; ++i) rq.writers[i] = (hipStream_t)writers[i];
    rq.marker = marker;
    rq.placement = placement;
    rq.ticket = g->submitted++;
    *ticket = rq.ticket;
    g->queue.push_back(rq);
    g->pushed.store(g->submitted, std::memory_order_release);
    g->cv.notify_one();
    return SL_OK;
}

int slhip_gather_poke(void *comm) {
    GatherComm *g = (GatherComm *)comm;
    if (!g) return fail(SL_E_ARG, "null pointer");
    if (!g->worker.joinable()) return SL_OK;            // (no asynchronous window yet: nobody to wake)
    {
        std::lock_guard<std::mutex> lock(g->m);
        g->poked = true;
    }
    g->cv.notify_one();
    return SL_OK;
}

int slhip_gather_done(void *comm, long long ticket, int block, int *done) {
    GatherComm *g = (GatherComm *)comm;
    if (!g || !done || ticket < 0) return fail(SL_E_ARG, "bad gather arguments");
    *done = 0;
    if (g->issued.load(std::memory_order_acquire) <= ticket) {
        if (!block) return SL_OK;
        // (polled first, for the same reason as in the worker)
        const auto t0 = std::chrono::steady_clock::now();
        while (g->issued.load(std::memory_order_acquire) <= ticket &&
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 3000.0)
            __builtin_ia32_pause();
        if (g->issued.load(std::memory_order_acquire) <= ticket) {
            std::unique_lock<std::mutex> lock(g->m);
            g->cv_done.wait(lock, [g, ticket] { return g->issued.load(std::memory_order_acquire) > ticket || g->error; });
        }
    }
    if (g->error) return fail(g->error, "asynchronous gather failed");
    if (g->submitted - ticket > GatherComm::RING) {     // its event slot has been reused: long finished
        *done = 1;
        return SL_OK;
    }
    hipEvent_t ev = g->done_ev[ticket % GatherComm::RING];
    if (block) {
        // (the runtime's own wait: polling hipEventQuery ahead of it contends with the runtime's completion handling --
        //  measured again in round 4, 11.3-13.6 instead of 10.8-11.0 us per step over a 20-step region)
        hipError_t err = hipEventSynchronize(ev);
        if (err != hipSuccess) return hip_fail(err, "hipEventSynchronize");
        *done = 1;
        return SL_OK;
    }
    *done = hipEventQuery(ev) == hipSuccess ? 1 : 0;
    (void)hipGetLastError();
    return SL_OK;
}

int slhip_gather_wait_streams(void *comm, long long ticket, void *const *streams, int n_streams) {
    GatherComm *g = (GatherComm *)comm;
    if (!g || ticket < 0 || n_streams < 0 || (n_streams && !streams)) return fail(SL_E_ARG, "bad gather arguments");
    int done = 0;
    {       // the group must have been enqueued before a stream can be made to wait for it
        std::unique_lock<std::mutex> lock(g->m);
        g->cv_done.wait(lock, [g, ticket] { return g->issued.load(std::memory_order_acquire) > ticket || g->error; });
    }
    if (g->error) return fail(g->error, "asynchronous gather failed");
    (void)done;
    if (g->submitted - ticket > GatherComm::RING) return SL_OK;        // slot reused: that window finished long ago
    for (int i = 0; i < n_streams; ++i) {
        hipError_t err = hipStreamWaitEvent((hipStream_t)streams[i], g->done_ev[ticket % GatherComm::RING], 0);
        if (err != hipSuccess) return hip_fail(err, "hipStreamWaitEvent");
    }
    return SL_OK;
}

int slhip_gather_destroy(void *comm) {
    GatherComm *g = (GatherComm *)comm;
    if (!g) return SL_OK;
    if (g->worker.joinable()) {
        {
            std::lock_guard<std::mutex> lock(g->m);
            g->stop = true;
        }
        g->cv.notify_all();
        g->worker.join();
        for (auto &e : g->order_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : g->done_ev)
            if (e) (void)hipEventDestroy(e);
    }
    const Rccl &r = rccl();
    const int rc = r.ok ? r.CommDestroy(g->comm) : 0;
    delete g;
    return rc ? rccl_fail(r, rc, "ncclCommDestroy") : SL_OK;
}

int slhip_side_effects(const sl_env_batch *env, const sl_episode_queue *queue, int num_samples, int derive_streams,
                       uint16_t *work_boards,
                       float *work_prob, int32_t *work_steps, sl_pcg64 *work_rng, int32_t *counts, uint16_t *keys,
                       double *life_dist, uint8_t *type_masks, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!queue || queue->capacity < 0) return fail(SL_E_ARG, "bad queue");
    if (num_samples < 1 || num_samples > 65535) return fail(SL_E_ARG, "num_samples outside 1..65535");
    if (queue->capacity == 0) return SL_OK;
    if (!queue->count || !queue->records || !queue->boards || !work_boards || !work_prob || !work_steps || !work_rng ||
        !counts || !keys || !life_dist || !type_masks)
        return fail(SL_E_ARG, "null pointer");
    const int H = env->H, W = env->W, C = queue->capacity;
    if (!sl::rowlane_supports(H, W) || force_generic() || (((uintptr_t)work_boards | (uintptr_t)queue->boards) & 15))
        return fail(SL_E_UNSUPPORTED, "side-effect pass: board shape without row kernels (or unaligned boards)");
    const sl::Jump *jump;
    if ((rc = jump_table(&jump))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t board_counts = (size_t)H * W * 8;
    hipError_t err;
    if (derive_streams) {
        // one fused launch over two runs of C boards: [roll b0 forward num_steps, then sample] and [sample the final
        // boards], every board with a stream of its own -- a chain of num_steps + num_samples CA steps instead of
        // num_steps + 2 * num_samples, and twice the wavefronts in flight
        err = sl::launch_se_gather(*env, *queue, work_boards, work_prob, work_steps, work_rng, true, st);
        if (err == hipSuccess)
            err = sl::launch_occupancy_rowlane(work_boards, counts, board_counts, 2 * C, queue->count, C, work_steps, H, W,
                                               work_prob, num_samples, work_rng, jump, st);
    } else {
        // the reference's order on ONE generator per entry: roll-forward, inaction tensor, action tensor
        err = sl::launch_se_gather(*env, *queue, work_boards, work_prob, work_steps, nullptr, false, st);
        if (err == hipSuccess)
            err = sl::launch_advance_rowlane(work_boards, work_boards, C, H, W, work_prob, 0, work_steps, queue->count,
                                             work_rng, jump, st);
        if (err == hipSuccess)
            err = sl::launch_occupancy_rowlane(work_boards, counts, board_counts, C, queue->count, 0, nullptr, H, W,
                                               work_prob, num_samples, work_rng, jump, st);
        if (err == hipSuccess)
            err = sl::launch_occupancy_rowlane(queue->boards, counts + (size_t)C * board_counts, board_counts, C,
                                               queue->count, 0, nullptr, H, W, work_prob, num_samples, work_rng, jump, st);
    }
    if (err == hipSuccess)
        err = sl::launch_se_distributions(*env, *queue, counts, (double)num_samples, keys, life_dist, type_masks, st);
    return err == hipSuccess ? SL_OK : hip_fail(err, "side_effects launch");
}

static int check_multi(const sl_env_batch *env, const sl_multi_agent *m) {
    int rc = check_env(env);
    if (rc) return rc;
    if (!m) return fail(SL_E_ARG, "null multi-agent description");
    if (m->n_agents < 1 || m->n_agents > SL_MAX_AGENTS) return fail(SL_E_ARG, "n_agents outside 1..SL_MAX_AGENTS");
    if (!m->agents || !m->pool_agents || !m->out) return fail(SL_E_ARG, "null pointer in sl_multi_agent");
    if (env->wrap.flags || env->finished.capacity > 0 || env->policy_obs)
        return fail(SL_E_UNSUPPORTED, "wrappers, the finished-episode queue and the policy layout are single-agent features");
    return SL_OK;
}

int slhip_env_step_multi(const sl_env_batch *env, const sl_multi_agent *multi, const int32_t *actions, void *stream) {
    int rc = check_multi(env, multi);
    if (rc) return rc;
    if (!actions) return fail(SL_E_ARG, "null actions");
    if (env->B == 0) return SL_OK;
    const sl::Jump *jump;
    if ((rc = jump_table(&jump))) return rc;
    hipError_t err = drop_goal_cache(env, (hipStream_t)stream);
    if (err == hipSuccess) err = sl::launch_env_step_multi(*env, *multi, actions, jump, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "env_step_multi launch");
}

int slhip_env_reset_multi(const sl_env_batch *env, const sl_multi_agent *multi, const uint8_t *mask, void *stream) {
    int rc = check_multi(env, multi);
    if (rc) return rc;
    if (env->B == 0) return SL_OK;
    hipError_t err = drop_goal_cache(env, (hipStream_t)stream);
    if (err == hipSuccess) err = sl::launch_env_reset_multi(*env, *multi, mask, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "env_reset_multi launch");
}

int slhip_env_obs(const sl_env_batch *env, void *stream) {
    int rc = check_env(env);
    if (rc) return rc;
    if (env->B == 0) return SL_OK;
    if (use_rowlane(env, 0)) {
        // the fused kernel with zero steps: load, observation epilogue, store (its action argument is only read,
        // never used, with T == 0: any B readable int32 do)
        const sl::Jump *jump;
        if ((rc = jump_table(&jump))) return rc;
        hipError_t err = sl::launch_env_rollout_rowlane(*env, 0, env->B, (const int32_t *)env->scalars, 0, env->B, nullptr,
                                                        nullptr, jump, (hipStream_t)stream);
        return err == hipSuccess ? SL_OK : hip_fail(err, "env_obs launch");
    }
    hipError_t err = sl::launch_env_obs_generic(*env, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "env_obs launch");
}

int slhip_sample_actions(const float *probs, int B, int n_actions, unsigned long long seed, unsigned long long counter,
                         int32_t *actions, void *stream) {
    if (B < 0 || n_actions < 1 || n_actions > 64) return fail(SL_E_ARG, "bad sizes");
    if (!probs || !actions) return fail(SL_E_ARG, "null pointer");
    if (B == 0) return SL_OK;
    hipError_t err = sl::launch_sample_actions(probs, B, n_actions, seed, counter, actions, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "sample_actions launch");
}

int slhip_obs_to_policy(const uint32_t *view, int B, int vh, int vw, const int32_t *channels, int C, void *out,
                        int dtype, void *stream) {
    if (B < 0 || vh < 1 || vw < 1) return fail(SL_E_ARG, "bad view shape");
    if (C < 1 || C > SL_MAX_CHANNELS || !channels) return fail(SL_E_ARG, "bad channel list");
    if (dtype != 0 && dtype != 1) return fail(SL_E_ARG, "dtype must be 0 (uint8) or 1 (float32)");
    if (!view || !out) return fail(SL_E_ARG, "null pointer");
    sl::sl_channel_list ch;
    for (int k = 0; k < C; ++k) {
        if (channels[k] < 0 || channels[k] > 31) return fail(SL_E_ARG, "channel outside 0..31");
        ch.c[k] = channels[k];
    }
    if (B == 0) return SL_OK;
    hipError_t err = sl::launch_obs_to_policy(view, B, vh, vw, ch, C, out, dtype, (hipStream_t)stream);
    return err == hipSuccess ? SL_OK : hip_fail(err, "obs_to_policy launch");
}

}  // extern "C"
