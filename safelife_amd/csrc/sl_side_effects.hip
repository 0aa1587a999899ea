// sl_side_effects.hip -- the episode-end pass of side_effect_score (safelife/side_effects.py:103-130) for the
// episodes the step kernels queued (sl_episode_queue), every stage on the device and sized by the queue's
// capacity: the number of valid entries is read on the device (min(*count, capacity)), never by the host.
//
//   k_se_gather          entry -> starting board b0 = pool_board[level], spawn_prob, num_steps, and the entry's own
//                        PCG64 stream (the reference draws all of this pass from its process-wide generator:
//                        side_effects.py runs outside use_rng -- nothing to be bit-compatible with; a stream per
//                        entry, derived from the level's generator, the env and the episode, keeps the pass
//                        reproducible and order-independent)
//   (slhip_advance_board_each / slhip_life_occupancy kernels: roll-forward :108, occupancy tensors :109-110)
//   k_se_distributions   :111-130: which life colours occur (total_counts[i] > 0) and their float64
//                        distributions counts / (num_runs * num_samples); the frozen-but-movable cell types of
//                        the starting board (np.unique order) with their 0/1 masks over b0 and the final board
#include "sl_device.h"
#include "sl_kernels.h"

namespace sl {

namespace {

constexpr int SE_THREADS = 256;
constexpr u32 MOVABLE = PUSHABLE | PULLABLE;       // CellTypes.movable
constexpr u32 LIFE = ALIVE | DESTRUCTIBLE;         // CellTypes.life
constexpr int COLOR_SHIFT = 9;                     // CellTypes.color_bit

__device__ __forceinline__ int se_valid(const sl_episode_queue &q) {
    const int n = *q.count;
    return n < q.capacity ? n : q.capacity;
}

// two_runs: the work arrays hold two runs of `capacity` boards -- run 0 the starting boards (to be rolled forward
// num_steps steps, then sampled), run 1 copies of the boards the agents left (sampled as they are) -- each with a
// random stream of its own, so that ONE fused launch can work on both at once.  Otherwise only run 0 is filled
// and the stream of an entry (if rng is given) is consumed by the three stages in the reference's order.
__global__ __launch_bounds__(SE_THREADS) void k_se_gather(sl_env_batch env, sl_episode_queue q, u16 *__restrict__ work_boards,
                                                         float *__restrict__ spawn_prob, int32_t *__restrict__ num_steps,
                                                         sl_pcg64 *__restrict__ rng, int two_runs) {
    const int slot = blockIdx.x, tid = threadIdx.x, HW = env.H * env.W, C = q.capacity;
    if (slot >= se_valid(q)) {
        if (tid == 0) {                              // an empty slot rolls nothing forward
            num_steps[slot] = 0;
            if (two_runs) num_steps[C + slot] = 0;
        }
        return;
    }
    const sl_episode_record rec = q.records[slot];
    const u16 *src = env.pool_board + (size_t)rec.level * HW;
    u16 *dst = work_boards + (size_t)slot * HW;
    for (int i = tid; i < HW; i += SE_THREADS) dst[i] = src[i];
    if (two_runs) {
        const u16 *fin = q.boards + (size_t)slot * HW;
        u16 *dst2 = work_boards + (size_t)(C + slot) * HW;
        for (int i = tid; i < HW; i += SE_THREADS) dst2[i] = fin[i];
    }
    if (tid == 0) {
        spawn_prob[slot] = rec.spawn_prob;
        num_steps[slot] = rec.num_steps;
        if (two_runs) {
            spawn_prob[C + slot] = rec.spawn_prob;
            num_steps[C + slot] = 0;
        }
        if (rng) {
            sl_pcg64 gen = env.pool_rng[rec.level];
            sl_episode_stream(gen.state_hi, gen.state_lo, 0x5EFFEC75 ^ rec.env, rec.episode_idx);
            rng[slot] = gen;
            if (two_runs) {
                sl_episode_stream(gen.state_hi, gen.state_lo, 0x2B0A2D5 ^ rec.env, rec.episode_idx);
                rng[C + slot] = gen;
            }
        }
    }
}

// One workgroup per entry.  counts: int32 [2, capacity, H, W, 8] (inaction, action).
__global__ __launch_bounds__(SE_THREADS) void k_se_distributions(sl_env_batch env, sl_episode_queue q,
                                                                const int32_t *__restrict__ counts, double denominator,
                                                                uint16_t *__restrict__ keys, double *__restrict__ life_dist,
                                                                uint8_t *__restrict__ type_masks) {
    __shared__ unsigned int bitmap[2048];            // one bit per uint16 cell value
    __shared__ int totals[8];
    __shared__ int scan[SE_THREADS];
    const int slot = blockIdx.x, tid = threadIdx.x, HW = env.H * env.W;
    uint16_t *my_keys = keys + (size_t)slot * SL_SE_MAX_KEYS;
    if (slot >= se_valid(q)) {
        if (tid < SL_SE_MAX_KEYS) my_keys[tid] = 0xFFFFu;
        return;
    }
    const sl_episode_record rec = q.records[slot];
    const u16 *b0 = env.pool_board + (size_t)rec.level * HW;
    const u16 *b2 = q.boards + (size_t)slot * HW;
    const int32_t *cnt0 = counts + (size_t)slot * HW * 8, *cnt1 = counts + ((size_t)q.capacity + slot) * HW * 8;
    for (int i = tid; i < 2048; i += SE_THREADS) bitmap[i] = 0u;
    if (tid < 8) totals[tid] = 0;
    __syncthreads();
    // total_counts[i] > 0 (side_effects.py:111): does colour i occur at all, in either run
    int seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 2 * HW; i += SE_THREADS) {
        const int32_t *cell = i < HW ? cnt0 + (size_t)i * 8 : cnt1 + (size_t)(i - HW) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) seen[c] |= cell[c] != 0;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (seen[c]) atomicOr(&totals[c], 1);
    // frozen things the agent may push or destroy (:125-128)
    for (int i = tid; i < HW; i += SE_THREADS) {
        const u32 c = b0[i];
        if ((c & FROZEN) && (c & (DESTRUCTIBLE | MOVABLE)) && !(c & AGENT)) atomicOr(&bitmap[c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    // keys: the eight life colours first (0xFFFF where absent), then the cell types in ascending order
    if (tid < 8) my_keys[tid] = totals[tid] ? (uint16_t)(LIFE | (tid << COLOR_SHIFT)) : (uint16_t)0xFFFFu;
    int mine = 0;
    for (int w = 0; w < 8; ++w) mine += __popc(bitmap[tid * 8 + w]);
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < SE_THREADS; off <<= 1) {            // inclusive scan
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int pos = scan[tid] - mine;
    const int n_types = min(scan[SE_THREADS - 1], SL_SE_MAX_KEYS - 8);
    for (int w = 0; w < 8 && mine > 0; ++w) {
        u32 bits = bitmap[tid * 8 + w];
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            if (pos < SL_SE_MAX_KEYS - 8) my_keys[8 + pos] = (uint16_t)((tid * 8 + w) * 32 + b);
            ++pos;
        }
    }
    if (tid < SL_SE_MAX_KEYS - 8 && tid >= n_types) my_keys[8 + tid] = 0xFFFFu;
    // how many such cell types the board really has (saturating): more than the key slots hold means the
    // device-side distributions of this entry are incomplete, and the host says so (SideEffectBatch)
    if (tid == 0) q.records[slot].n_cell_types = (uint8_t)min(scan[SE_THREADS - 1], 255);
    __syncthreads();
    // life distributions: float64 counts / denominator, laid out [2, 8, H, W]
    double *ld = life_dist + (size_t)slot * 2 * 8 * HW;
    for (int i = tid; i < 2 * HW; i += SE_THREADS) {
        const int run = i / HW, cell = i - run * HW;
        const int32_t *src = (run ? cnt1 : cnt0) + (size_t)cell * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) ld[((size_t)run * 8 + c) * HW + cell] = (double)src[c] / denominator;
    }
    // type masks: uint8 [2, SL_SE_MAX_KEYS - 8, H, W]: (b0 == c), (b2 == c)
    uint8_t *tm = type_masks + (size_t)slot * 2 * (SL_SE_MAX_KEYS - 8) * HW;
    for (int k = 0; k < n_types; ++k) {
        const u16 c = my_keys[8 + k];
        for (int i = tid; i < HW; i += SE_THREADS) {
            tm[((size_t)0 * (SL_SE_MAX_KEYS - 8) + k) * HW + i] = b0[i] == c;
            tm[((size_t)1 * (SL_SE_MAX_KEYS - 8) + k) * HW + i] = b2[i] == c;
        }
    }
}

}  // namespace

hipError_t launch_se_gather(const sl_env_batch &env, const sl_episode_queue &q, u16 *work_boards, float *spawn_prob,
                            int32_t *num_steps, sl_pcg64 *rng, bool two_runs, hipStream_t stream) {
    hipLaunchKernelGGL(k_se_gather, dim3(q.capacity), dim3(SE_THREADS), 0, stream, env, q, work_boards, spawn_prob,
                       num_steps, rng, two_runs ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_se_distributions(const sl_env_batch &env, const sl_episode_queue &q, const int32_t *counts,
                                   double denominator, uint16_t *keys, double *life_dist, uint8_t *type_masks,
                                   hipStream_t stream) {
    hipLaunchKernelGGL(k_se_distributions, dim3(q.capacity), dim3(SE_THREADS), 0, stream, env, q, counts, denominator,
                       keys, life_dist, type_masks);
    return hipGetLastError();
}

}  // namespace sl
