// sl_kernels.h -- launcher prototypes shared between the kernel translation units and the C-ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "sl_device.h"

namespace sl {

size_t generic_lds_bytes(int HW, int nbuf);

// sl_generic.hip : one workgroup per board, any 3 <= H, W with H*W <= SL_MAX_CELLS
hipError_t launch_advance_generic(const u16 *in, u16 *out, int B, int H, int W, const float *spawn_prob,
                                  int n_steps, sl_pcg64 *rng, const Jump *jump, int32_t *occupancy,
                                  hipStream_t stream, const int32_t *n_each = nullptr);
hipError_t launch_alive_counts(const u16 *board, const u16 *goals, int B, int HW, int64_t *out,
                               hipStream_t stream);
hipError_t launch_execute_actions(u16 *board, int B, int H, int W, int64_t *locs, const int64_t *actions,
                                  int A, int action_stride, int action_batch_stride, hipStream_t stream);
hipError_t launch_env_rollout_generic(const sl_env_batch &env, const int32_t *actions, int T,
                                      float *reward_t, uint8_t *done_t, const Jump *jump,
                                      hipStream_t stream);
hipError_t launch_env_reset_generic(const sl_env_batch &env, const uint8_t *mask, hipStream_t stream);
// SimpleSideEffectPenalty's "inaction" baseline, one CA step on (all envs of `env`)
hipError_t launch_inaction_generic(const sl_env_batch &env, const Jump *jump, hipStream_t stream);
hipError_t launch_env_obs_generic(const sl_env_batch &env, hipStream_t stream);
// multi-agent boards (sl_multi_agent): the fused step / reset, one workgroup per board
hipError_t launch_env_step_multi(const sl_env_batch &env, const sl_multi_agent &m, const int32_t *actions, const Jump *jump,
                                 hipStream_t stream);
hipError_t launch_env_reset_multi(const sl_env_batch &env, const sl_multi_agent &m, const uint8_t *mask, hipStream_t stream);
struct sl_channel_list {
    int32_t c[SL_MAX_CHANNELS];
};
hipError_t launch_sample_actions(const float *probs, int B, int A, unsigned long long seed, unsigned long long counter,
                                 int32_t *actions, hipStream_t stream);
hipError_t launch_obs_to_policy(const u32 *view, int B, int vh, int vw, const sl_channel_list &ch, int C, void *out,
                                int dtype, hipStream_t stream);

// sl_rowlane.hip : row-per-lane SWAR kernels for the shapes listed in SL_ROWLANE_SHAPES
bool rowlane_supports(int H, int W);
// size of sl_env_batch.goal_cache for a batch of B boards (0: no row kernels for the shape); zeroing it lowers every flag
// (spawn: the pool holds spawners -- which plain step kernel the batch runs, and whether that one keeps a cache)
size_t rowlane_goal_cache_bytes(int H, int W, int B, bool spawn, int *boards_per_block = nullptr);
bool rowlane_lean_takes_queue(int H, int W);       // the shape's plain step kernels also serve a finished-episode queue
int rowlane_policy_room(int H, int W);
hipError_t launch_build_score_lut(const int32_t *points_table, int n_tables, int8_t *lut, hipStream_t stream);
hipError_t launch_build_baseline(const sl_env_batch &env, hipStream_t stream);
hipError_t launch_idle(long long ticks, hipStream_t stream);
// n_each (device, optional): one step count per board instead of n_steps; n_valid (device, optional): only
// the first *n_valid boards exist
hipError_t launch_advance_rowlane(const u16 *in, u16 *out, int B, int H, int W, const float *spawn_prob,
                                  int n_steps, const int32_t *n_each, const int32_t *n_valid, sl_pcg64 *rng,
                                  const Jump *jump, hipStream_t stream);
// counts_stride: int32 elements between the outputs of consecutive boards (H*W*8 when dense).  n_valid / valid_period
// (device count, optional): boards come in runs of valid_period (0: one run) of which the first *n_valid exist.
// pre_steps (device, optional): per board, CA steps to run before counting starts.
hipError_t launch_occupancy_rowlane(const u16 *in, int32_t *counts, size_t counts_stride, int B, const int32_t *n_valid,
                                    int valid_period, const int32_t *pre_steps, int H, int W, const float *spawn_prob,
                                    int n_steps, sl_pcg64 *rng, const Jump *jump, hipStream_t stream);

// sl_side_effects.hip : the episode-end pass of side_effect_score for queued episodes
hipError_t launch_se_gather(const sl_env_batch &env, const sl_episode_queue &q, u16 *work_boards, float *spawn_prob,
                            int32_t *num_steps, sl_pcg64 *rng, bool two_runs, hipStream_t stream);
hipError_t launch_se_distributions(const sl_env_batch &env, const sl_episode_queue &q, const int32_t *counts,
                                   double denominator, uint16_t *keys, double *life_dist, uint8_t *type_masks,
                                   hipStream_t stream);
// envs [e_first, e_first + e_count) of the batch; actions / reward_t / done_t are indexed [t * tstride + e]
// with the env's index in the whole batch
// (sl_aql.hip dispatches the same kernel from queues of the library's own: PreparedStep below)
struct AqlLaunch {
    int queue;              // index of the queue (one per slice)
    bool head;              // first step after work of HIP streams: system-scope acquire
    bool release_free;      // opt-in: no release fence behind this step (valid while workgroup i of the slice's queue keeps
                            // its XCD: the kernel checks itself against the placement found when the queues were opened)
};
// A single-step launch of the fused kernel that is not issued but handed back: kernel handle, geometry and the packed
// argument block with the offsets of the fields that change from step to step -- what the queues dispatch, any number
// of times, with those fields patched in.
struct PreparedStep {
    hipFunction_t f;
    unsigned grid, threads, lds;
    size_t arg_bytes;
    size_t off_actions, off_out, off_base, off_flag, off_trace, off_next;
    alignas(16) unsigned char args[1024];
};
// prepared (optional, T == 1 only): fill it instead of launching
// T == -1: SafeLifeEnv.reset() of the envs with reset_mask[e] != 0 (device uint8 [B] indexed by the env's index in the
// batch; null: all) instead of steps; `actions` is then only a readable dummy
hipError_t launch_env_rollout_rowlane(const sl_env_batch &env, int e_first, int e_count, const int32_t *actions,
                                      int T, int tstride, float *reward_t, uint8_t *done_t, const Jump *jump,
                                      hipStream_t stream, PreparedStep *prepared = nullptr,
                                      const uint8_t *reset_mask = nullptr);

// sl_aql.hip : user-mode queues of the library's own next to HIP's streams
const char *aql_open(int n_queues);                 // null when usable, else why not
const char *aql_probe(hipFunction_t f);             // can HIP's kernel `f` be found in the HSA executables? (null: yes)
hipFunction_t rowlane_probe_function();             // any kernel of the library (sl_rowlane.hip)
// patch (optional): the argument block is `version` of `owner`'s (any non-zero ids the caller keeps unique); where the
// queue's next argument slot already holds exactly that, only the 8-byte words at `offset[0..n)` are written again
struct AqlPatch {
    uint32_t owner, version;
    int n;
    size_t offset[4];
};
hipError_t aql_dispatch(const AqlLaunch &a, hipFunction_t f, unsigned grid, unsigned threads, unsigned lds,
                        const void *args, size_t arg_bytes, const AqlPatch *patch = nullptr);
// write `owner`'s argument block into every idle slot of the queue's argument ring (dispatches then only patch)
void aql_warm(int queue, hipFunction_t f, const void *args, size_t arg_bytes, const AqlPatch &patch);
void aql_begin();                                   // dispatches between begin and commit go out in batches: their
void aql_flush();                                   // argument blocks are flushed once per batch (flush: hand over what
void aql_commit();                                  // has been written so far; commit: flush and leave batch mode)
// marker: a barrier packet with a system-scope release and a completion signal behind everything dispatched so far on
// queues [0, n_queues) that have work since their last marker (force: on every one of them); returns at once.
// *ticket = -1 when there was nothing to mark.  wait: the calling thread spins on the marker's signals (any thread; the
// queues stay usable meanwhile).  fence = marker + wait.
hipError_t aql_marker(int n_queues, bool force, long long *ticket);
hipError_t aql_wait(long long ticket);
hipError_t aql_fence(int n_queues);
hipError_t aql_drain(int n_queues);                // (self-test) every queue idle, no cache action
void aql_timeline_mark(const char *what);           // SL_AQL_TIMELINE=1 (debugging): a host-side mark ...
void aql_timeline_dump();                           // ... and the timeline since the last dump, on stderr
bool aql_poisoned();                                // a wait timed out: work may still be in flight

}  // namespace sl
