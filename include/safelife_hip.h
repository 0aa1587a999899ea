/*
 * safelife_hip.h -- C-ABI of libsafelife_hip.so: the MI355X (gfx950) implementation of the
 * SafeLife per-step hot path.  This is the drop-in boundary: every entry point replaces one
 * prototype of the reference's native layer (safelife/speedups_src/advance_board.h:3-16) or one
 * method of its Python step surface, batched over B independent boards.
 *
 * Conventions (all entry points)
 *   - Plain pointers and sizes only; no torch / Python types.  Every array pointer is a DEVICE
 *     pointer (HBM) unless the parameter is documented as "host".  The library allocates nothing
 *     per call and retains no pointer; the caller owns every buffer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work.
 *   - Return value: 0 on success, negative on failure (SL_E_*); slhip_last_error() gives a
 *     thread-local message.  The Python shim turns SL_E_SHAPE into the reference's ValueError text.
 *   - Boards are C-contiguous uint16 [B,H,W], the cell bit layout of constants.h:4-33
 *     (== safelife_game.py:75-123).  3 <= H, W; H*W <= SL_MAX_CELLS.
 *   - Random numbers: one numpy-compatible PCG64 (XSL-RR 128/64) stream per board replaces the
 *     reference's process-global bit generator (random.c:21-22,76-83).  Draws are consumed exactly
 *     as advance_board.c:115 consumes them (eligible cells only, row-major), so a board stepped
 *     here with the state of numpy generator G equals the reference stepped under G, and the
 *     state written back equals G's state afterwards.
 */
#ifndef SAFELIFE_HIP_H
#define SAFELIFE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL_ABI_VERSION 13       /* 13: slhip_queues_stage / slhip_queues_go, sl_env_batch.pool_ready / out_compact, sl_level_scalars.ready, slhip_env_step_multi / _reset_multi; 12: slhip_pool_write; 11: sl_env_batch.goal_cache, slhip_goal_cache_bytes; 10: slhip_queues_open_on, slhip_queues_stream_shares, slhip_gather_stream_shares, slhip_gather_poke */
#define SL_MAX_CELLS 16384        /* H*W limit of one board */
#define SL_MAX_CHANNELS 32

enum sl_status {
    SL_OK = 0,
    SL_E_SHAPE = -1,      /* bad board shape / sizes (module.c:32-35,112-114,170-179) */
    SL_E_ARG = -2,        /* null pointer or inconsistent argument */
    SL_E_HIP = -3,        /* HIP runtime error, see slhip_last_error() */
    SL_E_UNSUPPORTED = -4
};

/* numpy.random.PCG64 state as four 64-bit words (state = hi:lo, inc = hi:lo). */
typedef struct sl_pcg64 {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
} sl_pcg64;

int slhip_abi_version(void);
const char *slhip_last_error(void);
/* Number of HIP devices visible, or a negative sl_status. */
int slhip_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Batched primitives == the five prototypes of advance_board.h, one launch for B boards.
 * ------------------------------------------------------------------------------------------ */

/* advance_board_nstep (advance_board.h:6-7; Python advance_board(board, spawn_prob, n_step),
 * module.c:20-49).  out may alias in.  spawn_prob: [B] float (the reference parses a C float,
 * module.c:26, and compares the draw against it promoted to double, advance_board.c:35,115).
 * rng: [B], read and written back advanced by the number of draws consumed. */
int slhip_advance_board(const uint16_t *in, uint16_t *out, int B, int H, int W,
                        const float *spawn_prob, int n_steps, sl_pcg64 *rng, void *stream);

/* The same with one step count per board (n_steps: int32 [B]): side_effect_score rolls every
 * finished episode's starting board forward by that episode's length (side_effects.py:108). */
int slhip_advance_board_each(const uint16_t *in, uint16_t *out, int B, int H, int W,
                             const float *spawn_prob, const int32_t *n_steps, sl_pcg64 *rng, void *stream);

/* life_occupancy (advance_board.h:9-10, module.c:52-81).  counts: int32 [B,H,W,8], overwritten
 * (the reference accumulates into a zeroed array): per cell and colour, the number of steps
 * 1..n_steps after which the cell is ALIVE and not AGENT|EXIT|FROZEN (advance_board.c:153-161). */
int slhip_life_occupancy(const uint16_t *in, int32_t *counts, int B, int H, int W,
                         const float *spawn_prob, int n_steps, sl_pcg64 *rng, void *stream);

/* alive_counts (advance_board.h:12, module.c:99-132).  out: int64 [B,8,9], overwritten.
 * HW = cells per board (any shape, flat, as the reference). */
int slhip_alive_counts(const uint16_t *board, const uint16_t *goals, int B, int HW,
                       int64_t *out, void *stream);

/* execute_actions (advance_board.h:14-16, module.c:155-202).  board and locs are updated in
 * place; locs: int64 [B,A,2] as (row, col); actions: int64, element (b,k) at
 * actions[b*action_batch_stride + k*action_stride] (action_stride 0 broadcasts one action to all
 * agents of a board, module.c:185).  Agents of one board act in index order. */
int slhip_execute_actions(uint16_t *board, int B, int H, int W, int64_t *locs,
                          const int64_t *actions, int A, int action_stride,
                          int action_batch_stride, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused single-agent SafeLifeEnv.step()/reset() over B device-resident environments
 * (safelife_env.py:148-218 with the game glue of safelife_game.py:505-552,657-719,746-761 and
 * the observation of safelife_env.py:105-146 / helper_utils.py:42-75).
 * The struct lives on the HOST; every pointer inside is a device pointer.
 * ------------------------------------------------------------------------------------------ */
/* Per-env scalar state: one 64-byte record per env (a single wide load/store per step instead of a
 * dozen strided ones).  Field <-> reference attribute: */
typedef struct sl_env_scalars {
    int32_t agent_row, agent_col; /* SafeLifeGame.agent_locs[0] (row, col); row < 0 => no agent */
    int32_t num_steps;            /* GameState.num_steps */
    int32_t old_value;            /* SafeLifeEnv._old_game_value */
    int32_t required_points;      /* GameWithGoals.required_points() */
    int32_t initial_points;       /* sum(points_table * initial_counts) */
    int32_t table_idx;            /* row of points_table used by this env */
    int32_t level_idx;            /* pool level currently loaded */
    int32_t episode_idx;          /* episodes finished by this env */
    int32_t episode_length;       /* SafeLifeEnv.episode_length */
    float episode_reward;         /* SafeLifeEnv.episode_reward */
    float spawn_prob;             /* GameState.spawn_prob */
    int32_t goals_static;         /* SafeLifeGame._static_goals: 0 None, 1 True, 2 False */
    int32_t is_active;            /* SafeLifeEnv._is_active */
    int32_t exit_open_at_reset;   /* can_exit() during the reset's update_exit_colors (the exit paint of
                                     SimpleSideEffectPenalty's starting-state baseline) */
    int32_t loaded;               /* != 0 once a level has been loaded into this slot: slhip_env_reset() then moves on
                                     to the next level (level_idx += level_stride, episode_idx += 1), as
                                     SafeLifeEnv.reset() takes next(level_iterator) (safelife_env.py:204) */
} sl_env_scalars;

/* What one step() returns per env besides the observation (16 bytes). */
typedef struct sl_step_out {
    float reward;                 /* np.float32 reward of safelife_env.py:157,172 */
    uint8_t done;
    uint8_t success;              /* info['episode']['success'] */
    uint8_t times_up;             /* info['times_up'] */
    uint8_t reserved;
    float episode_reward;         /* info['episode']['reward'] (before any auto-reset) */
    int32_t episode_length;       /* info['episode']['length'] */
} sl_step_out;

/* Per-level constants of the pool (32 bytes). */
typedef struct sl_level_scalars {
    int32_t agent_row, agent_col;
    int32_t required_reset;       /* required_points while resetting (first observation) */
    int32_t required_step;        /* required_points for the steps that follow (differs under
                                     MinPerformanceScheduler, env_wrappers.py:142-145) */
    int32_t initial_points;
    int32_t table_idx;
    float spawn_prob;
    int32_t ready;                /* written by the library where sl_env_batch.pool_ready is kept (round 6), else unused:
                                     (old_value << 1) | exit_open -- what SafeLifeEnv.reset() computes on this level
                                     (safelife_env.py:203-218: the score of the fresh board + the exit bonus if the agent
                                     stands on an open exit; whether update_exit_colors opened the exits) */
} sl_level_scalars;

/* Training-wrapper math of safelife/env_wrappers.py, applied per env inside the step in the order
 * of training/env_factory.py:277-283:
 *   MovementBonusWrapper   :32-98   reward += movement_bonus * speed**power [- movement_bonus]
 *   ExtraExitBonus         :120-128 reward += done * bonus * episode_reward   (unless times_up)
 *   SimpleSideEffectPenalty:150-213 reward -= penalty_coef * (side_effect - last_side_effect),
 *                                   baseline = "starting-state" (the board right after reset) or, with
 *                                   SL_WRAP_INACTION, "inaction" (:179-180: that board advanced once per step)
 * (MinPerformanceScheduler :131-147 has no per-step arithmetic: sl_level_scalars.required_step.)
 * The wrapped reward is float64, as in the reference (np.float32 + np.float64), evaluated with the
 * same operations in the same order; speed**power comes from a host-built table so no pow() runs on
 * the device.  flags == 0 switches all of it off.
 * "inaction": the reference advances the baseline board outside any game method, so its spawners draw from the
 * process-wide generator (safelife/random.py:13).  With many envs that is one generator per env, inaction_rng[e];
 * an env whose generator starts in the state the process-wide one had reproduces the reference run exactly
 * (tests/golden/trace_wrap_inaction_*).  The baselines are advanced inside the step kernels (row-kernel shapes: a
 * third pass of the kernel's CA loop, every step of a T-step launch, through streams and queues alike; other shapes: a
 * launch of its own in front of each step, T = 1 only); an env whose num_steps is 0 takes its current board as the
 * baseline first, which is what the wrapper's reset() does. */
#define SL_WRAP_MOVEMENT 1
#define SL_WRAP_AS_PENALTY 2            /* MovementBonusWrapper.as_penalty */
#define SL_WRAP_EXIT_BONUS 4
#define SL_WRAP_SIDE_EFFECT 8
#define SL_WRAP_IGNORE_REWARD_CELLS 16  /* SimpleSideEffectPenalty.ignore_reward_cells */
#define SL_WRAP_INACTION 32             /* SimpleSideEffectPenalty.baseline == "inaction" (with SL_WRAP_SIDE_EFFECT) */
#define SL_WRAP_MAX_PERIOD 8

typedef struct sl_wrap_state {    /* per env, 48 bytes */
    int32_t n_prior;              /* positions held by MovementBonusWrapper's deque (<= period) */
    int32_t last_side_effect;     /* SimpleSideEffectPenalty.last_side_effect */
    int16_t prior[SL_WRAP_MAX_PERIOD][2];   /* (row, col), oldest first */
    int32_t reserved[2];
} sl_wrap_state;

typedef struct sl_wrappers {
    int32_t flags;                /* SL_WRAP_* */
    int32_t move_period;          /* movement_bonus_period, 1..SL_WRAP_MAX_PERIOD */
    int32_t move_table_len;       /* >= H + W + move_period */
    int32_t reserved;
    double move_bonus;            /* movement_bonus */
    double exit_bonus;            /* ExtraExitBonus.bonus */
    double penalty_coef;          /* SimpleSideEffectPenalty.penalty_coef */
    const double *move_table;     /* [move_table_len]: movement_bonus * (d / period) ** power, d = 0.. */
    sl_wrap_state *state;         /* [B] */
    double *shaped_reward;        /* [B] out: what the outermost wrapper's step() returns as reward */
    double *shaped_reward_t;      /* [T,B] per-step copy for slhip_env_rollout, or NULL */
    uint32_t *pool_baseline;      /* workspace [L, H, (W+1)/2]: every pool level as it stands right after
                                     reset, player bits cleared, in the row kernels' register layout;
                                     filled by slhip_env_prepare() (needed with SL_WRAP_SIDE_EFFECT) */
    uint16_t *inaction_board;     /* SL_WRAP_INACTION: [B,H,W] the baseline boards (state; 16-byte aligned) */
    sl_pcg64 *inaction_rng;       /* SL_WRAP_INACTION: [B] the baselines' generators (state) */
    uint32_t *inaction_rows;      /* (unused since ABI 9: the row kernels lay the advanced baselines out in LDS) */
} sl_wrappers;

/* Finished episodes, queued on the device by the step kernels for the side-effect pass (safelife_env.py:183-192
 * runs side_effect_score() inside the step that ends an episode; with thousands of envs the episode-end work
 * is batched instead).  The step that ends an env's episode -- done while the env was still active -- takes
 * slot = atomic_add(count, 1) and, if slot < capacity, writes a record and a copy of the board as the agent
 * left it (after update_exit_colors, before any auto-reset reloads the slot).  capacity == 0 switches the
 * queue off.  The consumer (slhip_side_effects) reads min(*count, capacity) entries; it never resets *count:
 * the caller alternates two queues and clears the idle one (hipMemsetAsync) before handing it back. */
typedef struct sl_episode_record {    /* 32 bytes */
    int32_t env;                  /* index of the env in the batch */
    int32_t level;                /* pool level the episode was played on: its starting board */
    int32_t num_steps;            /* GameState.num_steps when it ended */
    int32_t episode_idx;          /* the env's episode counter during that episode */
    float spawn_prob;
    float episode_reward;
    int32_t episode_length;
    uint8_t success, times_up;
    uint8_t n_cell_types;         /* written by slhip_side_effects: frozen movable / destructible cell types on the
                                     starting board (saturating); more than SL_SE_MAX_KEYS - 8 means keys and
                                     type_masks of this entry are cut short (rebuild them from `counts` on the host) */
    uint8_t reserved;
} sl_episode_record;

typedef struct sl_episode_queue {
    int32_t capacity;             /* slots; 0 = no queue */
    int32_t env_base;             /* added to the env index written into records (slices of a batch) */
    int32_t *count;               /* device int32: episodes pushed so far (beyond capacity: dropped) */
    sl_episode_record *records;   /* [capacity] */
    uint16_t *boards;             /* [capacity,H,W] */
} sl_episode_queue;

typedef struct sl_env_batch {
    int32_t B, H, W, E;          /* envs; board dims; exit slots per env (>= 1) */
    int32_t time_limit;          /* SafeLifeEnv.time_limit (safelife_env.py:65) */
    int32_t exit_points;         /* GameState.points_on_level_exit (safelife_game.py:156) */
    int32_t n_tables;
    int32_t auto_reset;          /* !=0: an env whose episode ends reloads its next pool level */
    int32_t remove_white_goals;  /* safelife_env.py:66,125-126 */
    int32_t view_h, view_w;      /* SafeLifeEnv.view_shape */
    int32_t n_channels;          /* len(output_channels); 0 => raw uint32 view */
    int32_t channels[SL_MAX_CHANNELS];
    int32_t spawner_free;        /* !=0: the caller guarantees that no board or goal array of the batch and
                                    of the pool holds a SPAWNING cell (the rules never create one), which
                                    lets the kernels drop the random-draw machinery; 0 = no promise */
    int32_t stream_salt;         /* 0: an env that loads pool level l starts from pool_rng[l] exactly (what replaying
                                    the reference's traces needs: every recorded level carries its own generator).
                                    != 0: the episode's generator is pool_rng[l] moved to a state derived from
                                    (stream_salt + env index, episode_idx), so envs that replay one pool level --
                                    and successive replays by one env -- see different random streams, as the
                                    reference's per-game SeedSequence children do (level_iterator.py:218).
                                    Use 1 + (global index of env 0) so that shards of one run do not collide */
    /* per-env state */
    uint16_t *board;             /* [B,H,W] */
    uint16_t *goals;             /* [B,H,W] */
    int32_t *exit_locs;          /* [B,E] flat cell index of GameState.exit_locs, -1 = unused */
    sl_pcg64 *rng;               /* [B] SafeLifeGame._rng */
    sl_env_scalars *scalars;     /* [B] */
    const int32_t *points_table; /* [n_tables,8,9] (safelife_game.py:595-605) */
    /* level pool: the device-resident counterpart of SafeLifeLevelIterator */
    int32_t L;
    int32_t level_stride;        /* next level of an env = (level_idx + level_stride) % L */
    const uint16_t *pool_board;  /* [L,H,W] as loaded (before update_exit_colors) */
    const uint16_t *pool_goals;  /* [L,H,W] */
    const int32_t *pool_exit_locs;      /* [L,E] */
    const sl_pcg64 *pool_rng;           /* [L] */
    const sl_level_scalars *pool_scalars; /* [L] */
    const int32_t *pool_next;    /* optional [L]: an env on slot s loads slot pool_next[s] at its next reset instead of
                                    (s + level_stride) % L.  What a pool that is REFRESHED while the envs step needs
                                    (safelife_amd.levels.LevelPool.stage / commit: new levels go into spare slots and the
                                    table is switched between two steps -- the device-resident counterpart of
                                    level_iterator.py:200-223 handing every reset a fresh level).  Read at launch time:
                                    the queue launcher patches the pointer into every dispatch.  NULL = the rule above */
    /* per-step outputs */
    sl_step_out *out;            /* [B] */
    uint8_t *obs;                /* [B,vh,vw,C] uint8, or uint32 [B,vh,vw] if n_channels == 0; NULL = skip */
    void *policy_obs;            /* the same observation as the policy network takes it (training/models.py:100-103
                                    transposes (h,w,c) -> (c,w,h); training/ppo.py:64 casts to float32), written by
                                    the step / reset kernels themselves: [B,C,vw,vh], policy_obs[b][c][x][y] = bit
                                    channels[c] of the view word at (y,x); uint8 or float32 (policy_dtype).  Needs
                                    n_channels >= 1.  NULL = skip */
    int32_t policy_dtype;        /* 0 = uint8, 1 = float32 */
    int32_t out_compact;         /* != 0: `out` takes 8-byte records -- the first half of sl_step_out: reward, done, success,
                                    times_up -- instead of 16-byte ones (what a learner on another rank needs of every
                                    step: half the bytes a gather window carries; sharding.RewardGather(record="compact")) */
    /* workspace */
    int8_t *score_lut;           /* [n_tables,4096+65536] per-cell score tables derived from points_table by
                                    slhip_env_prepare(); NULL => the size-generic kernels are used */
    uint32_t *goal_cache;        /* optional workspace of slhip_goal_cache_bytes() bytes, 16-byte aligned, ZEROED by the
                                    caller: the fused step kernels keep the goal colours of boards whose goals are
                                    static (safelife_game.py:753-760) in the form their lanes use, and a launch whose
                                    boards all have them there moves no goal array at all.  The kernels maintain it
                                    across steps, resets and slices; whoever writes `goals`, `scalars.goals_static` or
                                    `scalars.level_idx` from OUTSIDE the library zeroes it again.  Only batches without
                                    observation and wrappers use it (with a finished-episode queue: the wide board shapes).
                                    The kernels that keep it hold no goal image in LDS at all: WITHOUT the workspace
                                    (NULL) they fetch every lane's goal row from global memory at every step -- correct,
                                    and slower; give such batches their cache */
    uint16_t *pool_ready;        /* optional workspace [L,H,W] (round 6): every pool level as an episode STARTS on it --
                                    pool_board after the reset's update_exit_colors (safelife_game.py:537-552) -- kept by
                                    slhip_env_prepare() and slhip_pool_write() together with pool_scalars[l].ready.
                                    What a reset computes on a level depends on the level alone, so an env whose episode
                                    ends inside a step kernel COPIES its next level from here instead of scoring and
                                    repainting it (one pass over the board, the leaders' exit walk and a workgroup
                                    barrier less on the chain of the one workgroup its whole launch waits for).
                                    NULL: the step kernels work it out at every reset, as before */
    sl_wrappers wrap;            /* training wrappers; wrap.flags == 0 => none */
    sl_episode_queue finished;   /* episodes that ended, for the side-effect pass; finished.capacity == 0 => none */
} sl_env_batch;

/* Derive env->score_lut from env->points_table, and env->wrap.pool_baseline / env->pool_ready from the level pool
 * (call once, and again whenever points_table or the pool changes).
 * Synchronises the stream.  Returns SL_E_UNSUPPORTED when a table entry does not fit int8; the
 * caller then passes score_lut = NULL and every shape runs on the size-generic kernels. */
int slhip_env_prepare(const sl_env_batch *env, void *stream);

/* Rebuild env->wrap.pool_baseline from the level pool as it stands now, asynchronously on `stream` (what
 * slhip_env_prepare does once, synchronously): after pool slots have been rewritten. */
int slhip_pool_baseline(const sl_env_batch *env, void *stream);

/* New levels for pool slots no env can load at present (the spare bank of a refreshable pool: pool_next, above), in one
 * kernel on `stream`: the level iterator's hand-over (level_iterator.py:200-223) for a device-resident pool.  Every
 * source may be host memory the device can read (pinned): the kernel fetches it when it runs, so it stays untouched
 * until the stream has passed the launch. */
typedef struct sl_pool_rows {
    int32_t n;                   /* levels */
    const int32_t *slot;         /* [n] pool slots to write, distinct, < env->L */
    const uint16_t *board;       /* [n,H,W] */
    const uint16_t *goals;       /* [n,H,W] */
    const int32_t *exit_locs;    /* [n,E] */
    const sl_pcg64 *rng;         /* [n] */
    const sl_level_scalars *scalars; /* [n] */
    const int32_t *next;         /* optional [L]: a successor table, copied to next_dst by the same launch */
    int32_t *next_dst;
} sl_pool_rows;
int slhip_pool_write(const sl_env_batch *env, const sl_pool_rows *rows, void *stream);

/* Bytes of env->goal_cache for this batch as described by *env -- call it with the observation, wrapper and queue fields
 * already set (0: no step kernel of this batch keeps a cache -- leave goal_cache NULL).
 * *boards_per_block (optional): the cache is one block of bytes / ceil(B / boards_per_block) bytes per group of that many
 * consecutive envs; a block's first 32-bit word is its flag (1: the group steps on cached goal words). */
size_t slhip_goal_cache_bytes(const sl_env_batch *env, int *boards_per_block);

/* SafeLifeEnv.reset() for the envs with mask[e] != 0 (mask NULL = all): an env that has never been loaded
 * takes pool level level_idx[e]; any other moves on to (level_idx[e] + level_stride) % L and counts an
 * episode, exactly as the in-kernel auto-reset does.  Writes the first observation. */
int slhip_env_reset(const sl_env_batch *env, const uint8_t *mask, void *stream);

/* One SafeLifeEnv.step() for every env.  actions: int32 [B] in 0..8. */
int slhip_env_step(const sl_env_batch *env, const int32_t *actions, void *stream);

/* T consecutive steps in ONE launch (boards stay on chip between steps).  actions: int32 [T,B].
 * reward_t / done_t: optional [T,B] per-step outputs (NULL = only the last step's outputs in
 * env->reward/done are kept).  Observations are produced for the final state only. */
int slhip_env_rollout(const sl_env_batch *env, const int32_t *actions, int T,
                      float *reward_t, uint8_t *done_t, void *stream);

/* Do kernels on stream_a and stream_b overlap on the device?  HIP multiplexes streams onto a few hardware queues
 * (four by default); two streams that share one run their kernels one after the other, which silently turns sliced
 * stepping into serial stepping (measured: 24 instead of 13.5 us per step).  Which streams collide depends on what
 * else the process has created (RCCL, other envs).  The probe runs two idle 100 us one-wavefront kernels, one per
 * stream, and compares the wall time with one; *concurrent = 1 if they overlapped.  Synchronises both streams. */
int slhip_streams_concurrent(void *stream_a, void *stream_b, int *concurrent);

/* Stream ordering without a host wait: every stream of `after` waits for what has been enqueued so far on every
 * stream of `before` (HOST arrays of hipStream_t; a stream that appears on both sides is skipped).  The fence and join
 * of sliced stepping: order({caller}, slices) before a step whose actions the caller's stream produced, and
 * order(slices, {caller}) before the caller's stream reads the outputs.  Events come from a ring inside the library. */
int slhip_streams_order(void *const *before, int n_before, void *const *after, int n_after);

/* One step for every env, issued as n_slices launches: slice i = envs [bounds[i], bounds[i+1]) on
 * streams[i] (bounds: HOST int32 [n_slices+1], bounds[0] = 0, bounds[n_slices] = B; streams: HOST array of
 * hipStream_t).  Envs are independent, so the slices need no ordering among themselves: on distinct
 * streams the load / compute / store phases and the launch boundaries of one slice overlap those of the
 * others, step after step (a single launch per step leaves the chip idle at both ends of every launch).
 * The caller orders each stream against whoever produces `actions` and consumes the outputs.
 * actions: int32 [B].  Slice bounds should be multiples of 64
 * envs so that every slice keeps the 16-byte alignment the row kernels' DMA needs. */
int slhip_env_step_slices(const sl_env_batch *env, int n_slices, const int32_t *bounds, const int32_t *actions,
                          void *const *streams);

/* One step for the envs [first, first + count) only, one launch on `stream` (actions: int32 [B], indexed by the env's
 * index in the batch; `first` a multiple of 64 keeps the row kernels' alignment).  What a pipelined driver uses: the
 * policy of one group of envs runs while another group steps (training/base_algo.py:208-238 walks its envs one by one);
 * with a group's policy, action draw and step all on the group's own stream nothing needs a fence
 * (safelife_amd/runner.py: PipelinedRunner). */
int slhip_env_step_range(const sl_env_batch *env, int first, int count, const int32_t *actions, void *stream);

/* Sliced stepping WITHOUT HIP's launch path: slhip_env_step_slices hides one slice's kernel boundary under the other
 * slices' kernels, but a launch through a HIP stream costs the host 2.4-3 us, so one stepping thread feeds two slices per
 * ~8 us step and no more.  These entry points issue the SAME kernel from HSA queues of the library's own, one per slice
 * (csrc/sl_aql.hip): a dispatch is a 64-byte packet and a doorbell, and the packets and argument blocks of MANY steps
 * are written by one call.  Ordering is a stream's: every dispatch waits for its queue's previous one (barrier bit) with
 * agent-scope acquire / release fences -- independent of where workgroups run.
 *   open   : slices as in slhip_env_step_slices (1 to 8); row-kernel shapes only
 *            (SL_E_UNSUPPORTED otherwise, or when the HSA runtime offers no queue -- callers then keep to streams).
 *   steps  : n_steps consecutive steps of every env, all enqueued by this call (the queue rings give back-pressure: the
 *            call blocks while they are full).  Step t takes its actions from actions + t * action_stride (int32
 *            elements; device int32 [B] each, complete when the call is made) and writes its sl_step_out records to
 *            env->out + t * out_stride (records; 0: every step overwrites env->out).  head != 0 for the first step
 *            after anything OUTSIDE the queues wrote the envs' state or the actions through a HIP stream: the caller
 *            has synchronised those streams, and step 0 is dispatched with a system-scope acquire.
 *   step   : steps(handle, env, actions, 0, 0, 1, head).
 *   marker : a system-scope release behind every step dispatched so far, with a completion signal; returns at once
 *            (*ticket = -1: nothing was outstanding).  wait: the calling thread -- any thread -- waits for it.  Only then
 *            may HIP streams or the host read what the steps in front of the marker wrote.
 *   sync   : marker + wait.
 *   mode   : the flags in effect (why_not, optional: why SL_QUEUES_RELEASE_FREE was asked for and not granted, or NULL).
 *
 * SL_QUEUES_RELEASE_FREE (open's flags; OPT-IN, never the default): the steps of a queue go without the RELEASE fence
 * (it alone costs ~0.9 us of a 7.5 us step of 8192 25x25 envs: the write-back of the XCDs' L2s).  What a step wrote then
 * stays in the L2 of the XCD its workgroups ran on, and the next step's workgroup of the same index reads it there --
 * which is valid ONLY while a workgroup index of a QUEUE keeps running on the same XCD.  MI355X runs workgroup i of
 * every dispatch of a queue on XCD (q + i) mod 8, q a constant of the queue (tools/ubench/xcd_place.hip,
 * profiles/round4_a_xcd_placement.txt: whatever the grid, the kernel, the queue's history and the other queues are
 * doing), but no programming guide promises it, so
 *   - open probes it (a probe kernel, three dispatches per queue: q of every queue, and that every workgroup follows
 *     the formula) and falls back to the fenced mode where it does not hold (slhip_queues_mode reports that), and
 *   - EVERY step verifies it: every workgroup compares the XCD it runs on with (q + i) mod 8 of its slice's queue --
 *     scalar code, no memory access -- and raises a host-visible flag if it finds itself anywhere else; wait / sync
 *     then -- and from then on -- return SL_E_HIP: the envs' state since the queues were opened is not valid, and the
 *     caller starts over without the flag.  Use it where losing a run to that is acceptable (benchmarks, restartable
 *     roll-outs).
 *
 * selftest (tests only): SL_QUEUES_SELFTEST_PLANT tells slice 0 to expect its workgroups one XCD further on than they run;
 * SL_QUEUES_SELFTEST_SWAP (arg 1 / 0: on / off) dispatches step t's slice i on queue (i + t) mod n behind a host-side
 * drain of all queues that carries no release, so that every env is stepped -- in order -- by a workgroup of another
 * QUEUE than the step before, which on MI355X means another XCD -- harmless with a stream's fences, a real misplacement
 * for release-free stepping (tests/test_hip_parity.py shows both). */
#define SL_QUEUES_RELEASE_FREE 1
#define SL_QUEUES_SELFTEST_PLANT 1
#define SL_QUEUES_SELFTEST_SWAP 2
int slhip_queues_open(const sl_env_batch *env, int n_slices, const int32_t *bounds, int flags, void **handle);
/* The same with slice i on queue queue_ids[i] (distinct, 0 to 7; NULL: slice i on queue i) -- for callers that leave
 * out a queue another kernel of theirs would hold up (slhip_gather_stream_shares). */
int slhip_queues_open_on(const sl_env_batch *env, int n_slices, const int32_t *bounds, const int32_t *queue_ids, int flags,
                         void **handle);
/* Which of the queues 0 .. n_queues - 1 does a long one-wavefront kernel on HIP `stream` hold up (bit q of *mask)?
 * (On MI355X: none -- a kernel that fits next to the steps runs beside them whatever its queue.) */
int slhip_queues_stream_shares(int n_queues, void *stream, int *mask);
int slhip_queues_mode(void *handle, const char **why_not);
int slhip_queues_steps(void *handle, const sl_env_batch *env, const int32_t *actions, long long action_stride,
                       long long out_stride, int n_steps, int head);
int slhip_queues_step(void *handle, const sl_env_batch *env, const int32_t *actions, int head);
/* steps() split in two (round 6): stage() writes the argument blocks and packets of all n_steps (<= SL_QUEUES_STAGE_MAX)
 * WITHOUT handing anything to the device; go() makes them valid and rings one doorbell per queue.  The action buffers must
 * exist when the region is staged, their contents when it goes -- a caller stages the next region while it still waits
 * for the last one's results, and starts it with a few hundred nanoseconds of host work.  Nothing else is dispatched on
 * the handle in between (a marker or a steps() call hands the staged packets over early: harmless, just not deferred). */
#define SL_QUEUES_STAGE_MAX 48
int slhip_queues_stage(void *handle, const sl_env_batch *env, const int32_t *actions, long long action_stride,
                       long long out_stride, int n_steps, int head);
int slhip_queues_go(void *handle);
int slhip_queues_marker(void *handle, long long *ticket);
int slhip_queues_wait(void *handle, long long ticket);
int slhip_queues_sync(void *handle);
int slhip_queues_close(void *handle);
int slhip_queues_selftest(void *handle, int what, int arg);

/* The episode-end pass of side_effect_score() (side_effects.py:103-130) for every episode in `queue` (what
 * safelife_env.py:183-192 runs inside the step that ends an episode), all on the device and without a host
 * read: every stage is launched over queue->capacity entries and stops at min(*queue->count, capacity).
 *   1. b0 = pool_board[record.level]; b1 = advance_board(b0, spawn_prob, record.num_steps)       (:108)
 *   2. counts[:,0] = life_occupancy(b1, p, num_samples); counts[:,1] = life_occupancy(queue board, ...)  (:109-110)
 *   3. the distributions of :111-130: keys / life_dist / type_masks below.
 * Random draws: the reference takes them from its process-wide generator (side_effects.py runs outside
 * use_rng), so there is nothing to be stream-compatible with; every entry gets its own PCG64 stream derived
 * from its level's generator, env index and episode index, consumed in the reference's order (roll-forward,
 * inaction tensor, action tensor); derive_streams == 0 takes the caller's work_rng [C] as they are instead (how
 * the reference's recorded generator states are replayed).  The earth-mover distances stay on the host (pyemd:
 * parity unpinned).
 * All buffers are caller-owned device memory sized by the queue's capacity C:
 *   work_boards  uint16 [2C,H,W]     run 0: b0 (rolled forward in place when derive_streams == 0); run 1: copies of
 *                                    the final boards (derive_streams != 0: both runs are worked on by ONE fused
 *                                    launch, roll-forward and sampling in the same kernel)
 *   work_prob    float  [2C],  work_steps int32 [2C],  work_rng sl_pcg64 [2C]  (derive_streams == 0: the first C)
 *   counts       int32  [2,C,H,W,8]  out: the occupancy tensors, counts[0] = inaction, counts[1] = action
 *   keys         uint16 [C,SL_SE_MAX_KEYS] out: slots 0-7 = CellTypes.life | colour i where total_counts[i] > 0,
 *                else 0xFFFF; slots 8.. = the frozen, movable-or-destructible, non-agent cell values of b0 in
 *                ascending order (np.unique), 0xFFFF-padded (more than SL_SE_MAX_KEYS-8 of them: the rest are cut)
 *   life_dist    double [C,2,8,H,W]  out: counts / num_samples as float64 (num_runs = 1), colour-major
 *   type_masks   uint8  [C,2,SL_SE_MAX_KEYS-8,H,W] out: (b0 == key), (final board == key) for slots 8..
 * Supported for the board shapes of the row kernels (elsewhere SL_E_UNSUPPORTED: use the primitives). */
#define SL_SE_MAX_KEYS 24
int slhip_side_effects(const sl_env_batch *env, const sl_episode_queue *queue, int num_samples, int derive_streams,
                       uint16_t *work_boards, float *work_prob, int32_t *work_steps, sl_pcg64 *work_rng,
                       int32_t *counts, uint16_t *keys, double *life_dist, uint8_t *type_masks, void *stream);

/* ---- multi-GPU: the per-step records of every rank's envs -> rank 0 ----------------------------
 * Boards never cross GPUs; the only exchange of the path is what a learner on rank 0 needs from the other ranks:
 * their sl_step_out records.  A rank's step kernels write the records of a WINDOW of consecutive steps straight
 * into one device buffer; these entry points move a window to rank 0 with RCCL point-to-point calls
 * (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd: xGMI is point-to-point, the seven peers' windows arrive over
 * seven links at once) on a stream of the caller's choosing, so the exchange overlaps the following steps.  RCCL is
 * loaded on first use (dlopen of librccl.so.1 -- the copy the process already has, if any); a process that never
 * gathers never touches it.  Replaces, for this path, what training/base_algo.py does with Python lists of
 * per-env results (there is one process and no exchange in the reference).
 *
 *   slhip_gather_unique_id   rank 0: a fresh ncclUniqueId (SL_GATHER_ID_BYTES bytes, host memory) for the caller to
 *                            broadcast over whatever it has (the existing torch.distributed process group)
 *   slhip_gather_init        every rank, same id: ncclCommInitRank on the current device; *comm is an opaque handle
 *   slhip_gather_window      one window: rank r sends `bytes` from `send` to rank 0; rank 0 receives rank r's window
 *                            at recv + r * bytes (its own included), all inside one RCCL group, enqueued on `stream`.
 *                            recv is ignored on the other ranks.  Returns once the calls are enqueued.
 *   slhip_gather_destroy     releases the communicator */
#define SL_GATHER_ID_BYTES 128
int slhip_gather_unique_id(void *id_out);
int slhip_gather_init(const void *id, int world, int rank, void **comm);
int slhip_gather_window(void *comm, const void *send, void *recv, size_t bytes, void *stream);
int slhip_gather_destroy(void *comm);

/* Which of the library's step queues 0 .. n_queues - 1 (slhip_queues_*) does the exchange hold up when it runs on
 * `stream` (bit q of *mask)?  RCCL's kernel is dispatched from one of HIP's hardware queues; the step queue that takes
 * turns with it stands still for the length of the exchange (~40 us per window next to a running step loop) while the
 * others step on.  Measured with a real 8 MiB exchange per queue -- COLLECTIVE: every rank calls it, with the same
 * n_queues.  A caller that steps through queues then opens them with slhip_queues_open_on, leaving the marked queue out. */
int slhip_gather_stream_shares(void *comm, int n_queues, void *stream, int *mask);

/* The same hand-off off the caller's thread: the request is queued and a worker thread of the library issues the
 * stream ordering (the exchange's `stream` waits for what is enqueued on the window's `writers`, HOST array of up to 8
 * hipStream_t, at the time the WORKER gets to it -- later than the call, never earlier, which is safe because the
 * writers' later launches fill the OTHER window), the RCCL group, and an event behind it.  The stepping thread pays
 * for a queue push (~1 us) instead of the runtime calls (20 us, 35-80 us with busy queues).  *ticket identifies the
 * window: slhip_gather_done (is it finished?  block != 0 waits on the host) and slhip_gather_wait_streams (make streams
 * wait for it, e.g. before its buffers are written or read again; waits on the host only until the worker has
 * enqueued the group).  At most 8 windows may be outstanding. */
int slhip_gather_window_async(void *comm, const void *send, void *recv, size_t bytes, void *const *writers, int n_writers,
                              void *stream, long long *ticket);
/* A window will be handed over within the next few milliseconds: the worker thread, if it sleeps, wakes up now and polls
 * for the request (a wake-up at the hand-over itself costs tens of microseconds, milliseconds on a busy host). */
int slhip_gather_poke(void *comm);
int slhip_gather_done(void *comm, long long ticket, int block, int *done);
int slhip_gather_wait_streams(void *comm, long long ticket, void *const *streams, int n_streams);
/* A window that was written from the library's AQL queues (slhip_queues_steps): the call puts a marker behind the steps
 * dispatched so far on `queues` (the handle of slhip_queues_open) and returns; the worker thread waits for the marker
 * -- not the stepping thread, which goes on enqueuing the next window's steps -- and then issues the RCCL group on
 * `stream`.  Tickets as above. */
int slhip_gather_window_queued(void *comm, const void *send, void *recv, size_t bytes, void *queues, void *stream,
                               long long *ticket);

/* SafeLifeEnv.get_obs() for the current state. */
int slhip_env_obs(const sl_env_batch *env, void *stream);

/* ---- multi-agent batches (round 6): SafeLifeEnv(single_agent=False), safelife_env.py:148-218 with :162-170 NOT taken --
 * every board carries n_agents agents (advance_board.c:217-220: their actions in index order), each with its own points
 * table, exit condition, reward, done flag and episode accumulators (safelife_game.py:505-552,657-719 evaluated per
 * agent; the exits turn red when ANY agent may leave), and an observation centred on itself.  The fused step is the
 * size-generic family's (one workgroup per board, any shape).  `env` supplies the boards, goals, generators, exit tables,
 * the level pool, the points tables and the view; of sl_env_scalars only num_steps, level_idx, episode_idx, spawn_prob,
 * goals_static and loaded are used (the per-agent words live in `agents`).  env->obs and env->out are ignored: the
 * per-agent outputs are `multi`'s.  An env whose agents are ALL done reloads its next pool level inside the step when
 * env->auto_reset is set (training/base_algo.py:231-236 resets on np.all(done)); wrappers, the finished-episode queue
 * and the policy layout are single-agent features (SL_E_UNSUPPORTED). */
typedef struct sl_agent_state {   /* per env and agent (48 bytes) */
    int32_t row, col;             /* where the agent is -- or was when it left the board (GameState.agent_locs) */
    int32_t old_value;            /* SafeLifeEnv._old_game_value[a] */
    int32_t required_points;      /* SafeLifeGame.required_points()[a] */
    int32_t initial_points;       /* sum(points_table[a] * initial_counts) */
    int32_t table_idx;            /* the agent's table in env->points_table */
    int32_t episode_length;
    float episode_reward;
    int32_t is_active;            /* SafeLifeEnv._is_active[a] */
    int32_t reserved[3];
} sl_agent_state;
typedef struct sl_level_agent {   /* per pool level and agent (32 bytes); every level of the pool has exactly n_agents agents */
    int32_t row, col, required_reset, required_step, initial_points, table_idx, reserved[2];
} sl_level_agent;
typedef struct sl_multi_agent {
    int32_t n_agents;             /* A, 1 to SL_MAX_AGENTS */
    int32_t reserved;
    sl_agent_state *agents;       /* [B, A] */
    const sl_level_agent *pool_agents;   /* [L, A] */
    sl_step_out *out;             /* [B, A]: one record per agent and step */
    uint8_t *obs;                 /* [B, A, view_h, view_w, n_channels] uint8, or uint32 [B, A, view_h, view_w]; NULL = skip */
} sl_multi_agent;
#define SL_MAX_AGENTS 8
/* actions: int32 [B, A] in 0..8 (an agent that is done takes 0, as training/base_algo.py:216-219 feeds it). */
int slhip_env_step_multi(const sl_env_batch *env, const sl_multi_agent *multi, const int32_t *actions, void *stream);
/* SafeLifeEnv.reset() for the envs with mask[e] != 0 (NULL: all), as slhip_env_reset. */
int slhip_env_reset_multi(const sl_env_batch *env, const sl_multi_agent *multi, const uint8_t *mask, void *stream);

/* The raw uint32 view (output_channels=None, safelife_env.py:141) -> the tensor the policy network
 * convolves: channel-first with the spatial axes swapped, out[b][c][x][y] = (view[b][y][x] >> channels[c]) & 1,
 * i.e. what training/models.py:100-103 (obs.transpose(-1, -3)) and the float cast of training/ppo.py:64
 * produce from the (h, w, c) uint8 observation.  dtype: 0 = uint8, 1 = float32.  view: [B,vh,vw],
 * out: [B,C,vw,vh].  The step then only writes 4 bytes per view cell instead of C.
 * `channels` is a HOST pointer (C int32 values, copied into the launch); view and out are device pointers.
 * (sl_env_batch.policy_obs makes the step kernel write this layout itself.) */
int slhip_obs_to_policy(const uint32_t *view, int B, int vh, int vw, const int32_t *channels, int C,
                        void *out, int dtype, void *stream);

/* One categorical draw per env from policy probabilities, on the device, written as the int32 action the step kernels
 * read (training/ppo.py:66-69 draws on the host with numpy; there is no random stream to be compatible with).
 * probs: float32 [B, n_actions] (rows sum to 1); the draw of env e in call `counter` of a run seeded `seed` is a
 * function of (seed, counter, e) alone -- splitmix64, 24-bit uniform, inverse CDF -- so runs are reproducible and
 * shards may use their global env indices through `seed`. */
int slhip_sample_actions(const float *probs, int B, int n_actions, unsigned long long seed, unsigned long long counter,
                         int32_t *actions, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAFELIFE_HIP_H */
