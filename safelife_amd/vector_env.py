"""
``SafeLifeVectorEnv`` -- B device-resident single-agent SafeLife environments stepped by ONE
kernel launch per step (fast tier of SURVEY.md section 7).

Per env it reproduces the reference's ``SafeLifeEnv`` (safelife_env.py:148-218): execute the
action, advance board (and non-static goals) under the level's own PCG64 stream, recolour exits,
score, reward = float32(delta points) while active, done = agent gone or time limit, episode
accounting, observation.  With ``auto_reset`` an env whose episode ends immediately loads its
next level from the device-resident pool (the reference's driver calls ``env.reset()`` right
after a done step, training/base_algo.py:231-236) and the returned observation is the new
episode's first one; ``reward``/``done``/``info`` still describe the step that ended.

All state lives in HBM in torch tensors (buffer holders); the arithmetic is in
libsafelife_hip.so.  Nothing here falls back to the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import _hip
from .levels import LevelPool, PreparedLevels

_DEFAULT_CHANNELS = tuple(range(16)) + (25, 26, 27)      # safelife_env.py:71


class _EventSet(object):
    def __init__(self, events):
        self.events = list(events)

    def synchronize(self):
        for e in self.events:
            e.synchronize()


class SideEffectBatch(object):
    """What one ``side_effects_flush()`` produced, still on the device.

    ``count``      int32 [1]   episodes that ended since the previous flush (beyond ``capacity``: dropped)
    ``records``    int32 [C,8] ``struct sl_episode_record`` rows (env, level, num_steps, episode_idx, ...)
    ``boards``     the boards as the agents left them, uint16 payload [C,H,W]
    ``counts``     int32 [2,C,H,W,8]: the two ``life_occupancy`` tensors (inaction, action) of side_effects.py:109-110
    ``keys`` / ``life_dist`` / ``type_masks``: the distributions of :111-130 (include/safelife_hip.h)
    """

    def __init__(self, env, queue_bufs, out, num_samples, done=None):
        self.env, self.num_samples = env, num_samples
        self.count, self.rec_tensor, self.boards = queue_bufs["count"], queue_bufs["records"], queue_bufs["boards"]
        self.counts, self.keys = out["counts"], out["keys"]
        self.life_dist, self.type_masks = out["life_dist"], out["type_masks"]
        self._keep = out
        self._done = done       # event behind the pass when it ran on the env's side stream (overlap=True)
        # (the host fallback of distributions() reads an entry's starting board from the host pool by physical slot: which
        #  rewrite of every slot this batch belongs to)
        ver = getattr(env.pool, "slot_version", None)
        self._pool_versions = None if ver is None else ver.copy()

    def wait(self):
        """The caller's current stream waits for the pass (a no-op for a pass that ran on that stream); host reads
        through this class do it themselves."""
        if self._done is not None:
            self.env.side_effects_launch()          # (a deferred pass nobody has launched yet)
            self.env.torch.cuda.current_stream().wait_event(self._done)

    def __len__(self):
        """Valid entries (synchronises)."""
        self.wait()
        return min(int(self.count.item()), self.rec_tensor.shape[0])

    def records(self):
        """Host view of the valid records: dict of arrays (env, level, num_steps, episode_idx, spawn_prob,
        episode_reward, episode_length, success, times_up)."""
        n = len(self)
        raw = self.rec_tensor[:n].cpu().numpy()
        flags = raw[:, 7:8].copy().view(np.uint8)
        return dict(env=raw[:, 0], level=raw[:, 1], num_steps=raw[:, 2], episode_idx=raw[:, 3],
                    spawn_prob=raw[:, 4].copy().view(np.float32), episode_reward=raw[:, 5].copy().view(np.float32),
                    episode_length=raw[:, 6], success=flags[:, 0], times_up=flags[:, 1], n_cell_types=flags[:, 2])

    def dropped(self):
        """Episodes that ended while the queue was full (they were not recorded)."""
        self.wait()
        return max(0, int(self.count.item()) - self.rec_tensor.shape[0])

    def distributions(self, i):
        """(inaction, action): ``{cell type: float64 [H,W]}`` of entry i, as side_effects.py:113-130 builds them."""
        self.wait()
        rec = self.rec_tensor[i].cpu().numpy()
        n_types = int(rec[7:8].copy().view(np.uint8)[2])
        if n_types > _hip.SL_SE_MAX_KEYS - 8:
            # more frozen movable / destructible cell types on the starting board than the device-side key slots
            # hold: this entry's distributions are rebuilt on the host from the occupancy tensors (which are complete)
            from . import side_effects as se
            if self._pool_versions is not None and (int(rec[1]) < len(self._pool_versions) and
                                                    self.env.pool.slot_version[int(rec[1])] != self._pool_versions[int(rec[1])]):
                raise RuntimeError("entry %d: pool slot %d has been replaced since this batch was flushed -- its starting "
                                   "board is gone; evaluate a batch before its levels' slots are staged again" % (i, int(rec[1])))
            b0 = np.asarray(self.env.pool.arrays()["pool_board"][int(rec[1])], np.uint16)
            b2 = self.boards[i].cpu().numpy().view(np.uint16)
            found_in, found_act = se.distributions_from_counts(b0, b2, self.counts[:, i].cpu().numpy(), self.num_samples)
            return ({int(k): v for k, v in found_in.items()}, {int(k): v for k, v in found_act.items()})
        keys = self.keys[i].cpu().numpy().view(np.uint16)
        life = self.life_dist[i].cpu().numpy()
        masks = self.type_masks[i].cpu().numpy()
        inaction, action = {}, {}
        for k, key in enumerate(keys):
            if key == 0xFFFF:
                continue
            if k < 8:
                inaction[int(key)], action[int(key)] = life[0, k], life[1, k]
            else:
                inaction[int(key)], action[int(key)] = 1.0 * masks[0, k - 8], 1.0 * masks[1, k - 8]
        return inaction, action

    def scores(self, i, include=None, exclude=None, strkeys=True, weights=None):
        """``side_effect_score`` of entry i (earth-mover distances on the host; safelife_env.py:185-192 for
        ``weights``)."""
        from . import side_effects as se
        inaction, action = self.distributions(i)
        out = se._scores(inaction, action, tuple(self.boards.shape[1:]), include, exclude, strkeys)
        if weights is not None:
            total = np.zeros(2)
            for key, weight in weights.items():
                total += weight * np.array(out.get(key, 0))
            out["total"] = total.tolist()
        return out


class SafeLifeVectorEnv(object):
    """
    Parameters
    ----------
    pool : LevelPool
    num_envs : int                 envs held by THIS process (one process per GPU)
    time_limit, remove_white_goals, view_shape, output_channels : as SafeLifeEnv
        (safelife_env.py:63-73); ``output_channels=None`` yields the raw uint32 view.
    auto_reset : bool
    first_level : int or array     pool index loaded by env e at its first reset
                                   (default ``(env_offset + e) % len(pool)``)
    level_stride : int             an env's next level is ``(level + level_stride) % len(pool)``
    env_offset : int               global index of this process's env 0 (multi-GPU sharding)
    with_obs : bool                False skips observation writes entirely
    slices : int                   >1: the batch is cut into this many contiguous slices, each stepped by its
                                   own launch on its own HIP stream (``slhip_env_step_slices``; the streams are
                                   the env's own, none of them the caller's).  Envs are
                                   independent, so consecutive steps of different slices overlap on the chip
                                   (the load / store phases and the launch boundary of one slice hide under
                                   the compute phase of the other).  That is ``step_async()`` + ``join()``,
                                   what a pipelined driver uses.  ``step()`` keeps the one-stream semantics and
                                   is ONE launch on the caller's stream whatever ``slices`` says (fencing every
                                   step in and out of the slice streams costs four times the step); the env
                                   joins / fences by itself when a caller switches between the two.
    episode_streams : bool         True: every episode gets its own random stream, derived from the level's
                                   generator, the env's global index (``env_offset`` + e) and the env's episode
                                   count -- envs that replay one pool level, and successive replays by one env,
                                   then differ the way the reference's per-game ``SeedSequence.spawn`` children
                                   do (level_iterator.py:218).  False: an episode on pool level l starts from
                                   ``pool.rng[l]`` exactly (replaying recorded traces of the reference).
    policy_layout : None, "uint8" or "float32"
                                   the step / reset kernels also write the observation the way the policy network
                                   takes it -- ``env.policy_tensor`` ``[B, C, view_w, view_h]`` (channel first,
                                   spatial axes swapped: training/models.py:100-103; float32: ppo.py:64) -- so
                                   nothing has to be transposed or cast afterwards; combine with ``with_obs=False``
                                   to skip the (h, w, c) tensor altogether.
    side_effects : dict or None    ``dict(capacity=N, num_samples=1000)``: the step kernels queue every episode that
                                   ends (record + the board as the agent left it, taken before an auto-reset
                                   reloads the slot) and ``side_effects_flush()`` runs the episode-end pass of the
                                   reference's ``side_effect_score`` over the queue on the device -- see
                                   ``SideEffectBatch``.  ``capacity``: episodes held between two flushes; ``keep``
                                   (default 2): output sets cycled -- a batch's tensors are overwritten by the
                                   ``keep``-th flush after it.
    wrappers : dict or None        training-wrapper math of the reference's env_wrappers.py, fused into the
                                   step (stacked as training/env_factory.py:277-283 does); keys, all
                                   optional: ``movement_bonus``, ``movement_bonus_power``,
                                   ``movement_bonus_period``, ``as_penalty`` (MovementBonusWrapper),
                                   ``exit_bonus`` (ExtraExitBonus.bonus), ``penalty_coef``,
                                   ``ignore_reward_cells``, ``baseline`` ("starting-state" or "inaction":
                                   SimpleSideEffectPenalty; the inaction baseline's spawners draw from one
                                   generator per env where the reference has the process-wide one --
                                   ``inaction_seed``: a seed whose SeedSequence children seed them, or
                                   ``inaction_rng``: PCG64 words uint64 [B,4]).  A wrapper is active when its
                                   coefficient is not None.
                                   The wrapped float64 reward is ``env.shaped_reward`` after each step.
                                   (MinPerformanceScheduler = ``LevelPool(min_performance_fraction=...)``.)
    """

    def __init__(self, pool, num_envs, *, time_limit=1000, remove_white_goals=True,
                 view_shape=(15, 15), output_channels=_DEFAULT_CHANNELS, auto_reset=True,
                 first_level=None, level_stride=1, env_offset=0, with_obs=True,
                 points_on_level_exit=1, wrappers=None, slices=1, episode_streams=True, side_effects=None,
                 policy_layout=None):
        import torch
        self.torch = torch
        if not isinstance(pool, LevelPool):
            raise TypeError("pool must be a LevelPool")
        self.pool = pool
        self.device = _hip.device()
        self.num_envs = B = int(num_envs)
        H, W = pool.shape
        E = pool.exit_slots
        self.view_shape = tuple(int(v) for v in view_shape)
        self.output_channels = tuple(output_channels) if output_channels else None
        self.time_limit = int(time_limit)
        self.auto_reset = bool(auto_reset)
        self.single_agent = True
        chans = self.output_channels or ()
        if len(chans) > _hip.SL_MAX_CHANNELS:
            raise ValueError("too many output channels")

        dev = self.device
        self.t = t = {}
        t["board"] = torch.zeros((B, H, W), dtype=torch.int16, device=dev)      # uint16 payload
        t["goals"] = torch.zeros((B, H, W), dtype=torch.int16, device=dev)
        t["exit_locs"] = torch.full((B, E), -1, dtype=torch.int32, device=dev)
        t["rng"] = torch.zeros((B, 4), dtype=torch.int64, device=dev)           # uint64 payload
        t["scalars"] = torch.zeros((B, 16), dtype=torch.int32, device=dev)      # struct sl_env_scalars
        t["out"] = torch.zeros((B, 4), dtype=torch.int32, device=dev)           # struct sl_step_out
        pa = pool.arrays()
        for name in ("pool_board", "pool_goals"):
            t[name] = torch.from_numpy(np.ascontiguousarray(pa[name]).view(np.int16)).to(dev)
        t["pool_exit_locs"] = torch.from_numpy(np.ascontiguousarray(pa["pool_exit_locs"])).to(dev)
        t["pool_rng"] = torch.from_numpy(np.ascontiguousarray(pa["pool_rng"]).view(np.int64)).to(dev)
        t["points_table"] = torch.from_numpy(np.ascontiguousarray(pa["points_table"], dtype=np.int32)).to(dev)
        t["pool_scalars"] = torch.from_numpy(self._level_scalars(pa, slice(None))).to(dev)
        vh, vw = self.view_shape
        if not with_obs:
            self.obs = None
        elif chans:
            self.obs = torch.zeros((B, vh, vw, len(chans)), dtype=torch.uint8, device=dev)
        else:
            self.obs = torch.zeros((B, vh, vw), dtype=torch.int32, device=dev)   # uint32 payload
        self.policy_tensor = None
        if policy_layout is not None:
            if policy_layout not in ("uint8", "float32") or not chans:
                raise ValueError("policy_layout must be 'uint8' or 'float32' and needs output_channels")
            self.policy_tensor = torch.zeros((B, len(chans), vw, vh), device=dev,
                                             dtype=torch.uint8 if policy_layout == "uint8" else torch.float32)
        if first_level is None:
            first_level = (int(env_offset) + np.arange(B)) % len(pool)
        first = np.broadcast_to(np.asarray(first_level, np.int32), (B,)).copy()
        if B and (first.min() < 0 or first.max() >= len(pool)):
            raise ValueError("first_level must lie in 0..len(pool)-1")
        if int(level_stride) < 0:
            raise ValueError("level_stride must be >= 0")
        t["scalars"][:, _hip.SCALAR_COLS["level_idx"]] = torch.from_numpy(first).to(dev)

        s = self.struct = _hip.EnvBatch()
        s.B, s.H, s.W, s.E = B, H, W, E
        s.time_limit, s.exit_points = self.time_limit, int(points_on_level_exit)
        s.n_tables = int(pool.points_table.shape[0])
        s.auto_reset = int(self.auto_reset)
        s.remove_white_goals = int(bool(remove_white_goals))
        s.view_h, s.view_w, s.n_channels = vh, vw, len(chans)
        for i, c in enumerate(chans):
            s.channels[i] = int(c)
        s.L, s.level_stride = pool.n_slots, int(level_stride)
        # a refreshable pool (levels.LevelPool(refreshable=True)): the successor table instead of the stride rule, two
        # copies that pool_commit() alternates
        self._pool_next = None
        self._pool_refresh = None
        if getattr(pool, "refreshable", False):
            nt = torch.from_numpy(pool.next_table(level_stride)).to(dev)
            self._pool_next = [nt, nt.clone()]
            self._pool_refresh = {"which": 0, "staged": None, "fence": None, "stream": None}
        s.spawner_free = int(not pool.has_spawner)
        s.stream_salt = 1 + int(env_offset) if episode_streams else 0
        t["score_lut"] = torch.zeros((s.n_tables, 4096 + 65536), dtype=torch.int8, device=dev)
        if self._pool_next is not None:
            s.pool_next = self._pool_next[0].data_ptr()
        for name in _hip.ENV_STATE_PTRS + _hip.ENV_POOL_PTRS + _hip.ENV_OUT_PTRS:
            if name == "obs":
                s.obs = None if self.obs is None else self.obs.data_ptr()
                s.policy_obs = None if self.policy_tensor is None else self.policy_tensor.data_ptr()
                s.policy_dtype = 0 if policy_layout != "float32" else 1
            else:
                setattr(s, name, t[name].data_ptr())
        # every pool level as an episode starts on it (sl_env_batch.pool_ready: kept by slhip_env_prepare and
        # slhip_pool_write; in-kernel resets copy it instead of scoring and repainting the level).  SAFELIFE_POOL_READY=0
        # leaves it out (A/B runs).
        if os.environ.get("SAFELIFE_POOL_READY", "1") != "0":
            t["pool_ready"] = torch.zeros_like(t["pool_board"])
            s.pool_ready = t["pool_ready"].data_ptr()
        # views of the per-step output records (struct sl_step_out)
        out = t["out"]
        flags = out[:, 1:2].view(torch.uint8)                    # done, success, times_up, pad
        self.reward = out[:, 0].view(torch.float32)
        self.done = flags[:, 0]
        self.info = {"success": flags[:, 1], "times_up": flags[:, 2],
                     "episode_reward": out[:, 2].view(torch.float32), "episode_length": out[:, 3]}
        self.shaped_reward = None
        if wrappers:
            self._setup_wrappers(dict(wrappers))
        self._se = None
        if side_effects:
            self._se = dict(capacity=int(side_effects.get("capacity", 1024)),
                            num_samples=int(side_effects.get("num_samples", 1000)),
                            keep=max(1, int(side_effects.get("keep", 2))))
            self._se["queue"] = self._new_queue()
            s.finished = self._se["queue"][0]
        self._lib = _hip.lib()
        self._sref = C.byref(s)
        # slices: boundaries at multiples of 64 envs (keeps every slice 16-byte aligned for the row kernels)
        n_sl = max(1, min(int(slices), (B + 63) // 64))
        per = -(-B // n_sl)
        per = -(-per // 64) * 64
        bounds = [min(B, i * per) for i in range(n_sl)] + [B]
        self.slices = n_sl
        self.slice_bounds = tuple(bounds)
        self._bounds = (C.c_int32 * (n_sl + 1))(*bounds)
        # Every slice gets a stream of its own from torch's pool, at normal priority.  Measured alternatives:
        # slice 0 on the caller's stream (one stream fewer to fence and join) is as fast until a process group has
        # been initialised -- then the caller's (null) stream and the side stream serialise, 16.5 instead of 8.9 us
        # per C3 step; a high-priority side stream cures that but starves the other slice of 64x64 boards (67-84
        # instead of 36 us per navigation step).
        self._primary = torch.cuda.current_stream(dev)
        self._slice_streams = self._pick_streams(n_sl) if n_sl > 1 else []
        self._async_pending = False      # step_async() left work on the slice streams that the caller has not joined
        self._caller_ahead = True        # the caller's stream holds work the slice streams have not been fenced against
        self._queues = None              # step_queues(): the library's own AQL queues (opened on first use)
        self._queues_pending = False     # steps dispatched there since the last queues_sync()
        self.steps_dispatched = 0        # steps handed to the device so far, whatever the launcher
        self._slice_steps, self._slice_steps_min = [0] * max(1, n_sl), 0      # (step_slice(): per slice)
        self._queue_refs = []            # action tensors of those steps (kept alive until the sync)
        self._rf_recover, self._rf_ckpt, self._queue_log, self._queue_open_args = False, None, [], None
        self._stream_ptrs = (C.c_void_p * max(1, n_sl))(*[st.cuda_stream for st in self._slice_streams])
        self._primary_ptr = C.c_void_p(self._primary.cuda_stream)
        rc = self._lib.slhip_env_prepare(self._sref, _hip.current_stream_ptr())
        if rc == _hip.SL_E_UNSUPPORTED:
            s.score_lut = None          # points outside int8: every shape runs the size-generic kernels
        else:
            _hip.check(rc)
        # The goal-word cache (include/safelife_hip.h, sl_env_batch.goal_cache): plain batches -- no observation, no
        # wrappers, no finished-episode queue -- whose boards have static goals step without moving the goal array.
        # SAFELIFE_GOAL_CACHE=0 leaves it out (A/B runs).
        self.goal_cache_group = 0           # envs per block of the cache (one block per workgroup of the batch), 0 = none
        if s.score_lut and os.environ.get("SAFELIFE_GOAL_CACHE", "1") != "0":     # (the library says which batches keep one)
            group = C.c_int(0)
            total = int(self._lib.slhip_goal_cache_bytes(self._sref, C.byref(group)))
            if total > 0:
                t["goal_cache"] = torch.zeros(total // 4, dtype=torch.int32, device=dev)
                s.goal_cache = t["goal_cache"].data_ptr()
                self.goal_cache_group = int(group.value)

    @staticmethod
    def _level_scalars(pa, sel):
        """``struct sl_level_scalars`` rows (int32 [n,8]) of the pool slots `sel` from the pool's host arrays."""
        n = len(pa["pool_table_idx"][sel])
        lv = np.zeros((n, 8), np.int32)
        lv[:, 0:2] = pa["pool_agent_loc"][sel]
        lv[:, 2] = pa["pool_required_reset"][sel]
        lv[:, 3] = pa["pool_required_step"][sel]
        lv[:, 4] = pa["pool_initial_points"][sel]
        lv[:, 5] = pa["pool_table_idx"][sel]
        lv[:, 6] = pa["pool_spawn_prob"][sel].astype(np.float32).view(np.int32)
        return lv

    # ---- level-pool refresh while the envs step (levels.LevelPool(refreshable=True)) -------------------------------
    # The reference hands every reset a NEW level (level_iterator.py:200-223, safelife_env.py:203-218); a device-resident
    # pool that never changes makes 8192 envs cycle the same few dozen levels for ever.  pool_stage() takes new levels
    # for some of the pool's logical slots at any time: they are written into those slots' SPARE physical slots (host
    # arrays at once, device copies from pinned memory on a side stream) together with the successor table that names
    # them -- nothing an env can load refers to a spare slot, so the copies need no ordering against the steps.
    # pool_commit() makes them current: it waits for the copies' event (long past, normally) and hands the new table to
    # every step launched FROM THEN ON -- a kernel argument, patched per dispatch by the queue launcher -- so the switch
    # falls between two steps of the stepping thread's program order, deterministically, with no queue drain.

    def pool_stage(self, slots, levels, background=False):
        """New levels for the logical pool slots `slots`, staged (see above).  One staging at a time.  The spare slots it
        writes may still be read by resets of steps enqueued BEFORE the last ``pool_commit()``; it waits for those steps
        (a marker taken at that commit) -- so call it behind the step call that follows a commit, not right behind the
        commit: ``pool_commit(); step_queues_many(...); pool_stage(...)`` keeps the device busy meanwhile.
        ``background``: the work (the levels' cell counts through the HIP kernel, the host arrays, the copies) is done by
        a helper thread of the env, so that the stepping thread -- whose enqueueing is what bounds queue stepping -- only
        pays for handing it over; ``pool_commit()`` waits for it and re-raises what it raised.  Until then the caller
        must not touch ``self.pool``."""
        rf = self._pool_refresh
        if rf is None:
            raise ValueError("the env's pool was not built with refreshable=True")
        if rf["staged"] is not None:
            raise ValueError("pool_stage(): the previous staging has not been committed")
        slots = [int(l) for l in slots]
        if not isinstance(levels, PreparedLevels):
            levels = list(levels)
        w = self.struct.wrap
        if self._se is not None or (w.flags & _hip.WRAP_SIDE_EFFECT and not (w.flags & _hip.WRAP_INACTION)):
            # these read their level's slot long after the reset that loaded it (the episode-end pass takes the
            # starting board from it, the starting-state baseline its rows): a slot must outlive the episodes on it
            last = rf.setdefault("last_replaced", {})
            for l in slots:
                if self.steps_dispatched - last.get(int(l), -10 ** 9) < self.time_limit:
                    raise ValueError("pool slot %d was replaced less than one time limit ago: envs may still be playing "
                                     "its previous content, which this env's side-effect machinery reads from the pool"
                                     % int(l))
            if self._se is not None:
                # ... and the episodes that ENDED on the previous content sit in the finished queue until a flush: the
                # episode-end pass takes their starting boards from the pool when it RUNS.  So a flush must have been
                # issued after the last of them ended, and its pass -- deferred, or still running on the side stream --
                # must be through before the slot is written.
                flushed_at = self._se.get("flushed_at", -1)
                for l in slots:
                    if l in last and flushed_at < last[int(l)] + self.time_limit:
                        raise ValueError("pool slot %d: episodes that ended on its previous content may still be queued for the "
                                         "episode-end pass, which reads their starting boards from the pool -- call "
                                         "side_effects_flush() first" % int(l))
                self.side_effects_launch()
                ev = self._se.get("pass_done")
                if ev is not None:
                    ev.synchronize()
        if background:
            pool = rf.get("helper")
            if pool is None:
                from concurrent.futures import ThreadPoolExecutor
                pool = rf["helper"] = ThreadPoolExecutor(1, thread_name_prefix="safelife-pool-stage")
            rf["staged"] = pool.submit(self._pool_stage_now, slots, levels)
            return None
        rf["staged"] = self._pool_stage_now(slots, levels)
        return rf["staged"][4]

    def _pool_stage_now(self, slots, levels):
        rf, torch, dev, w = self._pool_refresh, self.torch, self.device, self.struct.wrap
        torch.cuda.set_device(dev)              # (the helper thread starts on device 0)
        # the spare slots about to be overwritten were current until the commit before last at the latest; every step
        # dispatched before the last commit has long completed -- make sure (a marker taken at that commit)
        fence = rf["fence"]
        if fence is not None:
            kind, what = fence
            if kind == "queues":
                self.queues_wait(what)
            else:
                what.synchronize()
            rf["fence"] = None
        phys = self.pool.replace(slots, levels)
        try:
            return self._pool_write_staged(slots, phys)
        except Exception:
            self.pool.undo_replace()        # (host and device pools stay the same: the copy did not happen)
            raise

    def _pool_write_staged(self, slots, phys):
        rf, torch, dev, w = self._pool_refresh, self.torch, self.device, self.struct.wrap
        pa = self.pool.arrays()
        sel = np.asarray(phys, np.int64)
        n = len(sel)
        side = rf["stream"]
        if side is None:
            side = rf["stream"] = torch.cuda.Stream(device=dev)
        # ONE pinned staging buffer, allocated once (pinning host memory costs milliseconds and stalls the device's
        # queues: a fresh set per staging made a refreshed run ten times slower), with a section per pool array sized for
        # every level of the pool at once; slhip_pool_write's single kernel reads it where it lies and scatters the rows
        # (and the successor table) into the pool.  One staging at a time, and a staging's kernel has completed when it
        # is committed: the buffer is free here.
        pin = rf.get("pinned")
        if pin is None:
            Lg, (H, W), E = len(self.pool), self.pool.shape, self.pool.exit_slots
            spec = [("board", (Lg, H, W), np.int16), ("goals", (Lg, H, W), np.int16), ("exits", (Lg, E), np.int32),
                    ("rng", (Lg, 4), np.int64), ("scalars", (Lg, 8), np.int32), ("slot", (Lg,), np.int32),
                    ("table", (self.pool.n_slots,), np.int32)]
            off, at = 0, {}
            for name, shape, ndt in spec:
                nbytes = int(np.prod(shape)) * np.dtype(ndt).itemsize
                at[name] = (off, nbytes, shape, ndt)
                off += -(-nbytes // 16) * 16
            host = torch.empty(off, dtype=torch.uint8).pin_memory()
            pin = rf["pinned"] = dict(host=host, h={}, ptr={})
            for name, (o, nbytes, shape, ndt) in at.items():
                pin["h"][name] = host.numpy()[o:o + nbytes].view(ndt).reshape(shape)
                pin["ptr"][name] = host.data_ptr() + o
        h, ptr = pin["h"], pin["ptr"]
        keep = []
        nxt = 1 - rf["which"]
        h["slot"][:n] = sel
        h["board"][:n] = pa["pool_board"][sel].view(np.int16)
        h["goals"][:n] = pa["pool_goals"][sel].view(np.int16)
        h["exits"][:n] = pa["pool_exit_locs"][sel]
        h["rng"][:n] = pa["pool_rng"][sel].view(np.int64)
        h["scalars"][:n] = self._level_scalars(pa, sel)
        h["table"][:] = self.pool.next_table(self.struct.level_stride)
        rows = _hip.PoolRows(n=n, slot=ptr["slot"], board=ptr["board"], goals=ptr["goals"], exit_locs=ptr["exits"],
                             rng=ptr["rng"], scalars=ptr["scalars"], next=ptr["table"],
                             next_dst=self._pool_next[nxt].data_ptr())
        _hip.check(self._lib.slhip_pool_write(self._sref, C.byref(rows), C.c_void_p(side.cuda_stream)))
        if w.flags & _hip.WRAP_SIDE_EFFECT and not (w.flags & _hip.WRAP_INACTION):
            _hip.check(self._lib.slhip_pool_baseline(self._sref, C.c_void_p(side.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(side)
        return (ev, nxt, keep, slots, phys)

    def pool_commit(self, wait=True):
        """Make the staged levels current for every step dispatched from now on.  Returns True when it did (or there was
        nothing staged).  ``wait=False``: if the staging is still on its way (the helper thread, the copy kernel), leave
        it staged and return False instead of waiting for it -- for a stepping loop that does not care at WHICH call the
        new levels arrive and tries again at the next one."""
        rf = self._pool_refresh
        if rf is None or rf["staged"] is None:
            return True
        staged = rf["staged"]
        if hasattr(staged, "result"):
            if not wait and not staged.done():
                return False
            staged = rf["staged"] = staged.result()     # (a background staging: re-raises what the helper raised)
        ev, nxt, keep, slots, _ = staged
        if not wait and not ev.query():
            return False
        ev.synchronize()
        rf["staged"] = None
        rf["which"] = nxt
        self.struct.pool_next = self._pool_next[nxt].data_ptr()
        for l in slots:
            rf.setdefault("last_replaced", {})[l] = self.steps_dispatched
        # what was current until now may be overwritten by the staging after next: not before every step dispatched so
        # far is through
        if self._queues is not None and self._queues_pending:
            rf["fence"] = ("queues", self.queues_marker())
        else:
            evs = []
            for st in list(self._slice_streams) + [self.torch.cuda.current_stream()]:
                e = self.torch.cuda.Event()
                e.record(st)
                evs.append(e)
            rf["fence"] = ("events", _EventSet(evs))
        return True

    def goal_cache_flags(self):
        """Per group of ``goal_cache_group`` consecutive envs: 1 where the group currently steps on cached goal words
        (host copy; None without a cache)."""
        if not self.goal_cache_group:
            return None
        self._settle()
        n = -(-self.num_envs // self.goal_cache_group)
        return self.t["goal_cache"].view(n, -1)[:, 0].cpu().numpy()

    def goal_cache_invalidate(self):
        """Call after writing ``goals``, or the ``goals_static`` / ``level_idx`` columns of the scalars, from outside the
        library (the kernels keep the cache themselves across steps, resets and slices)."""
        if self.goal_cache_group:
            self._settle()
            self.t["goal_cache"].zero_()
            self._caller_ahead = True

    def _setup_wrappers(self, cfg):
        torch, t, s, dev = self.torch, self.t, self.struct, self.device
        known = {"movement_bonus", "movement_bonus_power", "movement_bonus_period", "as_penalty", "exit_bonus",
                 "penalty_coef", "ignore_reward_cells", "baseline", "inaction_seed", "inaction_rng"}
        if set(cfg) - known:
            raise ValueError("unknown wrapper option(s): %s" % sorted(set(cfg) - known))
        B, (H, W) = self.num_envs, self.pool.shape
        bonus, period = cfg.get("movement_bonus"), int(cfg.get("movement_bonus_period", 4))
        power = cfg.get("movement_bonus_power", 1e-100)
        if not 1 <= period <= _hip.WRAP_MAX_PERIOD:
            raise ValueError("movement_bonus_period must be in 1..%d" % _hip.WRAP_MAX_PERIOD)
        w = s.wrap
        w.flags = ((_hip.WRAP_MOVEMENT if bonus is not None else 0)
                   | (_hip.WRAP_AS_PENALTY if cfg.get("as_penalty", True) else 0)
                   | (_hip.WRAP_EXIT_BONUS if cfg.get("exit_bonus") is not None else 0)
                   | (_hip.WRAP_SIDE_EFFECT if cfg.get("penalty_coef") is not None else 0)
                   | (_hip.WRAP_IGNORE_REWARD_CELLS if cfg.get("ignore_reward_cells") else 0))
        baseline = cfg.get("baseline", "starting-state")
        if baseline not in ("starting-state", "inaction"):
            raise ValueError('baseline must be "starting-state" or "inaction"')
        if baseline == "inaction" and cfg.get("penalty_coef") is not None:
            w.flags |= _hip.WRAP_INACTION
        if not w.flags & (_hip.WRAP_MOVEMENT | _hip.WRAP_EXIT_BONUS | _hip.WRAP_SIDE_EFFECT):
            w.flags = 0
            return
        # movement_bonus * speed**power for every reachable distance, evaluated on the host by numpy with
        # the reference's own expression (env_wrappers.py:83-87): no pow() ever runs on the device
        n_tab = (H + W + period + 2) & ~1
        table = np.zeros(n_tab, np.float64)
        if bonus is not None:
            for d in range(n_tab):
                speed = np.sum((np.array([d]) / period)[:1])
                table[d] = bonus * speed ** power
        t["move_table"] = torch.from_numpy(table).to(dev)
        t["wrap_state"] = torch.zeros((B, 12), dtype=torch.int32, device=dev)         # struct sl_wrap_state
        t["shaped_reward"] = torch.zeros(B, dtype=torch.float64, device=dev)
        t["pool_baseline"] = torch.zeros((self.pool.n_slots, H, (W + 1) // 2), dtype=torch.int32, device=dev)
        w.move_period, w.move_table_len = period, n_tab
        w.move_bonus = float(bonus or 0.0)
        w.exit_bonus = float(cfg.get("exit_bonus") or 0.0)
        w.penalty_coef = float(cfg.get("penalty_coef") or 0.0)
        w.move_table = t["move_table"].data_ptr()
        w.state = t["wrap_state"].data_ptr()
        w.shaped_reward = t["shaped_reward"].data_ptr()
        w.shaped_reward_t = None
        w.pool_baseline = t["pool_baseline"].data_ptr()
        self.shaped_reward = t["shaped_reward"]
        if w.flags & _hip.WRAP_INACTION:
            # env_wrappers.py:179-180: the baseline board advances once per step (inside the step kernel: a third pass
            # of its CA loop).  One generator per env.
            words = cfg.get("inaction_rng")
            if words is None:
                seq = cfg.get("inaction_seed")
                seq = seq if isinstance(seq, np.random.SeedSequence) else np.random.SeedSequence(seq)
                words = np.zeros((B, 4), np.uint64)
                for e, child in enumerate(seq.spawn(B)):
                    st = np.random.PCG64(child).state["state"]
                    words[e] = [st["state"] >> 64, st["state"] & (2 ** 64 - 1), st["inc"] >> 64, st["inc"] & (2 ** 64 - 1)]
            words = np.ascontiguousarray(words, dtype=np.uint64).reshape(B, 4)
            t["inaction_rng"] = torch.from_numpy(words.view(np.int64).copy()).to(dev)
            t["inaction_board"] = torch.zeros((B, H, W), dtype=torch.int16, device=dev)
            w.inaction_board = t["inaction_board"].data_ptr()
            w.inaction_rng = t["inaction_rng"].data_ptr()
            w.inaction_rows = None

    # ------------------------------------------------------------------ gym-like surface

    def reset(self, mask=None):
        """(Re)load every env -- or those with mask != 0 -- from its pool level; returns obs."""
        m = None
        if mask is not None:
            m = self.torch.as_tensor(mask, device=self.device).to(self.torch.uint8).contiguous()
        self._settle()
        rc = self._lib.slhip_env_reset(self._sref, _hip.ptr(m), _hip.current_stream_ptr())
        _hip.check(rc)
        self._caller_ahead = True
        return self.obs

    def _actions(self, actions, shape):
        torch = self.torch
        a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions))
        if a.device != self.device or a.dtype != torch.int32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        if tuple(a.shape) != shape:
            raise ValueError("actions must have shape %r" % (shape,))
        return a

    def step(self, actions):
        """actions: int [B] in 0..8.  Returns (obs, reward, done, info) as device tensors that are
        overwritten by the next call.  Ordered on the caller's current stream, sliced or not."""
        a = self._actions(actions, (self.num_envs,))
        # one-stream semantics leave nothing to overlap (every step would be fenced in and joined out: measured 51 us
        # per step with two slices against 12 with one launch), so this is ONE launch on the caller's stream whatever
        # `slices` says; the slices are for step_async()
        self._settle()
        rc = self._lib.slhip_env_step(self._sref, _hip.ptr(a), _hip.current_stream_ptr())
        _hip.check(rc)
        self._caller_ahead = True
        self.steps_dispatched += 1
        return self.obs, self.reward, self.done, self.info

    # ---- sliced stepping (slices > 1): no implicit ordering against the caller's stream

    def _pick_streams(self, n):
        """n streams whose kernels overlap on the device.  HIP multiplexes streams onto a few hardware queues and
        two streams of torch's pool may share one -- which streams collide depends on what else the process has
        created (RCCL's streams, other envs), and a collision silently serialises the slices (measured: 24 instead
        of 13.5 us per step).  So every candidate is probed against the streams already chosen
        (``slhip_streams_concurrent``: two idle 100 us kernels); a few hundred microseconds per probe, once."""
        torch, dev = self.torch, self.device
        chosen, spare = [], []
        for _ in range(4 * n + 4):
            cand = torch.cuda.Stream(device=dev)
            ok = C.c_int(1)
            for st in chosen:
                _hip.check(self._lib.slhip_streams_concurrent(st.cuda_stream, cand.cuda_stream, C.byref(ok)))
                if not ok.value:
                    break
            if ok.value:
                chosen.append(cand)
                if len(chosen) == n:
                    return chosen
            else:
                spare.append(cand)
        return chosen + spare[:n - len(chosen)]      # (no concurrent set found: correct, just not overlapped)

    def fence(self):
        """Slice streams wait for everything enqueued so far on the caller's current stream (call after
        producing actions, resetting, or touching env state there)."""
        if self._slice_streams:
            cur = (C.c_void_p * 1)(self.torch.cuda.current_stream().cuda_stream)
            _hip.check(self._lib.slhip_streams_order(cur, 1, self._stream_ptrs, len(self._slice_streams)))
        self._caller_ahead = False

    def join(self):
        """The caller's current stream waits for every slice's enqueued steps (call before consuming
        reward / done / obs / state there)."""
        if self._slice_streams:
            cur = (C.c_void_p * 1)(self.torch.cuda.current_stream().cuda_stream)
            _hip.check(self._lib.slhip_streams_order(self._stream_ptrs, len(self._slice_streams), cur, 1))
        self._async_pending = False

    def _settle(self):
        """Before the env is touched on the caller's stream: join the slices if step_async() left work on them, and
        wait for the steps step_queues() dispatched."""
        if self._async_pending:
            self.join()
        if self._queues_pending:
            # (what follows on the caller's stream -- even a read -- is not ordered against later queue steps: the next
            #  one waits for the streams first; queues_sync() notes that)
            self.queues_sync()

    # ---- sliced stepping on the library's own AQL queues (csrc/sl_aql.hip, slhip_queues_*): what step_async() does
    # ---- with stream slices, without HIP's per-launch host cost -- three to six slices per step become affordable

    _CKPT_NAMES = ("board", "goals", "rng", "scalars", "exit_locs", "out", "wrap_state", "inaction_board", "inaction_rng",
                   "shaped_reward")

    def queues_open(self, slices=None, release_free=None, queue_ids=None, recover=True):
        """Open one AQL queue per slice (default: SAFELIFE_QUEUE_SLICES or 4).  Raises SafeLifeHipError when the
        batch or the runtime does not support it -- callers keep to step_async() then.

        ``queue_ids``: which of the library's queues the slices go onto (default: slice i on queue i) -- a driver whose
        own kernels would hold one of them up leaves that one out (``sharding.RewardGather.free_queues``).

        ``release_free`` (default: True only if SAFELIFE_QUEUE_FENCES=none): OPT-IN to steps without a release fence
        (include/safelife_hip.h, SL_QUEUES_RELEASE_FREE) -- ~0.9 us faster per C3 step, valid only while a workgroup
        index keeps its XCD; the library probes that when the queues are opened (``queue_release_free`` tells whether
        it was granted, ``queue_mode_note`` why not) and every step verifies it.  The default keeps a stream's fences.

        ``recover`` (release-free stepping only): the env keeps a device-side copy of its state as of the last
        successful ``queues_sync()`` (one asynchronous copy per sync: 21 MB at C3) and a log of the step calls since.
        If a step then finds itself on the wrong XCD -- the placement is a property of the process's set of hardware
        queues: a stream or an RCCL communicator created after the queues were opened can change it -- the sync does
        not fail: it puts the copy back, reopens the queues with a stream's fences, replays the logged calls (each
        with the record destination and the pool's successor table it was made with) and warns; the run continues bit
        for bit as if the steps had carried fences all along -- PROVIDED the action buffers of the logged calls are
        unchanged (keep them until the next ``queues_sync()``).  ``recover=False``: the sync raises instead and the
        state since the previous sync is not valid."""
        if self._queues is not None:
            return
        self._rf_recover, self._rf_ckpt, self._queue_log = False, None, []
        B = self.num_envs
        n = int(slices if slices is not None else (len(queue_ids) if queue_ids is not None else
                                                   os.environ.get("SAFELIFE_QUEUE_SLICES", "4")))
        n = max(1, min(n, 8, (B + 63) // 64))
        if queue_ids is not None and len(queue_ids) < n:
            raise ValueError("queue_ids: one queue per slice")
        per = -(-(-(-B // n)) // 64) * 64
        bounds = [min(B, i * per) for i in range(n)] + [B]
        if release_free is None:
            release_free = os.environ.get("SAFELIFE_QUEUE_FENCES", "agent") == "none"
        handle = C.c_void_p()
        ids = (C.c_int32 * n)(*[int(q) for q in queue_ids[:n]]) if queue_ids is not None else None
        _hip.check(self._lib.slhip_queues_open_on(self._sref, n, (C.c_int32 * (n + 1))(*bounds), ids,
                                                  _hip.QUEUES_RELEASE_FREE if release_free else 0, C.byref(handle)))
        self.queue_ids = list(queue_ids[:n]) if queue_ids is not None else list(range(n))
        why = C.c_char_p()
        mode = self._lib.slhip_queues_mode(handle, C.byref(why))
        self._queues, self.queue_slices = handle, n
        self.queue_release_free = bool(mode & _hip.QUEUES_RELEASE_FREE)
        self.queue_mode_note = why.value.decode() if why.value else None
        if release_free and not self.queue_release_free:
            import warnings
            warnings.warn("safelife_amd: release-free queue stepping was asked for and not granted (%s); stepping with "
                          "agent-scope fences" % self.queue_mode_note, RuntimeWarning, stacklevel=2)
        self._queue_open_args = dict(slices=n, queue_ids=queue_ids)
        if self.queue_release_free and recover:
            self._rf_recover = True
            self._settle()
            self._rf_checkpoint()

    def _rf_checkpoint(self):
        """Device-side copy of everything a step changes, on the caller's stream (release-free stepping with recovery:
        taken when the queues are opened and after every successful sync)."""
        names = [k for k in self._CKPT_NAMES if k in self.t]
        if self._rf_ckpt is None:
            self._rf_ckpt = {k: self.torch.empty_like(self.t[k]) for k in names}
        for k in names:
            self._rf_ckpt[k].copy_(self.t[k], non_blocking=True)
        if self._se is not None:
            self._rf_ckpt["se_count"] = self._se["queue"][1]["count"].clone()
        # (the copies READ what the next queue steps write: those wait for this event whatever the caller vouches for)
        self._rf_ckpt_event = self.torch.cuda.Event()
        self._rf_ckpt_event.record()
        self._queue_log = []
        self._caller_ahead = True

    def _rf_recover_now(self, message):
        """The placement check fired: back to the last good state, stream fences from here on, the logged calls again."""
        import warnings
        log, args = self._queue_log, self._queue_open_args
        self._drain_staging()
        self._lib.slhip_queues_close(self._queues)
        self._queues, self._queues_pending = None, False
        self.torch.cuda.synchronize(self.device)
        if self._pool_refresh is not None:
            self._pool_refresh["fence"] = None      # (a ticket of the closed handle; the device is idle)
        for k, v in self._rf_ckpt.items():
            if k == "se_count":
                self._se["queue"][1]["count"].copy_(v)
            else:
                self.t[k].copy_(v)
        if self.goal_cache_group:
            self.t["goal_cache"].zero_()
        self.torch.cuda.synchronize(self.device)
        self._caller_ahead = True
        self._rf_recover, self._rf_ckpt, self._rf_ckpt_event = False, None, None
        warnings.warn("safelife_amd: release-free queue stepping was refused by its placement check (%s); restored the "
                      "state of the last sync, reopened the queues with agent-scope fences and replayed %d step call(s)"
                      % (message.split(";")[0][:200], len(log)), RuntimeWarning, stacklevel=3)
        self.queues_open(args["slices"], release_free=False, queue_ids=args["queue_ids"])
        self.steps_dispatched -= sum(c[2] for c in log)
        keep_out, keep_next = self.struct.out, self.struct.pool_next
        for actions, ptr, n_steps, stride, out_stride, out_ptr, pool_next in log:
            # (as the call was made: its records' destination and the successor table a pool_commit() has since replaced)
            self.struct.out, self.struct.pool_next = out_ptr, pool_next
            self.step_queues_many(actions if actions is not None else ptr, n_steps, stride, out_stride)
        self.struct.out, self.struct.pool_next = keep_out, keep_next
        self._queues_pending = False
        try:
            _hip.check(self._lib.slhip_queues_sync(self._queues))
        finally:
            self._queue_refs, self._queue_log = [], []
            self._caller_ahead = True

    def _queue_head(self, assume_ordered):
        """1 if the next queue step must take a system-scope acquire behind a device synchronize: HIP streams have
        touched the envs (reset, step(), rollout() ...) or may hold work on the actions / outputs (anything after a
        queues_sync()) since the queues last ran.  ``assume_ordered="untouched"``: the caller vouches that NOTHING
        outside the queues has written the envs' state, the actions of the coming steps or their outputs since the last
        ``queues_sync()`` (it only waited, or read) -- honoured only if this object itself has not put anything on a
        stream since (a reset, a step(), a checkpoint copy): the first step then is a step like any other, without the
        system-scope acquire (worth ~8 us: it drops every XCD's L2)."""
        ev = getattr(self, "_rf_ckpt_event", None)
        if ev is not None:          # the recovery copy of the last sync is still reading the state
            ev.synchronize()
            self._rf_ckpt_event = None
        if not (self._caller_ahead or self._async_pending):
            return 0
        if assume_ordered == "untouched" and self._caller_ahead == "sync" and not self._async_pending:
            self._caller_ahead = False
            return 0
        if self._async_pending:
            self.join()
        if not assume_ordered:
            self.torch.cuda.synchronize(self.device)
        self._caller_ahead = False
        return 1

    def step_queues(self, actions, assume_ordered=False):
        """One step per env, one dispatch per slice on the slice's own AQL queue.  `actions` as for step_async(),
        complete when the call is made; the env keeps a reference to the tensor until the next ``queues_sync()`` (an
        address passed as int must stay valid that long).  Outputs and state may be read (by the host or by any stream)
        only after ``queues_sync()``; every method of this class that touches the env does that itself.
        ``assume_ordered``: the caller vouches that no HIP stream holds unfinished work on the envs, the actions or
        the outputs (e.g. it has just synchronised the device itself): the first step after stream work then skips
        the device synchronize it would otherwise make."""
        self.step_queues_many(actions, 1, assume_ordered=assume_ordered)

    def step_queues_many(self, actions, n_steps=None, action_stride=None, out_stride=0, assume_ordered=False, defer=False):
        """``n_steps`` consecutive steps, ALL enqueued by this one call (``slhip_queues_steps``: the packets and
        argument blocks of every step and slice are written on the C side, the device starts on the first step while
        the rest is being written; the call blocks only while the queue rings are full).  `actions`: int32 device
        tensor [T, B] (or [B] with n_steps=1), complete when the call is made -- or its address with ``n_steps`` and
        ``action_stride`` (int32 elements between consecutive steps).  ``out_stride``: sl_step_out records between
        the outputs of consecutive steps (0: every step overwrites the env's own record tensor; sharding.RewardGather
        points it at a window).  Keeps a reference to `actions` until the next ``queues_sync()``.
        ``defer=True`` (``slhip_queues_stage``, at most ``_hip.QUEUES_STAGE_MAX`` steps): the steps are written --
        argument blocks, packets -- but not handed to the device until ``queues_go()``; the action buffers must exist
        now, their contents only then.  What a hipGraph's instantiate / launch split does for a stream."""
        if isinstance(actions, int):
            ptr = actions
            if n_steps is None:
                raise ValueError("n_steps is needed with an address")
            stride = int(action_stride if action_stride is not None else self.num_envs)
        else:
            t = actions
            if n_steps is None:
                n_steps = 1 if t.dim() == 1 else int(t.shape[0])
            if (t.dtype != self.torch.int32 or not t.is_contiguous() or t.device != self.device
                    or t.numel() < n_steps * self.num_envs or (t.dim() > 1 and t.shape[-1] != self.num_envs)):
                raise ValueError("step_queues() takes a contiguous int32 tensor [num_envs] (or [T, num_envs]) on the env's "
                                 "device (or its address); got %s %s on %s" % (t.dtype, tuple(t.shape), t.device))
            ptr = t.data_ptr()
            stride = int(action_stride if action_stride is not None else self.num_envs)
            self._queue_refs.append(t)
        if self._queues is None:
            self.queues_open()
        head = self._queue_head(assume_ordered)
        fn = self._lib.slhip_queues_stage if defer else self._lib.slhip_queues_steps
        rc = fn(self._queues, self._sref, ptr, stride, int(out_stride), int(n_steps), head)
        if rc:
            _hip.check(rc)
        self._queues_pending = True
        self.steps_dispatched += int(n_steps)
        if self._rf_recover:
            self._queue_log.append((None if isinstance(actions, int) else actions, ptr, int(n_steps), stride, int(out_stride),
                                    self.struct.out, self.struct.pool_next))

    def queues_go(self):
        """Hand the steps staged by ``step_queues_many(..., defer=True)`` to the device (one doorbell per queue)."""
        if self._queues is not None:
            _hip.check(self._lib.slhip_queues_go(self._queues))

    def queues_marker(self):
        """A system-scope release behind every queue step dispatched so far; returns a ticket at once (-1: nothing was
        outstanding).  ``queues_wait(ticket)`` waits for it.  The pair is ``queues_sync()`` split in two, for callers
        that have something else to do in between (or another thread to do the waiting: sharding.RewardGather)."""
        ticket = C.c_longlong(-1)
        if self._queues is not None:
            _hip.check(self._lib.slhip_queues_marker(self._queues, C.byref(ticket)))
        return ticket.value

    def queues_wait(self, ticket):
        """Wait for a ``queues_marker()``.  With release-free stepping and ``recover=True`` a placement failure found here
        is recovered from as in ``queues_sync()`` (which this then amounts to); on success the call log is KEPT -- steps
        behind the marker may still be in flight -- so action buffers handed to ``step_queues*`` must stay unchanged until
        the next ``queues_sync()`` while ``recover=True`` (the replay reads them again)."""
        if self._queues is not None:
            try:
                _hip.check(self._lib.slhip_queues_wait(self._queues, int(ticket)))
            except _hip.SafeLifeHipError as e:
                if not (self._rf_recover and "another XCD" in str(e)):
                    raise
                self._queues_pending = False
                self._rf_recover_now(str(e))
                self._queue_refs = []
                self._caller_ahead = True

    def queues_sync(self):
        """Wait for every step dispatched on the queues so far (system-scope release behind them): afterwards their
        outputs and the envs' state are visible to the host and to every HIP stream.  Whatever the caller then does on
        HIP streams is not ordered against LATER queue steps: the next one synchronises the device first (see
        ``assume_ordered``)."""
        self._queues_pending = False
        if self._queues is not None:
            try:
                _hip.check(self._lib.slhip_queues_sync(self._queues))
            except _hip.SafeLifeHipError as e:
                if not (self._rf_recover and "another XCD" in str(e)):
                    raise
                self._rf_recover_now(str(e))
                return
            finally:
                self._queue_refs = []
                # ("sync": nothing but this wait has happened since the queues last ran -- _queue_head)
                self._caller_ahead = self._caller_ahead or "sync"
            if self._rf_recover:
                self._rf_checkpoint()

    def _drain_staging(self):
        """A background pool staging waits on a marker of the CURRENT queue handle: let it finish before the handle goes
        (its result, or what it raised, stays with pool_commit())."""
        rf = self._pool_refresh
        if rf is not None and hasattr(rf.get("staged"), "exception"):
            rf["staged"].exception()

    def queues_close(self):
        if self._queues is not None:
            self._drain_staging()
            rf = self._pool_refresh
            if rf is not None and rf.get("fence") is not None and rf["fence"][0] == "queues":
                self._lib.slhip_queues_wait(self._queues, int(rf["fence"][1]))       # (a ticket of this handle: settle it now)
                rf["fence"] = None
            self._lib.slhip_queues_close(self._queues)
            self._queues, self._queues_pending, self._queue_refs = None, False, []

    def __del__(self):
        try:
            self.queues_close()
        except Exception:
            pass

    def step_async(self, actions):
        """One step per env, one launch per slice on the slice's own stream; nothing is fenced.  `actions`:
        a contiguous int32 device tensor [B] that is already complete (or ordered by ``fence()``), or its
        device address as an int.  Outputs are valid on the caller's stream after ``join()``."""
        if isinstance(actions, int):
            ptr = actions
        else:           # (a tensor: the checks step() makes through _actions(), without its conversions)
            if (actions.dtype != self.torch.int32 or not actions.is_contiguous() or actions.numel() != self.num_envs
                    or actions.device != self.device):
                raise ValueError("step_async() takes a contiguous int32 tensor [num_envs] on the env's device "
                                 "(or its address); got %s %s on %s" % (actions.dtype, tuple(actions.shape), actions.device))
            ptr = actions.data_ptr()
        if self._queues_pending:        # steps still running on the AQL queues: their writes are not visible to streams yet
            self._settle()
        if self.slices > 1:
            if self._caller_ahead:      # a reset / step() / rollout() on the caller's stream since the last fence
                self.fence()
            rc = self._lib.slhip_env_step_slices(self._sref, self.slices, self._bounds, ptr, self._stream_ptrs)
            self._async_pending = True
        else:
            rc = self._lib.slhip_env_step(self._sref, ptr, _hip.current_stream_ptr())
        if rc:
            _hip.check(rc)
        self.steps_dispatched += 1

    def step_slice(self, i, actions):
        """One step of slice i only (envs ``slice_bounds[i] .. slice_bounds[i+1]``), one launch on the slice's own
        stream (``slice_stream(i)``); `actions`: the int32 device tensor [num_envs] of the whole batch (or its address) --
        only the slice's entries are read.  No fence: whoever writes the slice's actions and reads its outputs does so
        on the same stream (runner.PipelinedRunner), or orders itself against it."""
        if not 0 <= i < self.slices or self.slices < 2:
            raise ValueError("no such slice (construct the env with slices >= 2)")
        ptr = actions if isinstance(actions, int) else actions.data_ptr()
        if self._queues_pending:
            self._settle()
        if self._caller_ahead:
            self.fence()
        lo, hi = self.slice_bounds[i], self.slice_bounds[i + 1]
        rc = self._lib.slhip_env_step_range(self._sref, lo, hi - lo, ptr, self._stream_ptrs[i])
        self._async_pending = True
        if rc:
            _hip.check(rc)
        # (steps_dispatched counts whole steps of the batch: the slowest slice's)
        c = self._slice_steps
        c[i] += 1
        m = min(c)
        if m > self._slice_steps_min:
            self.steps_dispatched += m - self._slice_steps_min
            self._slice_steps_min = m

    def slice_stream(self, i):
        """The torch stream slice i is stepped on."""
        return self._slice_streams[i]

    def rollout(self, actions, reward_out=None, done_out=None):
        """T steps in one launch.  actions: int [T,B].  Returns (reward[T,B], done[T,B])."""
        torch = self.torch
        T = int(actions.shape[0])
        a = self._actions(actions, (T, self.num_envs))
        if reward_out is None:
            reward_out = torch.empty((T, self.num_envs), dtype=torch.float32, device=self.device)
        if done_out is None:
            done_out = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device)
        if self.shaped_reward is not None:      # wrapped reward of every step: env.shaped_reward_t [T,B]
            self.shaped_reward_t = torch.empty((T, self.num_envs), dtype=torch.float64, device=self.device)
            self.struct.wrap.shaped_reward_t = self.shaped_reward_t.data_ptr()
        self._settle()
        rc = self._lib.slhip_env_rollout(self._sref, _hip.ptr(a), T, _hip.ptr(reward_out),
                                         _hip.ptr(done_out), _hip.current_stream_ptr())
        self.struct.wrap.shaped_reward_t = None
        _hip.check(rc)
        self._caller_ahead = True
        self.steps_dispatched += T
        return reward_out, done_out

    def set_step_outputs(self, out_ptr, compact=False):
        """Redirect the per-step output records (``sl_step_out[B]``, 16 bytes per env) to caller-owned
        device memory; ``None`` restores the env's own tensor.  Used by sharding.RewardGather to have
        the kernel fill a send buffer directly.  ``compact``: 8-byte records there -- reward, done, success,
        times_up: ``sl_env_batch.out_compact`` -- (the env's own tensor always takes whole records)."""
        self.struct.out = self.t["out"].data_ptr() if out_ptr is None else int(out_ptr)
        self.struct.out_compact = 1 if (compact and out_ptr is not None) else 0


    def get_obs(self):
        self._settle()
        rc = self._lib.slhip_env_obs(self._sref, _hip.current_stream_ptr())
        _hip.check(rc)
        self._caller_ahead = True
        return self.obs

    def policy_obs(self, channels=_DEFAULT_CHANNELS, dtype=None, out=None):
        """The observation as the policy network takes it (training/models.py:100-103): channel-first,
        spatial axes swapped, ``[B, C, view_w, view_h]``, uint8 or float32 -- unpacked on the device from
        the raw uint32 view (needs ``output_channels=None``), so a step writes 4 bytes per view cell and
        the (h, w, c) byte tensor is never materialised."""
        torch = self.torch
        if self.output_channels is not None or self.obs is None:
            raise ValueError("policy_obs() needs the raw view: construct with output_channels=None")
        dtype = dtype or torch.float32
        if dtype not in (torch.uint8, torch.float32):
            raise ValueError("dtype must be torch.uint8 or torch.float32")
        vh, vw = self.view_shape
        chans = np.ascontiguousarray(channels, dtype=np.int32)
        if out is None:
            out = torch.empty((self.num_envs, len(chans), vw, vh), dtype=dtype, device=self.device)
        rc = self._lib.slhip_obs_to_policy(_hip.ptr(self.obs), self.num_envs, vh, vw,
                                           chans.ctypes.data_as(C.c_void_p), len(chans), _hip.ptr(out),
                                           0 if dtype == torch.uint8 else 1, _hip.current_stream_ptr())
        _hip.check(rc)
        return out

    def _new_queue(self):
        torch, cap = self.torch, self._se["capacity"]
        H, W = self.pool.shape
        bufs = dict(count=torch.zeros(1, dtype=torch.int32, device=self.device),
                    records=torch.zeros((cap, 8), dtype=torch.int32, device=self.device),
                    boards=torch.zeros((cap, H, W), dtype=torch.int16, device=self.device))
        q = _hip.EpisodeQueue()
        q.capacity, q.env_base = cap, 0
        q.count, q.records, q.boards = (bufs[k].data_ptr() for k in ("count", "records", "boards"))
        return q, bufs

    def _side_stream(self):
        """A stream for work that runs UNDER the steps (the episode-end pass): one that shares a hardware queue with
        none of the slice streams -- HIP multiplexes streams onto four of them, and a 10 ms kernel on a shared one
        holds that slice's launches up for as long."""
        se = self._se
        if se.get("stream") is None:
            torch = self.torch
            pick = None
            for _ in range(12):
                cand = torch.cuda.Stream(device=self.device)
                ok = C.c_int(1)
                for st in [self._primary] + list(self._slice_streams):
                    _hip.check(self._lib.slhip_streams_concurrent(st.cuda_stream, cand.cuda_stream, C.byref(ok)))
                    if not ok.value:
                        break
                if ok.value:
                    pick = cand
                    break
                se.setdefault("spare_streams", []).append(cand)     # (kept: a released stream's queue slot is handed out again)
            se["stream"] = pick if pick is not None else torch.cuda.Stream(device=self.device)
        return se["stream"]

    def side_effects_flush(self, overlap=False, defer=False):
        """Run the episode-end pass (``slhip_side_effects``) over the episodes queued since the last flush and switch the
        step kernels to the other queue.  Nothing is read back: the returned ``SideEffectBatch`` takes the queue's
        buffers along (a fresh queue replaces it) next to the pass's outputs, all device tensors sized by the capacity,
        plus the device-side entry count; its ``records()`` / ``scores()`` synchronise when (and only when) the host wants
        the numbers.

        ``overlap=False``: the pass runs on the caller's current stream, behind every step enqueued so far, and later
        steps wait for it.  ``overlap=True``: it runs on a side stream of the env, behind the steps enqueued so far, and
        the steps that FOLLOW do not wait for it (nothing they touch is shared with it: the queue it reads has been
        replaced, the level pool is read-only) -- ``side_effects_join()`` / the batch's accessors order against it.
        ``defer=True`` (with overlap): everything but the launch -- ``side_effects_launch()`` makes it, so that the caller
        can put its next steps in front of the pass in the device's queues (it is also made by the next flush / join)."""
        if self._se is None:
            raise ValueError("construct the env with side_effects=dict(capacity=...) first")
        torch, se = self.torch, self._se
        self.side_effects_launch()
        side = None
        if overlap and self.device.type == "cuda":
            side = self._side_stream()
            if self._queues_pending:
                self.queues_sync()                    # (queue steps: their fence is a host wait)
            # the queue was filled by the steps enqueued so far: the side stream waits for them, nobody waits for it
            before = list(self._slice_streams) + [torch.cuda.current_stream()]
            b = (C.c_void_p * len(before))(*[st.cuda_stream for st in before])
            a = (C.c_void_p * 1)(side.cuda_stream)
            _hip.check(self._lib.slhip_streams_order(b, len(before), a, 1))
        else:
            self._settle()                            # the queue was filled on the slice streams
        q, bufs = se["queue"]
        se["queue"] = self._new_queue()               # the step kernels fill a fresh queue from here on
        self.struct.finished = se["queue"][0]
        self._caller_ahead = True                     # (the fresh queue's counter is zeroed on the caller's stream)
        H, W = self.pool.shape
        cap, K = se["capacity"], _hip.SL_SE_MAX_KEYS
        dev = self.device
        # The pass's outputs are gigabytes at C5's size (life_dist alone: capacity x 2 x 8 x H x W float64): they are
        # allocated once per output set and the sets are cycled -- a batch stays valid until `keep` further flushes
        # (allocating them afresh stalled the stepping thread for milliseconds per flush).
        sets = se.setdefault("outs", [])
        if len(sets) < se.get("keep", 2):
            sets.append(dict(work_boards=torch.empty((2 * cap, H, W), dtype=torch.int16, device=dev),
                             work_prob=torch.empty(2 * cap, dtype=torch.float32, device=dev),
                             work_steps=torch.empty(2 * cap, dtype=torch.int32, device=dev),
                             work_rng=torch.empty((2 * cap, 4), dtype=torch.int64, device=dev),
                             counts=torch.empty((2, cap, H, W, 8), dtype=torch.int32, device=dev),
                             keys=torch.empty((cap, K), dtype=torch.int16, device=dev),
                             life_dist=torch.empty((cap, 2, 8, H, W), dtype=torch.float64, device=dev),
                             type_masks=torch.empty((cap, 2, K - 8, H, W), dtype=torch.uint8, device=dev)))
            out = sets[-1]
            if side is not None:                      # (first use: the allocations above happened on the caller's stream)
                side.wait_stream(torch.cuda.current_stream())
        else:
            out = sets[se["flushes"] % len(sets)]
        se["flushes"] = se.get("flushes", 0) + 1
        stream_ptr = _hip.current_stream_ptr() if side is None else C.c_void_p(side.cuda_stream)
        done = torch.cuda.Event() if side is not None else None

        se["flushed_at"] = self.steps_dispatched
        pass_done = torch.cuda.Event() if self.device.type == "cuda" else None
        se["pass_done"] = pass_done         # (recorded behind the pass on whichever stream runs it: pool_stage waits for it)

        def launch():
            rc = self._lib.slhip_side_effects(self._sref, C.byref(q), se["num_samples"], 1,
                                              *[_hip.ptr(out[k]) for k in ("work_boards", "work_prob", "work_steps",
                                                                           "work_rng", "counts", "keys", "life_dist",
                                                                           "type_masks")],
                                              stream_ptr)
            _hip.check(rc)
            if side is not None:
                for tns in bufs.values():             # (the allocator must not hand these to someone else while the pass runs)
                    tns.record_stream(side)
                done.record(side)
            if pass_done is not None:
                pass_done.record(side if side is not None else torch.cuda.current_stream())
        if defer and side is not None:
            se["deferred"] = launch
        else:
            launch()
        if side is not None:
            se["last_done"] = done
        return SideEffectBatch(self, bufs, out, se["num_samples"], done)

    def side_effects_launch(self):
        """Launch the pass a ``side_effects_flush(overlap=True, defer=True)`` left prepared (no-op otherwise)."""
        if self._se is not None and self._se.get("deferred") is not None:
            launch, self._se["deferred"] = self._se["deferred"], None
            launch()

    def side_effects_join(self):
        """The caller's current stream waits for the last overlapped episode-end pass."""
        if self._se is not None and self._se.get("last_done") is not None:
            self.side_effects_launch()
            self.torch.cuda.current_stream().wait_event(self._se["last_done"])

    def side_effect_occupancy(self, env_ids, rng, num_samples=1000):
        """The two ``life_occupancy`` tensors of ``side_effect_score`` (side_effects.py:103-111) for the
        envs ``env_ids`` whose episodes have just ended: the episode's starting board (its pool level)
        rolled forward by the episode's length, and the board as the agent left it, each sampled for
        ``num_samples`` steps -- all on the device, int32 ``[n,H,W,8]`` each.

        Needs ``auto_reset=False`` (the terminal board must still be in place: call this right after the
        step that reported ``done``, then ``reset(mask)``).  ``rng``: int64 ``[n,4]`` PCG64 words, one
        stream per env, advanced in place; the reference draws all of this from its process-wide
        generator (roll-forward, then the inaction tensor, then the action tensor: the order kept here).
        Distances: ``safelife_amd.side_effects.side_effect_score_from_counts``."""
        from . import speedups
        torch = self.torch
        if self.auto_reset:
            raise ValueError("side_effect_occupancy() needs auto_reset=False: the terminal board is reloaded otherwise")
        ids = torch.as_tensor(env_ids, device=self.device, dtype=torch.int64)
        sc = self.t["scalars"][ids]
        level = sc[:, _hip.SCALAR_COLS["level_idx"]].to(torch.int64)
        steps = sc[:, _hip.SCALAR_COLS["num_steps"]].contiguous()
        prob = sc[:, _hip.SCALAR_COLS["spawn_prob"]].contiguous().view(torch.float32)
        b0 = self.t["pool_board"][level].contiguous()
        b2 = self.t["board"][ids].contiguous()
        b1 = speedups.advance_board_batch(b0, prob, rng, steps)
        occ0 = speedups.life_occupancy_batch(b1, prob, rng, num_samples)
        occ1 = speedups.life_occupancy_batch(b2, prob, rng, num_samples)
        return occ0, occ1, b0, b2

    # ------------------------------------------------------------------ host views

    def snapshot(self):
        """Device-side copy of the per-env state (board, goals, generator, per-env record), ordered after the steps
        enqueued so far; read it later with ``numpy(name, snapshot=...)`` -- nothing crosses to the host now."""
        self._settle()
        return {name: self.t[name].clone() for name in ("board", "goals", "rng", "scalars")}

    def numpy(self, name, snapshot=None):
        """Host copy of a state array under the reference's / oracle's name and dtype."""
        if snapshot is not None:
            live, self.t = self.t, dict(self.t, **snapshot)
            try:
                return self.numpy(name)
            finally:
                self.t = live
        self._settle()
        if name == "obs":
            a = self.obs.cpu().numpy()
            return a.view(np.uint32) if self.output_channels is None else a
        if name in ("board", "goals", "pool_board", "pool_goals", "inaction_board"):
            return self.t[name].cpu().numpy().view(np.uint16)
        if name in ("rng", "pool_rng", "inaction_rng"):
            return self.t[name].cpu().numpy().view(np.uint64)
        if name == "agent_loc":
            return self.t["scalars"][:, 0:2].cpu().numpy()
        if name in _hip.SCALAR_COLS:
            col = self.t["scalars"][:, _hip.SCALAR_COLS[name]].cpu().numpy()
            if name in _hip.SCALAR_FLOATS:
                return col.view(np.float32)
            if name in ("goals_static", "is_active", "loaded"):
                return col.astype(np.uint8)
            return col
        if name == "reward":
            return self.reward.cpu().numpy()
        if name == "done":
            return self.done.cpu().numpy()
        if name in self.info:
            return self.info[name].cpu().numpy()
        return self.t[name].cpu().numpy()
