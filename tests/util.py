"""Shared test helpers: golden-trace loading and a backend-neutral replay of SafeLifeEnv traces.

Two backends implement the same tiny adapter (reset / step / get):
  * OracleBackend  -- oracle.OracleEnv  (CPU checker; used by the `not gpu` tests to pin the oracle
                      against the reference's traces)
  * DeviceBackend  -- safelife_amd.SafeLifeVectorEnv (the HIP product path; `gpu` tests)
"""
import glob
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")

PALETTE = np.array(
    [0] * 12 + [9] * 5 + [1, 16, 17, 32788, 152, 152 | 0x200, 152 | 0x800, 144, 48, 53, 85, 32884,
                          272, 9 | 0x200, 9 | 0x400, 9 | 0x600, 9 | 0x800, 9 | 0xE00, 122,
                          0x8000 | 9, 4 | 8, 1 | 4 | 0x400, 32, 64, 128 | 0x400, 9 | 0xA00],
    dtype=np.uint16)


def random_boards(rng, B, H, W, kind=0):
    """Synthetic boards in the spirit of SURVEY.md section 8(d), config C2."""
    if kind == 0:
        return PALETTE[rng.integers(0, len(PALETTE), (B, H, W))]
    if kind == 1:
        b = np.where(rng.random((B, H, W)) < 0.3, 9, 0).astype(np.uint16)
        for k in range(B):
            for _ in range(4):
                b[k, rng.integers(0, H), rng.integers(0, W)] = 152 | (int(rng.integers(0, 8)) << 9)
        return b
    return rng.integers(0, 65536, (B, H, W)).astype(np.uint16)


def random_rng_words(rng, B):
    """B independent numpy PCG64 states as uint64 [B,4]."""
    out = np.zeros((B, 4), np.uint64)
    m = (1 << 64) - 1
    for k, child in enumerate(np.random.SeedSequence(int(rng.integers(0, 2**31))).spawn(B)):
        st = np.random.PCG64(child).state["state"]
        out[k] = [st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m]
    return out


def trace_names():
    """The single-agent SafeLifeEnv traces (the multi-agent ones, trace_multi_*.npz, have their own replay)."""
    return sorted(n for n in (os.path.basename(p)[len("trace_"):-4] for p in glob.glob(os.path.join(GOLDEN, "trace_*.npz")))
                  if not n.startswith("multi_"))


def load_trace(name):
    with np.load(os.path.join(GOLDEN, "trace_%s.npz" % name)) as d:
        return {k: d[k] for k in d.files}


def levels_from_trace(tr):
    from safelife_amd.levels import Level
    levels = []
    for i in range(int(tr["n_levels"])):
        pre = "level%d_" % i
        rec = {k[len(pre):]: tr[k] for k in tr if k.startswith(pre)}
        rng = rec.pop("rng")
        lv = Level.from_data(rec)
        lv.rng_words = np.array(rng, np.uint64)
        levels.append(lv)
    return levels


def env_kwargs_from_trace(tr):
    kw = {}
    if "env_view_shape" in tr:
        kw["view_shape"] = tuple(int(v) for v in tr["env_view_shape"])
    if "env_output_channels" in tr:
        oc = tr["env_output_channels"]
        kw["output_channels"] = None if oc.ndim == 0 else tuple(int(v) for v in oc)
    if "env_time_limit" in tr:
        kw["time_limit"] = int(tr["env_time_limit"])
    if "env_remove_white_goals" in tr:
        kw["remove_white_goals"] = bool(tr["env_remove_white_goals"])
    return kw


def wrappers_from_trace(tr):
    """Keyword arguments of the training-wrapper stack a wrap_* trace was recorded with, or None."""
    if not any(k.startswith("wrap_") for k in tr):
        return None
    w = {}
    if "wrap_movement" in tr:
        bonus, power, period, as_penalty = tr["wrap_movement"]
        w.update(movement_bonus=float(bonus), movement_bonus_power=float(power),
                 movement_bonus_period=int(period), as_penalty=bool(as_penalty))
    if "wrap_exit_bonus" in tr:
        w["exit_bonus"] = float(tr["wrap_exit_bonus"])
    if "wrap_side_effect" in tr:
        coef, ignore = tr["wrap_side_effect"]
        w.update(penalty_coef=float(coef), ignore_reward_cells=bool(ignore))
    if "wrap_inaction_rng" in tr:       # baseline="inaction"; the process-wide generator's state when the run began
        w.update(baseline="inaction", inaction_rng=np.asarray(tr["wrap_inaction_rng"], np.uint64)[None])
    return w


def pool_from_trace(tr, counts_fn):
    from safelife_amd.levels import LevelPool
    frac = float(tr["min_performance_fraction"]) if "min_performance_fraction" in tr else 1.0
    return LevelPool(levels_from_trace(tr), min_performance_fraction=frac, counts_fn=counts_fn)


class OracleBackend(object):
    def __init__(self, pool, B, first_level=0, **kw):
        import oracle
        from safelife_amd.levels import empty_env_arrays
        self.arrays = empty_env_arrays(pool, B)
        self.arrays["level_idx"][:] = first_level
        kw.setdefault("view_shape", (15, 15))
        wrappers = kw.pop("wrappers", None)
        kw.pop("slices", None)
        # the product's keyword -> the oracle's: salt 1 + env_offset (0 here), or 0 = exact pool generators
        kw["stream_salt"] = 1 + int(kw.pop("env_offset", 0)) if kw.pop("episode_streams", True) else 0
        self.env = oracle.OracleEnv(self.arrays, **kw)
        if wrappers is not None:
            self.env.set_wrappers(**wrappers)

    def reset(self):
        return self.env.reset().copy()

    def step(self, actions):
        obs, r, d = self.env.step(actions)
        return obs.copy(), r.copy(), d.copy()

    def get(self, name):
        if name in ("shaped_reward", "inaction_rng"):
            return self.env.wa[name].copy()
        if name == "inaction_board":
            return self.env.wa["baseline"].copy()
        return self.arrays[name].copy()


class DeviceBackend(object):
    def __init__(self, pool, B, first_level=0, **kw):
        from safelife_amd.vector_env import SafeLifeVectorEnv
        w = kw.get("wrappers")
        if w is not None:       # the oracle's keyword names are the product's; absent wrappers are None
            kw["wrappers"] = {k: v for k, v in w.items() if v is not None}
        self.env = SafeLifeVectorEnv(pool, B, first_level=first_level, **kw)

    def reset(self):
        self.env.reset()
        return self.env.numpy("obs")

    def step(self, actions):
        self.env.step(actions)
        return self.env.numpy("obs"), self.env.numpy("reward"), self.env.numpy("done")

    def get(self, name):
        return self.env.numpy(name)


def oracle_counts(boards, goals):
    import oracle
    return oracle.alive_counts_batch(boards, goals)


def replay_trace(tr, backend_cls, counts_fn):
    """Replay a reference SafeLifeEnv trace (B = 1, auto-reset) and assert every recorded output."""
    pool = pool_from_trace(tr, counts_fn)
    kw = env_kwargs_from_trace(tr)
    wrappers = wrappers_from_trace(tr)
    if wrappers is not None:
        kw["wrappers"] = wrappers
    be = backend_cls(pool, 1, first_level=0, auto_reset=True, level_stride=1, episode_streams=False, **kw)
    obs = be.reset()
    resets = list(tr["trace_reset_at"])
    n_resets = len(resets)
    assert np.array_equal(obs[0], tr["trace_reset_obs"][0]), "first reset obs"
    assert np.array_equal(be.get("board")[0], tr["trace_reset_board"][0])
    assert np.array_equal(be.get("rng")[0], tr["trace_reset_rng"][0])
    actions = tr["trace_actions"]
    T = len(tr["trace_reward"])
    episode = 0
    for t in range(T):
        obs, reward, done = be.step(np.array([actions[t]], np.int32))
        where = "step %d" % t
        assert reward[0] == tr["trace_reward"][t], where
        if wrappers is not None:        # float64, bit for bit: same operations in the same order
            assert be.get("shaped_reward")[0] == tr["trace_shaped_reward"][t], where + " shaped reward"
        if "trace_inaction_board" in tr and not tr["trace_done"][t]:   # the "inaction" baseline and its generator
            assert np.array_equal(be.get("inaction_board")[0], tr["trace_inaction_board"][t]), where + " baseline"
            assert np.array_equal(be.get("inaction_rng")[0], tr["trace_inaction_rng_after"][t]), where + " baseline rng"
        assert bool(done[0]) == bool(tr["trace_done"][t]), where
        assert bool(be.get("success")[0]) == bool(tr["trace_success"][t]), where
        assert bool(be.get("times_up")[0]) == bool(tr["trace_times_up"][t]), where
        if tr["trace_done"][t]:
            episode += 1
            if episode >= n_resets:
                break          # the reference ran out of levels here
            # auto-reset: state and observation are those of the next episode's reset
            assert np.array_equal(obs[0], tr["trace_reset_obs"][episode]), where + " reset obs"
            assert np.array_equal(be.get("board")[0], tr["trace_reset_board"][episode]), where
            assert np.array_equal(be.get("rng")[0], tr["trace_reset_rng"][episode]), where
        else:
            assert np.array_equal(be.get("board")[0], tr["trace_board"][t]), where + " board"
            assert np.array_equal(be.get("goals")[0], tr["trace_goals"][t]), where + " goals"
            assert np.array_equal(be.get("agent_loc")[0], tr["trace_agent_loc"][t]), where + " loc"
            assert np.array_equal(be.get("rng")[0], tr["trace_rng_after"][t]), where + " rng"
            assert np.array_equal(obs[0], tr["trace_obs"][t]), where + " obs"
            assert int(be.get("episode_length")[0]) == int(tr["trace_ep_length"][t]), where
            assert be.get("episode_reward")[0] == tr["trace_ep_reward"][t], where
            assert int(be.get("num_steps")[0]) == int(tr["trace_num_steps"][t]), where
    return T


# ---------------------------------------------------------------- multi-agent envs (single_agent=False)

class OracleMultiBackend(object):
    def __init__(self, pool, B, first_level=0, **kw):
        import oracle
        from safelife_amd.levels import empty_env_arrays
        self.arrays = empty_env_arrays(pool, B)
        self.arrays["level_idx"][:] = first_level
        kw.setdefault("view_shape", (15, 15))
        kw["stream_salt"] = 1 + int(kw.pop("env_offset", 0)) if kw.pop("episode_streams", True) else 0
        self.env = oracle.OracleMultiEnv(self.arrays, pool, **kw)

    def reset(self):
        return self.env.reset().copy()

    def step(self, actions):
        obs, r, d = self.env.step(actions)
        return obs.copy(), r.copy(), d.copy()

    def get(self, name):
        if name == "agent_locs":
            return self.env.ma["agent_loc"].copy()
        if name in self.env.ma:
            return self.env.ma[name].copy()
        if name == "times_up":
            return np.repeat(self.arrays["times_up"][:, None], self.env.A, axis=1)
        return self.arrays[name].copy()


class DeviceMultiBackend(object):
    def __init__(self, pool, B, first_level=0, **kw):
        from safelife_amd.multi_env import SafeLifeMultiAgentVectorEnv
        self.env = SafeLifeMultiAgentVectorEnv(pool, B, first_level=first_level, **kw)

    def reset(self):
        self.env.reset()
        return self.env.numpy("obs")

    def step(self, actions):
        self.env.step(actions)
        return self.env.numpy("obs"), self.env.numpy("reward"), self.env.numpy("done")

    def get(self, name):
        return self.env.numpy(name)


MULTI_STATE = ("board", "goals", "rng", "num_steps", "level_idx", "episode_idx", "goals_static", "exit_locs", "agent_locs",
               "old_value", "required_points", "initial_points", "table_idx", "is_active", "episode_reward",
               "episode_length", "reward", "done", "success")


def replay_trace_multi(tr, backend_cls, counts_fn):
    """Replay a reference SafeLifeEnv(single_agent=False) trace (B = 1; the env reloads once every agent is done)."""
    from safelife_amd.levels import LevelPool
    levels = levels_from_trace(tr)
    A = len(levels[0].agent_locs)
    pool = LevelPool(levels, counts_fn=counts_fn, n_agents=A)
    be = backend_cls(pool, 1, first_level=0, auto_reset=True, level_stride=1, episode_streams=False,
                     **env_kwargs_from_trace(tr))
    obs = be.reset()
    n_resets = len(tr["trace_reset_at"])
    assert np.array_equal(obs[0], tr["trace_reset_obs"][0]), "first reset obs"
    assert np.array_equal(be.get("board")[0], tr["trace_reset_board"][0])
    assert np.array_equal(be.get("rng")[0], tr["trace_reset_rng"][0])
    assert np.array_equal(be.get("required_points")[0], tr["trace_reset_required"][0])
    episode = 0
    T = len(tr["trace_reward"])
    for t in range(T):
        obs, reward, done = be.step(tr["trace_actions"][t][None].astype(np.int32))
        where = "step %d" % t
        assert np.array_equal(reward[0], tr["trace_reward"][t]), where
        assert np.array_equal(done[0].astype(bool), tr["trace_done"][t]), where
        assert np.array_equal(be.get("success")[0].astype(bool), tr["trace_success"][t]), where
        assert bool(be.get("times_up")[0][0]) == bool(tr["trace_times_up"][t]), where
        if np.all(tr["trace_done"][t]):
            episode += 1
            if episode >= n_resets:
                break
            assert np.array_equal(obs[0], tr["trace_reset_obs"][episode]), where + " reset obs"
            assert np.array_equal(be.get("board")[0], tr["trace_reset_board"][episode]), where
            assert np.array_equal(be.get("rng")[0], tr["trace_reset_rng"][episode]), where
            assert np.array_equal(be.get("required_points")[0], tr["trace_reset_required"][episode]), where
        else:
            assert np.array_equal(be.get("board")[0], tr["trace_board"][t]), where + " board"
            assert np.array_equal(be.get("goals")[0], tr["trace_goals"][t]), where + " goals"
            assert np.array_equal(be.get("agent_locs")[0], tr["trace_agent_locs"][t]), where + " locs"
            assert np.array_equal(be.get("rng")[0], tr["trace_rng_after"][t]), where + " rng"
            assert np.array_equal(obs[0], tr["trace_obs"][t]), where + " obs"
            assert np.array_equal(be.get("episode_length")[0], tr["trace_ep_length"][t]), where
            assert np.array_equal(be.get("episode_reward")[0], tr["trace_ep_reward"][t]), where
            assert int(be.get("num_steps")[0]) == int(tr["trace_num_steps"][t]), where
    return T


def replay_trace_terminal(tr, backend_cls, counts_fn):
    """Same trace without auto-reset: checks the terminal board/observation of every episode."""
    pool = pool_from_trace(tr, counts_fn)
    kw = env_kwargs_from_trace(tr)
    actions = tr["trace_actions"]
    resets = list(tr["trace_reset_at"]) + [len(tr["trace_reward"])]
    checked = 0
    for ep in range(len(resets) - 1):
        be = backend_cls(pool, 1, first_level=ep, auto_reset=False, episode_streams=False, **kw)
        be.reset()
        for t in range(resets[ep], resets[ep + 1]):
            obs, reward, done = be.step(np.array([actions[t]], np.int32))
        t = resets[ep + 1] - 1
        assert np.array_equal(be.get("board")[0], tr["trace_board"][t]), "terminal board ep %d" % ep
        assert np.array_equal(obs[0], tr["trace_obs"][t]), "terminal obs ep %d" % ep
        assert np.array_equal(be.get("agent_loc")[0], tr["trace_agent_loc"][t])
        assert reward[0] == tr["trace_reward"][t]
        checked += 1
    return checked


def pool_from_fixture(name, counts_fn, n=None, **kw):
    """LevelPool from tests/golden/pool_<name>.npz (reference procgen output)."""
    from safelife_amd.levels import Level, LevelPool
    with np.load(os.path.join(GOLDEN, "pool_%s.npz" % name)) as d:
        L = int(d["n_levels"]) if n is None else min(n, int(d["n_levels"]))
        levels = []
        for k in range(L):
            levels.append(Level(d["board"][k], d["goals"][k], d["agent_locs"][k],
                                spawn_prob=float(d["spawn_prob"][k]),
                                min_performance=float(d["min_performance"][k]),
                                points_table=d["points_table"][k], rng_words=d["rng"][k]))
        ref = {"required_points": d["required_points"][:L].copy(),
               "initial_available_points": d["initial_available_points"][:L].copy()}
    return LevelPool(levels, counts_fn=counts_fn, **kw), ref


def smoke_check():
    """One small fused step batch on the GPU against the oracle (used by __graft_entry__.smoke)."""
    pool, _ = pool_from_fixture("append_spawn_25", oracle_counts, n=8)
    B, T = 16, 12
    rng = np.random.default_rng(0)
    kw = dict(auto_reset=True, time_limit=8, view_shape=(25, 25),
              output_channels=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27))
    first = np.arange(B) % len(pool)
    dev = DeviceBackend(pool, B, first_level=first, **kw)
    cpu = OracleBackend(pool, B, first_level=first, **kw)
    assert np.array_equal(dev.reset(), cpu.reset())
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(o1, o2), t
        for name in ("board", "goals", "agent_loc", "rng", "num_steps", "level_idx"):
            assert np.array_equal(dev.get(name), cpu.get(name)), (t, name)
    # ... and the headline's launcher: a staged region on the library's AQL queues (slhip_queues_stage / _go)
    import torch
    from safelife_amd._hip import SafeLifeHipError
    acts = rng.integers(0, 9, (6, B)).astype(np.int32)
    try:
        dev.env.queues_open(2)
    except SafeLifeHipError:
        return
    d_acts = torch.from_numpy(acts).to(dev.env.device)
    torch.cuda.synchronize()
    dev.env.step_queues_many(d_acts, defer=True)
    dev.env.queues_go()
    dev.env.queues_sync()
    for t in range(len(acts)):
        cpu.env.step(acts[t])
    for name in ("board", "goals", "agent_loc", "rng", "num_steps", "level_idx", "reward", "done"):
        assert np.array_equal(dev.get(name), cpu.get(name)), ("queues", name)
    dev.env.queues_close()


# ---- the shipped benchmark levels in bulk (tests/golden/levels + bulk_levels.npz, make_golden.py gen_bulk) ----------

def bulk_hash(board, goals, rewards, dones, rng_words, agent_loc):
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a, dt in ((board, np.uint16), (goals, np.uint16), (rewards, np.float32), (dones, np.uint8),
                  (rng_words, np.uint64), (agent_loc, np.int32)):
        h.update(np.ascontiguousarray(a, dtype=dt).tobytes())
    return np.frombuffer(h.digest(), np.uint64)[0]


def bulk_levels_digests(backend_cls, counts_fn):
    """Every level of the shipped archives, loaded through safelife_amd.levels.load_levels, stepped in one batch per
    board shape with the fixture's action streams -> (digests, expected digests), in file order."""
    from safelife_amd.levels import LevelPool, load_levels
    with np.load(os.path.join(GOLDEN, "bulk_levels.npz")) as d:
        files, counts, want, T = [str(f) for f in d["files"]], d["counts"], d["digests"], int(d["steps"])
    levels = []
    for rel, n in zip(files, counts):
        got = load_levels(os.path.join(GOLDEN, "levels", rel))
        assert len(got) == n, rel
        levels += got
    for idx, lv in enumerate(levels):
        lv.seed = 5000 + idx                         # SafeLifeGame.seed of the generator
    out = np.zeros(len(levels), np.uint64)
    for shape in sorted(set(lv.shape for lv in levels)):
        ids = [i for i, lv in enumerate(levels) if lv.shape == shape]
        pool = LevelPool([levels[i] for i in ids], counts_fn=counts_fn)
        B = len(ids)
        be = backend_cls(pool, B, first_level=np.arange(B), auto_reset=False, episode_streams=False,
                         time_limit=1000, view_shape=(15, 15), output_channels=None)
        be.reset()
        acts = np.stack([np.random.default_rng(i).integers(0, 9, T) for i in ids], axis=1).astype(np.int32)
        rewards, dones = np.zeros((T, B), np.float32), np.zeros((T, B), np.uint8)
        for t in range(T):
            _, rewards[t], dones[t] = be.step(np.ascontiguousarray(acts[t]))
        board, goals, rng, loc = (be.get(k) for k in ("board", "goals", "rng", "agent_loc"))
        for k, i in enumerate(ids):
            out[i] = bulk_hash(board[k], goals[k], rewards[:, k], dones[:, k], rng[k], loc[k])
    return out, want
