#!/usr/bin/env python3
"""
Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference and oracle/_ref, see
oracle/Makefile).  Nothing here travels as code to the GPU box except this
script itself; the fixtures are data: inputs (level boards, seeds, action
streams) and the outputs the reference produced for them.

    python tests/golden/make_golden.py            # everything except the slow 64x64 pool
    python tests/golden/make_golden.py --nav64 8  # also 8 navigation 64x64 levels (~33 s each)

The reference's Python needs `gym` and `pyemd`, which are not installed; both
are replaced by in-memory placeholders (a base class / an unused function) so
that `safelife.safelife_env` imports.  The compiled reference extension from
oracle/_ref is registered as `safelife.speedups`.
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def import_reference():
    sys.path.insert(0, REPO)
    import oracle
    oracle.build()
    speedups = oracle.load_ref()
    assert speedups is not None, "run `make -C oracle ref` first"
    sys.modules["safelife.speedups"] = speedups

    gym = types.ModuleType("gym")

    class Env:
        pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

    class _Space:
        def __init__(self, *a, **kw):
            self.args, self.kw = a, kw

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete = spaces.Box = _Space
    gym.Env, gym.Wrapper, gym.spaces = Env, Wrapper, spaces
    gym.register = lambda *a, **kw: None
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    pyemd = types.ModuleType("pyemd")
    pyemd.emd = lambda *a, **kw: float("nan")
    sys.modules["pyemd"] = pyemd

    sys.path.insert(0, REFERENCE)
    import safelife  # noqa: F401
    from safelife import safelife_game, safelife_env, level_iterator, env_wrappers, side_effects
    return types.SimpleNamespace(
        speedups=speedups, game=safelife_game, env=safelife_env, levels=level_iterator,
        wrappers=env_wrappers, side_effects=side_effects, oracle=oracle)


def words(bitgen):
    st = bitgen.state["state"]
    m = (1 << 64) - 1
    return np.array([st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m], np.uint64)


PALETTE = np.array(
    [0] * 12 + [9] * 5 + [1, 16, 17, 32788, 152, 152 | 0x200, 152 | 0x800, 144, 48, 53, 85, 32884,
                          272, 9 | 0x200, 9 | 0x400, 9 | 0x600, 9 | 0x800, 9 | 0xE00, 122,
                          0x8000 | 9, 4 | 8, 1 | 4 | 0x400, 32, 64, 128 | 0x400, 9 | 0xA00],
    dtype=np.uint16)


def random_board(rng, h, w, kind):
    if kind == 0:
        return PALETTE[rng.integers(0, len(PALETTE), (h, w))]
    if kind == 1:  # sparse life + a few spawners
        b = np.where(rng.random((h, w)) < 0.3, 9, 0).astype(np.uint16)
        for _ in range(3):
            b[rng.integers(0, h), rng.integers(0, w)] = 152 | (int(rng.integers(0, 8)) << 9)
        return b
    return rng.integers(0, 65536, (h, w)).astype(np.uint16)


# ----------------------------------------------------------------- primitives

def gen_primitives(R, out):
    """(board, p, seed, n) -> board_out + rng state; counts; actions; occupancy."""
    sp = R.speedups
    rng = np.random.default_rng(20240928)
    cases = {}
    shapes = [(25, 25)] * 10 + [(26, 26)] * 6 + [(64, 64)] * 2 + [(3, 3), (3, 7), (7, 3), (4, 5),
                                                                 (15, 15), (10, 33), (33, 10), (20, 20)]
    k = 0
    for i, (h, w) in enumerate(shapes):
        for kind in (0, 1, 2):
            board = random_board(rng, h, w, kind)
            goals = (rng.integers(0, 8, (h, w)) << 9).astype(np.uint16)
            for p, n in ((0.3, 1), (0.3, 3), (1.0, 1), (0.0, 2), (0.05, 7)):
                if kind == 2 and n > 1 and i % 2:
                    continue
                bg = np.random.PCG64(1000 + k)
                w0 = words(bg)
                sp.set_bit_generator(bg)
                res = sp.advance_board(board, p, n)
                cases["adv_%03d_in" % k] = board
                cases["adv_%03d_out" % k] = res
                cases["adv_%03d_p_n" % k] = np.array([p, n], np.float64)
                cases["adv_%03d_rng0" % k] = w0
                cases["adv_%03d_rng1" % k] = words(bg)
                k += 1
            cases["cnt_%03d_board" % (i * 3 + kind)] = board
            cases["cnt_%03d_goals" % (i * 3 + kind)] = goals
            cases["cnt_%03d_out" % (i * 3 + kind)] = sp.alive_counts(board, goals)
    cases["n_adv"] = np.array(k)
    cases["n_cnt"] = np.array(len(shapes) * 3)

    # execute_actions: dense coverage of every branch, multi-agent, aliasing on 3-wide boards
    agent_cells = [122, 122 | 0x200, 122 | 4, 122 | 256, 2 | 8 | 0x8000, 122 | 0x3000, 2]
    m = 0
    for t in range(400):
        h, w = (3, 3) if t % 10 == 0 else ((3, 5) if t % 10 == 1 else
                                          tuple(int(x) for x in rng.integers(4, 12, 2)))
        board = PALETTE[rng.integers(0, len(PALETTE), (h, w))]
        if t % 3 == 0:
            board[rng.random((h, w)) < 0.5] = 0
        na = int(rng.integers(1, 4))
        locs = np.stack([rng.integers(0, h, na), rng.integers(0, w, na)], 1).astype(np.int64)
        for (y, x) in locs:
            if rng.random() < 0.85:
                board[y, x] = agent_cells[int(rng.integers(0, len(agent_cells)))]
        acts = rng.integers(0, 9, na).astype(np.int64)
        if t % 7 == 0:
            acts = acts[:1]  # broadcast form (module.c:185)
        b1, l1 = board.copy(), locs.copy()
        sp.execute_actions(b1, l1, acts)
        cases["act_%03d_board" % m] = board
        cases["act_%03d_locs" % m] = locs
        cases["act_%03d_acts" % m] = acts
        cases["act_%03d_board_out" % m] = b1
        cases["act_%03d_locs_out" % m] = l1
        m += 1
    cases["n_act"] = np.array(m)

    # life_occupancy
    q = 0
    for (h, w, n) in ((25, 25, 1000), (26, 26, 100), (9, 11, 50), (64, 64, 30)):
        for kind in (0, 1):
            board = random_board(rng, h, w, kind)
            bg = np.random.PCG64(5000 + q)
            w0 = words(bg)
            sp.set_bit_generator(bg)
            occ = sp.life_occupancy(board, 0.3, n)
            cases["occ_%02d_in" % q] = board
            cases["occ_%02d_n" % q] = np.array(n)
            cases["occ_%02d_rng0" % q] = w0
            cases["occ_%02d_rng1" % q] = words(bg)
            cases["occ_%02d_out" % q] = occ
            q += 1
    cases["n_occ"] = np.array(q)
    np.savez_compressed(os.path.join(out, "primitives.npz"), **cases)
    print("primitives:", k, "advance,", m, "actions,", q, "occupancy cases")


def gen_patterns(R, out):
    """Known-answer Life evolution of the shipped patterns (levels/patterns/*.npz)."""
    sp = R.speedups
    cases = {}
    for name in ("glider", "acorn", "rpentomino", "growth"):
        d = np.load(os.path.join(REFERENCE, "safelife/levels/patterns", name + ".npz"))
        board = d["board"]
        cases[name + "_in"] = board
        for n in (1, 4, 20, 100):
            cases["%s_n%d" % (name, n)] = sp.advance_board(board, 0.3, n)
    np.savez_compressed(os.path.join(out, "patterns.npz"), **cases)


# ------------------------------------------------------------------ env traces

def level_record(game):
    """Everything a from-scratch loader needs, as plain arrays (post-`loaddata`, pre-reset)."""
    d = game._init_data
    keys = d.dtype.fields if hasattr(d, "dtype") else d
    rec = {
        "board": np.array(d["board"], np.uint16),
        "goals": np.array(d["goals"], np.uint16),
        "spawn_prob": np.float64(d["spawn_prob"]) if "spawn_prob" in keys else np.float64(0.3),
        "min_performance": np.float64(d["min_performance"]) if "min_performance" in keys else np.float64(-1),
    }
    if "agent_loc" in keys:
        rec["agent_loc"] = np.array(d["agent_loc"], np.int64)
    if "agent_locs" in keys:
        rec["agent_locs"] = np.array(d["agent_locs"], np.int64).reshape(-1, 2)
    if "orientation" in keys:
        rec["orientation"] = np.int64(d["orientation"])
    if "points_table" in keys:
        rec["points_table"] = np.array(d["points_table"], np.int64)
    return rec


def run_trace(R, games, actions, env_kw, wrappers=None, min_perf_fraction=None, inaction_seed=None):
    """Drive the reference SafeLifeEnv over `games` (one episode each, in order) with the
    action stream; auto-reset on done like training/base_algo.py:231-236.

    `wrappers`: dict(movement=dict(...)|None, exit_bonus=float|None, side_effect=dict(...)|None):
    the reference's own env_wrappers stacked in the order of training/env_factory.py:277-283; the
    reward the outermost wrapper returns is recorded as `shaped_reward` (float64) next to the inner
    SafeLifeEnv reward.

    `inaction_seed`: seed of the process-wide generator (safelife/random.py:13) for the run -- the stream
    SimpleSideEffectPenalty's "inaction" baseline draws its spawners from (env_wrappers.py:179-180: advance_board
    outside any game method, i.e. after set_rng.__exit__ has put the global generator back)."""
    SafeLifeEnv = R.env.SafeLifeEnv
    global_gen = None
    if inaction_seed is not None:
        import safelife.random as sl_random
        global_gen = np.random.default_rng(inaction_seed)
        sl_random.random_gen = global_gen
        R.speedups.set_bit_generator(global_gen.bit_generator)
    it = iter(games)
    env = SafeLifeEnv(it, **env_kw)
    wrapped = env
    inner = {}
    if wrappers is not None:
        inner_step = env.step

        def recording_step(a):
            ret = inner_step(a)
            inner["reward"] = ret[1]
            return ret
        env.step = recording_step
        W = R.wrappers
        if wrappers.get("movement") is not None:
            wrapped = W.MovementBonusWrapper(wrapped, **wrappers["movement"])
        if wrappers.get("exit_bonus") is not None:
            wrapped = W.ExtraExitBonus(wrapped, bonus=wrappers["exit_bonus"])
        if wrappers.get("side_effect") is not None:
            wrapped = W.SimpleSideEffectPenalty(wrapped, **wrappers["side_effect"])
    if min_perf_fraction is not None:
        wrapped = R.wrappers.MinPerformanceScheduler(wrapped, min_performance_fraction=min_perf_fraction)
    rec = {k: [] for k in ("obs", "reward", "done", "board", "goals", "agent_loc", "times_up",
                           "ep_length", "ep_reward", "success", "reset_obs", "reset_board",
                           "reset_rng", "reset_required", "rng_after", "num_steps", "shaped_reward",
                           "inaction_board", "inaction_rng_after")}
    if global_gen is not None:
        rec["inaction_rng0"] = words(global_gen.bit_generator)

    def note_reset(obs):
        rec["reset_obs"].append(obs.copy())
        rec["reset_board"].append(env.game.board.copy())
        rec["reset_rng"].append(words(env.game._rng.bit_generator))
        rec["reset_required"].append(np.int64(env.game.required_points()[0])
                                     if len(env.game.agent_locs) else np.int64(0))

    obs = wrapped.reset()
    note_reset(obs)
    reset_at = [0]
    for t, a in enumerate(actions):
        obs, reward, done, info = wrapped.step(int(a))
        rec["obs"].append(obs.copy())
        if wrappers is not None:
            assert isinstance(reward, (float, np.floating)), type(reward)
            rec["shaped_reward"].append(np.float64(reward))
            reward = inner["reward"]
        rec["reward"].append(np.float32(reward))
        rec["done"].append(bool(done))
        rec["board"].append(info["board"].copy())
        rec["goals"].append(info["goals"].copy())
        loc = info["agent_locs"]
        rec["agent_loc"].append(loc[0].copy() if len(loc) else np.array([-1, -1]))
        rec["times_up"].append(bool(info["times_up"]))
        rec["ep_length"].append(int(info["episode"]["length"]))
        rec["ep_reward"].append(np.float32(info["episode"]["reward"]))
        rec["success"].append(bool(info["episode"]["success"]))
        rec["rng_after"].append(words(env.game._rng.bit_generator))
        rec["num_steps"].append(env.game.num_steps)
        if global_gen is not None:
            w_ = wrapped
            while not isinstance(w_, R.wrappers.SimpleSideEffectPenalty):
                w_ = w_.env
            rec["inaction_board"].append(w_.baseline_board.copy())
            rec["inaction_rng_after"].append(words(global_gen.bit_generator))
        if done:
            try:
                obs = wrapped.reset()
            except StopIteration:
                break
            note_reset(obs)
            reset_at.append(t + 1)
    if global_gen is None:
        del rec["inaction_board"], rec["inaction_rng_after"]
    out = {k: np.array(v) for k, v in rec.items()}
    out["reset_at"] = np.array(reset_at)
    out["actions"] = np.array(actions[:len(rec["reward"])], np.int32)
    return out


def seeded(game, seed):
    game.seed = np.random.SeedSequence(seed)
    return game


def normalize_level(data):
    """Legacy single-agent key `agent_loc` (x, y) -> `agent_locs` [[row, col]].

    safelife_game.py:219-221 turns the legacy key into a negative-stride VIEW; the reference's
    execute_actions wrapper relies on numpy's write-back-if-copy for non-contiguous `locations`
    (module.c:163-166,188-189), which numpy >= 1.23 no longer performs for NPY_ARRAY_INOUT_ARRAY,
    so under the numpy 2.2 of this image the agent location would silently stop updating.
    Feeding the modern key (a contiguous array) makes the reference behave as designed.
    """
    names = data.dtype.names if hasattr(data, "dtype") and data.dtype.names else list(data.keys())
    out = {k: np.array(data[k]) for k in names}
    # Some shipped examples name the experimental, pure-Python `GameOfLife` class
    # (safelife_game.py:768-838), which is outside the hot path; dropping the key makes
    # `loaddata` build a SafeLifeGame, i.e. the C physics this repository restates.
    out.pop("class", None)
    if "agent_loc" in out:
        out["agent_locs"] = np.ascontiguousarray(np.array(out.pop("agent_loc"))[None, ::-1])
    return out


def load_level(R, rel):
    path = os.path.join(REFERENCE, "safelife/levels", rel)
    with np.load(path) as d:
        return normalize_level({k: d[k] for k in d.keys()})


def greedy_actions(R, game, rng, n, p_random=0.35):
    """Action stream that wanders but heads for the exit (BFS over empty cells on a private
    copy of the game) so that episodes actually end by success."""
    from collections import deque
    acts = []
    g = R.game.SafeLifeGame.loaddata(game._init_data)
    moves = ((1, -1, 0), (2, 0, 1), (3, 1, 0), (4, 0, -1))
    for _ in range(n):
        a = None
        if len(g.agent_locs) and len(g.exit_locs[0]) and rng.random() >= p_random:
            h, w = g.board.shape
            y0, x0 = (int(v) for v in g.agent_locs[0])
            target = (int(g.exit_locs[0][0]), int(g.exit_locs[1][0]))
            first = {(y0, x0): None}
            dq = deque([(y0, x0)])
            while dq and target not in first:
                y, x = dq.popleft()
                for act, dy, dx in moves:
                    q = ((y + dy) % h, (x + dx) % w)
                    if q in first or not (g.board[q] == 0 or q == target):
                        continue
                    first[q] = first[(y, x)] or act
                    dq.append(q)
            a = first.get(target)
        if a is None:
            a = int(rng.integers(0, 9))
        acts.append(a)
        g.execute_actions(a)
        g.advance_board()
        g.update_exit_colors()
        if len(g.agent_locs) and not g.agent_is_active()[0]:
            break
    return acts


def trace_blob(R, name, level_datas, seeds, actions, env_kw, min_perf_fraction=None, wrappers=None,
               inaction_seed=None):
    Game = R.game.SafeLifeGame
    games = [seeded(Game.loaddata(d), s) for d, s in zip(level_datas, seeds)]
    tr = run_trace(R, games, actions, env_kw, wrappers=wrappers, min_perf_fraction=min_perf_fraction,
                   inaction_seed=inaction_seed)
    rng0 = tr.pop("inaction_rng0", None)
    games2 = [seeded(Game.loaddata(d), s) for d, s in zip(level_datas, seeds)]
    blob = {}
    for i, g in enumerate(games2):
        for k, v in level_record(g).items():
            blob["level%d_%s" % (i, k)] = v
        blob["level%d_rng" % i] = words(g._rng.bit_generator)
    blob["n_levels"] = np.array(len(games2))
    for k, v in tr.items():
        blob["trace_" + k] = v
    for k, v in env_kw.items():
        blob["env_" + k] = np.array(-1 if v is None else v)
    if min_perf_fraction is not None:
        blob["min_performance_fraction"] = np.array(min_perf_fraction)
    if wrappers is not None:
        mv, se = wrappers.get("movement"), wrappers.get("side_effect")
        if mv is not None:
            blob["wrap_movement"] = np.array([mv.get("movement_bonus", 0.1), mv.get("movement_bonus_power", 1e-100),
                                              mv.get("movement_bonus_period", 4), float(mv.get("as_penalty", True))])
        if wrappers.get("exit_bonus") is not None:
            blob["wrap_exit_bonus"] = np.array(float(wrappers["exit_bonus"]))
        if se is not None:
            assert (se.get("baseline", "starting-state") == "inaction") == (inaction_seed is not None)
            blob["wrap_side_effect"] = np.array([se.get("penalty_coef", 0.0), float(se.get("ignore_reward_cells", False))])
            if inaction_seed is not None:
                blob["wrap_inaction_rng"] = rng0
    print("trace %-28s steps=%4d episodes=%d sum_reward=%.1f success=%d%s" % (
        name, len(tr["reward"]), len(tr["reset_at"]), tr["reward"].sum(), tr["success"].sum(),
        "" if wrappers is None else " shaped_sum=%.4f" % tr["shaped_reward"].sum()))
    return blob


def gen_wrapper_traces(R, out, only_prefix=""):
    """Reference env_wrappers stacked as in training/env_factory.py:277-283 over the reference env."""
    Game = R.game.SafeLifeGame
    rng = np.random.default_rng(177)
    no_se = dict(should_calculate_side_effects=False)
    kw = dict(view_shape=(25, 25), output_channels=None, time_limit=100, **no_se)
    traces = {}
    training = dict(movement=dict(as_penalty=True), exit_bonus=0.5,
                    side_effect=dict(baseline="starting-state", penalty_coef=0.3))
    # the training stack on archives where the agent reaches open exits and on stochastic ones
    for arch, n_lv, first, frac in (("prune-still", 6, 20, None), ("append-spawn", 5, 20, None),
                                    ("append-still", 4, 40, 0.02), ("navigation", 3, 10, 0.5)):
        with np.load(os.path.join(REFERENCE, "safelife/levels/benchmarks/v1.0/%s.npz" % arch)) as d:
            levels = [normalize_level(d["levels"][first + i]) for i in range(n_lv)]
        if frac is None:
            for l in levels:
                l["min_performance"] = np.float64(-1)
        seeds = [500 + i for i in range(n_lv)]
        games = [seeded(Game.loaddata(l), sd) for l, sd in zip(levels, seeds)]
        acts = []
        for g in games:
            acts += greedy_actions(R, g, rng, 120, p_random=0.15)
        traces["wrap_train_" + arch] = trace_blob(R, "wrap_train_" + arch, levels, seeds, acts, kw,
                                                  min_perf_fraction=frac, wrappers=training)
    # other parameterisations: bonus instead of penalty, real exponent, period 3, reward cells ignored
    with np.load(os.path.join(REFERENCE, "safelife/levels/benchmarks/v1.0/prune-still.npz")) as d:
        levels = [normalize_level(d["levels"][60 + i]) for i in range(4)]
    for l in levels:
        l["min_performance"] = np.float64(-1)
    seeds = [700 + i for i in range(4)]
    games = [seeded(Game.loaddata(l), sd) for l, sd in zip(levels, seeds)]
    acts = []
    for g in games:
        acts += greedy_actions(R, g, rng, 120, p_random=0.25)
    other = dict(movement=dict(as_penalty=False, movement_bonus=0.25, movement_bonus_power=0.5,
                               movement_bonus_period=3),
                 exit_bonus=1.5, side_effect=dict(penalty_coef=0.125, ignore_reward_cells=True))
    traces["wrap_other_prune-still"] = trace_blob(R, "wrap_other_prune-still", levels, seeds, acts, kw, wrappers=other)
    only_se = dict(side_effect=dict(penalty_coef=1.0))
    lv = load_level(R, "benchmarks/v0.1/append-stochastic-1.npz")
    traces["wrap_se_append-stochastic-1"] = trace_blob(
        R, "wrap_se_append-stochastic-1", [lv], [5], rng.integers(0, 9, 200),
        dict(view_shape=(25, 25), output_channels=None, **no_se), wrappers=only_se)
    # the "inaction" baseline (env_wrappers.py:179-180): levels with spawners (the baseline's draws come from the
    # process-wide generator, seeded here and recorded) and still ones with ignore_reward_cells
    for arch, n_lv, first, se_kw, seed0 in (("append-spawn", 4, 30, dict(penalty_coef=0.3), 4242),
                                            ("prune-still", 4, 70, dict(penalty_coef=0.5, ignore_reward_cells=True), 4343),
                                            ("prune-spawn", 3, 12, dict(penalty_coef=1.0), 4444)):
        with np.load(os.path.join(REFERENCE, "safelife/levels/benchmarks/v1.0/%s.npz" % arch)) as d:
            levels = [normalize_level(d["levels"][first + i]) for i in range(n_lv)]
        for l in levels:
            l["min_performance"] = np.float64(-1)
        seeds = [900 + i for i in range(n_lv)]
        games = [seeded(Game.loaddata(l), sd) for l, sd in zip(levels, seeds)]
        acts = []
        for g in games:
            acts += greedy_actions(R, g, rng, 100, p_random=0.2)
        stack = dict(movement=dict(as_penalty=True), exit_bonus=0.5, side_effect=dict(baseline="inaction", **se_kw))
        traces["wrap_inaction_" + arch] = trace_blob(R, "wrap_inaction_" + arch, levels, seeds, acts, kw,
                                                     wrappers=stack, inaction_seed=seed0)
    only_mv = dict(movement=dict(movement_bonus_period=8, movement_bonus_power=1.0))
    lv = load_level(R, "patterns/glider.npz")
    traces["wrap_mv_noagent"] = trace_blob(R, "wrap_mv_noagent", [lv, lv], [0, 1], [0, 3, 5],
                                           dict(view_shape=(9, 9), output_channels=None, **no_se), wrappers=only_mv)
    for name, blob in traces.items():
        if name.startswith(only_prefix):
            np.savez_compressed(os.path.join(out, "trace_%s.npz" % name), **blob)


def gen_env_traces(R, out):
    Game = R.game.SafeLifeGame
    rng = np.random.default_rng(77)
    traces = {}

    def add(name, level_datas, seeds, actions, env_kw, min_perf_fraction=None):
        traces[name] = trace_blob(R, name, level_datas, seeds, actions, env_kw, min_perf_fraction)

    full19 = tuple(range(16)) + (25, 26, 27)
    train15 = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27)
    no_se = dict(should_calculate_side_effects=False)

    # C1: 25x25 append-still benchmark level, 1000 random actions (BASELINE.json configs[0])
    lv = load_level(R, "benchmarks/v0.1/append-still-1.npz")
    acts = np.random.default_rng(0).integers(0, 9, 1000)
    add("c1_append_still_1", [lv], [11], acts,
        dict(view_shape=(25, 25), output_channels=None, **no_se))
    # same level, channel observation, default 15x15 view, short
    add("append_still_1_chan19", [lv], [11], acts[:120],
        dict(view_shape=(15, 15), output_channels=full19, **no_se))
    add("append_still_1_chan15_v25", [lv], [11], acts[:60],
        dict(view_shape=(25, 25), output_channels=train15, **no_se))

    # stochastic levels (spawners): RNG order matters
    for nm in ("append-stochastic-1", "prune-stochastic-2", "append-stochastic-osc-1", "prune-dynamic-3"):
        lv = load_level(R, "benchmarks/v0.1/%s.npz" % nm)
        acts = rng.integers(0, 9, 300)
        add("v01_" + nm, [lv], [5], acts, dict(view_shape=(25, 25), output_channels=None, **no_se))

    # mechanics showcases
    for nm in ("containment", "sokuban", "one way", "rainbow spawn", "color test", "super weed",
               "spawn and oscillate 1", "predator"):
        lv = load_level(R, "examples/%s.npz" % nm)
        acts = rng.integers(0, 9, 250)
        add("ex_" + nm.replace(" ", "_"), [lv], [3], acts,
            dict(view_shape=(15, 15), output_channels=None, time_limit=200, **no_se))

    # several episodes with exits reached, time-outs and auto-reset; v1.0 archives (26x26, legacy keys)
    for arch, n_lv, tl in (("prune-still", 6, 80), ("append-still", 6, 80), ("append-spawn", 5, 100),
                           ("prune-spawn", 4, 100), ("navigation", 4, 150), ("append-dynamic", 3, 60)):
        with np.load(os.path.join(REFERENCE, "safelife/levels/benchmarks/v1.0/%s.npz" % arch)) as d:
            levels = [normalize_level(d["levels"][i]) for i in range(n_lv)]
        games = [seeded(Game.loaddata(l), 100 + i) for i, l in enumerate(levels)]
        acts = []
        for g in games:
            acts += greedy_actions(R, g, rng, tl + 20)
        add("v10_" + arch, levels, [100 + i for i in range(n_lv)], acts,
            dict(view_shape=(25, 25), output_channels=None, time_limit=tl, **no_se),
            min_perf_fraction=0.01)

    # exits open from the start (min_performance = -1) so that episodes end by success
    for arch, n_lv in (("prune-still", 8), ("append-spawn", 6)):
        with np.load(os.path.join(REFERENCE, "safelife/levels/benchmarks/v1.0/%s.npz" % arch)) as d:
            levels = [normalize_level(d["levels"][20 + i]) for i in range(n_lv)]
        for l in levels:
            l["min_performance"] = np.float64(-1)
        games = [seeded(Game.loaddata(l), 300 + i) for i, l in enumerate(levels)]
        acts = []
        for g in games:
            acts += greedy_actions(R, g, rng, 120, p_random=0.12)
        add("v10_%s_open" % arch, levels, [300 + i for i in range(n_lv)], acts,
            dict(view_shape=(25, 25), output_channels=None, time_limit=100, **no_se))

    # bigger-than-board view and tiny view
    lv = load_level(R, "benchmarks/v0.1/prune-still-2.npz")
    add("view33_prune_still_2", [lv], [1], rng.integers(0, 9, 80),
        dict(view_shape=(33, 33), output_channels=None, **no_se))
    add("view5x9_prune_still_2", [lv], [1], rng.integers(0, 9, 80),
        dict(view_shape=(5, 9), output_channels=None, remove_white_goals=False, **no_se))

    # the worked known-answer example of SURVEY Appendix C (7x7, scripted)
    g = Game((7, 7))
    g.board[3, 5] = 272
    g.board[1, 1] = 9 | 0x200
    g.min_performance = -1
    data = g.serialize()
    add("worked_7x7_exit", [data], [0], [2, 2, 2, 0], dict(time_limit=5, output_channels=None,
                                                          view_shape=(7, 7), **no_se))
    add("worked_7x7_noop", [data], [0], [0] * 7, dict(time_limit=5, output_channels=None,
                                                    view_shape=(7, 7), **no_se))
    # no-agent level (patterns): reward 0, done immediately
    lv = load_level(R, "patterns/glider.npz")
    add("pattern_glider_noagent", [lv, lv], [0, 1], [0, 3, 5],
        dict(view_shape=(9, 9), output_channels=None, **no_se))

    for name, blob in traces.items():
        np.savez_compressed(os.path.join(out, "trace_%s.npz" % name), **blob)


# --------------------------------------------------------------- multi-agent traces (single_agent=False)

def run_trace_multi(R, games, actions, env_kw):
    """The reference SafeLifeEnv with single_agent=False (safelife_env.py:148-218) over `games`, one episode each:
    per-agent observation, reward, done, episode accumulators; the env is reset when EVERY agent is done
    (training/base_algo.py:231-236).  actions: int [T, A]."""
    env = R.env.SafeLifeEnv(iter(games), single_agent=False, **env_kw)
    rec = {k: [] for k in ("obs", "reward", "done", "board", "goals", "agent_locs", "times_up", "ep_length", "ep_reward",
                           "success", "reset_obs", "reset_board", "reset_rng", "reset_required", "rng_after", "num_steps")}

    def note_reset(obs):
        rec["reset_obs"].append(obs.copy())
        rec["reset_board"].append(env.game.board.copy())
        rec["reset_rng"].append(words(env.game._rng.bit_generator))
        rec["reset_required"].append(np.array(env.game.required_points(), np.int64))

    note_reset(env.reset())
    reset_at = [0]
    for t, a in enumerate(actions):
        obs, reward, done, info = env.step(np.array(a, np.int64))
        assert reward.dtype == np.float32 and obs.shape[0] == len(a)
        rec["obs"].append(obs.copy())
        rec["reward"].append(reward.copy())
        rec["done"].append(np.array(done, bool))
        rec["board"].append(info["board"].copy())
        rec["goals"].append(info["goals"].copy())
        rec["agent_locs"].append(np.array(info["agent_locs"], np.int64).copy())
        rec["times_up"].append(bool(info["times_up"]))
        rec["ep_length"].append(np.array(info["episode"]["length"], np.int64).copy())
        rec["ep_reward"].append(np.array(info["episode"]["reward"], np.float32).copy())
        rec["success"].append(np.array(info["episode"]["success"], bool))
        rec["rng_after"].append(words(env.game._rng.bit_generator))
        rec["num_steps"].append(env.game.num_steps)
        if np.all(done):
            try:
                obs = env.reset()
            except StopIteration:
                break
            note_reset(obs)
            reset_at.append(t + 1)
    out = {k: np.array(v) for k, v in rec.items()}
    out["reset_at"] = np.array(reset_at)
    out["actions"] = np.array(actions[:len(rec["reward"])], np.int32)
    return out


def gen_multi_agent_traces(R, out):
    """Two-agent levels of the reference's own multi-agent specs (levels/random/multi-agent/*.yaml: distinct agent
    colours, flags and points tables) at 26x26, and a hand-made 10x10 level on which one agent walks into the open exit
    while the other plays on -- the multi-agent half of SafeLifeEnv.step (safelife_env.py:162-170)."""
    Game = R.game.SafeLifeGame
    rng = np.random.default_rng(99)

    def blob_of(name, games_fn, actions, env_kw):
        games = games_fn()
        tr = run_trace_multi(R, games, actions, env_kw)
        blob = {}
        for i, g in enumerate(games_fn()):
            for k, v in level_record(g).items():
                blob["level%d_%s" % (i, k)] = v
            blob["level%d_rng" % i] = words(g._rng.bit_generator)
        blob["n_levels"] = np.array(len(games))
        for k, v in tr.items():
            blob["trace_" + k] = v
        for k, v in env_kw.items():
            blob["env_" + k] = np.array(-1 if v is None else v)
        np.savez_compressed(os.path.join(out, "trace_multi_%s.npz" % name), **blob)
        print("multi-agent trace %-18s steps=%4d episodes=%d sum_reward=%s exits=%d" % (
            name, len(tr["reward"]), len(tr["reset_at"]), tr["reward"].sum(0), int(tr["success"].any(1).sum())))

    train15 = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27)
    for spec, seed, n_levels, env_kw in (
            ("asym1", 5, 3, dict(view_shape=(9, 9), output_channels=None, time_limit=45, should_calculate_side_effects=False)),
            ("build-coop", 6, 2, dict(view_shape=(15, 15), output_channels=train15, time_limit=60,
                                      should_calculate_side_effects=False)),
            ("build-compete", 7, 2, dict(view_shape=(25, 25), output_channels=None, time_limit=50,
                                         should_calculate_side_effects=False))):
        def games_fn(spec=spec, seed=seed, n_levels=n_levels):
            it = R.levels.SafeLifeLevelIterator("random/multi-agent/" + spec, seed=seed, num_workers=0)
            return [next(it) for _ in range(n_levels)]
        n_agents = len(games_fn()[0].agent_locs)
        assert n_agents == 2
        acts = rng.integers(0, 9, (150, n_agents))
        blob_of(spec.replace("-", "_"), games_fn, acts, env_kw)

    # hand-made: exits open from the start (min_performance -1); agent 0 walks right into the exit at step 2 and is
    # done (success) while agent 1 keeps playing until the time limit; then a second episode in which both walk out
    CT = R.game.CellTypes
    b = np.zeros((10, 10), np.uint16)
    b[5, 5] = CT.level_exit
    b[2, 7] = CT.level_exit
    b[5, 3] = CT.player | CT.color_r
    b[2, 2] = CT.player | CT.color_b
    b[7, 7] = CT.life | CT.color_g
    b[7, 8] = CT.life | CT.color_g
    b[8, 7] = CT.life | CT.color_g
    b[8, 8] = CT.life | CT.color_g
    goals = np.zeros_like(b)
    goals[1, 1] = CT.color_g

    def hand_games():
        gs = []
        for k in range(2):
            g = Game(board_size=(10, 10))
            g.board = b.copy()
            g.goals = goals.copy()
            g.agent_locs = np.array([[5, 3], [2, 2]])
            g.min_performance = -1
            g.reset_points_table()
            g.update_exit_locs()
            data = g.serialize()
            gs.append(seeded(Game.loaddata(data), 40 + k))
        return gs
    # actions: 1 up, 2 right, 3 down, 4 left (safelife_game.py action table); 0 = no-op once an agent has left
    ep1 = [[2, 0], [2, 5], [0, 2], [0, 2], [0, 6], [0, 3], [0, 0], [0, 1]]
    ep2 = [[2, 2], [2, 2], [0, 2], [0, 2], [0, 2], [0, 0]]
    blob_of("hand_exit", hand_games, np.array(ep1 + ep2), dict(view_shape=(7, 7), output_channels=None, time_limit=8,
                                                              should_calculate_side_effects=False))


# --------------------------------------------------------------- level pools

def gen_pool(R, out, spec, shape, n, seed, tag):
    """Procgen levels from the reference generator (proc_gen.py) -> one .npz pool."""
    it = R.levels.SafeLifeLevelIterator("random/" + spec, seed=seed, num_workers=0)
    for fd in it.file_data:
        fd[2]["board_shape"] = list(shape)
    recs = []
    for i in range(n):
        game = next(it)
        rec = level_record(game)
        rec["rng"] = words(game._rng.bit_generator)
        rec["required_points"] = np.int64(game.required_points()[0])
        rec["initial_available_points"] = np.int64(game.initial_available_points()[0])
        recs.append(rec)
    blob = {"n_levels": np.array(n), "spec": np.array(spec), "seed": np.array(seed)}
    for k in recs[0]:
        blob[k] = np.stack([np.asarray(r[k]) for r in recs])
    np.savez_compressed(os.path.join(out, "pool_%s.npz" % tag), **blob)
    print("pool %s: %d levels %s" % (tag, n, shape))


def gen_side_effect_inputs(R, out):
    """Occupancy tensors of side_effect_score (side_effects.py:103-113); EMD itself is unpinned."""
    Game = R.game.SafeLifeGame
    sp = R.speedups
    lv = load_level(R, "benchmarks/v0.1/prune-stochastic-1.npz")
    game = seeded(Game.loaddata(lv), 9)
    rng = np.random.default_rng(5)
    for a in rng.integers(0, 9, 40):
        game.execute_actions(int(a))
        game.advance_board()
        game.update_exit_colors()
    glob = np.random.PCG64(4242)
    w0 = words(glob)
    sp.set_bit_generator(glob)
    b0 = game._init_data["board"]
    b1 = sp.advance_board(b0, game.spawn_prob, game.num_steps)
    w1 = words(glob)
    occ0 = sp.life_occupancy(b1, game.spawn_prob, 1000)
    w2 = words(glob)
    occ1 = sp.life_occupancy(game.board, game.spawn_prob, 1000)
    w3 = words(glob)
    np.savez_compressed(os.path.join(out, "side_effect_inputs.npz"),
                        b0=b0, b2=game.board, num_steps=np.array(game.num_steps),
                        spawn_prob=np.array(game.spawn_prob), b1=b1, occ0=occ0, occ1=occ1,
                        rng0=w0, rng1=w1, rng2=w2, rng3=w3)


def gen_nav64(R, out, n):
    """64x64 navigation (BASELINE.json configs[4]): the level pool, one SafeLifeEnv trace on two of its levels, and the
    internals of side_effect_score (side_effects.py:103-113) on an episode played on a third -- the sizes the
    C5 numbers are quoted on.  ~33 s of procgen per level."""
    Game = R.game.SafeLifeGame
    sp = R.speedups
    it = R.levels.SafeLifeLevelIterator("random/navigation", seed=2027, num_workers=0)
    for fd in it.file_data:
        fd[2]["board_shape"] = [64, 64]
    games, recs = [], []
    for i in range(n):
        game = next(it)
        games.append(game)
        rec = level_record(game)
        rec["rng"] = words(game._rng.bit_generator)
        rec["required_points"] = np.int64(game.required_points()[0])
        rec["initial_available_points"] = np.int64(game.initial_available_points()[0])
        recs.append(rec)
        print("  level %d / %d" % (i + 1, n), flush=True)
    blob = {"n_levels": np.array(n), "spec": np.array("navigation"), "seed": np.array(2027)}
    for k in recs[0]:
        blob[k] = np.stack([np.asarray(r[k]) for r in recs])
    np.savez_compressed(os.path.join(out, "pool_navigation_64.npz"), **blob)
    print("pool navigation_64: %d levels" % n)

    # a SafeLifeEnv trace on two of the levels: heads for the exit, so the first episode ends by success
    rng = np.random.default_rng(64)
    datas = [normalize_level(games[k]._init_data) for k in (0, 1)]
    acts = []
    for d in datas:
        acts += greedy_actions(R, Game.loaddata(d), rng, 110, p_random=0.3)
    acts += list(rng.integers(0, 9, 40))
    tr = trace_blob(R, "nav64", datas, [21, 22], np.array(acts), dict(view_shape=(25, 25), output_channels=None,
                                                                     time_limit=150, should_calculate_side_effects=False))
    np.savez_compressed(os.path.join(out, "trace_nav64.npz"), **tr)

    # side_effect_score internals at 64x64 (spawners all over the board: the draws dominate)
    game = seeded(Game.loaddata(normalize_level(games[2]._init_data)), 9)
    for a in rng.integers(0, 9, 60):
        game.execute_actions(int(a))
        game.advance_board()
        game.update_exit_colors()
    glob = np.random.PCG64(6464)
    w0 = words(glob)
    sp.set_bit_generator(glob)
    b0 = game._init_data["board"]
    b1 = sp.advance_board(b0, game.spawn_prob, game.num_steps)
    w1 = words(glob)
    occ0 = sp.life_occupancy(b1, game.spawn_prob, 1000)
    w2 = words(glob)
    occ1 = sp.life_occupancy(game.board, game.spawn_prob, 1000)
    w3 = words(glob)
    np.savez_compressed(os.path.join(out, "side_effect_inputs_64.npz"),
                        b0=b0, b2=game.board, num_steps=np.array(game.num_steps),
                        spawn_prob=np.array(game.spawn_prob), b1=b1, occ0=occ0, occ1=occ1,
                        rng0=w0, rng1=w1, rng2=w2, rng3=w3)
    print("side_effect_inputs_64: %d steps, %d cells ever alive" % (game.num_steps, int((occ1.sum(-1) > 0).sum())))


BULK_STEPS = 100


def bulk_hash(board, goals, rewards, dones, rng_words, agent_loc):
    """64-bit digest of what an env looks like after its action stream (tests/util.py restates it)."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a, dt in ((board, np.uint16), (goals, np.uint16), (rewards, np.float32), (dones, np.uint8),
                  (rng_words, np.uint64), (agent_loc, np.int32)):
        h.update(np.ascontiguousarray(a, dtype=dt).tobytes())
    return np.frombuffer(h.digest(), np.uint64)[0]


def gen_bulk(R, out):
    """Every shipped benchmark level (8 x 100 levels of benchmarks/v1.0, the 31 single-level files of v0.1) under the
    reference's SafeLifeEnv: 100 seeded random actions each, no reset -- an env that finishes keeps stepping, as
    SafeLifeEnv does -- and one 64-bit digest per level of (board, goals, reward stream, done stream, generator
    state, agent location).  The archives themselves (level DATA) are copied next to the digests so that the tests can
    load them through safelife_amd.levels.load_levels, legacy keys and all."""
    import glob
    import shutil
    Game = R.game.SafeLifeGame
    dst_root = os.path.join(out, "levels")
    files, digests, counts = [], [], []
    idx = 0
    for ver in ("v1.0", "v0.1"):
        os.makedirs(os.path.join(dst_root, ver), exist_ok=True)
        for path in sorted(glob.glob(os.path.join(REFERENCE, "safelife/levels/benchmarks", ver, "*.npz"))):
            rel = os.path.join(ver, os.path.basename(path))
            shutil.copyfile(path, os.path.join(dst_root, rel))
            os.chmod(os.path.join(dst_root, rel), 0o644)
            with np.load(path) as d:
                recs = list(d["levels"]) if "levels" in d.files else [{k: d[k] for k in d.files}]
            n = 0
            for rec in recs:
                game = seeded(Game.loaddata(normalize_level(rec)), 5000 + idx)
                env = R.env.SafeLifeEnv(iter([game]), time_limit=1000, view_shape=(15, 15), output_channels=None,
                                        should_calculate_side_effects=False)
                env.reset()
                acts = np.random.default_rng(idx).integers(0, 9, BULK_STEPS)
                rewards, dones = [], []
                for a in acts:
                    _, r, dn, info = env.step(int(a))
                    rewards.append(np.float32(r))
                    dones.append(bool(dn))
                loc = env.game.agent_locs
                digests.append(bulk_hash(env.game.board, env.game.goals, rewards, dones,
                                         words(env.game._rng.bit_generator), loc[0] if len(loc) else [-1, -1]))
                idx += 1
                n += 1
            files.append(rel)
            counts.append(n)
            print("  %-32s %3d levels" % (rel, n), flush=True)
    np.savez_compressed(os.path.join(out, "bulk_levels.npz"), files=np.array(files), counts=np.array(counts),
                        digests=np.array(digests, np.uint64), steps=np.array(BULK_STEPS))
    print("bulk: %d levels in %d files" % (idx, len(files)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nav64", type=int, default=0, help="number of 64x64 navigation levels (slow)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    R = import_reference()
    out = HERE
    todo = args.only.split(",") if args.only else ["primitives", "patterns", "traces", "wrappers", "pools", "side"]
    if "primitives" in todo:
        gen_primitives(R, out)
    if "patterns" in todo:
        gen_patterns(R, out)
    if "traces" in todo:
        gen_env_traces(R, out)
    if "wrappers" in todo:
        gen_wrapper_traces(R, out)
    elif "inaction" in todo:            # only the wrap_inaction_* traces (the others would be rewritten unchanged)
        gen_wrapper_traces(R, out, only_prefix="wrap_inaction_")
    if "side" in todo:
        gen_side_effect_inputs(R, out)
    if "pools" in todo:
        gen_pool(R, out, "prune-still", (25, 25), 96, 2024, "prune_still_25")
        gen_pool(R, out, "append-spawn", (25, 25), 64, 2025, "append_spawn_25")
        gen_pool(R, out, "append-still", (26, 26), 32, 2026, "append_still_26")
    if "multi" in todo:
        gen_multi_agent_traces(R, out)
    if "bulk" in todo:
        gen_bulk(R, out)
    if args.nav64:
        gen_nav64(R, out, args.nav64)


if __name__ == "__main__":
    main()
