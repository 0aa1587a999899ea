"""GPU suite: the HIP path (through the C-ABI of libsafelife_hip.so) against the oracle and the
reference's golden vectors.  Integer work: every comparison is bit-exact.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    from safelife_amd import speedups
    return speedups


@pytest.fixture(scope="module")
def prim():
    with np.load(os.path.join(util.GOLDEN, "primitives.npz")) as d:
        return {k: d[k] for k in d.files}


def _dev_advance(sp, boards, p, n, words):
    import torch
    d_b = sp._to_device(boards, np.uint16)
    d_r = sp._to_device(words, np.uint64)
    d_p = torch.as_tensor(np.broadcast_to(np.asarray(p, np.float32), (len(boards),)).copy()).to(d_b.device)
    out = sp.advance_board_batch(d_b, d_p, d_r, n)
    return sp._to_host(out, np.uint16), sp._to_host(d_r, np.uint64)


# ------------------------------------------------------------------ primitives vs golden vectors

def test_advance_board_golden(sp, prim):
    """Reference-shaped entry point, draws taken from (and returned to) a numpy generator."""
    for k in range(int(prim["n_adv"])):
        p, steps = prim["adv_%03d_p_n" % k]
        bg = np.random.PCG64(0)
        oracle.pcg64_set_state_words(bg, prim["adv_%03d_rng0" % k])
        sp.set_bit_generator(bg)
        src = prim["adv_%03d_in" % k]
        keep = src.copy()
        out = sp.advance_board(src, float(p), int(steps))
        assert out.dtype == np.uint16 and np.array_equal(src, keep)
        assert np.array_equal(out, prim["adv_%03d_out" % k]), k
        assert np.array_equal(oracle.pcg64_state_words(bg), prim["adv_%03d_rng1" % k]), k


def test_alive_counts_golden(sp, prim):
    for k in range(int(prim["n_cnt"])):
        got = sp.alive_counts(prim["cnt_%03d_board" % k], prim["cnt_%03d_goals" % k])
        assert got.dtype == np.int64 and got.shape == (8, 9)
        assert np.array_equal(got, prim["cnt_%03d_out" % k]), k
    with pytest.raises(ValueError, match="same size"):
        sp.alive_counts(np.zeros((3, 3), np.uint16), np.zeros((3, 4), np.uint16))


def test_execute_actions_golden(sp, prim):
    for k in range(int(prim["n_act"])):
        board = prim["act_%03d_board" % k].copy()
        locs = prim["act_%03d_locs" % k].copy()
        assert sp.execute_actions(board, locs, prim["act_%03d_acts" % k]) is None
        assert np.array_equal(board, prim["act_%03d_board_out" % k]), k
        assert np.array_equal(locs, prim["act_%03d_locs_out" % k]), k
    with pytest.raises(ValueError, match="at least 3x3"):
        sp.execute_actions(np.zeros((2, 5), np.uint16), np.zeros((1, 2), np.int64), 1)
    with pytest.raises(ValueError, match="2-dimensional"):
        sp.execute_actions(np.zeros(9, np.uint16), np.zeros((1, 2), np.int64), 1)
    with pytest.raises(ValueError, match="n_agent"):
        sp.execute_actions(np.zeros((5, 5), np.uint16), np.zeros((3, 2), np.int64), [1, 2])


def test_life_occupancy_golden(sp, prim):
    for k in range(int(prim["n_occ"])):
        bg = np.random.PCG64(0)
        oracle.pcg64_set_state_words(bg, prim["occ_%02d_rng0" % k])
        sp.set_bit_generator(bg)
        got = sp.life_occupancy(prim["occ_%02d_in" % k], 0.3, int(prim["occ_%02d_n" % k]))
        assert got.dtype == np.int32
        assert np.array_equal(got, prim["occ_%02d_out" % k]), k
        assert np.array_equal(oracle.pcg64_state_words(bg), prim["occ_%02d_rng1" % k]), k


def test_patterns_known_answers(sp):
    with np.load(os.path.join(util.GOLDEN, "patterns.npz")) as d:
        for name in ("glider", "acorn", "rpentomino", "growth"):
            for n in (1, 4, 20, 100):
                assert np.array_equal(sp.advance_board(d[name + "_in"], 0.3, n), d["%s_n%d" % (name, n)])


@pytest.mark.parametrize("fixture", ["side_effect_inputs.npz", "side_effect_inputs_64.npz"])
def test_side_effect_occupancy_tensors(sp, fixture):
    """Config C5's pinned part: advance(n) + both life_occupancy tensors of side_effect_score
    (side_effects.py:103-113), all on one generator, in the reference's order (25x25 and 64x64)."""
    with np.load(os.path.join(util.GOLDEN, fixture)) as d:
        bg = np.random.PCG64(0)
        oracle.pcg64_set_state_words(bg, d["rng0"])
        sp.set_bit_generator(bg)
        p = float(d["spawn_prob"])
        b1 = sp.advance_board(d["b0"], p, int(d["num_steps"]))
        assert np.array_equal(b1, d["b1"])
        assert np.array_equal(oracle.pcg64_state_words(bg), d["rng1"])
        assert np.array_equal(sp.life_occupancy(b1, p, 1000), d["occ0"])
        assert np.array_equal(sp.life_occupancy(d["b2"], p, 1000), d["occ1"])
        assert np.array_equal(oracle.pcg64_state_words(bg), d["rng3"])


# ------------------------------------------------------------------ primitives vs oracle, seeded

@pytest.mark.parametrize("shape,B", [((25, 25), 1024), ((26, 26), 300), ((64, 64), 64), ((3, 3), 50),
                                     ((15, 15), 130), ((20, 20), 77), ((10, 10), 100), ((25, 25), 5),
                                     ((5, 64), 33), ((64, 5), 33), ((10, 33), 40), ((31, 17), 40),
                                     ((100, 100), 6), ((128, 128), 2),
                                     ((8, 8), 70), ((12, 12), 45), ((16, 16), 37), ((24, 24), 19), ((30, 30), 21),
                                     ((32, 32), 18), ((40, 40), 9), ((48, 48), 7)])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_advance_board_vs_oracle(sp, shape, B, kind):
    rng = np.random.default_rng(hash((shape, kind)) % 2**31)
    boards = util.random_boards(rng, B, shape[0], shape[1], kind)
    words = util.random_rng_words(rng, B)
    p = rng.choice([0.3, 0.0, 1.0, 0.05], B).astype(np.float32)
    for n in (1, 3):
        w_cpu = words.copy()
        want = oracle.advance_board_batch(boards, p, n, w_cpu, n_threads=8)
        got, w_dev = _dev_advance(sp, boards, p, n, words)
        assert np.array_equal(got, want)
        assert np.array_equal(w_dev, w_cpu)


@pytest.mark.parametrize("shape,B,n_step", [((25, 25), 70, 1000), ((26, 26), 33, 300), ((64, 64), 9, 400),
                                            ((20, 20), 10, 100), ((15, 15), 13, 100), ((10, 10), 25, 100),
                                            ((9, 31), 6, 60), ((25, 25), 3, 0), ((8, 8), 40, 50), ((12, 12), 11, 50),
                                            ((16, 16), 9, 50), ((24, 24), 5, 50), ((30, 30), 5, 50), ((32, 32), 6, 50),
                                            ((40, 40), 5, 50), ((48, 48), 3, 50)])
@pytest.mark.parametrize("kind", [0, 1])
def test_life_occupancy_vs_oracle(sp, shape, B, n_step, kind):
    import torch
    rng = np.random.default_rng(hash((shape, kind, n_step)) % 2**31)
    boards = util.random_boards(rng, B, shape[0], shape[1], kind)
    words = util.random_rng_words(rng, B)
    p = rng.choice([0.3, 0.05, 1.0], B).astype(np.float32)
    w_cpu = words.copy()
    want = oracle.life_occupancy_batch(boards, p, n_step, w_cpu, n_threads=8)
    d_rng = sp._to_device(words.copy(), np.uint64)
    got = sp.life_occupancy_batch(sp._to_device(boards, np.uint16), torch.from_numpy(p).to(d_rng.device), d_rng, n_step)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(sp._to_host(d_rng, np.uint64), w_cpu)


@pytest.mark.parametrize("shape,n_step", [((64, 64), 300), ((48, 48), 120), ((40, 40), 60), ((25, 25), 60)])
def test_life_occupancy_colour_classes(sp, shape, n_step):
    """The counters-per-cell classes of the wide occupancy kernels (sl_rowlane.hip, OccGeom): a board takes two, four
    or eight colour slots by the colour bits that VARY between its live and spawner cells.  One batch mixes boards of
    every class -- a single colour, two colours one bit apart, a bit every source has next to varying ones, all eight --
    and walls whose colours belong to no source; more than 255 steps at 64x64 so that the 8-bit counters drain."""
    import torch
    H, W = shape
    rng = np.random.default_rng(1234 + H)
    sets = [[3], [2, 3], [6], [5, 7], [1, 2], [4, 5, 6, 7], [0, 7], [0, 1, 2, 3, 4, 5, 6, 7], [0], [1, 3, 5, 7], [2, 4]]
    boards = np.zeros((2 * len(sets), H, W), np.uint16)
    for k in range(len(boards)):
        cols = np.array(sets[k % len(sets)])
        pick = lambda n: (cols[rng.integers(0, len(cols), n)].astype(np.uint16) << 9)
        b = boards[k].reshape(-1)
        alive = rng.random(H * W) < 0.3
        b[alive] = 9 | pick(int(alive.sum()))
        for _ in range(6):
            b[rng.integers(0, H * W)] = 152 | pick(1)[0]                   # a spawner of one of the board's colours
        for _ in range(5):
            b[rng.integers(0, H * W)] = 17 | pick(1)[0]                    # frozen life: a source as well
        walls = rng.random(H * W) < 0.03
        b[walls] = 16 | (rng.integers(0, 8, int(walls.sum())).astype(np.uint16) << 9)   # any colour: not a source
    B = len(boards)
    words = util.random_rng_words(rng, B)
    p = rng.choice([0.3, 0.05, 1.0], B).astype(np.float32)
    w_cpu = words.copy()
    want = oracle.life_occupancy_batch(boards, p, n_step, w_cpu, n_threads=8)
    d_rng = sp._to_device(words.copy(), np.uint64)
    got = sp.life_occupancy_batch(sp._to_device(boards, np.uint16), torch.from_numpy(p).to(d_rng.device), d_rng, n_step)
    got = got.cpu().numpy()
    for k in range(B):
        assert np.array_equal(got[k], want[k]), (k, sets[k % len(sets)])
    assert np.array_equal(sp._to_host(d_rng, np.uint64), w_cpu)
    assert want.sum() > 0


def test_alive_counts_and_actions_vs_oracle(sp):
    import torch
    rng = np.random.default_rng(11)
    for (H, W, B, A) in ((25, 25, 500, 1), (26, 26, 100, 3), (3, 3, 100, 2), (64, 64, 20, 4)):
        boards = util.random_boards(rng, B, H, W, 0)
        goals = (rng.integers(0, 8, (B, H, W)) << 9).astype(np.uint16)
        got = sp._to_host(sp.alive_counts_batch(sp._to_device(boards, np.uint16),
                                                sp._to_device(goals, np.uint16)), np.int64)
        assert np.array_equal(got, oracle.alive_counts_batch(boards, goals))
        locs = np.stack([rng.integers(0, H, (B, A)), rng.integers(0, W, (B, A))], -1).astype(np.int64)
        for b in range(B):
            for k in range(A):
                if rng.random() < 0.9:
                    boards[b, locs[b, k, 0], locs[b, k, 1]] = rng.choice([122, 122 | 0x200, 122 | 4, 122 | 256])
        acts = rng.integers(0, 9, (B, A)).astype(np.int64)
        b_cpu, l_cpu = boards.copy(), locs.copy()
        oracle.execute_actions_batch(b_cpu, l_cpu, acts)
        d_b, d_l = sp._to_device(boards, np.uint16), sp._to_device(locs, np.int64)
        sp.execute_actions_batch(d_b, d_l, torch.from_numpy(acts).to(d_b.device))
        assert np.array_equal(sp._to_host(d_b, np.uint16), b_cpu)
        assert np.array_equal(sp._to_host(d_l, np.int64), l_cpu)


# ------------------------------------------------------------------ env: traces of the reference

def _device_counts(boards, goals):
    from safelife_amd.levels import _device_counts as f
    return f(boards, goals)


@pytest.mark.parametrize("name", util.trace_names())
def test_env_trace(name):
    """SafeLifeVectorEnv (B=1) replays every recorded step of the reference's SafeLifeEnv; level
    constants come from the device alive_counts kernel (the product's own pool builder)."""
    tr = util.load_trace(name)
    assert util.replay_trace(tr, util.DeviceBackend, _device_counts) == len(tr["trace_reward"])


@pytest.mark.parametrize("name", ["v10_prune-still_open", "v10_navigation", "worked_7x7_exit",
                                  "v10_append-spawn", "pattern_glider_noagent"])
def test_env_trace_terminal_states(name):
    tr = util.load_trace(name)
    assert util.replay_trace_terminal(tr, util.DeviceBackend, _device_counts) >= 1


# ------------------------------------------------------------------ env: batched vs oracle

ENV_STATE = ("board", "goals", "agent_loc", "exit_locs", "rng", "num_steps", "old_value",
             "required_points", "initial_points", "goals_static", "is_active", "episode_reward",
             "episode_length", "level_idx", "episode_idx", "success", "times_up")


@pytest.mark.parametrize("pool_name,B,T,kw", [
    ("prune_still_25", 512, 150, dict(time_limit=60, view_shape=(25, 25),
                                      output_channels=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27))),
    ("append_spawn_25", 512, 150, dict(time_limit=70, view_shape=(15, 15))),
    ("append_still_26", 200, 120, dict(time_limit=50, view_shape=(9, 33), output_channels=None,
                                       remove_white_goals=False)),
    ("navigation_64", 96, 90, dict(time_limit=40, view_shape=(25, 25),
                                   output_channels=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27))),
    ("navigation_64", 40, 60, dict(time_limit=25, view_shape=(64, 64), output_channels=None)),
])
def test_env_batch_vs_oracle(pool_name, B, T, kw):
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 7) % len(pool)
    dev = util.DeviceBackend(pool, B, first_level=first, auto_reset=True, level_stride=5, **kw)
    cpu = util.OracleBackend(pool, B, first_level=first, auto_reset=True, level_stride=5, **kw)
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(3)
    n_done = 0
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2), t
        assert np.array_equal(d1, d2), t
        assert np.array_equal(o1, o2), t
        n_done += int(d1.sum())
        if t % 10 == 0 or t == T - 1:
            for name in ENV_STATE:
                assert np.array_equal(dev.get(name), cpu.get(name)), (t, name)
    assert n_done > B      # every env went through at least one auto-reset on average


TRAINING_WRAPPERS = dict(movement_bonus=0.1, movement_bonus_power=1e-100, movement_bonus_period=4,
                         as_penalty=True, exit_bonus=0.5, penalty_coef=0.3, ignore_reward_cells=False)


@pytest.mark.parametrize("pool_name,B,T,wrappers,kw", [
    ("prune_still_25", 300, 120, TRAINING_WRAPPERS, dict(time_limit=50)),
    ("append_spawn_25", 300, 120, TRAINING_WRAPPERS, dict(time_limit=60)),
    ("append_still_26", 120, 100, dict(movement_bonus=0.25, movement_bonus_power=0.5, movement_bonus_period=3,
                                       as_penalty=False, exit_bonus=1.5, penalty_coef=0.125,
                                       ignore_reward_cells=True), dict(time_limit=40)),
    ("navigation_64", 48, 70, TRAINING_WRAPPERS, dict(time_limit=30)),
    ("prune_still_25", 64, 60, dict(penalty_coef=1.0, movement_bonus=None, exit_bonus=None), dict(time_limit=25)),
    ("prune_still_25", 64, 60, dict(movement_bonus=0.1, movement_bonus_period=8, movement_bonus_power=1.0,
                                    penalty_coef=None, exit_bonus=None), dict(time_limit=25)),
])
def test_env_batch_wrappers_vs_oracle(pool_name, B, T, wrappers, kw):
    """Training-wrapper math fused into the step: float64 shaped reward, bit for bit, every step,
    across auto-resets; then the same through the T-step rollout."""
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 5) % len(pool)
    common = dict(first_level=first, auto_reset=True, level_stride=3, view_shape=(15, 15), wrappers=wrappers, **kw)
    dev = util.DeviceBackend(pool, B, **common)
    cpu = util.OracleBackend(pool, B, **common)
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(13)
    n_done = 0
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        assert np.array_equal(dev.get("shaped_reward"), cpu.get("shaped_reward")), t
        assert np.array_equal(o1, o2), t
        n_done += int(d1.sum())
    assert n_done > B // 2
    assert np.array_equal(dev.get("board"), cpu.get("board"))
    # rollout: T2 more steps in one launch
    T2 = 12
    a = rng.integers(0, 9, (T2, B)).astype(np.int32)
    dev.env.rollout(a)
    want = []
    for t in range(T2):
        cpu.step(a[t])
        want.append(cpu.get("shaped_reward"))
    assert np.array_equal(dev.env.shaped_reward_t.cpu().numpy(), np.stack(want))
    assert np.array_equal(dev.get("board"), cpu.get("board"))


@pytest.mark.parametrize("pool_name,B,T,extra,kw", [
    ("append_spawn_25", 300, 100, dict(), dict(time_limit=40)),
    ("append_spawn_25", 200, 60, dict(), dict(time_limit=30, slices=2)),
    ("prune_still_25", 200, 60, dict(ignore_reward_cells=True, penalty_coef=0.5), dict(time_limit=25)),
    ("navigation_64", 40, 50, dict(), dict(time_limit=30)),
])
def test_env_batch_inaction_baseline_vs_oracle(pool_name, B, T, extra, kw):
    """SimpleSideEffectPenalty(baseline="inaction") in the batch (env_wrappers.py:179-180): every env's baseline
    board advanced once per step with its own generator -- shaped reward float64 bit for bit, baseline boards and
    generators, across auto-resets (a fresh episode's baseline is the board its reset left)."""
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    words = np.random.default_rng(5).integers(0, 2 ** 63, (B, 4), dtype=np.uint64)
    words[:, 3] |= 1                    # a PCG64 increment is odd
    wrappers = dict(TRAINING_WRAPPERS, baseline="inaction", inaction_rng=words, **extra)
    first = (np.arange(B) * 5) % len(pool)
    common = dict(first_level=first, auto_reset=True, level_stride=3, view_shape=(15, 15), wrappers=wrappers, **kw)
    dev = util.DeviceBackend(pool, B, **common)
    cpu = util.OracleBackend(pool, B, **common)
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(14)
    n_done = 0
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        if kw.get("slices", 1) > 1:
            import torch
            a_dev = torch.as_tensor(a, device=dev.env.device)
            dev.env.step_async(a_dev)
            r1, d1 = dev.env.numpy("reward"), dev.env.numpy("done")
        else:
            _, r1, d1 = dev.step(a)
        _, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2), t
        assert np.array_equal(dev.get("shaped_reward"), cpu.get("shaped_reward")), t
        n_done += int(d1.sum())
        if t % 10 == 0 or t == T - 1:
            assert np.array_equal(dev.get("inaction_rng"), cpu.get("inaction_rng")), t
            # (the oracle copies the board into the baseline when it resets an env, the library when the env next
            #  steps: compare the envs that did not just finish)
            keep = ~d1.astype(bool)
            assert np.array_equal(dev.get("inaction_board")[keep], cpu.get("inaction_board")[keep]), t
    assert n_done > B // 2
    assert np.array_equal(dev.get("board"), cpu.get("board"))
    # the baseline advances inside the step kernel: T-step launches and the library's queues carry it too
    import torch
    T2 = 14
    a = rng.integers(0, 9, (T2, B)).astype(np.int32)
    dev.env.rollout(a)
    want = []
    for t in range(T2):
        cpu.step(a[t])
        want.append(cpu.get("shaped_reward"))
    assert np.array_equal(dev.env.shaped_reward_t.cpu().numpy(), np.stack(want))
    assert np.array_equal(dev.get("inaction_rng"), cpu.get("inaction_rng"))
    a = rng.integers(0, 9, (T2, B)).astype(np.int32)
    d_a = torch.from_numpy(a).to(dev.env.device)
    try:
        dev.env.queues_open(2)
    except Exception as e:              # no HSA queue on this box
        print("queues unavailable:", e)
    else:
        dev.env.step_queues_many(d_a)
        for t in range(T2):
            cpu.step(a[t])
        assert np.array_equal(dev.get("shaped_reward"), cpu.get("shaped_reward"))
        assert np.array_equal(dev.get("inaction_rng"), cpu.get("inaction_rng"))
        assert np.array_equal(dev.get("board"), cpu.get("board"))
        dev.env.queues_close()


def test_generic_kernels_with_wrappers():
    """The size-generic kernels carry the same wrapper math: rerun the wrapper traces with the row
    kernels switched off (the switch is read once per process, hence the subprocess)."""
    import subprocess, sys
    env = dict(os.environ, SAFELIFE_HIP_FORCE_GENERIC="1")
    subprocess.check_call([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu",
                           "-k", "test_env_trace and wrap"], env=env, cwd=util.REPO)


def test_rollout_equals_steps():
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=16)
    B, T = 128, 40
    kw = dict(time_limit=25, view_shape=(25, 25), output_channels=None, auto_reset=True)
    a = np.random.default_rng(8).integers(0, 9, (T, B)).astype(np.int32)
    one = util.DeviceBackend(pool, B, first_level=np.arange(B) % 16, **kw)
    many = util.DeviceBackend(pool, B, first_level=np.arange(B) % 16, **kw)
    one.reset()
    many.reset()
    rewards, dones = [], []
    for t in range(T):
        _, r, d = one.step(a[t])
        rewards.append(r)
        dones.append(d)
    r_t, d_t = many.env.rollout(a)
    assert np.array_equal(r_t.cpu().numpy(), np.stack(rewards))
    assert np.array_equal(d_t.cpu().numpy(), np.stack(dones))
    for name in ENV_STATE:
        assert np.array_equal(one.get(name), many.get(name)), name
    assert np.array_equal(one.get("obs"), many.get("obs"))


# ------------------------------------------------------------------ full-size properties (C2/C3)

def test_full_size_translation_invariance_and_batch_independence(sp):
    """8192 x 25 x 25 (BASELINE configs[2] size): a torus has no preferred origin, so rolling every
    board and stepping equals stepping and rolling (RNG-free boards); and a board's result does not
    depend on its position in the batch."""
    rng = np.random.default_rng(21)
    B = 8192
    boards = util.random_boards(rng, B, 25, 25, 0)
    boards[(boards & 128) > 0] = 0                      # no spawners: deterministic
    words = np.zeros((B, 4), np.uint64)
    out, _ = _dev_advance(sp, boards, 0.3, 2, words)
    dy, dx = 7, -3
    rolled, _ = _dev_advance(sp, np.roll(boards, (dy, dx), (1, 2)), 0.3, 2, words)
    assert np.array_equal(np.roll(out, (dy, dx), (1, 2)), rolled)
    perm = rng.permutation(B)
    out_p, _ = _dev_advance(sp, boards[perm], 0.3, 2, words)
    assert np.array_equal(out_p, out[perm])
    sub = rng.choice(B, 64, replace=False)
    want = oracle.advance_board_batch(boards[sub], 0.3, 2, np.zeros((64, 4), np.uint64), n_threads=8)
    assert np.array_equal(out[sub], want)


@pytest.mark.parametrize("pool_name,B,slices,kw", [
    # BASELINE configs[2]: 8192 x 25x25 prune-still, one launch per step and the bench's two slices
    ("prune_still_25", 8192, 1, dict(time_limit=20, view_shape=(25, 25), output_channels=None)),
    ("prune_still_25", 8192, 2, dict(time_limit=20, view_shape=(25, 25), with_obs=False)),
    # configs[3] per-GPU share: 8192 x 25x25 append-spawn (dynamic spawners: every env draws)
    ("append_spawn_25", 8192, 2, dict(time_limit=20, view_shape=(25, 25), with_obs=False)),
    # configs[4] per-GPU share: 4096 x 64x64 navigation
    ("navigation_64", 4096, 2, dict(time_limit=15, view_shape=(25, 25), with_obs=False)),
])
def test_full_size_env_vs_oracle(pool_name, B, slices, kw):
    """The per-GPU shares of BASELINE.json's configs at their full size: 30 steps with time limits short enough
    that every env resets at least once; reward and done of ALL envs every step, then boards, goals, generator
    states, agent locations and episode counters of all envs at the end (and the observation where there is one)."""
    pool, _ = util.pool_from_fixture(pool_name, _device_counts)
    T = 30
    kw = dict(auto_reset=True, level_stride=3, **kw)
    first = np.arange(B) % len(pool)
    dev = util.DeviceBackend(pool, B, first_level=first, slices=slices, **kw)
    cpu = util.OracleBackend(pool, B, first_level=first, **kw)
    dev.env.reset()
    cpu.env.reset()
    rng = np.random.default_rng(4)
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        dev.env.step(a)
        cpu.env.step(a, n_threads=8)
        assert np.array_equal(dev.get("reward"), cpu.get("reward")) and np.array_equal(dev.get("done"), cpu.get("done")), t
    for name in ("board", "goals", "rng", "agent_loc", "episode_idx", "level_idx", "num_steps", "episode_reward"):
        assert np.array_equal(dev.get(name), cpu.get(name)), name
    assert cpu.get("episode_idx").min() >= 1
    if kw.get("with_obs", True):
        assert np.array_equal(dev.get("obs"), cpu.env.obs)


# ------------------------------------------------------------------ compat tier (one env at a time)

def _compat_games(tr):
    from safelife_amd.game import SafeLifeGame
    games = []
    for lv in util.levels_from_trace(tr):
        game = SafeLifeGame.loaddata(lv.as_data())
        bg = np.random.PCG64(0)
        oracle.pcg64_set_state_words(bg, lv.rng_words)
        game._rng = np.random.Generator(bg)
        games.append(game)
    return games


@pytest.mark.parametrize("name", ["c1_append_still_1", "append_still_1_chan19", "v01_append-stochastic-1",
                                  "ex_sokuban", "ex_containment", "v10_prune-still_open", "v10_append-dynamic",
                                  "worked_7x7_exit", "pattern_glider_noagent", "view33_prune_still_2"])
def test_compat_env_trace(name):
    """safelife_amd.env.SafeLifeEnv + game.SafeLifeGame (reference-shaped classes over the GPU
    speedups) reproduce the reference's own SafeLifeEnv step for step, info dict included."""
    from safelife_amd.env import SafeLifeEnv
    tr = util.load_trace(name)
    kw = util.env_kwargs_from_trace(tr)
    env = SafeLifeEnv(iter(_compat_games(tr)), **kw)
    obs = env.reset()
    assert np.array_equal(obs, tr["trace_reset_obs"][0])
    episode = 0
    T = min(len(tr["trace_reward"]), 400)
    for t in range(T):
        obs, reward, done, info = env.step(int(tr["trace_actions"][t]))
        where = "step %d" % t
        assert isinstance(reward, np.float32) and reward == tr["trace_reward"][t], where
        assert bool(done) == bool(tr["trace_done"][t]), where
        assert np.array_equal(obs, tr["trace_obs"][t]), where
        assert np.array_equal(info["board"], tr["trace_board"][t]), where
        assert np.array_equal(info["goals"], tr["trace_goals"][t]), where
        assert bool(info["times_up"]) == bool(tr["trace_times_up"][t]), where
        assert int(info["episode"]["length"]) == int(tr["trace_ep_length"][t]), where
        assert info["episode"]["reward"] == tr["trace_ep_reward"][t], where
        assert bool(info["episode"]["success"]) == bool(tr["trace_success"][t]), where
        if len(info["agent_locs"]):
            assert np.array_equal(info["agent_locs"][0], tr["trace_agent_loc"][t]), where
        if done:
            episode += 1
            if episode >= len(tr["trace_reset_at"]):
                break
            obs = env.reset()
            assert np.array_equal(obs, tr["trace_reset_obs"][episode]), where


MULTI_TRACES = ["multi_asym1", "multi_build_coop", "multi_build_compete", "multi_hand_exit"]


@pytest.mark.parametrize("name", MULTI_TRACES)
def test_compat_env_trace_multi_agent(name):
    """single_agent=False (safelife_env.py:162-170, advance_board.c:217-220): two-agent levels of the reference's own
    multi-agent specs -- distinct colours, flags and points tables per agent -- and a hand-made level on which the agents
    leave through the exit one after the other; every agent's observation, reward, done, episode accumulators and the
    shared board against the reference's SafeLifeEnv, step for step, resets when all agents are done."""
    from safelife_amd.env import SafeLifeEnv
    tr = util.load_trace(name)
    kw = util.env_kwargs_from_trace(tr)
    env = SafeLifeEnv(iter(_compat_games(tr)), single_agent=False, **kw)
    obs = env.reset()
    assert np.array_equal(obs, tr["trace_reset_obs"][0])
    assert np.array_equal(env.game.required_points(), tr["trace_reset_required"][0])
    episode = 0
    for t in range(len(tr["trace_reward"])):
        obs, reward, done, info = env.step(tr["trace_actions"][t].astype(np.int64))
        where = "step %d" % t
        assert reward.dtype == np.float32 and np.array_equal(reward, tr["trace_reward"][t]), where
        assert np.array_equal(np.asarray(done, bool), tr["trace_done"][t]), where
        assert np.array_equal(obs, tr["trace_obs"][t]), where
        assert np.array_equal(info["board"], tr["trace_board"][t]), where
        assert np.array_equal(info["goals"], tr["trace_goals"][t]), where
        assert np.array_equal(info["agent_locs"], tr["trace_agent_locs"][t]), where
        assert bool(info["times_up"]) == bool(tr["trace_times_up"][t]), where
        assert np.array_equal(info["episode"]["length"], tr["trace_ep_length"][t]), where
        assert np.array_equal(info["episode"]["reward"], tr["trace_ep_reward"][t]), where
        assert np.array_equal(np.asarray(info["episode"]["success"], bool), tr["trace_success"][t]), where
        if np.all(done):
            episode += 1
            if episode >= len(tr["trace_reset_at"]):
                break
            obs = env.reset()
            assert np.array_equal(obs, tr["trace_reset_obs"][episode]), where
            assert np.array_equal(env.game.required_points(), tr["trace_reset_required"][episode]), where


@pytest.mark.parametrize("name", MULTI_TRACES)
def test_multi_agent_env_trace(name):
    """slhip_env_step_multi / _reset_multi (SafeLifeMultiAgentVectorEnv): the fused multi-agent step on the device
    against the reference's SafeLifeEnv(single_agent=False) traces, as the oracle's test does on the CPU."""
    tr = util.load_trace(name)
    assert util.replay_trace_multi(tr, util.DeviceMultiBackend, _device_counts) > 0


@pytest.mark.parametrize("B,view,chans", [(96, (9, 9), None), (33, (25, 25), tuple(range(12)) + (25, 26, 27))])
def test_multi_agent_batch_vs_oracle(B, view, chans):
    """A batch of two-agent envs over the reference's multi-agent levels (asym1 / build-coop / build-compete at 26x26:
    distinct points tables, colours and flags per agent), random actions for both agents, reloads inside the step when
    all agents of an env are done, per-episode random streams: device against oracle, every array, every step."""
    levels = []
    for name in ("multi_asym1", "multi_build_coop", "multi_build_compete"):
        levels += util.levels_from_trace(util.load_trace(name))
    from safelife_amd.levels import LevelPool
    pool_d = LevelPool(levels, counts_fn=_device_counts, n_agents=2, min_performance_fraction=0.1)
    pool_c = LevelPool(levels, counts_fn=util.oracle_counts, n_agents=2, min_performance_fraction=0.1)
    for k in ("pool_agent_initial_points", "pool_agent_required_step", "pool_agent_table_idx"):
        assert np.array_equal(getattr(pool_d, k), getattr(pool_c, k))
    kw = dict(first_level=(np.arange(B) * 3) % len(levels), auto_reset=True, level_stride=2, time_limit=17, view_shape=view,
              output_channels=chans)
    dev = util.DeviceMultiBackend(pool_d, B, **kw)
    cpu = util.OracleMultiBackend(pool_c, B, **kw)
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(12)
    for t in range(60):
        acts = rng.integers(0, 9, (B, 2)).astype(np.int32)
        acts[cpu.get("is_active") == 0] = 0             # (training/base_algo.py:216-219: a done agent gets no action)
        od, rd, dd = dev.step(acts)
        oc, rc, dc = cpu.step(acts)
        assert np.array_equal(rd, rc) and np.array_equal(dd, dc), t
        assert np.array_equal(od, oc), t
        for name in util.MULTI_STATE:
            assert np.array_equal(dev.get(name), cpu.get(name)), (t, name)
    assert cpu.get("episode_idx").min() >= 2


@pytest.mark.parametrize("name", ["wrap_train_prune-still", "wrap_train_append-still", "wrap_other_prune-still",
                                  "wrap_se_append-stochastic-1", "wrap_mv_noagent"])
def test_compat_wrappers_trace(name):
    """safelife_amd.env_wrappers stacked over the compat SafeLifeEnv as training/env_factory.py does:
    the reward the outermost wrapper returns equals the reference's, float64 bit for bit."""
    from safelife_amd import env_wrappers as W
    from safelife_amd.env import SafeLifeEnv
    tr = util.load_trace(name)
    env = SafeLifeEnv(iter(_compat_games(tr)), **util.env_kwargs_from_trace(tr))
    if "wrap_movement" in tr:
        bonus, power, period, as_penalty = tr["wrap_movement"]
        env = W.MovementBonusWrapper(env, movement_bonus=float(bonus), movement_bonus_power=float(power),
                                     movement_bonus_period=int(period), as_penalty=bool(as_penalty))
    if "wrap_exit_bonus" in tr:
        env = W.ExtraExitBonus(env, bonus=float(tr["wrap_exit_bonus"]))
    if "wrap_side_effect" in tr:
        coef, ignore = tr["wrap_side_effect"]
        env = W.SimpleSideEffectPenalty(env, penalty_coef=float(coef), ignore_reward_cells=bool(ignore))
    if "min_performance_fraction" in tr:
        env = W.MinPerformanceScheduler(env, min_performance_fraction=float(tr["min_performance_fraction"]))
    obs = env.reset()
    assert np.array_equal(obs, tr["trace_reset_obs"][0])
    episode = 0
    for t in range(len(tr["trace_reward"])):
        obs, reward, done, info = env.step(int(tr["trace_actions"][t]))
        assert isinstance(reward, np.float64) and reward == tr["trace_shaped_reward"][t], t
        assert bool(done) == bool(tr["trace_done"][t]), t
        assert np.array_equal(info["board"], tr["trace_board"][t]), t
        if done:
            episode += 1
            if episode >= len(tr["trace_reset_at"]):
                break
            assert np.array_equal(env.reset(), tr["trace_reset_obs"][episode]), t


def test_policy_obs_layout():
    """policy_obs() == the reference pipeline on the (h, w, c) observation: transpose(-1, -3) and a
    float cast (training/models.py:100-103, ppo.py:64), for the same state."""
    import torch
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=16)
    B = 96
    chans = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27)
    kw = dict(time_limit=30, view_shape=(13, 21), auto_reset=True)
    a = util.DeviceBackend(pool, B, first_level=np.arange(B) % 16, output_channels=chans, **kw)
    b = util.DeviceBackend(pool, B, first_level=np.arange(B) % 16, output_channels=None, **kw)
    a.reset()
    b.reset()
    rng = np.random.default_rng(2)
    for t in range(25):
        acts = rng.integers(0, 9, B).astype(np.int32)
        a.step(acts)
        b.step(acts)
    want = torch.from_numpy(a.get("obs")).transpose(-1, -3)
    for dtype in (torch.uint8, torch.float32):
        got = b.env.policy_obs(chans, dtype=dtype).cpu()
        assert got.shape == (B, len(chans), 21, 13) and got.is_contiguous()
        assert torch.equal(got, want.to(dtype))


def test_side_effect_score_pipeline(sp):
    """side_effects.side_effect_score on the pinned inputs of the reference's side_effect_score: the
    distributions are the golden occupancy tensors / num_samples; the distances come from the LP."""
    from safelife_amd import side_effects as se
    from safelife_amd.cell_types import CellTypes as CT

    class Game(object):
        pass
    with np.load(os.path.join(util.GOLDEN, "side_effect_inputs.npz")) as d:
        game = Game()
        game._init_data = {"board": d["b0"]}
        game.board = d["b2"]
        game.spawn_prob = float(d["spawn_prob"])
        game.num_steps = int(d["num_steps"])
        bg = np.random.PCG64(0)
        oracle.pcg64_set_state_words(bg, d["rng0"])
        sp.set_bit_generator(bg)
        inaction, action = se.side_effect_distributions(game, num_samples=1000, num_runs=1)
        assert np.array_equal(oracle.pcg64_state_words(bg), d["rng3"])
        seen = 0
        for i in range(8):
            key = CT.life + (i << CT.color_bit)
            if d["occ0"][..., i].sum() + d["occ1"][..., i].sum() > 0:
                assert np.array_equal(inaction[key], d["occ0"][..., i] / 1000)
                assert np.array_equal(action[key], d["occ1"][..., i] / 1000)
                seen += 1
            else:
                assert key not in inaction
        assert seen >= 1
        oracle.pcg64_set_state_words(bg, d["rng0"])
        sp.set_bit_generator(bg)
        scores = se.side_effect_score(game, strkeys=True)
        assert scores and all(k.rpartition("-")[2] in ("gray", "red", "green", "blue", "yellow", "magenta", "cyan", "white")
                              for k in scores)
        for dist, mass in scores.values():
            assert dist >= 0 and mass >= 0


@pytest.mark.parametrize("fixture", ["side_effect_inputs.npz", "side_effect_inputs_64.npz"])
def test_side_effect_pass_reproduces_reference_inputs(sp, fixture):
    """slhip_side_effects (the batched episode-end pass) on a queue of 80 entries that all replay a pinned case
    of the reference's side_effect_score (tests/golden/side_effect_inputs*.npz: starting board, final board,
    episode length, generator state -- a 25x25 benchmark level, and a 64x64 navigation level, the size C5's
    number is quoted on: compact counters, pre-roll and the two-run launch at W = 64): roll-forward, both
    occupancy tensors and the distributions of side_effects.py:111-130, every entry -- with a device-side entry
    count below the capacity."""
    import ctypes as C
    import torch
    from safelife_amd import _hip, side_effects as se
    from safelife_amd.levels import Level, LevelPool
    from safelife_amd.vector_env import SafeLifeVectorEnv
    with np.load(os.path.join(util.GOLDEN, "side_effect_inputs.npz")) as d:
        d = {k: d[k] for k in d.files}
    p, n_steps = float(d["spawn_prob"]), int(d["num_steps"])
    filler = Level(np.zeros_like(d["b0"]), agent_locs=np.zeros((0, 2), int))
    start = Level(d["b0"], agent_locs=np.zeros((0, 2), int), spawn_prob=p)
    pool = LevelPool([filler, start], counts_fn=_device_counts)
    env = SafeLifeVectorEnv(pool, 8, with_obs=False)
    dev, cap, n = env.device, 96, 80
    H, W = d["b0"].shape
    rec = np.zeros((cap, 8), np.int32)
    rec[:n, 0] = np.arange(n)
    rec[:n, 1] = 1
    rec[:n, 2] = n_steps
    rec[:n, 4] = np.float32(p).view(np.int32)
    bufs = dict(count=torch.tensor([n], dtype=torch.int32, device=dev), records=torch.from_numpy(rec).to(dev),
                boards=torch.from_numpy(np.broadcast_to(d["b2"], (cap, H, W)).copy().view(np.int16)).to(dev))
    q = _hip.EpisodeQueue()
    q.capacity, q.env_base = cap, 0
    q.count, q.records, q.boards = (bufs[k].data_ptr() for k in ("count", "records", "boards"))
    K = _hip.SL_SE_MAX_KEYS
    out = dict(work_boards=torch.zeros((2 * cap, H, W), dtype=torch.int16, device=dev),
               work_prob=torch.zeros(2 * cap, dtype=torch.float32, device=dev),
               work_steps=torch.zeros(2 * cap, dtype=torch.int32, device=dev),
               work_rng=sp._to_device(np.broadcast_to(d["rng0"], (2 * cap, 4)).copy(), np.uint64),
               counts=torch.zeros((2, cap, H, W, 8), dtype=torch.int32, device=dev),
               keys=torch.zeros((cap, K), dtype=torch.int16, device=dev),
               life_dist=torch.zeros((cap, 2, 8, H, W), dtype=torch.float64, device=dev),
               type_masks=torch.zeros((cap, 2, K - 8, H, W), dtype=torch.uint8, device=dev))
    rc = _hip.lib().slhip_side_effects(env._sref, C.byref(q), 1000, 0,
                                       *[_hip.ptr(out[k]) for k in ("work_boards", "work_prob", "work_steps", "work_rng",
                                                                    "counts", "keys", "life_dist", "type_masks")],
                                       _hip.current_stream_ptr())
    _hip.check(rc)
    b1 = out["work_boards"].cpu().numpy().view(np.uint16)
    counts = out["counts"].cpu().numpy()
    after = sp._to_host(out["work_rng"], np.uint64)
    assert np.array_equal(b1[:n], np.broadcast_to(d["b1"], (n, H, W)))
    assert np.array_equal(counts[0, :n], np.broadcast_to(d["occ0"], (n, H, W, 8)))
    assert np.array_equal(counts[1, :n], np.broadcast_to(d["occ1"], (n, H, W, 8)))
    assert np.array_equal(after[:n], np.broadcast_to(d["rng3"], (n, 4)))
    assert np.array_equal(after[n:], np.broadcast_to(d["rng0"], (2 * cap - n, 4)))     # entries past the count: untouched
    # the distributions against the host restatement of side_effects.py:111-130 on the golden tensors
    want_in, want_act = se.distributions_from_counts(d["b0"], d["b2"], np.stack([d["occ0"], d["occ1"]]), 1000)
    batch = type("B", (), {})()
    from safelife_amd.vector_env import SideEffectBatch
    batch = SideEffectBatch(env, bufs, out, 1000)
    assert len(batch) == n
    for i in (0, 37, n - 1):
        got_in, got_act = batch.distributions(i)
        assert set(got_in) == set(int(k) for k in want_in)
        for k in want_in:
            assert np.array_equal(got_in[int(k)], want_in[k]) and np.array_equal(got_act[int(k)], want_act[k]), (i, k)
    assert (out["keys"].cpu().numpy().view(np.uint16)[n:] == 0xFFFF).all()
    assert isinstance(batch.scores(0), dict)


@pytest.mark.parametrize("pool_name,n_levels,B,time_limit,capacity,n_samples", [
    ("append_spawn_25", 12, 128, 11, 600, 60),
    ("navigation_64", 32, 64, 9, 300, 40),         # C5's shape: 64-wide rows, ~70 spawners per level
])
def test_side_effect_queue_end_to_end(sp, pool_name, n_levels, B, time_limit, capacity, n_samples):
    """auto_reset=True, short episodes: the step kernels queue every finished episode (record + the
    board as the agent left it, before the reset reloads the slot); side_effects_flush() runs the pass without
    any host read.  Checked: the queue against a replay on the oracle (which episode ended when, with which
    board), and every entry's occupancy tensors against the one-board primitives under the entry's own stream."""
    from safelife_amd import side_effects as se
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, n=n_levels, min_performance_fraction=0.05)
    T = 34
    first = np.arange(B) % len(pool)
    kw = dict(first_level=first, auto_reset=True, level_stride=1, time_limit=time_limit, view_shape=(9, 9))
    dev = util.DeviceBackend(pool, B, slices=2, side_effects=dict(capacity=capacity, num_samples=n_samples), **kw)
    cpu = util.OracleBackend(pool, B, **kw)
    dev.reset(), cpu.reset()
    rng = np.random.default_rng(31)
    want = {}                      # (env, episode_idx) -> (level, num_steps, final board)
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        before = {k: cpu.get(k) for k in ("level_idx", "episode_idx", "is_active")}
        boards_before_reset = None
        cpu.env.s.auto_reset = 0           # look at the terminal boards, then let the oracle reset
        _, _, d2 = cpu.step(a)
        boards_before_reset = cpu.get("board")
        steps_now = cpu.get("num_steps")
        for e in np.nonzero(d2 & (before["is_active"] != 0))[0]:
            want[(int(e), int(before["episode_idx"][e]))] = (int(before["level_idx"][e]), int(steps_now[e]),
                                                            boards_before_reset[e].copy())
        cpu.env.s.auto_reset = 1
        for e in np.nonzero(d2)[0]:        # the oracle's auto-reset, by hand
            cpu.arrays["level_idx"][e] = (cpu.arrays["level_idx"][e] + 1) % len(pool)
            cpu.arrays["episode_idx"][e] += 1
        if d2.any():
            m = d2.astype(np.uint8)
            cpu.arrays["loaded"][m != 0] = 0
            cpu.env.reset(m)
        _, _, d1 = dev.step(a)
        assert np.array_equal(d1, d2), t
    batch = dev.env.side_effects_flush()
    recs = batch.records()
    assert len(batch) == len(want) and len(want) >= 3 * B
    boards = batch.boards.cpu().numpy().view(np.uint16)
    counts = batch.counts.cpu().numpy()
    pool_rng = pool.arrays()["pool_rng"]

    def mix64(z):
        m = (1 << 64) - 1
        z = (z + 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return z ^ (z >> 31)
    for i in range(len(batch)):
        key = (int(recs["env"][i]), int(recs["episode_idx"][i]))
        level, steps, board = want.pop(key)
        assert (int(recs["level"][i]), int(recs["num_steps"][i])) == (level, steps), key
        assert np.array_equal(boards[i], board), key
        if i % 9 == 0:             # the pass itself, entry by entry, under the entry's derived stream
            words = pool_rng[level].copy()          # run 0: roll-forward + inaction tensor on the entry's first stream
            a_ = mix64((((0x5EFFEC75 ^ key[0]) & 0xFFFFFFFF) << 32) | key[1])
            words[0] ^= np.uint64(a_)
            words[1] ^= np.uint64(mix64(a_))
            bg = np.random.PCG64(0)
            oracle.pcg64_set_state_words(bg, words)
            sp.set_bit_generator(bg)
            lv = pool.levels[level]
            c0 = sp.life_occupancy(sp.advance_board(lv.board, lv.spawn_prob, steps), lv.spawn_prob, n_samples)
            a_ = mix64((((0x2B0A2D5 ^ key[0]) & 0xFFFFFFFF) << 32) | key[1])     # run 1: the action tensor on its second
            words[0] ^= np.uint64(a_)
            words[1] ^= np.uint64(mix64(a_))
            oracle.pcg64_set_state_words(bg, words)
            c1 = sp.life_occupancy(board, lv.spawn_prob, n_samples)
            assert np.array_equal(counts[0, i], c0) and np.array_equal(counts[1, i], c1), key
    assert not want
    assert len(dev.env.side_effects_flush()) == 0          # the fresh queue starts empty


def test_side_effects_flush_overlapped_and_deferred():
    """side_effects_flush(overlap=True) -- the pass on the env's side stream, later steps not waiting for it -- and
    defer=True + side_effects_launch() -- the launch itself left until the caller has put further steps in front of it --
    return what the plain flush returns (records, boards, counts, keys, distributions), and the steps that went in
    between are the same steps: two envs, same pool, same actions, one flushed each way."""
    import torch
    from safelife_amd.vector_env import SafeLifeVectorEnv
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=6, min_performance_fraction=0.05)
    B, T = 40, 31
    kw = dict(auto_reset=True, level_stride=1, time_limit=7, view_shape=(9, 9), slices=2,
              side_effects=dict(capacity=256, num_samples=40))
    envs = [SafeLifeVectorEnv(pool, B, **kw) for _ in range(3)]
    for e in envs:
        e.reset()
    rng = np.random.default_rng(8)
    acts = torch.from_numpy(rng.integers(0, 9, (T + 12, B)).astype(np.int32)).to(envs[0].device)

    def same(a, b, what):
        assert len(a) == len(b) and len(a) > 0, what
        ra, rb = a.records(), b.records()
        order_a = np.lexsort((ra["episode_idx"], ra["env"]))
        order_b = np.lexsort((rb["episode_idx"], rb["env"]))
        for k in ("env", "episode_idx", "level", "num_steps"):
            assert np.array_equal(ra[k][order_a], rb[k][order_b]), (what, k)
        n = len(a)
        for name in ("boards", "keys", "life_dist"):
            x = getattr(a, name).cpu().numpy()[:n][order_a]
            y = getattr(b, name).cpu().numpy()[:n][order_b]
            assert np.array_equal(x, y), (what, name)
        ca, cb = a.counts.cpu().numpy(), b.counts.cpu().numpy()
        assert np.array_equal(ca[:, :n][:, order_a], cb[:, :n][:, order_b]), (what, "counts")

    for t in range(T):
        for e in envs:
            e.step_async(acts[t])
    plain = envs[0].side_effects_flush()
    over = envs[1].side_effects_flush(overlap=True)
    late = envs[2].side_effects_flush(overlap=True, defer=True)
    for t in range(T, T + 6):
        for e in envs:
            e.step_async(acts[t])
    envs[2].side_effects_launch()
    for t in range(T + 6, T + 12):
        for e in envs:
            e.step_async(acts[t])
    same(plain, over, "overlapped")
    same(plain, late, "deferred")
    # the second window: the deferred one is never launched by hand -- the flush / the accessors do it
    p2 = envs[0].side_effects_flush()
    l2 = envs[2].side_effects_flush(overlap=True, defer=True)
    same(p2, l2, "second window")
    for name in ("board", "rng"):
        assert torch.equal(envs[0].t[name], envs[2].t[name]), name


def test_vector_env_side_effect_occupancy(sp):
    """Batched device pipeline of side_effect_score for finished episodes == the one-env pipeline
    (advance_board(b0, n) -> life_occupancy x2 under one generator), env by env, generator state included."""
    import torch
    from safelife_amd import side_effects as se
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=8)
    B = 6
    dev = util.DeviceBackend(pool, B, first_level=np.arange(B), auto_reset=False, time_limit=9 + 4,
                             view_shape=(15, 15), output_channels=None)
    dev.reset()
    rng = np.random.default_rng(5)
    for t in range(13):
        _, _, done = dev.step(rng.integers(0, 9, B).astype(np.int32))
    assert done.all()
    words = util.random_rng_words(rng, B)
    d_rng = sp._to_device(words.copy(), np.uint64)
    occ0, occ1, b0, b2 = dev.env.side_effect_occupancy(np.arange(B), d_rng, num_samples=200)
    after = sp._to_host(d_rng, np.uint64)
    boards, levels, steps = dev.get("board"), dev.get("level_idx"), dev.get("num_steps")
    for e in range(B):
        bg = np.random.PCG64(0)
        oracle.pcg64_set_state_words(bg, words[e])
        sp.set_bit_generator(bg)
        lv = pool.levels[levels[e]]
        c0, c1 = se.occupancy_pair(lv.board, boards[e], lv.spawn_prob, int(steps[e]), 200)
        assert np.array_equal(occ0[e].cpu().numpy(), c0) and np.array_equal(occ1[e].cpu().numpy(), c1), e
        assert np.array_equal(oracle.pcg64_state_words(bg), after[e]), e
    scores = se.side_effect_score_from_counts(b0[0].cpu().numpy(), b2[0].cpu().numpy(), occ0[0].cpu().numpy(),
                                              occ1[0].cpu().numpy(), 200, strkeys=True)
    assert isinstance(scores, dict)


# ------------------------------------------------------------------ edges: empty, maximal, ragged

def test_empty_batches_and_size_limits(sp):
    """B = 0 is a no-op for every primitive; H*W = 16384 is the largest board; beyond it, and below
    3x3, the C-ABI reports SL_E_SHAPE, which the Python layer turns into the reference's ValueError."""
    import torch
    from safelife_amd import _hip
    dev = _hip.device()
    z16 = torch.zeros((0, 25, 25), dtype=torch.int16, device=dev)
    zf = torch.zeros((0,), dtype=torch.float32, device=dev)
    zr = torch.zeros((0, 4), dtype=torch.int64, device=dev)
    assert sp.advance_board_batch(z16, zf, zr, 3).shape == (0, 25, 25)
    assert sp.life_occupancy_batch(z16, zf, zr, 5).shape == (0, 25, 25, 8)
    assert sp.alive_counts_batch(z16, z16).shape == (0, 8, 9)
    rng = np.random.default_rng(3)
    big = util.random_boards(rng, 2, 128, 128, 1)                      # 16384 cells
    words = util.random_rng_words(rng, 2)
    w_cpu = words.copy()
    want = oracle.advance_board_batch(big, 0.3, 2, w_cpu, n_threads=2)
    got, w_dev = _dev_advance(sp, big, 0.3, 2, words)
    assert np.array_equal(got, want) and np.array_equal(w_dev, w_cpu)
    too_big = torch.zeros((1, 128, 129), dtype=torch.int16, device=dev)
    one = torch.full((1,), 0.3, dtype=torch.float32, device=dev)
    r1 = torch.zeros((1, 4), dtype=torch.int64, device=dev)
    with pytest.raises(ValueError):
        sp.advance_board_batch(too_big, one, r1, 1)
    with pytest.raises(ValueError):
        sp.advance_board_batch(torch.zeros((1, 2, 9), dtype=torch.int16, device=dev), one, r1, 1)
    with pytest.raises(ValueError):
        sp.advance_board(np.zeros((2, 2), np.uint16))


def test_ragged_tail_workgroups_and_many_exits():
    """Batch sizes that leave partly filled workgroups / waves in the row kernels (8 boards per
    workgroup, 2 per wave), and a level with more exit cells than the row kernels take (E > 8 runs the
    size-generic kernels): both against the oracle."""
    from safelife_amd.levels import Level, LevelPool
    from safelife_amd.cell_types import CellTypes as CT
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=12)
    for B in (1, 2, 7, 9, 15, 17):
        kw = dict(first_level=np.arange(B) % 12, auto_reset=True, time_limit=9, view_shape=(25, 25),
                  output_channels=None, level_stride=5)
        dev, cpu = util.DeviceBackend(pool, B, **kw), util.OracleBackend(pool, B, **kw)
        assert np.array_equal(dev.reset(), cpu.reset())
        rng = np.random.default_rng(B)
        for t in range(25):
            a = rng.integers(0, 9, B).astype(np.int32)
            o1, r1, d1 = dev.step(a)
            o2, r2, d2 = cpu.step(a)
            assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(o1, o2), (B, t)
        for name in ENV_STATE:
            assert np.array_equal(dev.get(name), cpu.get(name)), (B, name)
    board = np.zeros((25, 25), np.uint16)
    board[12, 12] = CT.player
    for k in range(11):
        board[3, 2 * k + 1] = CT.level_exit
    board[8:10, 8:10] = CT.life
    many = LevelPool([Level(board, np.zeros_like(board), np.array([[12, 12]]), min_performance=-1)],
                     counts_fn=_device_counts)
    assert many.exit_slots == 11
    kw = dict(first_level=0, auto_reset=True, time_limit=30, view_shape=(9, 9), output_channels=None)
    dev, cpu = util.DeviceBackend(many, 5, **kw), util.OracleBackend(many, 5, **kw)
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(1)
    for t in range(60):
        a = rng.integers(0, 9, 5).astype(np.int32)
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(o1, o2), t
    assert np.array_equal(dev.get("board"), cpu.get("board"))


@pytest.mark.parametrize("pool_name,B,slices,kw", [
    ("prune_still_25", 1000, 3, dict(time_limit=40, view_shape=(25, 25),
                                     output_channels=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27))),
    ("append_spawn_25", 700, 2, dict(time_limit=50, view_shape=(15, 15), output_channels=None,
                                     wrappers=TRAINING_WRAPPERS)),
    ("navigation_64", 200, 4, dict(time_limit=30, view_shape=(25, 25))),
    ("append_still_26", 130, 8, dict(time_limit=30, view_shape=(9, 9))),       # more slices asked than 64-env blocks
])
def test_sliced_stepping_vs_oracle(pool_name, B, slices, kw):
    """slhip_env_step_slices: the batch cut into slices, one launch per slice on its own stream -- the
    drop-in step() (fenced on both sides) every step against the oracle, then a run of step_async() calls
    (nothing fenced until join()) against the same oracle steps."""
    import torch
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 3) % len(pool)
    common = dict(first_level=first, auto_reset=True, level_stride=5, **kw)
    dev = util.DeviceBackend(pool, B, slices=slices, **common)
    cpu = util.OracleBackend(pool, B, **common)
    env = dev.env
    assert env.slices == min(slices, (B + 63) // 64) and env.slice_bounds[-1] == B
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(23)
    for t in range(45):
        a = rng.integers(0, 9, B).astype(np.int32)
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(o1, o2), t
    acts = rng.integers(0, 9, (40, B)).astype(np.int32)
    d_acts = torch.from_numpy(acts).to(env.device)
    torch.cuda.synchronize()                    # the actions are complete: nothing to fence per step
    for t in range(40):
        env.step_async(d_acts[t])
        cpu.step(acts[t])
    env.join()
    for name in ENV_STATE:
        assert np.array_equal(dev.get(name), cpu.get(name)), name
    assert np.array_equal(env.numpy("obs"), cpu.env.obs)
    if "wrappers" in kw:
        assert np.array_equal(env.shaped_reward.cpu().numpy(), cpu.env.wa["shaped_reward"])


@pytest.mark.parametrize("pool_name,B", [("navigation_64", 200), ("append_spawn_25", 1500)])
def test_sliced_stepping_soak(pool_name, B):
    """Slices of one batch run as concurrent launches, so from step to step an env's workgroup may land on
    another XCD (with one launch per step the placement never changes): repeated unfenced runs of 2, 3 and 4
    slices against the one-launch env on the same actions -- boards, generators and episode state.
    (Write-through board stores failed exactly this on 64x64 spawner levels.)"""
    import torch
    from safelife_amd.vector_env import SafeLifeVectorEnv
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 3) % len(pool)
    kw = dict(first_level=first, auto_reset=True, level_stride=5, time_limit=30, view_shape=(15, 15), with_obs=False)
    T = 50
    acts = torch.from_numpy(np.random.default_rng(5).integers(0, 9, (T, B)).astype(np.int32)).to("cuda")
    whole = SafeLifeVectorEnv(pool, B, **kw)
    whole.reset()
    for t in range(T):
        whole.step_async(acts[t])
    want = {name: whole.numpy(name) for name in ("board", "goals", "rng", "agent_loc", "episode_idx", "level_idx",
                                                 "episode_reward", "num_steps")}
    for trial in range(3):
        for n in (2, 3, 4):
            env = SafeLifeVectorEnv(pool, B, slices=n, **kw)
            env.reset()
            torch.cuda.synchronize()
            for t in range(T):
                env.step_async(acts[t])
            env.join()
            for name, ref in want.items():
                assert np.array_equal(env.numpy(name), ref), (trial, n, name)


def _queues_or_skip(env, slices=None, release_free=False, queue_ids=None, recover=True):
    from safelife_amd._hip import SafeLifeHipError
    try:
        env.queues_open(slices, release_free=release_free, queue_ids=queue_ids, recover=recover)
    except SafeLifeHipError as e:           # no HSA queue to be had (not an MI355X box as the driver's): say why
        pytest.skip("AQL queues unavailable: %s" % e)
    if release_free and not env.queue_release_free:
        pytest.skip("release-free stepping not granted on this box: %s" % env.queue_mode_note)


@pytest.mark.parametrize("release_free", [False, True], ids=["agent-fences", "release-free"])
@pytest.mark.parametrize("pool_name,B,queue_slices,kw", [
    ("prune_still_25", 700, 1, dict(time_limit=12, view_shape=(9, 9))),
    ("append_spawn_25", 1500, 4, dict(time_limit=20, view_shape=(25, 25), output_channels=tuple(range(15)))),
    ("navigation_64", 300, 2, dict(time_limit=15, view_shape=(15, 15), with_obs=False)),
    ("prune_still_25", 333, 3, dict(time_limit=9, view_shape=(5, 7), wrappers=TRAINING_WRAPPERS)),
    # slices on chosen queues (slhip_queues_open_on): what a driver does that leaves out the queue its exchange holds up
    ("prune_still_25", 900, (3, 0, 2), dict(time_limit=14, view_shape=(9, 9), with_obs=False)),
    # the headline batch itself: all 8192 envs of C3 on four queues against the oracle, every step of the runs
    ("prune_still_25", 8192, 4, dict(time_limit=40, view_shape=(25, 25), with_obs=False)),
])
def test_queue_stepping_vs_oracle(pool_name, B, queue_slices, kw, release_free):
    """slhip_queues_*: the slices of a step dispatched from the library's own AQL queues instead of HIP streams (same
    kernel, found in HIP's executables; barrier bit + agent-scope fences for ordering, or -- opt-in -- no release
    between steps).  A synchronised step at a time against the oracle (reward, done), then runs of unsynchronised steps
    -- one call per step, and many steps per call (slhip_queues_steps) -- against the same oracle steps, with a reset
    through a HIP stream in the middle (the next queue step takes a system-scope acquire)."""
    import torch
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 3) % len(pool)
    common = dict(first_level=first, auto_reset=True, level_stride=5, **kw)
    dev = util.DeviceBackend(pool, B, **common)
    cpu = util.OracleBackend(pool, B, **common)
    env = dev.env
    if isinstance(queue_slices, tuple):
        _queues_or_skip(env, None, release_free, queue_ids=list(queue_slices))
        assert env.queue_ids == list(queue_slices) and env.queue_slices == len(queue_slices)
    else:
        _queues_or_skip(env, queue_slices, release_free)
    assert env.queue_release_free == release_free
    dev.env.reset()
    cpu.env.reset()
    rng = np.random.default_rng(31)
    acts = rng.integers(0, 9, (130, B)).astype(np.int32)
    d_acts = torch.from_numpy(acts).to(env.device)
    torch.cuda.synchronize()
    t = 0
    for _ in range(25):
        env.step_queues(d_acts[t])
        cpu.env.step(acts[t])
        env.queues_sync()
        assert np.array_equal(env.numpy("reward"), cpu.get("reward")) and np.array_equal(env.numpy("done"), cpu.get("done")), t
        t += 1
    for run in (40, 5, 60):
        if run == 5:
            for _ in range(run):
                env.step_queues(d_acts[t])
                cpu.env.step(acts[t])
                t += 1
        else:                                   # one call for the whole run
            if run == 40:
                # staged, then released (slhip_queues_stage / _go), and -- nothing but waits and reads since the last sync --
                # without the first step's system-scope acquire
                env.step_queues_many(d_acts[t:t + run], assume_ordered="untouched", defer=True)
                env.queues_go()
            else:
                env.step_queues_many(d_acts[t:t + run])
            for _ in range(run):
                cpu.env.step(acts[t])
                t += 1
        for name in ENV_STATE:
            assert np.array_equal(dev.get(name), cpu.get(name)), (run, name)
        if kw.get("with_obs", True):
            assert np.array_equal(env.numpy("obs"), cpu.env.obs)
        if "wrappers" in kw:
            assert np.array_equal(env.shaped_reward.cpu().numpy(), cpu.env.wa["shaped_reward"])
        if run == 5:                            # a reset on the caller's stream: the queues start again behind it
            mask = (np.arange(B) % 3 == 0).astype(np.uint8)
            dev.env.reset(mask)
            cpu.env.reset(mask)
    assert cpu.get("episode_idx").min() >= 1


@pytest.mark.parametrize("compact", [False, True], ids=["16-byte records", "8-byte records"])
def test_queue_steps_write_one_record_set_per_step(compact):
    """slhip_queues_steps with out_stride: step t's sl_step_out records land in slot t of a caller-owned window (what
    sharding.RewardGather hands to RCCL), against the oracle's reward / done of every step.  compact: the 8-byte
    record -- reward + flags, sl_env_batch.out_compact -- that halves what a window carries to rank 0; also through
    one launch per step on a stream, and through the size-generic kernels."""
    import torch
    B, T = 512, 24
    W = 2 if compact else 4
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts, min_performance_fraction=0.05)
    common = dict(first_level=np.arange(B) % len(pool), auto_reset=True, level_stride=3, time_limit=10, view_shape=(9, 9),
                  with_obs=False)
    dev = util.DeviceBackend(pool, B, **common)
    cpu = util.OracleBackend(pool, B, **common)
    env = dev.env
    _queues_or_skip(env, 3)
    env.reset()
    cpu.env.reset()
    acts = np.random.default_rng(8).integers(0, 9, (T, B)).astype(np.int32)
    d_acts = torch.from_numpy(acts).to(env.device)
    window = torch.full((T + 1, B, W), -1, dtype=torch.int32, device=env.device)
    torch.cuda.synchronize()
    env.set_step_outputs(window.data_ptr(), compact=compact)
    env.step_queues_many(d_acts[:T - 4], out_stride=B)
    env.queues_sync()
    for t in range(T - 4, T):           # the last steps one launch at a time on the caller's stream, same records
        env.set_step_outputs(window[t].data_ptr(), compact=compact)
        env.step_async(d_acts[t])
    env.join()
    env.set_step_outputs(None)
    rec = window.cpu().numpy()
    assert (rec[T] == -1).all()         # nothing written past the last step's slot
    for t in range(T):
        cpu.env.step(acts[t])
        assert np.array_equal(rec[t, :, 0].view(np.float32), cpu.get("reward")), t
        assert np.array_equal(rec[t, :, 1].astype(np.uint32) & 0xFF, cpu.get("done").astype(np.uint32)), t
    assert env.struct.out_compact == 0


def test_queues_stream_shares_probe():
    """slhip_queues_stream_shares: a long one-wavefront kernel on a HIP stream against a one-workgroup dispatch on each of
    the library's queues -- returns a mask over the queues asked about (on MI355X such a kernel holds none of them up:
    it is RCCL's exchange kernel that does, which slhip_gather_stream_shares measures; tests/rccl_gather_check.py)."""
    import ctypes as C
    import torch
    from safelife_amd import _hip
    lib = _hip.lib()
    mask = C.c_int(-1)
    rc = lib.slhip_queues_stream_shares(4, C.c_void_p(torch.cuda.Stream().cuda_stream), C.byref(mask))
    if rc == _hip.SL_E_UNSUPPORTED:
        pytest.skip("AQL queues unavailable: %s" % lib.slhip_last_error().decode())
    assert rc == 0, lib.slhip_last_error()
    assert 0 <= mask.value < 16
    assert lib.slhip_queues_stream_shares(0, None, C.byref(mask)) != 0            # argument errors are errors


@pytest.mark.gpu
def test_queue_stepping_refuses_a_planted_placement_record():
    """Release-free queue stepping (opt-in) rests on workgroup i of a slice's queue always running on the same XCD;
    every workgroup of every step compares where it runs with where the probe at open found that index and raises a
    flag when it differs.  Told to expect slice 0 one XCD further on (self-test hook) the next sync must refuse, and
    keep refusing."""
    import torch
    from safelife_amd import _hip
    from safelife_amd.vector_env import SafeLifeVectorEnv
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts, n=8, min_performance_fraction=0.05)
    B = 256
    acts = torch.zeros((B,), dtype=torch.int32, device="cuda")
    kw = dict(auto_reset=True, time_limit=20, view_shape=(9, 9), with_obs=False)
    env = SafeLifeVectorEnv(pool, B, **kw)
    _queues_or_skip(env, 2, release_free=True, recover=False)
    env.reset()
    env.step_queues(acts)
    env.queues_sync()                       # an honest record: nothing to refuse
    _hip.check(env._lib.slhip_queues_selftest(env._queues, _hip.QUEUES_SELFTEST_PLANT, 0))
    env.step_queues(acts)
    with pytest.raises(_hip.SafeLifeHipError, match="another XCD"):
        env.queues_sync()
    env.step_queues(acts)
    with pytest.raises(_hip.SafeLifeHipError, match="another XCD"):
        env.queues_sync()
    env.queues_close()
    # the default mode has no record and nothing to plant
    env = SafeLifeVectorEnv(pool, B, **kw)
    env.queues_open(2)
    assert not env.queue_release_free
    assert env._lib.slhip_queues_selftest(env._queues, _hip.QUEUES_SELFTEST_PLANT, 0) == _hip.SL_E_UNSUPPORTED
    env.reset()
    env.step_queues(acts)
    env.queues_sync()
    env.queues_close()


def test_queue_stepping_with_misplaced_workgroups():
    """A REAL misplacement, not a planted word: step t dispatches slice i on queue (i + t) mod n (self-test hook; the
    host drains the queues between steps WITHOUT any cache action, so the steps of an env stay in order).  A workgroup
    index of another queue runs -- on MI355X -- on another XCD (profiles/round4_a_xcd_placement.txt), so every env is
    stepped on another XCD than the step before.
      * with a stream's fences (the default) that is harmless: bit exact against the oracle;
      * release-free, the placement check must fire (sync raises), and the state it refuses really is wrong: boards
        differ from the oracle's, because steps read what another XCD's L2 had not written back."""
    import torch
    from safelife_amd import _hip
    B, T = 8192, 60
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts, min_performance_fraction=0.05)
    common = dict(first_level=np.arange(B) % len(pool), auto_reset=True, level_stride=3, time_limit=25, view_shape=(9, 9),
                  with_obs=False)
    cpu = util.OracleBackend(pool, B, **common)
    cpu.env.reset()
    acts = np.random.default_rng(12).integers(0, 9, (T, B)).astype(np.int32)
    for t in range(T):
        cpu.env.step(acts[t], n_threads=8)
    names = ("board", "goals", "rng", "agent_loc", "episode_idx", "num_steps")
    want = {name: cpu.get(name).copy() for name in names}
    d_acts = torch.from_numpy(acts).to("cuda")
    torch.cuda.synchronize()

    dev = util.DeviceBackend(pool, B, **common)
    env = dev.env
    _queues_or_skip(env, 4)
    _hip.check(env._lib.slhip_queues_selftest(env._queues, _hip.QUEUES_SELFTEST_SWAP, 1))
    env.reset()
    env.step_queues_many(d_acts)
    env.queues_sync()
    for name in names:
        assert np.array_equal(dev.get(name), want[name]), ("agent fences, queues swapped", name)
    env.queues_close()

    dev = util.DeviceBackend(pool, B, **common)
    env = dev.env
    _queues_or_skip(env, 4, release_free=True, recover=False)
    env.reset()
    env.step_queues_many(d_acts[:10])       # honest placement first: nothing to refuse
    env.queues_sync()
    _hip.check(env._lib.slhip_queues_selftest(env._queues, _hip.QUEUES_SELFTEST_SWAP, 1))
    env.step_queues_many(d_acts[10:])
    with pytest.raises(_hip.SafeLifeHipError, match="another XCD"):
        env.queues_sync()
    # what the check refused (the fence itself completed: everything is visible): compare it anyway
    wrong = int((dev.get("board") != want["board"]).any(axis=(1, 2)).sum())
    print("release-free stepping with misplaced workgroups: %d of %d boards differ from the oracle" % (wrong, B))
    assert wrong > 0
    env.queues_close()


@pytest.mark.parametrize("release_free", [False, True], ids=["agent-fences", "release-free"])
def test_queue_stepping_soak(release_free):
    """Many unsynchronised queue steps at the bench's size (8192 envs; two, four and six queues; one call per step and
    one call for all of them) against the ORACLE on the same actions: boards, generators, episode state of every env."""
    import torch
    from safelife_amd.vector_env import SafeLifeVectorEnv
    B, T = 8192, 300
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 3) % len(pool)
    kw = dict(first_level=first, auto_reset=True, level_stride=5, time_limit=40, view_shape=(15, 15), with_obs=False)
    host_acts = np.random.default_rng(6).integers(0, 9, (T, B)).astype(np.int32)
    cpu = util.OracleBackend(pool, B, **kw)
    cpu.env.reset()
    for t in range(T):
        cpu.env.step(host_acts[t], n_threads=8)
    names = ("board", "goals", "rng", "agent_loc", "episode_idx", "level_idx", "episode_reward", "num_steps")
    want = {name: cpu.get(name).copy() for name in names}
    acts = torch.from_numpy(host_acts).to("cuda")
    for trial in range(2):
        for n in (2, 4, 6):
            env = SafeLifeVectorEnv(pool, B, **kw)
            _queues_or_skip(env, n, release_free)
            env.reset()
            if trial == 0:
                for t in range(T):
                    env.step_queues(acts[t])
            else:
                env.step_queues_many(acts)
            for name, ref in want.items():
                assert np.array_equal(env.numpy(name), ref), (trial, n, name)
            env.queues_close()


def test_queue_steps_then_stream_steps_are_ordered():
    """step_queues() followed by step_async() / step() / a direct queues_sync() and more queue steps: the env settles
    the queues before anything goes onto a HIP stream and takes a system-scope acquire behind a device synchronize
    when it comes back."""
    import torch
    B, T = 640, 40
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, min_performance_fraction=0.05)
    common = dict(first_level=np.arange(B) % len(pool), auto_reset=True, level_stride=3, time_limit=15, view_shape=(9, 9),
                  with_obs=False)
    dev = util.DeviceBackend(pool, B, slices=2, **common)
    cpu = util.OracleBackend(pool, B, **common)
    env = dev.env
    _queues_or_skip(env, 2)
    env.reset()
    cpu.env.reset()
    acts = np.random.default_rng(3).integers(0, 9, (T, B)).astype(np.int32)
    d_acts = torch.from_numpy(acts).to(env.device)
    torch.cuda.synchronize()
    for t in range(T):
        kind = t % 5
        if kind in (0, 1):
            env.step_queues(d_acts[t])
        elif kind == 2:
            env.step_async(d_acts[t])           # (queues pending: settled first)
        elif kind == 3:
            env.step(d_acts[t])
        else:
            env.queues_sync()                   # a direct sync, then actions made on a stream right before the step
            fresh = (d_acts[t] + 0).contiguous()
            env.step_queues(fresh)
            del fresh                           # (the env holds on to the tensor until its next sync)
        cpu.env.step(acts[t])
    for name in ENV_STATE:
        assert np.array_equal(dev.get(name), cpu.get(name)), name


def test_manual_reset_moves_on_and_episode_streams():
    """auto_reset=False, the flow of INTEGRATION.md section 3: the driver resets finished envs itself.  Every such
    reset takes the env's NEXT pool level (SafeLifeEnv.reset() -> next(level_iterator)) and, with
    episode_streams, its own random stream -- device against the oracle over several episodes; and two envs
    on the same spawner level must not be stochastic replicas of each other."""
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=6, min_performance_fraction=0.05)
    B = 40
    kw = dict(first_level=np.zeros(B, np.int32), auto_reset=False, level_stride=2, time_limit=12, view_shape=(9, 9))
    dev, cpu = util.DeviceBackend(pool, B, **kw), util.OracleBackend(pool, B, **kw)
    assert np.array_equal(dev.reset(), cpu.reset())
    rng = np.random.default_rng(9)
    for t in range(40):
        a = np.zeros(B, np.int32) if t < 6 else rng.integers(0, 9, B).astype(np.int32)
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(o1, o2), t
        if t == 5:      # same level, same (no-op) actions, different streams: the spawners must have diverged
            boards = dev.get("board")
            assert len({boards[e].tobytes() for e in range(B)}) > B // 2
        if d1.any():
            dev.env.reset(d1)
            cpu.env.reset(d1)
            assert np.array_equal(dev.get("obs"), cpu.env.obs)
    for name in ENV_STATE + ("loaded",):
        assert np.array_equal(dev.get(name), cpu.get(name)), name
    episodes = dev.get("episode_idx")
    assert episodes.min() >= 2 and np.array_equal(dev.get("level_idx"), (2 * episodes) % len(pool))


@pytest.mark.parametrize("record", ["full", "compact"])
def test_reward_gather_through_rccl(record):
    """The real env (two slices, outputs redirected into the gather windows) + RewardGather with the RCCL
    collective forced on for a single rank (SAFELIFE_FORCE_GATHER=1): every window against the rewards / dones
    of an identical env read directly.  Own process: it initialises torch.distributed with the nccl backend.
    record: whole sl_step_out records, or their first 8 bytes (sl_env_batch.out_compact)."""
    import subprocess, sys
    out = subprocess.check_output([sys.executable, os.path.join(util.REPO, "tests", "rccl_gather_check.py")],
                                  env=dict(os.environ, SL_GATHER_RECORD=record), stderr=subprocess.STDOUT, timeout=300).decode()
    assert "rccl gather ok" in out, out


@pytest.mark.parametrize("pool_name,B,kw", [
    ("prune_still_25", 200, dict(view_shape=(25, 25), time_limit=25,
                                 output_channels=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27))),
    ("append_spawn_25", 90, dict(view_shape=(15, 9), time_limit=20)),
    ("navigation_64", 40, dict(view_shape=(25, 25), time_limit=20)),
    ("append_still_26", 50, dict(view_shape=(33, 33), time_limit=20, remove_white_goals=False)),
])
@pytest.mark.parametrize("layout", ["uint8", "float32"])
def test_policy_layout_from_the_step(pool_name, B, kw, layout):
    """f4: the step / reset kernels write the observation as the policy network takes it, [B,C,view_w,view_h]
    (training/models.py:100-103 transposes (h,w,c) -> (c,w,h); ppo.py:64 casts to float32): every step and
    after resets against the ORACLE's (h,w,c) observation transposed on the host; with and without the (h,w,c)
    tensor next to it."""
    import torch
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    first = (np.arange(B) * 3) % len(pool)
    common = dict(first_level=first, auto_reset=True, level_stride=2, **kw)
    dev = util.DeviceBackend(pool, B, policy_layout=layout, with_obs=(layout == "uint8"), **common)
    cpu = util.OracleBackend(pool, B, **common)
    dev.env.reset()
    want = cpu.reset()

    def check(tag):
        got = dev.env.policy_tensor.cpu().numpy()
        assert got.dtype == (np.uint8 if layout == "uint8" else np.float32)
        assert np.array_equal(got, np.transpose(want, (0, 3, 2, 1)).astype(got.dtype)), tag
        if layout == "uint8":
            assert np.array_equal(dev.get("obs"), want), tag
    check("reset")
    rng = np.random.default_rng(8)
    for t in range(45):
        a = rng.integers(0, 9, B).astype(np.int32)
        dev.env.step(a)
        want, _, _ = cpu.step(a)
        check(t)


def test_policy_layout_generic_kernels():
    """... and through the size-generic kernels (a board shape without row kernels)."""
    from safelife_amd.cell_types import CellTypes as CT
    from safelife_amd.levels import Level, LevelPool
    rng = np.random.default_rng(2)
    levels = []
    for _ in range(3):
        b = util.random_boards(rng, 1, 9, 13, 1)[0]
        b[4, 6] = CT.player
        levels.append(Level(b, (rng.integers(0, 8, (9, 13)) << 9).astype(np.uint16), [[4, 6]], min_performance=-1))
    pool = LevelPool(levels, counts_fn=_device_counts)
    kw = dict(first_level=np.arange(7) % 3, auto_reset=True, time_limit=9, view_shape=(7, 5))
    dev = util.DeviceBackend(pool, 7, policy_layout="float32", **kw)
    cpu = util.OracleBackend(pool, 7, **kw)
    dev.env.reset()
    want = cpu.reset()
    for t in range(25):
        assert np.array_equal(dev.env.policy_tensor.cpu().numpy(), np.transpose(want, (0, 3, 2, 1)).astype(np.float32)), t
        a = rng.integers(0, 9, 7).astype(np.int32)
        dev.env.step(a)
        want, _, _ = cpu.step(a)


def test_every_shipped_benchmark_level():
    """All 830 shipped benchmark levels loaded through levels.load_levels and stepped in ONE batch per board shape
    (800 x 26x26, 30 x 25x25), 100 seeded random actions each: every level's digest of (board, goals, reward stream,
    done stream, generator state, agent location) as the reference's SafeLifeEnv left it."""
    got, want = util.bulk_levels_digests(util.DeviceBackend, _device_counts)
    assert len(want) == 830
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


@pytest.mark.parametrize("name", ["wrap_train_append-still", "wrap_train_navigation"])
def test_vector_runner_hands_on_the_wrapped_reward(name):
    """A learner driven by VectorRunner must see what the reference's trainers see: the reward AFTER the wrapper
    stack (movement bonus, exit bonus, side-effect penalty), float64 -- against traces recorded through the
    reference's own wrappers."""
    import torch
    from safelife_amd.runner import VectorRunner
    from safelife_amd.vector_env import SafeLifeVectorEnv
    tr = util.load_trace(name)
    pool = util.pool_from_trace(tr, _device_counts)
    kw = util.env_kwargs_from_trace(tr)
    kw["output_channels"] = kw.get("output_channels") or tuple(range(16)) + (25, 26, 27)
    env = SafeLifeVectorEnv(pool, 1, first_level=0, auto_reset=True, level_stride=1, episode_streams=False,
                            policy_layout="uint8", with_obs=False, wrappers=util.wrappers_from_trace(tr), **kw)
    actions = tr["trace_actions"]
    step_no = [0]

    def scripted(obs):
        probs = torch.zeros((obs.shape[0], 9), device=obs.device)
        probs[:, int(actions[step_no[0]])] = 1.0
        return torch.zeros(obs.shape[0], device=obs.device), probs
    runner = VectorRunner(env, scripted)
    n_levels, n_resets = len(tr["trace_reset_obs"]), 0
    for t in range(len(tr["trace_reward"])):
        step_no[0] = t
        res = runner.take_one_step()
        assert res.rewards.dtype == torch.float64
        assert float(res.rewards[0]) == float(tr["trace_shaped_reward"][t]), t
        assert float(env.reward[0]) == float(tr["trace_reward"][t]), t
        if bool(res.done[0]):
            n_resets += 1
            if n_resets >= n_levels:
                break


def test_step_async_rejects_what_it_would_misread():
    """step_async() takes the tensor's address: an int64 tensor (torch's default for randint / argmax / multinomial)
    or a strided view would be read as something else -- it has to say so, like step() does."""
    import torch
    from safelife_amd.vector_env import SafeLifeVectorEnv
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts, n=4)
    env = SafeLifeVectorEnv(pool, 128, with_obs=False, slices=2)
    env.reset()
    good = torch.zeros(128, dtype=torch.int32, device=env.device)
    env.step_async(good)
    env.step_async(good.data_ptr())
    for bad in (torch.zeros(128, dtype=torch.int64, device=env.device),
                torch.zeros((128, 2), dtype=torch.int32, device=env.device)[:, 0],
                torch.zeros(64, dtype=torch.int32, device=env.device),
                torch.zeros(128, dtype=torch.int32)):
        with pytest.raises(ValueError):
            env.step_async(bad)
    env.join()


def test_side_effect_keys_overflow_is_reported_and_survived():
    """A starting board with more frozen movable / destructible cell types than the device-side key slots hold
    (SL_SE_MAX_KEYS - 8 = 16): the record says how many there are, and the batch rebuilds that entry's
    distributions on the host instead of silently dropping types."""
    from safelife_amd import _hip, side_effects as se
    from safelife_amd.cell_types import CellTypes as CT
    from safelife_amd.levels import Level, LevelPool
    H = W = 25
    b = np.zeros((H, W), np.uint16)
    kinds = [int(CT.frozen) | int(CT.destructible) | (c << 9) | extra for c in range(8) for extra in (0, int(CT.pushable), int(CT.pullable))]
    for i, v in enumerate(kinds):                         # 24 distinct frozen cell types
        b[2 + i // 8 * 3, 2 + (i % 8) * 2] = v
    b[20, 20] = CT.player
    b[10, 10:13] = CT.life | CT.color_g                  # a blinker for the occupancy tensors
    lv = Level(b, np.zeros_like(b), [[20, 20]], min_performance=-1)
    pool = LevelPool([lv], counts_fn=_device_counts)
    dev = util.DeviceBackend(pool, 4, auto_reset=True, time_limit=3, view_shape=(9, 9), with_obs=False,
                             side_effects=dict(capacity=16, num_samples=20))
    dev.env.reset()
    for t in range(3):
        dev.env.step(np.zeros(4, np.int32))
    batch = dev.env.side_effects_flush()
    assert len(batch) == 4 and batch.dropped() == 0
    recs = batch.records()
    assert (recs["n_cell_types"] == len(kinds)).all() and len(kinds) > _hip.SL_SE_MAX_KEYS - 8
    got_in, got_act = batch.distributions(1)
    want_in, want_act = se.distributions_from_counts(b, batch.boards[1].cpu().numpy().view(np.uint16),
                                                     batch.counts[:, 1].cpu().numpy(), 20)
    assert set(got_in) == set(int(k) for k in want_in) and len([k for k in got_in if k & CT.frozen]) == len(kinds)
    for k in want_in:
        assert np.array_equal(got_in[int(k)], want_in[k]) and np.array_equal(got_act[int(k)], want_act[k])


def test_sample_actions_kernel():
    """slhip_sample_actions: one-hot rows give that action; a fixed distribution is hit within sampling error; a call
    is a function of (seed, counter) -- the same arguments give the same draws, another counter gives others."""
    import torch
    from safelife_amd import _hip
    lib = _hip.lib()
    B = 65536
    dev = torch.device("cuda")
    st = _hip.current_stream_ptr()
    onehot = torch.zeros((B, 9), device=dev)
    want = torch.arange(B, device=dev) % 9
    onehot[torch.arange(B, device=dev), want] = 1.0
    out = torch.full((B,), -1, dtype=torch.int32, device=dev)
    _hip.check(lib.slhip_sample_actions(onehot.data_ptr(), B, 9, 5, 0, out.data_ptr(), st))
    assert torch.equal(out.to(torch.int64), want)
    p = torch.tensor([0.05, 0.3, 0.0, 0.15, 0.1, 0.1, 0.2, 0.02, 0.08], device=dev)
    probs = p.repeat(B, 1).contiguous()
    a1 = torch.empty(B, dtype=torch.int32, device=dev)
    a2 = torch.empty_like(a1)
    a3 = torch.empty_like(a1)
    _hip.check(lib.slhip_sample_actions(probs.data_ptr(), B, 9, 5, 7, a1.data_ptr(), st))
    _hip.check(lib.slhip_sample_actions(probs.data_ptr(), B, 9, 5, 7, a2.data_ptr(), st))
    _hip.check(lib.slhip_sample_actions(probs.data_ptr(), B, 9, 5, 8, a3.data_ptr(), st))
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)
    freq = torch.bincount(a1.to(torch.int64), minlength=9).double() / B
    assert float((freq - p.double()).abs().max()) < 0.01 and int((a1 == 2).sum()) == 0
    assert int(a1.min()) >= 0 and int(a1.max()) <= 8


def test_pipelined_runner_vs_oracle():
    """PipelinedRunner: the envs in two groups, each group's observation -> policy -> draw -> step on the group's own
    stream (slhip_env_step_range), the groups overlapping.  A scripted policy that DEPENDS on the observation it is
    handed (so a step that ran ahead of its policy, or a policy ahead of its observation, would show) against the
    oracle stepped with the same rule on the host: rewards and dones of every step, then the full state."""
    import torch
    from safelife_amd.runner import PipelinedRunner
    B, T = 448, 40
    chans = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27)
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, min_performance_fraction=0.05)
    common = dict(first_level=np.arange(B) % len(pool), auto_reset=True, level_stride=3, time_limit=15, view_shape=(9, 9),
                  output_channels=chans)
    dev = util.DeviceBackend(pool, B, slices=2, policy_layout="uint8", **common)
    cpu = util.OracleBackend(pool, B, **common)
    env = dev.env
    counters = {}

    def rule(total, t):                     # the action: a hash of the observation's bit count and the step number
        return (total * 7 + t * 3) % 9

    def scripted(obs):                      # obs: this group's [n, C, W, H] uint8 (the env's policy tensor as it is)
        key = obs.shape[0]                  # (the two groups have different sizes: 256 and 192 envs)
        t = counters.get(key, 0)
        counters[key] = t + 1
        total = obs.to(torch.int64).sum(dim=(1, 2, 3))
        probs = torch.zeros((obs.shape[0], 9), device=obs.device)
        probs[torch.arange(obs.shape[0], device=obs.device), rule(total, t)] = 1.0
        return torch.zeros(obs.shape[0], device=obs.device), probs
    seen = []

    def on_step(g, lo, hi):
        seen.append((g, env.reward[lo:hi].clone(), env.done[lo:hi].clone()))
    runner = PipelinedRunner(env, scripted, on_step=on_step)
    assert env.slice_bounds == (0, 256, B)
    obs = cpu.reset()                       # (h, w, c) uint8
    runner.run(T)
    runner.finish()
    torch.cuda.synchronize()
    k = 0
    for t in range(T):
        a = rule(obs.reshape(B, -1).sum(axis=1).astype(np.int64), t).astype(np.int32)
        obs, r, d = cpu.step(a)
        for g, (lo, hi) in enumerate(((0, 256), (256, B))):
            gg, rw, dn = seen[k]
            k += 1
            assert gg == g
            assert np.array_equal(rw.cpu().numpy(), r[lo:hi]) and np.array_equal(dn.cpu().numpy().astype(bool), d[lo:hi].astype(bool)), (t, g)
    for name in ENV_STATE:
        assert np.array_equal(dev.get(name), cpu.get(name)), name
    assert cpu.get("episode_idx").min() >= 1


@pytest.mark.parametrize("name", ["v10_append-spawn", "v10_prune-still_open", "v10_navigation", "append_still_1_chan19"])
def test_vector_runner_replays_reference_trace(name):
    """VectorRunner (obs -> policy -> device-side action draw -> fused step -> reset bookkeeping, the batched
    training/base_algo.py:152-244) driven by a scripted policy that puts all probability on the action the
    reference's trace took: observations (policy layout), rewards, dones and agent ids step by step against
    the recorded SafeLifeEnv run, across its resets."""
    import torch
    from safelife_amd.runner import VectorRunner
    from safelife_amd.vector_env import SafeLifeVectorEnv
    tr = util.load_trace(name)
    pool = util.pool_from_trace(tr, _device_counts)
    kw = util.env_kwargs_from_trace(tr)
    chans = kw.get("output_channels") or tuple(range(16)) + (25, 26, 27)
    raw = not kw.get("output_channels")
    kw["output_channels"] = chans

    def policy_view(obs):        # the trace's observation -> [C, W, H] float32
        if raw:                  # recorded as the uint32 view: unpack the channels here
            obs = np.stack([(obs >> c) & 1 for c in chans], axis=-1)
        return np.transpose(obs, (2, 1, 0)).astype(np.float32)
    env = SafeLifeVectorEnv(pool, 1, first_level=0, auto_reset=True, level_stride=1, episode_streams=False,
                            policy_layout="float32", with_obs=False, **kw)
    actions = tr["trace_actions"]
    step_no = [0]

    def scripted(obs):
        probs = torch.zeros((obs.shape[0], 9), device=obs.device)
        probs[:, int(actions[step_no[0]])] = 1.0
        return torch.zeros(obs.shape[0], device=obs.device), probs
    runner = VectorRunner(env, scripted)
    n_levels, n_resets = len(tr["trace_reset_obs"]), 0
    for t in range(len(tr["trace_reward"])):
        step_no[0] = t
        res = runner.take_one_step()
        assert int(res.actions[0]) == int(actions[t])
        assert float(res.rewards[0]) == float(tr["trace_reward"][t]), t
        assert bool(res.done[0]) == bool(tr["trace_done"][t]), t
        assert int(res.agent_ids[0][0]) == 0 and int(res.agent_ids[1][0]) == n_resets, t
        if t == 0:
            assert np.array_equal(res.obs[0].cpu().numpy(), policy_view(tr["trace_reset_obs"][0]))
        nxt = res.next_obs[0].cpu().numpy()
        if bool(res.done[0]):
            n_resets += 1
            if n_resets >= n_levels:
                break                   # the reference ran out of levels here
            assert np.array_equal(nxt, policy_view(tr["trace_reset_obs"][n_resets])), t
        else:
            assert np.array_equal(nxt, policy_view(tr["trace_obs"][t])), t
    assert int(runner.num_resets[0]) == n_resets


@pytest.mark.parametrize("pool_name,B,kw", [
    ("prune_still_25", 130, dict(view_shape=(25, 25), output_channels=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27))),
    ("navigation_64", 21, dict(view_shape=(15, 15), output_channels=None)),
    ("append_still_26", 33, dict(view_shape=(9, 9), wrappers=TRAINING_WRAPPERS)),
])
def test_get_obs_leaves_state_alone(pool_name, B, kw):
    """SafeLifeEnv.get_obs() on the batch (slhip_env_obs: the fused kernel with zero steps): the observation of
    the current state is rebuilt, and nothing else moves."""
    pool, _ = util.pool_from_fixture(pool_name, _device_counts, min_performance_fraction=0.05)
    common = dict(first_level=np.arange(B) % len(pool), auto_reset=True, time_limit=14, **kw)
    dev, cpu = util.DeviceBackend(pool, B, **common), util.OracleBackend(pool, B, **common)
    dev.reset(), cpu.reset()
    rng = np.random.default_rng(6)
    for t in range(20):
        a = rng.integers(0, 9, B).astype(np.int32)
        dev.env.step(a)
        want, _, _ = cpu.step(a)
    before = {name: dev.get(name) for name in ENV_STATE}
    dev.env.obs.zero_()
    dev.env.get_obs()
    assert np.array_equal(dev.get("obs"), want)
    for name in ENV_STATE:
        assert np.array_equal(dev.get(name), before[name]), name
    a = rng.integers(0, 9, B).astype(np.int32)          # and the run goes on as if nothing had happened
    dev.env.step(a)
    want, r2, d2 = cpu.step(a)
    assert np.array_equal(dev.get("obs"), want) and np.array_equal(dev.get("reward"), r2)


@pytest.mark.parametrize("pool_name,B,kw", [
    ("prune_still_25", 8192 + 3, dict(time_limit=12)),          # C3's kernel, a ragged last workgroup
    ("append_spawn_25", 1027, dict(time_limit=9)),               # C4's kernel (spawner draws)
    ("navigation_64", 258, dict(time_limit=16)),                 # C5's kernel (one board per wavefront, four per workgroup)
    ("append_still_26", 515, dict(time_limit=13)),
])
def test_goal_word_cache_vs_oracle(pool_name, B, kw):
    """The goal-word cache of the plain step kernels (sl_env_batch.goal_cache): single steps, masked resets, T-step
    launches, slices and queue steps in one run, every step's reward / done and the full state against the oracle --
    and the cache must actually be in use: a raised flag implies static goals on every board of its workgroup, and
    some flags are raised."""
    import torch
    from safelife_amd import _hip
    pool, _ = util.pool_from_fixture(pool_name, _device_counts)
    common = dict(auto_reset=True, level_stride=3, view_shape=(15, 15), with_obs=False, **kw)
    first = (np.arange(B) * 5) % len(pool)
    dev = util.DeviceBackend(pool, B, first_level=first, slices=2, **common)
    cpu = util.OracleBackend(pool, B, first_level=first, **common)
    env = dev.env
    assert env.goal_cache_group > 0 and env.struct.goal_cache
    env.reset()
    cpu.env.reset()
    rng = np.random.default_rng(21)

    def check(tag):
        for name in ("board", "goals", "rng", "agent_loc", "episode_idx", "level_idx", "num_steps", "goals_static"):
            assert np.array_equal(dev.get(name), cpu.get(name)), (tag, name)

    def flags():
        torch.cuda.synchronize()
        return env.goal_cache_flags()

    def both(a):
        env.step(a)
        cpu.env.step(a)
        assert np.array_equal(dev.get("reward"), cpu.get("reward")) and np.array_equal(dev.get("done"), cpu.get("done"))

    for t in range(10):                                         # single steps on the caller's stream
        both(rng.integers(0, 9, B).astype(np.int32))
    check("steps")
    f = flags()
    assert f.max() == 1 and f.sum() >= 1, "no workgroup runs on cached goal words"
    mask = (np.arange(B) % 3 == 0).astype(np.uint8)            # masked reset: every third env takes its next level now
    env.reset(mask)
    cpu.env.reset(mask)
    check("masked reset")
    for t in range(6):                                          # the slices, each on its own stream
        a = torch.from_numpy(rng.integers(0, 9, B).astype(np.int32)).to(env.device)
        env.step_async(a)
        env.join()
        cpu.env.step(a.cpu().numpy())
        assert np.array_equal(dev.get("reward"), cpu.get("reward")), ("slices", t)
    check("slices")
    T = 7                                                       # one T-step launch (levels load in its middle)
    a = rng.integers(0, 9, (T, B)).astype(np.int32)
    r_t, d_t = env.rollout(a)
    want = []
    for k in range(T):
        cpu.env.step(a[k])
        want.append(cpu.get("reward"))
    assert np.array_equal(r_t.cpu().numpy(), np.stack(want))
    check("rollout")
    for t in range(3):
        both(rng.integers(0, 9, B).astype(np.int32))
    try:                                                        # queue steps, where the runtime offers queues
        env.queues_open(4)
    except _hip.SafeLifeHipError:
        pass
    else:
        acts = torch.from_numpy(rng.integers(0, 9, (15, B)).astype(np.int32)).to(env.device)
        env.step_queues_many(acts)
        env.queues_sync()
        for k in range(15):
            cpu.env.step(acts[k].cpu().numpy())
        assert np.array_equal(dev.get("reward"), cpu.get("reward")) and np.array_equal(dev.get("done"), cpu.get("done"))
        check("queues")
        env.queues_close()
    for t in range(8):
        both(rng.integers(0, 9, B).astype(np.int32))
    check("end")
    assert cpu.get("episode_idx").min() >= 1
    f = flags()
    nb = env.goal_cache_group
    static = cpu.get("goals_static") == 1
    for w in np.nonzero(f)[0]:
        assert static[w * nb:(w + 1) * nb].all(), w


def test_release_free_stepping_recovers_from_a_misplacement():
    """Release-free queue stepping that a trainer survives: the env keeps a device-side copy of its state as of the last
    good sync and a log of the step calls since; when steps then run on the wrong XCD (the self-test hook that swaps the
    queues: a REAL misplacement, the check fires, boards are wrong) the sync restores the copy, reopens the queues with a
    stream's fences, replays the log and warns -- and the run goes on bit for bit with the oracle, training wrappers'
    state included."""
    import torch
    from safelife_amd import _hip
    B, T = 4096 + 11, 75
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts, min_performance_fraction=0.05)
    wr = dict(movement_bonus=0.1, movement_bonus_power=1e-100, movement_bonus_period=4, as_penalty=True, exit_bonus=0.5,
              penalty_coef=0.3)
    for wrappers in (None, wr):
        common = dict(first_level=np.arange(B) % len(pool), auto_reset=True, level_stride=3, time_limit=25, view_shape=(9, 9),
                      with_obs=False, wrappers=wrappers)
        cpu = util.OracleBackend(pool, B, **common)
        cpu.env.reset()
        acts = np.random.default_rng(14).integers(0, 9, (T, B)).astype(np.int32)
        d_acts = torch.from_numpy(acts).to("cuda")
        dev = util.DeviceBackend(pool, B, **common)
        env = dev.env
        _queues_or_skip(env, 4, release_free=True)          # (recover=True is the default)
        assert env._rf_recover
        env.reset()
        env.step_queues_many(d_acts[:25])
        env.queues_sync()                                   # a good sync: the state copy is taken here
        for t in range(25):
            cpu.env.step(acts[t], n_threads=8)
        assert np.array_equal(dev.get("board"), cpu.get("board"))
        _hip.check(env._lib.slhip_queues_selftest(env._queues, _hip.QUEUES_SELFTEST_SWAP, 1))
        env.step_queues_many(d_acts[25:40])                 # two calls in the log
        env.step_queues_many(d_acts[40:50])
        with pytest.warns(RuntimeWarning, match="placement check"):
            env.queues_sync()
        assert not env.queue_release_free and env._queues is not None
        for t in range(25, 50):
            cpu.env.step(acts[t], n_threads=8)
        names = ("reward", "done", "board", "goals", "rng", "agent_loc", "episode_idx", "level_idx", "num_steps")
        for name in names:
            assert np.array_equal(dev.get(name), cpu.get(name)), ("after the recovery", name)
        if wrappers:
            assert np.array_equal(dev.get("shaped_reward"), cpu.get("shaped_reward"))
        env.step_queues_many(d_acts[50:])                   # and on, with a stream's fences
        env.queues_sync()
        for t in range(50, T):
            cpu.env.step(acts[t], n_threads=8)
        for name in names:
            assert np.array_equal(dev.get(name), cpu.get(name)), ("after the recovery, later", name)
        env.queues_close()


@pytest.mark.parametrize("pool_name,B,wrappers,background", [
    ("prune_still_25", 2048 + 5, None, False),
    ("prune_still_25", 2048 + 5, None, True),
    ("append_spawn_25", 515, None, False),
    ("append_spawn_25", 515, None, True),
    ("prune_still_25", 300, dict(movement_bonus=0.1, movement_bonus_power=1e-100, movement_bonus_period=4, as_penalty=True,
                                 exit_bonus=0.5, penalty_coef=None), False),
])
def test_level_pool_refresh_while_stepping(pool_name, B, wrappers, background):
    """levels.LevelPool(refreshable=True) + SafeLifeVectorEnv.pool_stage / pool_commit (level_iterator.py:200-223,
    safelife_env.py:203-218: every reset takes a fresh level): half of the pool's levels are replaced every ten steps
    while the envs step through the library's queues -- staged ten steps ahead, committed between two queue calls, no
    queue drain for the refresh itself -- and the oracle, given the same replacements at the same steps, agrees bit
    for bit after 200 steps (state compared every fifty).  ``background``: the staging is handed to the env's helper
    thread BEHIND the step call (the order bench.py measures), while the queues are stepping."""
    import torch
    from safelife_amd import _hip
    _, _ = util.pool_from_fixture(pool_name, _device_counts, n=1)
    from safelife_amd.levels import LevelPool

    def build():
        p, _ = util.pool_from_fixture(pool_name, _device_counts)
        levels = list(p.levels)
        return levels, LevelPool(levels[:32], counts_fn=_device_counts, refreshable=True, min_performance_fraction=0.05)

    levels, pool_dev = build()
    _, pool_cpu = build()
    assert pool_dev.n_slots == 64 and len(pool_dev) == 32
    kw = dict(auto_reset=True, level_stride=3, view_shape=(15, 15), time_limit=13, with_obs=False)
    first = (np.arange(B) * 5) % 32
    dev = util.DeviceBackend(pool_dev, B, first_level=first, slices=2, wrappers=wrappers, **kw)
    cpu = util.OracleBackend(pool_cpu, B, first_level=first, wrappers=wrappers, **kw)
    env = dev.env
    cpu.env.set_pool_next(pool_cpu.next_table(3))
    env.reset()
    cpu.env.reset()
    use_queues = True
    try:
        env.queues_open(4)
    except _hip.SafeLifeHipError:
        use_queues = False
    rng = np.random.default_rng(77)
    T, CH = 200, 10
    acts = rng.integers(0, 9, (T, B)).astype(np.int32)
    d_acts = torch.from_numpy(acts).to(env.device)
    staged = None
    for t0 in range(0, T, CH):
        if staged is not None:                      # what was staged ten steps ago becomes current now, on both sides
            env.pool_commit()
            pool_cpu.replace(*staged)
            cpu.env.set_pool_next(pool_cpu.next_table(3))
        slots = rng.choice(32, 16, replace=False)
        news = [levels[int(k)] for k in rng.integers(0, len(levels), 16)]
        staged = (slots, news)
        if not background:
            env.pool_stage(slots, news)
        if use_queues:
            env.step_queues_many(d_acts[t0:t0 + CH])
        else:
            for t in range(t0, t0 + CH):
                env.step_async(d_acts[t])
        if background:
            assert env.pool_stage(slots, pool_dev.prepare(news), background=True) is None
        for t in range(t0, t0 + CH):
            cpu.env.step(acts[t])
        if (t0 + CH) % 50 == 0:
            for name in ("reward", "done", "board", "goals", "rng", "agent_loc", "episode_idx", "level_idx", "num_steps",
                         "exit_locs"):
                assert np.array_equal(dev.get(name), cpu.get(name)), (t0 + CH, name)
            if wrappers:
                assert np.array_equal(dev.get("shaped_reward"), cpu.get("shaped_reward")), t0 + CH
    assert cpu.get("episode_idx").min() >= 10
    lvl = cpu.get("level_idx")
    assert (lvl >= 32).any() and (lvl < 32).any()           # both banks in use
    env.queues_close()


def test_pool_slot_reuse_waits_for_the_episode_end_pass():
    """A refreshable pool under an env with the side-effect queue: the episode-end pass takes an episode's
    STARTING board from the pool when it runs, so a slot may only be staged again once (a) a time limit of steps has
    passed since its last replacement and (b) a flush has been issued since -- whose pass is waited for.  Also:
    step_slice() rounds count as dispatched steps (the guard's clock under PipelinedRunner)."""
    import torch
    from safelife_amd import _hip
    from safelife_amd.levels import LevelPool
    p, _ = util.pool_from_fixture("prune_still_25", _device_counts)
    levels = list(p.levels)
    pool = LevelPool(levels[:8], counts_fn=_device_counts, refreshable=True, min_performance_fraction=0.05)
    B, TL = 256, 6
    env = util.DeviceBackend(pool, B, first_level=np.arange(B) % 8, slices=2, auto_reset=True, level_stride=1, view_shape=(9, 9),
                             time_limit=TL, with_obs=False, side_effects=dict(capacity=1024, num_samples=20)).env
    env.reset()
    acts = torch.randint(0, 9, (B,), device=env.device, dtype=torch.int32)

    def steps(n):
        for _ in range(n):
            env.step_slice(0, acts)
            env.step_slice(1, acts)
        env.join()
    steps(1)
    assert env.steps_dispatched == 1                    # one round of both slices = one step of the batch
    env.pool_stage([0, 1], [levels[8], levels[9]])
    env.pool_commit()
    steps(2)
    with pytest.raises(ValueError, match="less than one time limit ago"):
        env.pool_stage([0], [levels[10]])
    steps(TL)
    with pytest.raises(ValueError, match="side_effects_flush"):
        env.pool_stage([0], [levels[10]])               # ended episodes of the old content are still queued
    batch = env.side_effects_flush(overlap=True, defer=True)
    env.pool_stage([0], [levels[10]])                   # launches the deferred pass and waits for it, then writes the slot
    env.pool_commit()
    assert len(batch) > 0
    rec = batch.records()
    assert set(np.unique(rec["level"])) <= set(range(16))
    # a staging whose device copy fails leaves the host pool as it was
    bank = pool.bank.copy()
    keep = env._lib.slhip_pool_write
    try:
        env._lib.slhip_pool_write = lambda *a: _hip.SL_E_HIP
        steps(TL)
        env.side_effects_flush()
        with pytest.raises(_hip.SafeLifeHipError):
            env.pool_stage([2], [levels[11]])
    finally:
        env._lib.slhip_pool_write = keep
    assert np.array_equal(pool.bank, bank) and env._pool_refresh["staged"] is None


def test_sharded_equals_unsharded():
    """env e behaves the same whichever rank owns it -- two half-size envs with
    env_offset 0 / B/2 (what ranks 0 and 1 of a 2-GPU run hold) against one env of size B."""
    from safelife_amd.vector_env import SafeLifeVectorEnv
    pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=24)
    B, T = 96, 60
    kw = dict(time_limit=25, view_shape=(15, 15), output_channels=None, auto_reset=True, level_stride=7)
    whole = SafeLifeVectorEnv(pool, B, **kw)
    parts = [SafeLifeVectorEnv(pool, B // 2, env_offset=k * (B // 2), **kw) for k in range(2)]
    whole.reset()
    for p_ in parts:
        p_.reset()
    rng = np.random.default_rng(17)
    for t in range(T):
        a = rng.integers(0, 9, B).astype(np.int32)
        whole.step(a)
        for k, p_ in enumerate(parts):
            p_.step(a[k * (B // 2):(k + 1) * (B // 2)])
        for name in ("reward", "done", "obs"):
            joined = np.concatenate([p_.numpy(name) for p_ in parts])
            assert np.array_equal(whole.numpy(name), joined), (t, name)
    for name in ("board", "goals", "rng", "agent_loc", "episode_idx", "level_idx"):
        assert np.array_equal(whole.numpy(name), np.concatenate([p_.numpy(name) for p_ in parts])), name


@pytest.mark.parametrize("seed", [1, 2])
def test_randomised_soak(seed):
    """tools/soak.py for a few seconds: random synthetic pools (shapes incl. 10x10 with 24 boards per
    workgroup, 0..8 exits, spawners, evolving goals, agent-less levels), random env options and wrapper
    settings, odd batch sizes -- device vs oracle on everything.  (Seed 1 is the run that caught the
    wrapper-state write-back of workgroups holding more than 21 boards.)"""
    import subprocess, sys
    out = subprocess.check_output([sys.executable, os.path.join(util.REPO, "tools", "soak.py"), "6", str(seed)],
                                  cwd=util.REPO, stderr=subprocess.STDOUT).decode()
    assert "soak ok" in out, out[-2000:]


@pytest.mark.parametrize("shape", [(25, 25), (64, 64), (7, 11)])
def test_advance_board_per_board_step_counts(sp, shape):
    """slhip_advance_board_each: every board advanced by its own number of steps (what
    side_effect_score needs for a batch of finished episodes), generator states included."""
    import torch
    rng = np.random.default_rng(shape[0])
    B = 37
    boards = util.random_boards(rng, B, shape[0], shape[1], 1)
    words = util.random_rng_words(rng, B)
    p = rng.choice([0.3, 0.05], B).astype(np.float32)
    steps = rng.integers(0, 40, B).astype(np.int32)
    steps[0] = 0
    want, w_cpu = np.zeros_like(boards), words.copy()
    for b in range(B):
        wb = w_cpu[b:b + 1].copy()
        want[b] = oracle.advance_board_batch(boards[b:b + 1], p[b:b + 1], int(steps[b]), wb)[0]
        w_cpu[b] = wb[0]
    d_rng = sp._to_device(words.copy(), np.uint64)
    d_b = sp._to_device(boards, np.uint16)
    got = sp.advance_board_batch(d_b, torch.from_numpy(p).to(d_b.device), d_rng, torch.from_numpy(steps).to(d_b.device))
    assert np.array_equal(sp._to_host(got, np.uint16), want)
    assert np.array_equal(sp._to_host(d_rng, np.uint64), w_cpu)


def test_slice_streams_are_probed_for_concurrency():
    """slhip_streams_concurrent: a stream cannot overlap itself (its two idle kernels run back to back), and the
    streams a sliced env ends up with were all found to overlap pairwise."""
    import ctypes as C
    import torch
    from safelife_amd import _hip
    pool, _ = util.pool_from_fixture("prune_still_25", _device_counts)
    env = util.DeviceBackend(pool, 256, slices=3, auto_reset=True).env
    lib = _hip.lib()
    ok = C.c_int(-1)
    s0 = env._slice_streams[0].cuda_stream
    assert lib.slhip_streams_concurrent(s0, s0, C.byref(ok)) == 0 and ok.value == 0
    free = torch.cuda.Stream()
    pairs = 0
    for i, a in enumerate(env._slice_streams):
        for b in env._slice_streams[i + 1:]:
            assert lib.slhip_streams_concurrent(a.cuda_stream, b.cuda_stream, C.byref(ok)) == 0
            pairs += ok.value
    # (a box whose runtime maps every stream onto one queue would leave nothing to choose from: then the env
    #  falls back to whatever streams it has and this only checks that the probe ran)
    assert pairs == 3 or lib.slhip_streams_concurrent(s0, free.cuda_stream, C.byref(ok)) == 0
    assert lib.slhip_streams_concurrent(s0, s0, None) != 0
