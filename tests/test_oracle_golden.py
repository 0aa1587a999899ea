"""CPU suite, part 1: pin the oracle (oracle/sl_oracle.c) to the reference.

Sources of truth, in order: golden vectors produced by the reference itself
(tests/golden/*.npz, generator: tests/golden/make_golden.py) and, when present, the reference
C extension compiled from its own sources (oracle/_ref).
"""
import os

import numpy as np
import pytest

import oracle
from tests import util


@pytest.fixture(scope="module")
def prim():
    with np.load(os.path.join(util.GOLDEN, "primitives.npz")) as d:
        return {k: d[k] for k in d.files}


def test_advance_board_golden(prim):
    n = int(prim["n_adv"])
    assert n > 300
    for k in range(n):
        p, steps = prim["adv_%03d_p_n" % k]
        out, words = oracle.advance_board(prim["adv_%03d_in" % k], float(p), int(steps),
                                          rng_words=prim["adv_%03d_rng0" % k])
        assert np.array_equal(out, prim["adv_%03d_out" % k]), k
        assert np.array_equal(words, prim["adv_%03d_rng1" % k]), k


def test_alive_counts_golden(prim):
    for k in range(int(prim["n_cnt"])):
        got = oracle.alive_counts(prim["cnt_%03d_board" % k], prim["cnt_%03d_goals" % k])
        assert got.dtype == np.int64 and got.shape == (8, 9)
        assert np.array_equal(got, prim["cnt_%03d_out" % k]), k


def test_execute_actions_golden(prim):
    for k in range(int(prim["n_act"])):
        board = prim["act_%03d_board" % k].copy()
        locs = prim["act_%03d_locs" % k].copy()
        oracle.execute_actions(board, locs, prim["act_%03d_acts" % k])
        assert np.array_equal(board, prim["act_%03d_board_out" % k]), k
        assert np.array_equal(locs, prim["act_%03d_locs_out" % k]), k


def test_life_occupancy_golden(prim):
    for k in range(int(prim["n_occ"])):
        got = oracle.life_occupancy(prim["occ_%02d_in" % k], 0.3, int(prim["occ_%02d_n" % k]),
                                    rng_words=prim["occ_%02d_rng0" % k])
        assert got.dtype == np.int32
        assert np.array_equal(got, prim["occ_%02d_out" % k]), k


def test_patterns_known_answers():
    """Pure-Life evolution of the reference's shipped patterns, incl. the glider's period-4 shift."""
    with np.load(os.path.join(util.GOLDEN, "patterns.npz")) as d:
        for name in ("glider", "acorn", "rpentomino", "growth"):
            for n in (1, 4, 20, 100):
                assert np.array_equal(oracle.advance_board(d[name + "_in"], 0.3, n), d["%s_n%d" % (name, n)])
        # the level holds a glider, a blinker and a loaf: after 4 steps the glider (rows 7-9,
        # cols 8-10) has moved one cell down-right, the other two are back in phase
        g0, g4 = d["glider_in"].copy(), d["glider_n4"]
        glider = np.zeros_like(g0, dtype=bool)
        glider[7:10, 8:11] = g0[7:10, 8:11] > 0
        assert glider.sum() == 5
        expect = np.where(glider, 0, g0) | np.roll(np.where(glider, g0, 0), (1, 1), (0, 1))
        assert np.array_equal(expect, g4)


def test_batch_matches_single():
    rng = np.random.default_rng(5)
    boards = util.random_boards(rng, 12, 25, 25, kind=1)
    words = util.random_rng_words(rng, 12)
    w2 = words.copy()
    out = oracle.advance_board_batch(boards, 0.3, 3, w2, n_threads=4)
    for b in range(12):
        o, w = oracle.advance_board(boards[b], 0.3, 3, rng_words=words[b])
        assert np.array_equal(o, out[b]) and np.array_equal(w, w2[b])


def test_pcg64_matches_numpy():
    L = oracle.lib()
    bg = np.random.PCG64(12345)
    g = oracle.Pcg64(*[int(x) for x in oracle.pcg64_state_words(bg)])
    gen = np.random.Generator(bg)
    import ctypes as C
    for _ in range(100):
        assert L.slo_pcg64_next_double(C.byref(g)) == gen.random()
    bg2 = np.random.PCG64(99)
    g2 = oracle.Pcg64(*[int(x) for x in oracle.pcg64_state_words(bg2)])
    L.slo_pcg64_advance(C.byref(g2), 1000)
    bg2.advance(1000)
    assert [g2.state_hi, g2.state_lo] == [int(x) for x in oracle.pcg64_state_words(bg2)[:2]]


def test_reference_timing_harness():
    """bench.py's C2 reference leg calls the reference's compiled advance_board_nstep from a C loop
    (oracle.ref_advance_batch): same boards, same generator stream as its CPython wrapper board by board."""
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    boards = util.random_boards(np.random.default_rng(4), 64, 25, 25, kind=0)
    g1, g2 = np.random.PCG64(7), np.random.PCG64(7)
    ref.set_bit_generator(g1)
    got = oracle.ref_advance_batch(ref, boards, 0.3, 2)
    ref.set_bit_generator(g2)
    want = np.stack([ref.advance_board(b, 0.3, 2) for b in boards])
    assert np.array_equal(got, want) and g1.state == g2.state


def test_against_compiled_reference():
    """Random differential test against oracle/_ref (the reference C built from its own sources)."""
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    rng = np.random.default_rng(2)
    for trial in range(150):
        h, w = (25, 25) if trial % 3 == 0 else tuple(int(v) for v in rng.integers(3, 40, 2))
        board = util.random_boards(rng, 1, h, w, kind=trial % 3)[0]
        n, p = int(rng.integers(1, 5)), float(rng.choice([0.3, 0.0, 1.0, 0.1]))
        g1, g2 = np.random.PCG64(trial), np.random.PCG64(trial)
        ref.set_bit_generator(g1)
        assert np.array_equal(ref.advance_board(board, p, n), oracle.advance_board(board, p, n, bitgen=g2))
        assert g1.state == g2.state
        goals = (rng.integers(0, 8, (h, w)) << 9).astype(np.uint16)
        assert np.array_equal(ref.alive_counts(board, goals), oracle.alive_counts(board, goals))


@pytest.mark.parametrize("name", util.trace_names())
def test_env_trace(name):
    """The oracle's batched SafeLifeEnv against traces of the reference's SafeLifeEnv."""
    tr = util.load_trace(name)
    n = util.replay_trace(tr, util.OracleBackend, util.oracle_counts)
    assert n == len(tr["trace_reward"])


@pytest.mark.parametrize("name", ["v10_prune-still_open", "v10_navigation", "worked_7x7_exit",
                                  "v10_append-spawn", "pattern_glider_noagent"])
def test_env_trace_terminal_states(name):
    tr = util.load_trace(name)
    assert util.replay_trace_terminal(tr, util.OracleBackend, util.oracle_counts) >= 1


@pytest.mark.parametrize("fixture", ["side_effect_inputs.npz", "side_effect_inputs_64.npz"])
def test_side_effect_score_internals(fixture):
    """The inputs of the reference's side_effect_score (side_effects.py:103-113) -- roll the starting board forward by
    the episode's length, then 1000 steps of life_occupancy from there and from the board the agent left -- as the
    reference computed them under one generator (25x25 benchmark level and 64x64 navigation level): the oracle
    reproduces boards, both occupancy tensors and the generator state after each stage."""
    with np.load(os.path.join(util.GOLDEN, fixture)) as d:
        d = {k: d[k] for k in d.files}
    p, n = float(d["spawn_prob"]), int(d["num_steps"])
    words = d["rng0"].copy()[None]                      # [1, 4] uint64, advanced in place by the batch primitives
    b1 = oracle.advance_board_batch(d["b0"][None], p, n, words)
    assert np.array_equal(b1[0], d["b1"]) and np.array_equal(words[0], d["rng1"])
    occ0 = oracle.life_occupancy_batch(b1, p, 1000, words)
    assert np.array_equal(occ0[0], d["occ0"]) and np.array_equal(words[0], d["rng2"])
    occ1 = oracle.life_occupancy_batch(d["b2"][None], p, 1000, words)
    assert np.array_equal(occ1[0], d["occ1"]) and np.array_equal(words[0], d["rng3"])


def test_every_shipped_benchmark_level():
    """All 830 shipped benchmark levels (8 x 100 of v1.0 -- legacy `agent_loc` (x, y) key, 26x26 -- and the 30
    single-level files of v0.1, 25x25), 100 seeded random actions each: the oracle reproduces the reference's
    digest of (board, goals, reward stream, done stream, generator state, agent location) for every one."""
    got, want = util.bulk_levels_digests(util.OracleBackend, util.oracle_counts)
    assert len(want) == 830
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_refreshable_pool_on_the_oracle():
    """levels.LevelPool(refreshable=True): a replacement goes into the level's spare slot, the successor table names it
    from then on, and the oracle env that follows the table loads exactly the new content at its next reset -- while an
    env that is still playing the old content keeps it (its slot is untouched until the same level is replaced again)."""
    from safelife_amd.levels import LevelPool
    pool0, _ = util.pool_from_fixture("prune_still_25", util.oracle_counts)
    levels = list(pool0.levels)
    L = 8
    pool = LevelPool(levels[:L], counts_fn=util.oracle_counts, refreshable=True)
    assert len(pool) == L and pool.n_slots == 2 * L
    nt = pool.next_table(3)
    assert np.array_equal(nt[:L], (np.arange(L) + 3) % L) and np.array_equal(nt[L:], nt[:L])
    B = 2 * L
    first = np.arange(B) % L
    kw = dict(auto_reset=True, level_stride=3, time_limit=5, with_obs=False)
    cpu = util.OracleBackend(pool, B, first_level=first, **kw)
    cpu.env.set_pool_next(pool.next_table(3))
    cpu.env.reset()
    assert np.array_equal(cpu.get("level_idx"), first)
    old4 = pool.pool_board[4].copy()
    phys = pool.replace([4, 6], [levels[20], levels[21]])
    assert phys == [4 + L, 6 + L] and pool.slot_of(4) == 4 + L and pool.slot_of(5) == 5
    assert np.array_equal(pool.pool_board[4], old4) and np.array_equal(pool.pool_board[4 + L], levels[20].board)
    cpu.env.set_pool_next(pool.next_table(3))
    noop = np.zeros(B, np.int32)
    for t in range(5):              # time limit 5: every env resets at the fifth step, onto (level + 3) % L
        cpu.env.step(noop)
    lvl = cpu.get("level_idx")
    want = (first + 3) % L
    want = np.where(want == 4, 4 + L, np.where(want == 6, 6 + L, want))
    assert np.array_equal(lvl, want)
    e = int(np.nonzero(lvl == 4 + L)[0][0])
    got = cpu.get("board")[e]
    keep = (levels[20].board & 256) == 0            # (exit cells are repainted by the reset)
    assert np.array_equal(got[keep], levels[20].board[keep])
    # replacing level 4 AGAIN goes back to slot 4; envs on slot 4 + L keep their content
    phys = pool.replace([4], [levels[22]])
    assert phys == [4] and np.array_equal(pool.pool_board[4 + L], levels[20].board)
    with pytest.raises(ValueError):
        LevelPool(levels[:L], counts_fn=util.oracle_counts).replace([0], [levels[9]])
    # prepared levels (LevelPool.prepare: counts, points, RNG words ahead of time; replace only copies) == plain ones
    a = LevelPool(levels[:L], counts_fn=util.oracle_counts, refreshable=True, seed=5)
    b = LevelPool(levels[:L], counts_fn=util.oracle_counts, refreshable=True, seed=5)
    ready = b.prepare(levels[10:30])
    assert len(ready) == 20 and len(ready.take([3, 3, 7])) == 3
    for slots, idx in (([1, 5, 2], [0, 7, 19]), ([5, 0], [4, 4])):
        assert a.replace(slots, [levels[10 + k] for k in idx]) == b.replace(slots, ready.take(idx))
        for k in LevelPool.ARRAYS:
            if k != "pool_rng":         # (levels without a generator of their own draw theirs from the pool's seed, in the
                #                          order they are prepared: the words differ by construction here)
                assert np.array_equal(getattr(a, k), getattr(b, k)), k
        assert np.array_equal(a.bank, b.bank) and np.array_equal(a.initial_counts, b.initial_counts)
    with pytest.raises(ValueError):
        b.replace([1, 1], ready.take([0, 1]))
    with pytest.raises(ValueError):
        b.replace([L], ready.take([0]))


@pytest.mark.parametrize("name", ["multi_asym1", "multi_build_coop", "multi_build_compete", "multi_hand_exit"])
def test_oracle_multi_agent_env_traces(name):
    """slo_env_step_multi / _reset_multi -- the multi-agent half of SafeLifeEnv.step (single_agent=False,
    safelife_env.py:148-218) -- against traces of the reference's own env on its own two-agent levels: every agent's
    observation, reward, done, success, episode accumulators, the shared board, goals and generator, step for step,
    reloading when all agents are done."""
    tr = util.load_trace(name)
    assert util.replay_trace_multi(tr, util.OracleMultiBackend, util.oracle_counts) > 0
