"""The bit-plane CA step (safelife_amd/csrc/sl_planes.h) on a CPU model of a wavefront, against the oracle.

`tests/sim/plane_sim.cpp` compiles the kernel text with every value widened to the 64 lanes of a wave and the
vertical lane moves replaced by index tables (DPP wave shift with halo lanes, DPP wave rotate, ds_bpermute).  No GPU:
this pins the transposition network, the whole-row rule, the inheritance planes, the spawner draws' order and the
merge before any of it reaches the device (the `-m gpu` parity tests then check the device build itself)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plane_sim") / "plane_sim")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "sim", "plane_sim.cpp")])
    return exe


def _run(sim, tmp_path, boards, prob, words, steps, spawn, keep_planes=False):
    B, H, W = boards.shape
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("5i", H, W, B, steps, int(spawn) | (2 if keep_planes else 0)))
        f.write(boards.tobytes())
        f.write(prob.tobytes())
        f.write(words.tobytes())
    subprocess.check_call([sim, fin, fout])
    raw = open(fout, "rb").read()
    nb = boards.nbytes
    return (np.frombuffer(raw[:nb], np.uint16).reshape(B, H, W), np.frombuffer(raw[nb:], np.uint64).reshape(B, 4))


# (H, W): square shapes of the row kernels, plus shapes that force each vertical lane layout and both row parities
SHAPES = [(25, 25), (26, 26), (8, 8), (10, 10), (12, 12), (15, 15), (16, 16), (20, 20), (24, 24),
          (10, 25), (25, 10), (9, 27), (64, 28), (40, 5), (30, 4), (64, 64), (20, 64),
          # even rows of 30 to 48 cells: two plane words, one per half of the split layout (round 4)
          (30, 30), (32, 32), (40, 40), (48, 48), (12, 40), (64, 30)]


@pytest.mark.parametrize("H,W", SHAPES)
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_plane_step_matches_oracle(sim, tmp_path, H, W, kind):
    rng = np.random.default_rng(1000 * H + 10 * W + kind)
    for spawn in (False, True):
        boards = util.random_boards(rng, 7, H, W, kind)
        if not spawn:
            boards = boards & ~np.uint16(128)       # spawner-free instantiation: no SPAWNING bit anywhere
        words = util.random_rng_words(rng, 7)
        prob = np.full(7, 0.3, np.float32)
        got, got_rng = _run(sim, tmp_path, boards, prob, words, 4, spawn)
        exp_rng = words.copy()
        exp = oracle.advance_board_batch(boards, prob, 4, exp_rng)
        assert np.array_equal(got, exp), (H, W, kind, spawn)
        assert np.array_equal(got_rng, exp_rng), (H, W, kind, spawn)


def test_plane_step_known_patterns(sim, tmp_path):
    """Blinker, block and glider across the seams of a 25-wide torus (cells 0, 12, 13, 24 are the halves' edges)."""
    H = W = 25
    boards = np.zeros((4, H, W), np.uint16)
    life = 9 | 0x400
    for x in (24, 0, 1):
        boards[0, 0, x] = life                      # blinker over the row seam and the board's corner
    for (y, x) in ((5, 12), (5, 13), (6, 12), (6, 13)):
        boards[1, y, x] = life                      # block on the split between the halves
    for (y, x) in ((0, 12), (1, 13), (2, 11), (2, 12), (2, 13)):
        boards[2, y, x] = life                      # glider
    boards[3, 12, 12] = 122                         # agent alone: freezes its neighbourhood
    boards[3, 11, 12] = life
    words = util.random_rng_words(np.random.default_rng(5), 4)
    prob = np.zeros(4, np.float32)
    got, _ = _run(sim, tmp_path, boards, prob, words, 30, False)
    exp = oracle.advance_board_batch(boards, prob, 30, words.copy())
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("H,W", [(25, 25), (26, 26), (10, 10), (25, 10), (9, 27), (40, 5), (64, 64), (20, 64),
                                 (32, 32), (40, 40), (48, 48), (30, 30), (12, 40)])
def test_planes_kept_across_steps_match_oracle(sim, tmp_path, H, W):
    """The multi-step form (life_occupancy's thousand steps): one transposition in, the row stays in plane form --
    seam bits and the V_SHIFT halo lanes follow through the masks each step applies -- and one merge out."""
    rng = np.random.default_rng(77 * H + W)
    for spawn, kind in ((False, 0), (True, 1), (True, 2)):
        boards = util.random_boards(rng, 5, H, W, kind)
        if not spawn:
            boards = boards & ~np.uint16(128)
        words = util.random_rng_words(rng, 5)
        prob = np.full(5, 0.25, np.float32)
        got, got_rng = _run(sim, tmp_path, boards, prob, words, 12, spawn, keep_planes=True)
        exp_rng = words.copy()
        exp = oracle.advance_board_batch(boards, prob, 12, exp_rng)
        assert np.array_equal(got, exp), (H, W, kind, spawn)
        assert np.array_equal(got_rng, exp_rng), (H, W, kind, spawn)
