#!/usr/bin/env python3
"""Randomised differential soak of the batched primitives (HIP) against the CPU oracle:
advance_board, life_occupancy, alive_counts, execute_actions over random shapes / batch sizes / steps.
    python tools/soak_prims.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from tests import util
from safelife_amd import speedups as sp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SHAPES = [(25, 25), (26, 26), (64, 64), (15, 15), (20, 20), (10, 10), (3, 3), (5, 40), (33, 7), (9, 9), (100, 100),
          (8, 8), (12, 12), (16, 16), (24, 24), (30, 30), (32, 32), (40, 40), (48, 48)]
t_end, n = time.time() + budget, 0
while time.time() < t_end:
    H, W = SHAPES[rng.integers(0, len(SHAPES))]
    B = int(rng.choice([1, 2, 5, 8, 9, 24, 25, 47, 100]))
    kind = int(rng.integers(0, 3))
    boards = util.random_boards(rng, B, H, W, kind)
    words = util.random_rng_words(rng, B)
    p = rng.choice([0.3, 0.0, 1.0, 0.07], B).astype(np.float32)
    d_b, d_p = sp._to_device(boards, np.uint16), torch.from_numpy(p).to(sp._to_device(boards, np.uint16).device)
    what = int(rng.integers(0, 4))
    desc = (what, (H, W), B, kind)
    if what == 0:
        ns = int(rng.choice([0, 1, 2, 7]))
        w_cpu, d_r = words.copy(), sp._to_device(words.copy(), np.uint64)
        want = oracle.advance_board_batch(boards, p, ns, w_cpu, n_threads=4)
        got = sp.advance_board_batch(d_b, d_p, d_r, ns)
        assert np.array_equal(sp._to_host(got, np.uint16), want) and np.array_equal(sp._to_host(d_r, np.uint64), w_cpu), desc
    elif what == 1:
        ns = int(rng.choice([0, 3, 40, 300]))
        w_cpu, d_r = words.copy(), sp._to_device(words.copy(), np.uint64)
        want = oracle.life_occupancy_batch(boards, p, ns, w_cpu, n_threads=4)
        got = sp.life_occupancy_batch(d_b, d_p, d_r, ns)
        assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(sp._to_host(d_r, np.uint64), w_cpu), desc
    elif what == 2:
        goals = (rng.integers(0, 8, (B, H, W)) << 9).astype(np.uint16)
        got = sp._to_host(sp.alive_counts_batch(d_b, sp._to_device(goals, np.uint16)), np.int64)
        assert np.array_equal(got, oracle.alive_counts_batch(boards, goals)), desc
    else:
        A = int(rng.integers(1, 4))
        locs = np.stack([rng.integers(0, H, (B, A)), rng.integers(0, W, (B, A))], -1).astype(np.int64)
        for b in range(B):
            for k in range(A):
                if rng.random() < 0.9:
                    boards[b, locs[b, k, 0], locs[b, k, 1]] = rng.choice([122, 122 | 0x200, 122 | 4, 122 | 256, 122 | 0x1000])
        acts = rng.integers(0, 9, (B, A)).astype(np.int64)
        b_cpu, l_cpu = boards.copy(), locs.copy()
        oracle.execute_actions_batch(b_cpu, l_cpu, acts)
        d_b2, d_l = sp._to_device(boards, np.uint16), sp._to_device(locs, np.int64)
        sp.execute_actions_batch(d_b2, d_l, torch.from_numpy(acts).to(d_b2.device))
        assert np.array_equal(sp._to_host(d_b2, np.uint16), b_cpu) and np.array_equal(sp._to_host(d_l, np.int64), l_cpu), desc
    n += 1
print("primitive soak ok: %d cases" % n)
