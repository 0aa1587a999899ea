#!/usr/bin/env python3
"""life_occupancy / advance_board(n) throughput on random boards of a given shape: python tools/exp/occ_shapes.py H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util
from safelife_amd import speedups as sp, _hip
H, W = int(sys.argv[1]), int(sys.argv[2])
nb, n = max(256, (4096 * 4096 // (H * W)) // 64 * 64), 300
rng = np.random.default_rng(3)
boards = sp._to_device(util.random_boards(rng, nb, H, W, 1), np.uint16)
probs = torch.full((nb,), 0.1, dtype=torch.float32, device=boards.device)
rngs = sp._to_device(util.random_rng_words(rng, nb), np.uint64)
for name, fn in (("life_occupancy", lambda: sp.life_occupancy_batch(boards, probs, rngs, n)),
                 ("advance_board", lambda: sp.advance_board_batch(boards, probs, rngs, n))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%dx%d %s: %d boards x %d steps %.2f ms = %.3g board-steps/s" % (H, W, name, nb, n, ms, nb * n / (ms * 1e-3)))
