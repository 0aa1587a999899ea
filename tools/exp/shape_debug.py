#!/usr/bin/env python3
"""advance_board on the device against the oracle for chosen shapes: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from tests import util
from safelife_amd import speedups as sp

def dev_advance(boards, p, n, words):
    d_b = sp._to_device(boards, np.uint16)
    d_r = sp._to_device(words, np.uint64)
    d_p = torch.as_tensor(np.asarray(p, np.float32).copy()).to(d_b.device)
    out = sp.advance_board_batch(d_b, d_p, d_r, n)
    return sp._to_host(out, np.uint16), sp._to_host(d_r, np.uint64)

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(40, 40), (48, 48)]
for (H, W) in shapes:
    for kind in (1, 0, 2):
        for B in (9, 400):
            for n in (1, 3):
                rng = np.random.default_rng(7 + kind)
                boards = util.random_boards(rng, B, H, W, kind)
                words = util.random_rng_words(rng, B)
                p = rng.choice([0.3, 0.0, 1.0, 0.05], B).astype(np.float32)
                if kind == 1 and n == 1 and B == 9:
                    p[:] = 0.0          # pure CA first
                w_cpu = words.copy()
                want = oracle.advance_board_batch(boards, p, n, w_cpu, n_threads=8)
                got, w_dev = dev_advance(boards, p, n, words)
                bad = got != want
                msg = "ok" if not bad.any() and np.array_equal(w_dev, w_cpu) else "MISMATCH"
                print("%dx%d kind %d B %d n %d: %s  cells %d boards %d rng %s" % (H, W, kind, B, n, msg, bad.sum(), bad.any((1, 2)).sum(),
                      "ok" if np.array_equal(w_dev, w_cpu) else "differs"))
                if bad.any():
                    cols = np.nonzero(bad.any((0, 1)))[0]
                    rows = np.nonzero(bad.any((0, 2)))[0]
                    print("    columns:", cols[:40].tolist(), " rows:", rows[:40].tolist())
                    b0 = np.nonzero(bad.any((1, 2)))[0][0]
                    y, x = np.argwhere(bad[b0])[0]
                    print("    first: board %d (%d,%d) got %#06x want %#06x was %#06x" % (b0, y, x, got[b0, y, x], want[b0, y, x], boards[b0, y, x]))
