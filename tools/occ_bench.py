#!/usr/bin/env python3
"""life_occupancy throughput: python tools/occ_bench.py [pool] [boards] [n_steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from safelife_amd import speedups, _hip
from safelife_amd.levels import _device_counts
name = sys.argv[1] if len(sys.argv) > 1 else "navigation_64"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
pool = bench.load_pool(name, _device_counts)
dev = _hip.device()
boards = torch.from_numpy(np.ascontiguousarray(pool.arrays()["pool_board"][np.arange(nb) % len(pool)]).view(np.int16)).to(dev)
probs = torch.full((nb,), 0.3, dtype=torch.float32, device=dev)
rngs = torch.arange(nb * 4, dtype=torch.int64, device=dev).reshape(nb, 4) * 2 + 1
speedups.life_occupancy_batch(boards[:16], probs[:16], rngs[:16].clone(), 10)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
speedups.life_occupancy_batch(boards, probs, rngs, n)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print("%s %d boards x %d steps: %.2f ms, %.3g board-steps/s" % (name, nb, n, ms, nb * n / (ms * 1e-3)))
# the roll-forward alone (advance_board, same boards, same step count): what the CA + draws cost without counting
out = torch.empty_like(boards)
speedups.advance_board_batch(boards[:16], probs[:16], rngs[:16].clone(), 10)
torch.cuda.synchronize()
e0.record()
speedups.advance_board_batch(boards, probs, rngs, n, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print("%s advance_board %d boards x %d steps: %.2f ms, %.3g board-steps/s" % (name, nb, n, ms, nb * n / (ms * 1e-3)))
