#!/usr/bin/env python3
"""Step time of the fused env on a synthetic pool of a given board shape (two stream slices, K steps):
    python tools/exp/shape_bench.py H W [envs] [spawners]"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from safelife_amd.levels import Level, LevelPool, _device_counts
from safelife_amd.cell_types import CellTypes as CT
from safelife_amd.vector_env import SafeLifeVectorEnv

H, W = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else max(256, (8192 * 625 // (H * W)) // 64 * 64)
n_spawn = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(5)
levels = []
for l in range(32):
    b = np.zeros((H, W), np.uint16)
    for _ in range(H * W // 40):                      # blocks (still lifes) and the odd blinker
        y, x = int(rng.integers(1, H - 3)), int(rng.integers(1, W - 3))
        if rng.random() < 0.8:
            b[y:y + 2, x:x + 2] = CT.life | (int(rng.integers(1, 8)) << 9)
        else:
            b[y, x:x + 3] = CT.life | (int(rng.integers(1, 8)) << 9)
    for _ in range(n_spawn):
        b[rng.integers(0, H), rng.integers(0, W)] = 152 | 0x200
    g = ((rng.integers(0, 8, (H, W)) << 9) * (rng.random((H, W)) < 0.2)).astype(np.uint16)
    b[0, 0] = CT.level_exit
    y, x = H // 2, W // 2
    b[y, x] = CT.player
    levels.append(Level(b, g, np.array([[y, x]]), spawn_prob=0.1, min_performance=0.5))
pool = LevelPool(levels, counts_fn=_device_counts)
env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(15, 15), auto_reset=True, with_obs=False, slices=2)
env.reset()
K = 300
acts = torch.randint(0, 9, (K + 40, B), device=env.device, dtype=torch.int32)
for t in range(40):
    env.step_async(acts[t])
env.join()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0 = env._slice_streams[0]
e0.record(s0)
for t in range(40, 40 + K):
    env.step_async(acts[t])
e1.record(s0)
env.join()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / K * 1e3
print("%dx%d  %d envs  spawners %d : %.2f us/step = %.3g env-steps/s, %.2f TB/s of board traffic (3*H*W*2 B per env-step)"
      % (H, W, B, n_spawn, us, B / us * 1e6, 3 * H * W * 2 * B / us / 1e6))
