#!/usr/bin/env python3
"""A/B helper: step time of one configuration under the library SAFELIFE_HIP_LIB points at.

    SL_MODE=wrap (training wrappers fused in, default) | obs15 | obs19 (uint8 observation) | plain
    SL_POOL, SL_ENVS, SL_VIEW as in bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool = bench.load_pool(os.environ.get("SL_POOL", "prune_still_25"), _device_counts)
B = int(os.environ.get("SL_ENVS", "8192"))
mode = os.environ.get("SL_MODE", "wrap")
vs = int(os.environ.get("SL_VIEW", "25"))
chans = tuple(range(16)) + (25, 26, 27) if mode == "obs19" else bench.TRAIN_CHANNELS
env = SafeLifeVectorEnv(pool, B, view_shape=(vs, vs), output_channels=chans, with_obs=mode.startswith("obs"),
                        wrappers=dict(movement_bonus=0.1, exit_bonus=0.5, penalty_coef=0.3) if mode == "wrap" else None)
env.reset()
acts = torch.randint(0, 9, (440, B), device=env.device, dtype=torch.int32)
for t in range(40):
    env.step(acts[t])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(40, 440):
    env.step(acts[t])
e1.record()
torch.cuda.synchronize()
print(os.environ.get("SAFELIFE_HIP_LIB", "current"), round(e0.elapsed_time(e1) / 400 * 1e3, 2), "us/step, mode", mode)
