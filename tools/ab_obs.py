#!/usr/bin/env python3
"""Step time with a uint8 observation of SL_CHANNELS channels (15 or 19): SAFELIFE_HIP_LIB=... python tools/ab_obs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool = bench.load_pool("prune_still_25", _device_counts)
B = 8192
chans = bench.TRAIN_CHANNELS if os.environ.get("SL_CHANNELS", "19") == "15" else tuple(range(16)) + (25, 26, 27)
vs = int(os.environ.get("SL_VIEW", "25"))
env = SafeLifeVectorEnv(pool, B, view_shape=(vs, vs), output_channels=chans)
env.reset()
acts = torch.randint(0, 9, (240, B), device=env.device, dtype=torch.int32)
for t in range(40):
    env.step(acts[t])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(40, 240):
    env.step(acts[t])
e1.record()
torch.cuda.synchronize()
print(os.environ.get("SAFELIFE_HIP_LIB", "current"), len(chans), "channels, view", vs, round(e0.elapsed_time(e1) / 200 * 1e3, 2), "us/step")
