#!/usr/bin/env python3
"""Phase timeline of the fused step kernel (needs a -DSL_TRACE build of libsafelife_hip.so):
   hipcc ... -DSL_TRACE ... -o safelife_amd/libsafelife_hip.so ; python tools/trace_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv

pool = bench.load_pool("prune_still_25", _device_counts)
B = 8192
env = SafeLifeVectorEnv(pool, B, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, with_obs=False)
env.reset()
acts = torch.randint(0, 9, (64, B), device=env.device, dtype=torch.int32)
for t in range(20):
    env.step(acts[t])
grid = B // 8
trace = torch.zeros((grid, 16), dtype=torch.int64, device=env.device)
lib = _hip.lib()
names = ["start", "loads issued", "after barrier", "goal rows ready", "after act", "after CA", "after score",
         "loop end", "after end barrier", "stores issued"]
rows = []
for t in range(20, 40):
    rc = lib.slhip_env_rollout(env._sref, _hip.ptr(acts[t]), 1, _hip.ptr(trace), None, _hip.current_stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    tr = trace.cpu().numpy()[:, :10].astype(np.float64)
    rel = (tr - tr[:, 0].min()) * 10.0          # s_memrealtime: 100 MHz, chip-wide -> ns
    own = (tr - tr[:, :1]) * 10.0
    rows.append(np.stack([rel.mean(0), rel.min(0), rel.max(0), own.mean(0), np.percentile(rel, 90, axis=0)]))
m = np.mean(rows, axis=0)
print("ns since the first workgroup started (mean / min / max / p90 over workgroups); last: mean ns since own start")
for i, n in enumerate(names):
    print("%-20s %9.0f %9.0f %9.0f %9.0f   %9.0f" % (n, m[0, i], m[1, i], m[2, i], m[4, i], m[3, i]))
