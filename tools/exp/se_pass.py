"""The episode-end pass of C5 by itself: queue ~2000 finished 64x64 navigation episodes, time side_effects_flush()."""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
p2 = bench.load_pool("navigation_64", _device_counts)
n = 4096
env = SafeLifeVectorEnv(p2, n, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, auto_reset=True,
                        with_obs=False, slices=2, side_effects=dict(capacity=2 * (n * 512 // 1000 + 64), num_samples=1000))
env.reset()
dev = env.device
env.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = (torch.arange(n, device=dev, dtype=torch.int32) * 997) % 1000
acts = torch.randint(0, 9, (532, n), device=dev, dtype=torch.int32)
for t in range(20):
    env.step_async(acts[t])
env.side_effects_flush()
torch.cuda.synchronize()
for t in range(20, 532):
    env.step_async(acts[t])
env.join()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
b = env.side_effects_flush()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
recs = b.records()
steps = recs["num_steps"].astype(np.int64)
print("pass: %d episodes, %.2f ms; CA steps %.3g -> %.3g board-steps/s; mean episode length %.0f" % (
    len(b), ms, (steps + 2000).sum(), (steps + 2000).sum() / (ms * 1e-3), steps.mean()))
