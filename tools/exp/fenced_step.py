"""Host and device cost of the fenced drop-in step() with slices (fence + launches + join per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool = bench.load_pool("prune_still_25", _device_counts)
for slices in (1, 2):
    env = SafeLifeVectorEnv(pool, 8192, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                            auto_reset=True, with_obs=False, slices=slices)
    env.reset()
    acts = torch.randint(0, 9, (300, 8192), device=env.device, dtype=torch.int32)
    for t in range(50):
        env.step(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(50, 300):
        env.step(acts[t])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("slices", slices, "fenced step(): host %.2f us/step, wall %.2f us/step" % ((t1 - t0) / 250 * 1e6, (t2 - t0) / 250 * 1e6))
