"""C5's stepping alone through the queues (4096 x navigation 64x64, no side-effect queue): us per step"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
name = sys.argv[1] if len(sys.argv) > 1 else "navigation_64"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nsl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
p2 = bench.load_pool(name, _device_counts)
env = SafeLifeVectorEnv(p2, n, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, auto_reset=True,
                        with_obs=False, slices=nsl)
env.reset()
dev = env.device
if os.environ.get("NO_SPREAD") != "1":        # episode ends spread evenly over the envs (else: none within the run)
    env.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = (torch.arange(n, device=dev, dtype=torch.int32) * 997) % 1000
acts = torch.randint(0, 9, (440, n), device=dev, dtype=torch.int32)
env.queues_open(min(nsl, 4), release_free=True, recover=False)
env.step_queues_many(acts[:40]); env.queues_sync(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    env.step_queues_many(acts[40:440], assume_ordered=True)
    env.queues_sync()
    us = (time.perf_counter() - t0) / 400 * 1e6
    print("%s %s x %d, %d slices: %.2f us/step = %.2f ns per env-step" % (
        os.path.basename(os.environ.get("SAFELIFE_HIP_LIB", "tree")), name, n, nsl, us, us * 1e3 / n), flush=True)
