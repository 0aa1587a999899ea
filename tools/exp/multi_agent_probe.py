import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests import util
from safelife_amd.levels import LevelPool, _device_counts
from safelife_amd.multi_env import SafeLifeMultiAgentVectorEnv
from safelife_amd.vector_env import SafeLifeVectorEnv
levels = []
for name in ("multi_asym1", "multi_build_coop", "multi_build_compete"):
    levels += util.levels_from_trace(util.load_trace(name))
B = 8192
pool = LevelPool(levels, counts_fn=_device_counts, n_agents=2)
def timeit(env, acts, n=30):
    env.reset()
    for t in range(5): env.step(acts[t])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(n): env.step(acts[5 + t])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
acts = torch.randint(0, 9, (40, B, 2), device='cuda', dtype=torch.int32)
for kw in (dict(with_obs=False), dict(view_shape=(25, 25), output_channels=None), dict(view_shape=(25, 25), output_channels=tuple(range(12)) + (25, 26, 27))):
    env = SafeLifeMultiAgentVectorEnv(pool, B, time_limit=1000, **kw)
    print("multi", kw, "%.1f us/step" % timeit(env, acts))
