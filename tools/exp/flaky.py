import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool, _ = util.pool_from_fixture("navigation_64", _device_counts, min_performance_fraction=0.05)
B = 200
first = (np.arange(B) * 3) % len(pool)
kw = dict(first_level=first, auto_reset=True, level_stride=5, time_limit=30, view_shape=(25, 25))
for trial in range(int(os.environ.get("SL_TRIALS", "6"))):
    a_env = SafeLifeVectorEnv(pool, B, slices=1, **kw)
    b_env = SafeLifeVectorEnv(pool, B, slices=int(os.environ.get("SL_N", "4")), **kw)
    a_env.reset(); b_env.reset()
    rng = np.random.default_rng(23)
    bad = 0
    for t in range(60):
        a = rng.integers(0, 9, B).astype(np.int32)
        a_env.step(a); b_env.step(a)
        o1, o2 = a_env.numpy("obs"), b_env.numpy("obs")
        if not np.array_equal(o1, o2):
            idx = np.argwhere(o1 != o2)
            envs = np.unique(idx[:, 0])
            print("trial %d step %d: %d bytes differ, envs %s, first %s, vals %s vs %s" % (trial, t, len(idx), envs[:10], idx[:4].tolist(),
                  o1[tuple(idx[0])], o2[tuple(idx[0])]), flush=True)
            for name in ("board", "goals", "agent_loc"):
                print("   state", name, "equal:", np.array_equal(a_env.numpy(name), b_env.numpy(name)))
            bad += 1
            if bad > 3: break
    print("trial", trial, "bad steps", bad, flush=True)
