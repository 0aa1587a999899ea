"""Where a level-pool refresh spends its time (dev probe; GPU)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import util
from safelife_amd.levels import LevelPool, _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool, _ = util.pool_from_fixture("prune_still_25", _device_counts)
lv_all = list(pool.levels); n_half = len(lv_all) // 2
B = 8192
pool_r = LevelPool(lv_all[:n_half], counts_fn=_device_counts, refreshable=True)
env = SafeLifeVectorEnv(pool_r, B, time_limit=1000, view_shape=(25, 25), auto_reset=True, with_obs=False, slices=4)
env.reset()
dev = env.device
chunk, n_calls = 100, 8
acts = torch.randint(0, 9, (chunk * (n_calls + 1), B), device=dev, dtype=torch.int32)
env.queues_open(4, release_free=True, recover=False)
rr = np.random.default_rng(5)
env.pool_stage([0], [lv_all[n_half]]); env.pool_commit()
ready = pool_r.prepare(lv_all)

def run(mode):
    env.step_queues_many(acts[:chunk]); env.queues_sync(); torch.cuda.synchronize()
    tc = ts = tg = 0.0
    t0 = time.perf_counter()
    for c in range(n_calls):
        a = time.perf_counter()
        free = True
        if mode in ("fg", "bg", "marker", "bgnw"):
            if mode == "marker":
                env.queues_marker()
            else:
                free = env.pool_commit(wait=(mode != "bgnw"))
        b = time.perf_counter()
        env.step_queues_many(acts[chunk * (c + 1):chunk * (c + 2)], assume_ordered=True)
        d = time.perf_counter()
        if mode in ("fg", "bg", "bgnw") and free:
            slots = rr.choice(n_half, n_half // 6, replace=False)
            env.pool_stage(slots, ready.take(rr.integers(0, len(ready), len(slots))), background=(mode != "fg"))
        e = time.perf_counter()
        tc += b - a; ts += d - b; tg += e - d
    env.queues_sync()
    tot = time.perf_counter() - t0
    env.pool_commit()
    print("%-7s %.2f us/step   per call: commit %.0f us, steps %.0f us, stage %.0f us" % (
        mode, tot / (chunk * n_calls) * 1e6, tc / n_calls * 1e6, ts / n_calls * 1e6, tg / n_calls * 1e6), flush=True)

for rep in range(2):
    for mode in ("static", "marker", "fg", "bg", "bgnw"):
        run(mode)
# the pieces of one foreground staging
import cProfile, pstats
env.step_queues_many(acts[:chunk]); env.queues_sync()
pr = cProfile.Profile(); pr.enable()
for c in range(5):
    slots = rr.choice(n_half, n_half // 6, replace=False)
    env.pool_stage(slots, ready.take(rr.integers(0, len(ready), len(slots))))
    env.pool_commit()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
