"""The bench's timed region as a function of its length: wall clock of K queue steps of C3 (8192 x 25x25 prune-still)
bracketed exactly like bench.py's region (device idle, one library call, torch.cuda.synchronize + queues_sync), for
K = 1 .. 400, median of `reps` repetitions each; least-squares line  elapsed = fixed + per_step * K  over the K's.

    python tools/exp/kfit.py [spread=0|1] [fences=none|agent] [reps] [queues]

SAFELIFE_HIP_LIB=<other .so> runs the same against another build (A/B)."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gc
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv

spread = int(sys.argv[1]) if len(sys.argv) > 1 else 0
fences = sys.argv[2] if len(sys.argv) > 2 else "none"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 4
B = 8192
NO_HEAD = os.environ.get("KFIT_NO_HEAD") == "1"
NO_TORCH_SYNC = os.environ.get("KFIT_NO_TORCH_SYNC") == "1"
STAGE = os.environ.get("KFIT_STAGE") == "1"         # staged regions + "untouched" first steps, as bench.py times them
pool = bench.load_pool("prune_still_25", _device_counts)
env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, auto_reset=True,
                        with_obs=False, slices=2)
env.reset()
dev = env.device
if spread:
    env.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = (torch.arange(B, device=dev, dtype=torch.int32) * 997) % 1000
acts = torch.randint(0, 9, (440, B), device=dev, dtype=torch.int32)
env.queues_open(nq, release_free=(fences == "none"), recover=False)
print("queues %d release_free %s (%s)" % (env.queue_slices, env.queue_release_free, env.queue_mode_note))
env.step_queues_many(acts[:40]); env.queues_sync(); torch.cuda.synchronize()
Ks = (1, 2, 5, 10, 20, 50, 100, 400)
med = {}
gc.collect(); gc.disable()
for K in Ks:
    ts = []
    for rep in range(reps):
        env.step_queues_many(acts[:5], assume_ordered=True)         # the bench's warm-up, then idle
        env.queues_sync(); torch.cuda.synchronize()
        if NO_HEAD:
            env._caller_ahead = False           # (experiment: the first step of the region without its system-scope acquire)
        if STAGE and K <= 48:                   # (bench.py's region: staged ahead of the clock, released inside it)
            env.step_queues_many(acts[40:40 + K], assume_ordered="untouched", defer=True)
            t0 = time.perf_counter()
            env.queues_go()
        else:
            t0 = time.perf_counter()
            env.step_queues_many(acts[40:40 + K], assume_ordered="untouched" if STAGE else True)
        t1 = time.perf_counter()
        if not NO_TORCH_SYNC:
            torch.cuda.synchronize()
        env.queues_sync()
        t2 = time.perf_counter()
        ts.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6))
    ts.sort()
    med[K] = ts[len(ts) // 2]
    print("K=%3d  region %8.1f us  = %6.2f us/step   (enqueue call %6.1f us)   min %.1f max %.1f" % (
        K, med[K][0], med[K][0] / K, med[K][1], ts[0][0], ts[-1][0]), flush=True)
x = np.array(Ks, float); y = np.array([med[K][0] for K in Ks])
A = np.stack([np.ones_like(x), x], 1)
(fixed, per), *_ = np.linalg.lstsq(A, y, rcond=None)
print("%s spread=%d fences=%s queues=%d%s%s: elapsed = %.1f us + %.3f us x K   (K=20 -> %.2f us/step)" % (
    os.path.basename(os.environ.get("SAFELIFE_HIP_LIB", "tree")), spread, fences, env.queue_slices, (" no-head" if NO_HEAD else "") + (" staged+untouched" if STAGE else ""), " no-torch-sync" if NO_TORCH_SYNC else "", fixed, per,
    (fixed + 20 * per) / 20))
