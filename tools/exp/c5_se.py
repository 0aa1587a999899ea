"""C5 with the side-effect score in the timed region (bench.py's extra, by itself): python tools/exp/c5_se.py [queues=1]"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
p2 = bench.load_pool("navigation_64", _device_counts)
n, flush_every, n_meas = 4096, int(os.environ.get("FLUSH_EVERY", "512")), 2048
overlap = os.environ.get("OVERLAP", "1") != "0"
env = SafeLifeVectorEnv(p2, n, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, auto_reset=True,
                        with_obs=False, slices=4, side_effects=dict(capacity=2 * (n * flush_every // 1000 + 64), num_samples=1000))
env.reset()
dev = env.device
env.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = (torch.arange(n, device=dev, dtype=torch.int32) * 997) % 1000
acts = torch.randint(0, 9, (n_meas + 20, n), device=dev, dtype=torch.int32)
for t in range(20):
    env.step_async(acts[t])
env.side_effects_flush()
torch.cuda.synchronize()
# plain stepping, no pass, same envs (what the steps alone cost at this setting)
env.queues_open(4, release_free=True, recover=False)
env.step_queues_many(acts[:20]); env.queues_sync(); torch.cuda.synchronize()
for rep in range(2):
    env.side_effects_flush(); torch.cuda.synchronize()
    batches = []
    t0 = time.perf_counter()
    head = int(os.environ.get("DEFER", "0"))        # > 0: that many steps of the next window go in front of the pass
    for w0 in range(0, n_meas, flush_every):
        torch.cuda.current_stream().synchronize()
        if head:
            env.step_queues_many(acts[20 + w0:20 + w0 + head], assume_ordered=True)
            env.side_effects_launch()
            env.step_queues_many(acts[20 + w0 + head:20 + w0 + flush_every], assume_ordered=True)
        else:
            env.step_queues_many(acts[20 + w0:20 + w0 + flush_every], assume_ordered=True)
        batches.append(env.side_effects_flush(overlap=overlap, defer=head > 0))
    env.queues_sync()
    env.side_effects_join()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n_meas * 1e6
    print("LDS_MIN=%s prio=%s defer=%d overlap=%d flush_every=%d: %.2f us/step with the pass (%d episodes)" % (
        os.environ.get("SAFELIFE_STEP_LDS_MIN", "-"), os.environ.get("SL_AQL_PRIORITY", "-"), head, overlap, flush_every, us,
        sum(len(b) for b in batches)), flush=True)
# steps alone
env.side_effects_flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
env.step_queues_many(acts[20:20 + 400], assume_ordered=True)
env.queues_sync()
us = (time.perf_counter() - t0) / 400 * 1e6
print("   steps alone: %.2f us/step" % us, flush=True)
