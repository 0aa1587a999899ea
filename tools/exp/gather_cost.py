"""Host cost of one gather window through slhip_gather_* (one rank: self send/recv), next to a stepping loop."""
import os, sys, time
os.environ["SAFELIFE_FORCE_GATHER"] = "1"
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
from safelife_amd.sharding import RewardGather
pool = bench.load_pool("prune_still_25", _device_counts)
env = SafeLifeVectorEnv(pool, 8192, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                        auto_reset=True, with_obs=False, slices=2)
env.reset()
every = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = RewardGather(env, every=every, world=1, rank=0)
g.prime()
acts = torch.randint(0, 9, (400, 8192), device="cuda", dtype=torch.int32)
ptrs = [acts[t].data_ptr() for t in range(400)]
tb, ts, ta = [], [], []
for t in range(400):
    a = time.perf_counter(); g.before_step(t); b = time.perf_counter(); env.step_async(ptrs[t]); c = time.perf_counter(); g.after_step(t); d = time.perf_counter()
    tb.append(b - a); ts.append(c - b); ta.append(d - c)
torch.cuda.synchronize()
tb, ts, ta = (np.array(x) * 1e6 for x in (tb, ts, ta))
close = np.arange(400) % every == every - 1
first = np.arange(400) % every == 0
print("every", every, "| step %.1f us (median) %.1f (mean)" % (np.median(ts), ts.mean()),
      "| after_step at window close: median %.1f mean %.1f" % (np.median(ta[close]), ta[close].mean()),
      "| before_step at window start: median %.1f mean %.1f, else %.2f" % (np.median(tb[first]), tb[first].mean(), np.median(tb[~first])),
      "| steps right after a close: %.1f" % np.median(ts[np.roll(close, 1)]))
# pieces
import ctypes as C
lib = g._lib
ws = g._writer_streams()
t0 = time.perf_counter()
for _ in range(50): g._order(ws, [g._stream])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(50): lib.slhip_gather_window(g._comm, g.buf[0].data_ptr(), g.recv[0].data_ptr(), g.buf[0].numel() * 4, g._gptr[0])
t3 = time.perf_counter()
torch.cuda.synchronize()
print("streams_order %.1f us, gather_window %.1f us per call (idle GPU)" % ((t1 - t0) / 50 * 1e6, (t3 - t2) / 50 * 1e6))
