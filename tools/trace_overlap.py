#!/usr/bin/env python3
"""Timeline of consecutive two-slice steps from the kernels' own clocks (needs a -DSL_TRACE build, tools/build_trace.sh):

    SAFELIFE_HIP_LIB=tools/lib_trace.so python tools/trace_overlap.py [steps]

Every wave stamps s_memrealtime (100 MHz) at its start and when its stores are acknowledged; every launch of
slhip_env_step_slices writes its stamps to its own buffer.  The counters of the eight XCDs are offset against each
other but run at one rate, so each XCD is put on its own time axis (zero = its first wave of the first launch shown)
and the table gives, per launch, the median over the XCDs.  rocprofv3's kernel trace cannot show this: it serialises
the dispatches of the two streams."""
import ctypes as C, os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, SL = 8192, 2
pool = bench.load_pool("prune_still_25", _device_counts)
env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                        auto_reset=True, with_obs=False, slices=SL)
env.reset()
acts = torch.randint(0, 9, (64 + N, B), device=env.device, dtype=torch.int32)
ptrs = [acts[t].data_ptr() for t in range(64 + N)]      # (addresses: the loop below is all the host does per step)
step = env.step_async
for t in range(64):
    step(ptrs[t])
torch.cuda.synchronize()
lib = _hip.lib()
waves = (B // SL // 8) * 4
trace = torch.zeros((N * SL, waves, 16), dtype=torch.int64, device=env.device)
lib.slhip_trace_set.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
assert lib.slhip_trace_set(trace.data_ptr(), waves * 16 * 8, N * SL) == 0
import gc
gc.disable()
for t in range(64, 64 + N):
    step(ptrs[t])
torch.cuda.synchronize()
tr = trace.cpu().numpy()
xcc = (tr[:, :, 11] & 0xF).astype(int)
start, end = tr[:, :, 0].astype(np.float64) * 10.0, tr[:, :, 10].astype(np.float64) * 10.0     # ns
rows = []
for x in range(8):
    m = xcc == x
    if not m.any():
        continue
    t0 = start[0][m[0]].min() if m[0].any() else start[m].min()
    rows.append([(start[i][m[i]].min() - t0, start[i][m[i]].max() - t0, end[i][m[i]].max() - t0) for i in range(N * SL)])
med = np.median(np.array(rows), axis=0)
print("two-slice steps of %d envs, per launch: first wave start / last wave start / last store acknowledged, ns since the "
      "first wave of the first launch (median over the XCDs)" % B)
prev_end = {}
for i in range(N * SL):
    s, step = i % SL, i // SL
    gap = med[i][0] - prev_end[s] if s in prev_end else float("nan")
    print("step %d slice %d | first wave %8.0f | last wave starts %8.0f | last store acked %8.0f | busy %6.0f | since the "
          "slice's previous launch ended %6.0f" % (step, s, med[i][0], med[i][1], med[i][2], med[i][2] - med[i][0], gap))
    prev_end[s] = med[i][2]
per_step = (med[-1][2] - med[SL - 1][2]) / (N - 1)
print("steady state: %.2f us per step (end of the last launch of a step to the end of the last launch of the next)" % (per_step / 1e3))
