#!/usr/bin/env python3
"""Timeline of consecutive sliced steps from the kernels' own clocks (needs a -DSL_TRACE build, tools/build_trace.sh):

    SAFELIFE_HIP_LIB=tools/lib_trace.so python tools/trace_overlap.py [steps] [--queues N] [--fences none|agent]

Default: two slices on two HIP streams (slhip_env_step_slices).  --queues N: N slices dispatched from the library's AQL
queues (slhip_queues_steps, all steps enqueued by one call), with a stream's fences (agent) or release-free (none).

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

import argparse
if "--gather-every" in sys.argv:
    os.environ["SAFELIFE_FORCE_GATHER"] = "1"
ap = argparse.ArgumentParser()
ap.add_argument("steps", nargs="?", type=int, default=6)
ap.add_argument("--queues", type=int, default=0)
ap.add_argument("--fences", default="agent")
ap.add_argument("--stage", type=int, default=0, help="queue mode: 1 = the steps are staged, then released by queues_go (bench.py's region)")
ap.add_argument("--spread", type=int, default=0, help="1: envs spread over their episodes (resets at every step)")
ap.add_argument("--gather-every", type=int, default=0,
                help="queue mode: hand a window of this many steps to the RCCL exchange (one rank, to itself) inside the trace")
args = ap.parse_args()
N = args.steps
B, SL = 8192, (args.queues or 2)
pool = bench.load_pool("prune_still_25", _device_counts)
env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                        auto_reset=True, with_obs=False, slices=2 if args.queues else SL)
env.reset()
if args.spread:
    env.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = (torch.arange(B, device=env.device, dtype=torch.int32) * 997) % 1000
if args.queues:
    env.queues_open(args.queues, release_free=(args.fences == "none"))
    print("AQL queues: %d, release-free: %s" % (env.queue_slices, env.queue_release_free))
acts = torch.randint(0, 9, (64 + N, B), device=env.device, dtype=torch.int32)
ptrs = [acts[t].data_ptr() for t in range(64 + N)]      # (addresses: the loop below is all the host does per step)
step = env.step_queues if args.queues else env.step_async
for t in range(64):
    step(ptrs[t])
if args.queues:
    env.queues_sync()
torch.cuda.synchronize()
lib = _hip.lib()
waves = (B // SL // 8) * 4
trace = torch.zeros((N * SL, waves, 16), dtype=torch.int64, device=env.device)
lib.slhip_trace_set.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
assert lib.slhip_trace_set(trace.data_ptr(), waves * 16 * 8, N * SL) == 0
import gc
gc.disable()
if args.queues and args.gather_every:
    from safelife_amd.sharding import RewardGather
    gather = RewardGather(env, every=args.gather_every, world=1, rank=0)
    gather.queued = True
    gather.prime()
    torch.cuda.synchronize()
    gather.run_queued(0, N, acts[64].data_ptr(), B, shift=0, assume_ordered=True)
    gather.flush()
    env.queues_sync()
    print("windows of %d steps handed to the exchange: %d" % (args.gather_every, gather.windows))
elif args.queues:
    if args.stage:
        env.step_queues_many(acts[64:64 + N], assume_ordered="untouched", defer=True)
        import time
        time.sleep(0.001)
        env.queues_go()
    else:
        env.step_queues_many(acts[64:64 + N], assume_ordered=True)
    env.queues_sync()
else:
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
print("%d-slice steps of %d envs (%s), per launch: first wave start / last wave start / last store acknowledged, ns since "
      "the first wave of the first launch (median over the XCDs)"
      % (SL, B, ("AQL queues, fences: " + ("none" if env.queue_release_free else "agent")) if args.queues else "HIP streams"))
prev_end = {}
for i in range(N * SL):
    s, step = i % SL, i // SL
    gap = med[i][0] - prev_end[s] if s in prev_end else float("nan")
    print("step %d slice %d | first wave %8.0f | last wave starts %8.0f | last store acked %8.0f | busy %6.0f | since the "
          "slice's previous launch ended %6.0f" % (step, s, med[i][0], med[i][1], med[i][2], med[i][2] - med[i][0], gap))
    prev_end[s] = med[i][2]
per_step = (med[-1][2] - med[SL - 1][2]) / (N - 1)
print("steady state: %.2f us per step (end of the last launch of a step to the end of the last launch of the next)" % (per_step / 1e3))

# per-wave phases of the launches above (steady state: the first two steps left out), time spent in each
names = ["loads issued", "loads landed (barrier)", "goal rows", "move in the image", "CA", "score + barrier", "leaders",
         "end barrier", "stores issued", "stores acknowledged"]
st = tr[2 * SL:, :, :11].astype(np.float64) * 10.0
ph = np.diff(st, axis=2)
print("phases, ns per wave (mean / p50 / p90 over the waves of %d launches): " % (st.shape[0]))
for i, n in enumerate(names):
    v = ph[:, :, i].reshape(-1)
    print("  %-24s %6.0f %6.0f %6.0f" % (n, v.mean(), np.percentile(v, 50), np.percentile(v, 90)))
life = (st[:, :, 10] - st[:, :, 0]).reshape(-1)
print("  %-24s %6.0f %6.0f %6.0f" % ("wave lifetime", life.mean(), np.percentile(life, 50), np.percentile(life, 90)))

# the load barrier: who it waits for (stamps 13 / 14 of trace builds in workgroups that do not reload a level -- the reload
# block reuses the two slots; wave 0 of a workgroup is its leader wave)
allst = tr[2 * SL:, :, :16].astype(np.float64) * 10.0
lead = (np.arange(allst.shape[1]) % 4) == 0
if (tr[2 * SL:, :, 13] != 0).any():
    def pct(v):
        v = v.reshape(-1)
        return "%5.0f / %5.0f / %5.0f" % (v.mean(), np.percentile(v, 50), np.percentile(v, 90))
    t0 = allst[:, :, 0]
    print("load barrier, ns since the wave's start (mean / p50 / p90):")
    print("  loading waves: loads issued and flag seen  %s" % pct((allst[:, :, 1] - t0)[:, ~lead]))
    # (a workgroup that reloads a level stamps 13-15 again, behind its score barrier: left out)
    late = (allst[:, :, 13] > allst[:, :, 6]) | (allst[:, :, 14] > allst[:, :, 6]) | (allst[:, :, 15] > allst[:, :, 6])
    ok11 = (allst[:, :, 13] != 0) & ~late & ~lead[None, :]
    ok12 = (allst[:, :, 14] != 0) & ~late & lead[None, :]
    print("  loading waves: own loads landed            %s   (%d of %d waves stamped)" % (pct((allst[:, :, 13] - t0)[ok11]), int(ok11.sum()), int((~lead).sum()) * allst.shape[0]))
    print("  leader waves:  first round trip back       %s   (%d of %d)" % (pct((allst[:, :, 14] - t0)[ok12]), int(ok12.sum()), int(lead.sum()) * allst.shape[0]))
    print("  leader waves:  goal-word flag seen         %s" % pct((allst[:, :, 13] - t0)[ok12]))
    print("  leader waves:  kernel arguments in         %s" % pct((allst[:, :, 15] - t0)[ok12]))
    print("  leader waves:  move decided, at the barrier %s" % pct((allst[:, :, 1] - t0)[:, lead]))
    print("  all waves:     barrier released            %s" % pct(allst[:, :, 2] - t0))

# what a workgroup's CU slot does between two steps: a wave's start against the end (stores acknowledged) of the same wave
# of the same slice one step earlier (same workgroup: same XCD, same clock)
succ = (tr[3 * SL:, :, 0] - tr[2 * SL:-SL, :, 10]).astype(np.float64).reshape(-1) * 10.0
print("  %-24s %6.0f %6.0f %6.0f   (same workgroup, one step earlier: its stores acknowledged -> this wave's start)"
      % ("successor starts after", succ.mean(), np.percentile(succ, 50), np.percentile(succ, 90)))
cyc = (tr[3 * SL:, :, 0] - tr[2 * SL:-SL, :, 0]).astype(np.float64).reshape(-1) * 10.0
print("  %-24s %6.0f %6.0f %6.0f   (start to start of the same workgroup's consecutive steps)" % ("workgroup cycle", cyc.mean(), np.percentile(cyc, 50), np.percentile(cyc, 90)))

# the waves a launch waits for: phase breakdown of the slowest 0.5 % (with --spread 1: the workgroups that reload a level)
lt = (st[:, :, 10] - st[:, :, 0])
cut = np.percentile(lt.reshape(-1), 99.5)
slow = lt >= cut
print("slowest 0.5 %% of waves (lifetime >= %.0f ns, %d waves): phases, mean ns" % (cut, int(slow.sum())))
for i, n in enumerate(names):
    print("  %-24s %6.0f   (all waves %6.0f)" % (n, ph[:, :, i][slow].mean(), ph[:, :, i].mean()))
print("  %-24s %6.0f   (all waves %6.0f)" % ("wave lifetime", lt[slow].mean(), lt.mean()))
per_launch_max = lt.max(axis=1)
print("per launch: slowest wave %.0f ns mean (median wave %.0f)" % (per_launch_max.mean(), np.median(lt, axis=1).mean()))

# the reload block of those waves (stamps 13-15, written only by workgroups that reload a level in the launch)
full = tr[2 * SL:, :, :16].astype(np.float64) * 10.0
# (slots 13-15 also carry the load barrier's stamps of every wave: a reload's are the ones behind the score barrier)
has = ((tr[2 * SL:, :, 13] > tr[2 * SL:, :, 6]) & (tr[2 * SL:, :, 14] > tr[2 * SL:, :, 13]) &
       (tr[2 * SL:, :, 15] > tr[2 * SL:, :, 14]))
if has.any():
    f = full[has]
    print("reload block, %d waves: score barrier -> block entered %.0f | rows fetched and placed %.0f | barrier %.0f | "
          "leaders' new record + last barrier %.0f | (whole 'leaders' phase %.0f) ns"
          % (int(has.sum()), (f[:, 13] - f[:, 6]).mean(), (f[:, 14] - f[:, 13]).mean(), (f[:, 15] - f[:, 14]).mean(),
             (f[:, 7] - f[:, 15]).mean(), (f[:, 7] - f[:, 6]).mean()))
