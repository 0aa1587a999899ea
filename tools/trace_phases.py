#!/usr/bin/env python3
"""Phase timeline of the fused step kernel, per wave (needs a -DSL_TRACE build: tools/build_trace.sh):

    SAFELIFE_HIP_LIB=tools/lib_trace.so [SL_TRACE_ENVS=8192] python tools/trace_phases.py

s_memrealtime (100 MHz) counters of the eight XCDs are offset against each other, so every XCD is
aligned on its own earliest wave start before anything is compared across the chip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv

pool = bench.load_pool(os.environ.get("SL_TRACE_POOL", "prune_still_25"), _device_counts)
B = int(os.environ.get("SL_TRACE_ENVS", "8192"))
wrappers = dict(movement_bonus=0.1, exit_bonus=0.5, penalty_coef=0.3) if os.environ.get("SL_TRACE_WRAP") == "1" else None
env = SafeLifeVectorEnv(pool, B, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, with_obs=False,
                        wrappers=wrappers)
env.reset()
acts = torch.randint(0, 9, (64, B), device=env.device, dtype=torch.int32)
for t in range(20):
    env.step(acts[t])
H = pool.shape[0]
boards_per_wave = 64 // (H + 2) if 64 // (H + 2) == 64 // H else max(1, 64 // H)
nw = -(-B // (4 * boards_per_wave)) * 4             # 4 waves per workgroup
trace = torch.zeros((nw, 16), dtype=torch.int64, device=env.device)
lib = _hip.lib()
names = ["start", "loads issued", "after barrier", "goal rows ready", "after act", "after CA", "after score",
         "loop end", "after end barrier", "stores issued", "stores acked"]
NP = len(names)
acc_al, acc_own = [], []
for t in range(20, 40):
    rc = lib.slhip_env_rollout(env._sref, _hip.ptr(acts[t]), 1, _hip.ptr(trace), None, _hip.current_stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    full = trace.cpu().numpy()
    xcc = (full[:, 11] & 0xF).astype(int)
    ts = full[:, :NP].astype(np.float64) * 10.0
    al = np.zeros_like(ts)
    for x in np.unique(xcc):
        m_ = xcc == x
        al[m_] = ts[m_] - ts[m_, 0].min()
    acc_al.append(al)
    acc_own.append(ts - ts[:, :1])
al = np.concatenate(acc_al)
own = np.concatenate(acc_own)
step = np.diff(own, axis=1, prepend=0.0)
print("%d envs, %d waves; ns.  aligned = since the XCD's first wave start; phase = time spent in the phase" % (B, nw))
print("%-20s | %s | %s" % ("", "aligned  mean   p10   p50   p90   max", "phase  mean   p10   p50   p90   max"))
for i, n in enumerate(names):
    a = [al[:, i].mean()] + np.percentile(al[:, i], [10, 50, 90, 100]).tolist()
    p = [step[:, i].mean()] + np.percentile(step[:, i], [10, 50, 90, 100]).tolist()
    print("%-20s |        %6.0f%6.0f%6.0f%6.0f%6.0f |       %6.0f%6.0f%6.0f%6.0f%6.0f" % tuple([n] + a + p))

# placement (last launch): waves per SIMD, and duration against the crowding of the wave's SIMD
hw = full[:, 12]
simd = ((hw >> 4) & 0x3).astype(int)
cu = (((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)).astype(int)
key = cu * 4 + simd
uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("SIMDs used: %d; waves per SIMD histogram:" % len(uniq), np.bincount(cnt).tolist())
dur = (ts - ts[:, :1])[:, 9]
crowd = cnt[inv]
for c in np.unique(crowd):
    print("  waves on a SIMD shared by %d: %5d   duration to stores issued mean %.0f max %.0f"
          % (c, (crowd == c).sum(), dur[crowd == c].mean(), dur[crowd == c].max()))
wave_in_wg = np.arange(nw) % 4
print("mean duration by wave index in the workgroup:", [round(float(dur[wave_in_wg == k].mean())) for k in range(4)])
