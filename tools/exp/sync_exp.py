#!/usr/bin/env python3
"""Experiment: fixed cost of a short timed region (first launch from idle + the final synchronize)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
if os.environ.get("SL_SPIN") == "1":
    hip = C.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
pool = bench.load_pool("prune_still_25", _device_counts)
B = 8192
lib = _hip.lib()
lib.slhip_exp_pipeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = _hip.device()
acts = torch.randint(0, 9, (405, B), device=dev, dtype=torch.int32)
tag = "spin=%s intr=%s" % (os.environ.get("SL_SPIN", "0"), os.environ.get("HSA_ENABLE_INTERRUPT", "-"))
for n in (1, 2):
    per = B // n
    envs = [SafeLifeVectorEnv(pool, per, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS, with_obs=False,
                              env_offset=i * per) for i in range(n)]
    for e in envs:
        e.reset()
    torch.cuda.synchronize()
    arr = (_hip.EnvBatch * n)()
    for i, e in enumerate(envs):
        C.memmove(C.byref(arr, i * C.sizeof(_hip.EnvBatch)), C.byref(e.struct), C.sizeof(_hip.EnvBatch))
    aptr = (C.c_void_p * n)(*[acts.data_ptr() + 4 * i * per for i in range(n)])
    out = (C.c_double * 2)()
    for K in (1, 5, 20, 100):
        for mode in (0, 2):
            res = []
            for rep in range(5):
                rc = lib.slhip_exp_pipeline(arr, n, aptr, K, B, mode, out)
                assert rc == 0
                res.append(out[1])
            print("%s slices=%d K=%3d query_spin=%d : total us %s" % (tag, n, K, mode >> 1, " ".join("%.1f" % r for r in sorted(res))), flush=True)
    # python-level: env.step loop + torch.cuda.synchronize, as bench.py does
    if n == 1:
        e = envs[0]
        for K in (20,):
            res = []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for t in range(K):
                    e.step(acts[t])
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                res.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
            print("%s python loop K=%d: (enqueue, total) us %s" % (tag, K, " ".join("(%.0f,%.0f)" % r for r in res)), flush=True)
    del envs
