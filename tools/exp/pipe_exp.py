#!/usr/bin/env python3
"""Experiment: the C3 batch cut into n slices, each stepped on its own HIP stream (cross-launch overlap)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv

pool = bench.load_pool(os.environ.get("SL_POOL", "prune_still_25"), _device_counts)
B = int(os.environ.get("SL_ENVS", "8192"))
lib = _hip.lib()
lib.slhip_exp_pipeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = _hip.device()
KMAX = 405
acts = torch.randint(0, 9, (KMAX, B), device=dev, dtype=torch.int32)
tag = os.environ.get("SAFELIFE_HIP_LIB", "current")
for n in [int(x) for x in os.environ.get("SL_SLICES", "1,2,4,8").split(",")]:
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
    for K in (400, 20):
        for threaded in (0, 1):
            res = []
            for rep in range(3):
                rc = lib.slhip_exp_pipeline(arr, n, aptr, K, B, threaded, out)
                assert rc == 0, lib.slhip_last_error()
                res.append((out[0] / K, out[1] / K))
            best = min(res, key=lambda r: r[1])
            print("%s slices=%d K=%3d threaded=%d : enqueue %.2f us/step, total %.2f us/step  (all: %s)" % (
                tag, n, K, threaded, best[0], best[1], " ".join("%.2f" % r[1] for r in res)), flush=True)
    del envs
