#!/usr/bin/env python3
"""Which of the library's AQL queues share a hardware pipe with which HIP streams?  (slhip_queues_stream_shares)"""
import ctypes as C, os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ["SAFELIFE_FORCE_GATHER"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
from safelife_amd.sharding import RewardGather

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pool = bench.load_pool("prune_still_25", _device_counts)
env = SafeLifeVectorEnv(pool, 8192, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                        auto_reset=True, with_obs=False, slices=2)
env.reset()
gather = RewardGather(env, every=8, world=1, rank=0)
env.queues_open(nq, release_free=True)
lib = _hip.lib()
lib.slhip_queues_stream_shares.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int)]
def shares(stream):
    m = C.c_int(0)
    _hip.check(lib.slhip_queues_stream_shares(nq, C.c_void_p(stream.cuda_stream), C.byref(m)))
    return m.value
print("queues:", nq)
print("current stream      mask %s" % bin(shares(torch.cuda.current_stream())))
print("gather's stream     mask %s" % bin(shares(gather._stream)))
for i, s in enumerate(env._slice_streams or []):
    print("slice stream %d      mask %s" % (i, bin(shares(s))))
keep = []
for i in range(10):
    s = torch.cuda.Stream()
    keep.append(s)
    print("new stream %2d       mask %s" % (i, bin(shares(s))))
