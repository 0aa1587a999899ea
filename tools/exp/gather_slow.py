"""Which part of the window hand-off makes the following launches slow?  (1 rank through RCCL)
usage: python tools/exp/gather_slow.py <mode>   mode: full | nowait | nofence | nojoin | copy"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
import torch, torch.distributed as dist
import bench
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
mode = sys.argv[1]
dev = _hip.device()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pool = bench.load_pool("prune_still_25", _device_counts)
B, every, K = 8192, 32, 200
env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                        auto_reset=True, with_obs=False, slices=2)
env.reset()
acts = torch.randint(0, 9, (K, B), device=dev, dtype=torch.int32)
ptr = [acts[t].data_ptr() for t in range(K)]
buf = [torch.zeros((every, B, 4), dtype=torch.int32, device=dev) for _ in range(2)]
recv = [[torch.zeros_like(buf[0])] for _ in range(2)]
work = [None, None]
for w in (0, 1):
    dist.gather(buf[w], recv[w], dst=0, async_op=True).wait()
torch.cuda.synchronize()
times = []
for t in range(K):
    slot, which = t % every, (t // every) % 2
    a = time.perf_counter()
    if slot == 0 and work[which] is not None:
        if mode == "perstream":
            for st in env._slice_streams:
                with torch.cuda.stream(st):
                    work[which].wait()
        elif mode != "nowait":
            work[which].wait()
        work[which] = None
        if mode not in ("nofence", "nowait", "perstream"):
            env.fence()
    env.set_step_outputs(buf[which].data_ptr() + 16 * slot * B)
    b = time.perf_counter()
    env.step_async(ptr[t])
    c = time.perf_counter()
    if slot == every - 1:
        if mode != "nojoin":
            env.join()
        if mode == "copy":
            recv[which][0].copy_(buf[which], non_blocking=True)
            work[which] = None
        else:
            work[which] = dist.gather(buf[which], recv[which], dst=0, async_op=True)
    times.append(((b - a) * 1e6, (c - b) * 1e6, (time.perf_counter() - c) * 1e6))
torch.cuda.synchronize()
tot = sum(sum(x) for x in times)
print(mode, "host us/step %.2f" % (tot / K), " ".join("%d:%.0f/%.0f/%.0f" % ((t,) + times[t]) for t in range(60, 110)))
dist.destroy_process_group()
