import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, '.')
import torch, bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool = bench.load_pool("prune_still_25", _device_counts)
B = 8192
acts = torch.randint(0, 9, (240, B), device="cuda", dtype=torch.int32)
for layout, obs in (("uint8", False), ("float32", False), (None, True)):
    env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                            auto_reset=True, level_stride=1, with_obs=obs, policy_layout=layout)
    env.reset()
    for t in range(20): env.step(acts[t])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(20, 220): env.step(acts[t])
    e1.record(); torch.cuda.synchronize()
    print(layout or "hwc u8", round(e0.elapsed_time(e1) / 200 * 1e3, 2), "us/step")
