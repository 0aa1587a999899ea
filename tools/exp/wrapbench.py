import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, '.')
import torch, bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
pool = bench.load_pool("prune_still_25", _device_counts)
B = 8192
acts = torch.randint(0, 9, (440, B), device="cuda", dtype=torch.int32)
for name, w in (("wrappers", dict(movement_bonus=0.1, exit_bonus=0.5, penalty_coef=0.3)),
                ("inaction", dict(movement_bonus=0.1, exit_bonus=0.5, penalty_coef=0.3, baseline="inaction", inaction_seed=3))):
    try:
        env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                                auto_reset=True, with_obs=False, slices=2, wrappers=w)
    except Exception as ex:
        print(name, "unsupported:", ex)
        continue
    env.reset()
    for t in range(40): env.step_async(acts[t])
    env.join(); torch.cuda.synchronize()
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(40, 440): env.step_async(acts[t])
        env.join(); e1.record(); torch.cuda.synchronize()
        print(name, round(e0.elapsed_time(e1) / 400 * 1e3, 2), "us/step")
