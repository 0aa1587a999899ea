"""Run by tests/test_hip_parity.py::test_reward_gather_through_rccl in a process of its own: the real
SafeLifeVectorEnv (two slices) + RewardGather with the exchange forced on for a single rank -- the library's own
RCCL entry points (slhip_gather_*: rank 0 sends to and receives from itself inside one RCCL group, on the gather's
side stream).  Two phases: step_async() (the windows are written on the slice streams) and step() on the SAME
sliced env (one launch on the caller's stream: the gather has to follow the writer), then step_queues() (the
library's AQL queues: a marker behind each window, waited for by the library's worker thread), one call per step
and whole windows per call (RewardGather.run_queued -> slhip_queues_steps with one record slot per step)."""
import os
import sys

os.environ["SAFELIFE_FORCE_GATHER"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from safelife_amd import _hip
from safelife_amd.levels import _device_counts
from safelife_amd.sharding import RewardGather
from safelife_amd.vector_env import SafeLifeVectorEnv
from tests import util

dev = _hip.device()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pool, _ = util.pool_from_fixture("append_spawn_25", _device_counts, n=16)
B, every, T = 256, 8, 40
kw = dict(time_limit=12, view_shape=(9, 9), auto_reset=True, with_obs=False)
env = SafeLifeVectorEnv(pool, B, slices=2, **kw)
ref = SafeLifeVectorEnv(pool, B, **kw)          # same envs, outputs read directly every step
env.reset(), ref.reset()
gather = RewardGather(env, every=every, world=1, rank=0, record=os.environ.get("SL_GATHER_RECORD", "full"))
assert gather.backend == "rccl" and gather.collective
assert gather.buf[0].shape[-1] == (2 if gather.compact else 4)
gather.prime()
rng = np.random.default_rng(3)
want_r, want_d = [], []
for t in range(T):
    a = torch.from_numpy(rng.integers(0, 9, B).astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    gather.before_step(t)
    env.step_async(a)
    gather.after_step(t)
    ref.step(a)
    want_r.append(ref.numpy("reward"))
    want_d.append(ref.numpy("done"))
    if t % every == every - 1:
        rw, dn = gather.latest()                # waits for the collective that filled the window
        assert rw.shape == (1, every, B)
        assert np.array_equal(rw[0].cpu().numpy(), np.stack(want_r[-every:])), t
        assert np.array_equal(dn[0].cpu().numpy(), np.stack(want_d[-every:])), t
# phase 2: step() on the sliced env -- ONE launch on the caller's stream; a gather that still followed the slice
# streams would ship half-written windows
side = torch.cuda.Stream()
for t in range(T, 2 * T):
    a = torch.from_numpy(rng.integers(0, 9, B).astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        gather.before_step(t)
        env.step(a)
        gather.after_step(t)
    ref.step(a)
    want_r.append(ref.numpy("reward"))
    want_d.append(ref.numpy("done"))
    if t % every == every - 1:
        rw, dn = gather.latest()
        assert np.array_equal(rw[0].cpu().numpy(), np.stack(want_r[-every:])), t
        assert np.array_equal(dn[0].cpu().numpy(), np.stack(want_d[-every:])), t
# phase 3: the same env stepped from the library's own AQL queues (slhip_queues_*): a window is handed over with a
# marker behind its last step (slhip_gather_window_queued: the worker thread waits for it, then issues the RCCL group);
# a blocking query before a buffer is reused
gather.flush()
torch.cuda.synchronize()
try:
    # the queue RCCL's kernel would hold up is measured (slhip_gather_stream_shares) and left out: three slices on
    # three of the library's four queues (slhip_queues_open_on)
    free = gather.free_queues(4)
    # (a host-timer measurement: usually one queue shares the exchange's pipe; a noisy box may flag more or none)
    assert set(free) <= {0, 1, 2, 3}, free
    if len(free) < 3:
        free = [0, 1, 2, 3]
    env.queues_open(queue_ids=free[:3])
    assert env.queue_slices == 3 and env.queue_ids == free[:3]
    print("step queues the exchange does not touch:", free)
    queued = True
except _hip.SafeLifeHipError as e:
    print("queues unavailable:", e)
    queued = False
if queued:
    gather.queued = True
    for t in range(2 * T, 3 * T):
        a = torch.from_numpy(rng.integers(0, 9, B).astype(np.int32)).to(dev)
        torch.cuda.synchronize()
        gather.before_step(t)
        env.step_queues(a)
        gather.after_step(t)
        ref.step(a)
        want_r.append(ref.numpy("reward"))
        want_d.append(ref.numpy("done"))
        if t % every == every - 1:
            rw, dn = gather.latest()
            assert np.array_equal(rw[0].cpu().numpy(), np.stack(want_r[-every:])), t
            assert np.array_equal(dn[0].cpu().numpy(), np.stack(want_d[-every:])), t
    # phase 4: whole windows per library call (two and a half windows at a time: the split at window ends is
    # run_queued's), every step's records in their own slot
    t, keep = 3 * T, []
    for chunk in (2 * every + every // 2, every - every // 2, 3 * every):
        acts = torch.from_numpy(rng.integers(0, 9, (chunk, B)).astype(np.int32)).to(dev)
        torch.cuda.synchronize()
        gather.run_queued(t, chunk, acts.data_ptr(), B)
        for k in range(chunk):
            ref.step(acts[k])
            want_r.append(ref.numpy("reward"))
            want_d.append(ref.numpy("done"))
        t += chunk
        keep.append(acts)                       # (addresses were handed over: the tensors live until the queues are synced)
        if t % every == 0:
            rw, dn = gather.latest()
            assert np.array_equal(rw[0].cpu().numpy(), np.stack(want_r[-every:])), t
            assert np.array_equal(dn[0].cpu().numpy(), np.stack(want_d[-every:])), t
    gather.flush()
    env.queues_sync()
    gather.queued = False
gather.flush()
torch.cuda.synchronize()
assert np.array_equal(env.numpy("board"), ref.numpy("board"))
gather.close()
dist.destroy_process_group()
print("rccl gather ok")
