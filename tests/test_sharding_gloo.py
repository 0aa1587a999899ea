"""CPU suite, part 3: the multi-process path (one process per GPU in production) exercised with
world_size = 2 over gloo: env partition and the batched (reward, done) gather to rank 0.
The step kernel is replaced by a stand-in that writes through the very pointers the kernel is
given, so buffer packing, double buffering and unpacking are the real code."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from safelife_amd.sharding import RewardGather, shard_bounds


def test_shard_bounds_cover_everything():
    for total, world in ((65536, 8), (32768, 8), (10, 3), (7, 8), (8192, 1)):
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


class _FakeEnv(object):
    """Stands in for SafeLifeVectorEnv: owns reward/done outputs and lets them be redirected."""

    def __init__(self, B):
        self.num_envs = B
        self.device = torch.device("cpu")
        self.own_reward = np.zeros(B, np.float32)
        self.own_done = np.zeros(B, np.uint8)
        self.set_step_outputs(None, None)

    def set_step_outputs(self, reward_ptr, done_ptr):
        self.reward_ptr = self.own_reward.ctypes.data if reward_ptr is None else int(reward_ptr)
        self.done_ptr = self.own_done.ctypes.data if done_ptr is None else int(done_ptr)

    def step(self, t, rank):
        r = np.ctypeslib.as_array(C.cast(self.reward_ptr, C.POINTER(C.c_float)), (self.num_envs,))
        d = np.ctypeslib.as_array(C.cast(self.done_ptr, C.POINTER(C.c_uint8)), (self.num_envs,))
        r[:] = 1000.0 * rank + t + np.arange(self.num_envs) / 64.0
        d[:] = (np.arange(self.num_envs) + t + rank) % 3 == 0


def _worker(rank, world, port, B, every, steps, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = _FakeEnv(B)
    gather = RewardGather(env, every=every, world=world, rank=rank)
    seen = []
    for t in range(steps):
        gather.before_step(t)
        env.step(t, rank)
        gather.after_step(t)
        if t % every == every - 1:
            gather.flush()            # make the window visible before reading it in the test
            if rank == 0:
                rw, dn = gather.latest()
                seen.append((t, rw.clone().numpy(), dn.clone().numpy()))
    gather.flush()
    dist.barrier()
    if rank == 0:
        out_q.put(seen)
    dist.destroy_process_group()


def test_reward_gather_world2():
    world, B, every, steps = 2, 48, 4, 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, every, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [t for t, _, _ in seen] == [3, 7, 11]
    for t_end, rw, dn in seen:
        assert rw.shape == (world, every, B) and dn.shape == (world, every, B)
        for rank in range(world):
            for k in range(every):
                t = t_end - every + 1 + k
                assert np.array_equal(rw[rank, k], (1000.0 * rank + t + np.arange(B) / 64.0).astype(np.float32))
                assert np.array_equal(dn[rank, k], ((np.arange(B) + t + rank) % 3 == 0).astype(np.uint8))


def test_reward_gather_single_process():
    env = _FakeEnv(16)
    gather = RewardGather(env, every=2, world=1, rank=0)
    for t in range(4):
        gather.before_step(t)
        env.step(t, 0)
        gather.after_step(t)
    rw, dn = gather.latest()
    assert rw.shape == (1, 2, 16)
    assert np.array_equal(rw[0, 1].numpy(), (3 + np.arange(16) / 64.0).astype(np.float32))
    gather.flush()
    assert env.reward_ptr == env.own_reward.ctypes.data
