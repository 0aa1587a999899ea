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
    """Stands in for SafeLifeVectorEnv: owns the output records and lets them be redirected."""

    def __init__(self, B):
        self.num_envs = B
        self.device = torch.device("cpu")
        self.own = np.zeros((B, 4), np.int32)
        self.set_step_outputs(None)

    def set_step_outputs(self, out_ptr, compact=False):
        self.out_ptr = self.own.ctypes.data if out_ptr is None else int(out_ptr)
        self.compact = bool(compact and out_ptr is not None)

    def step(self, t, rank):
        words = 2 if self.compact else 4            # (sl_env_batch.out_compact: the record's first 8 bytes only)
        rec = np.ctypeslib.as_array(C.cast(self.out_ptr, C.POINTER(C.c_int32)), (self.num_envs, words))
        rec[:, 0] = (1000.0 * rank + t + np.arange(self.num_envs) / 64.0).astype(np.float32).view(np.int32)
        rec[:, 1] = ((np.arange(self.num_envs) + t + rank) % 3 == 0).astype(np.int32)      # done in byte 0
        if not self.compact:
            rec[:, 2] = np.float32(t).view(np.int32)
            rec[:, 3] = t


def _worker(rank, world, port, B, every, steps, out_q, record="full"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = _FakeEnv(B)
    gather = RewardGather(env, every=every, world=world, rank=rank, record=record)
    assert gather.buf[0].shape == (every, B, 2 if record == "compact" else 4)
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


@pytest.mark.parametrize("record", ["full", "compact"])
def test_reward_gather_world2(record):
    world, B, every, steps = 2, 48, 4, 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, every, steps, q, record)) for r in range(world)]
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
    assert env.out_ptr == env.own.ctypes.data


def test_bench_launches_itself_and_gathers_inside_the_region():
    """`python bench.py --gpus 2` without a launcher (how the driver starts the N = 1 line, and what it would
    type for N > 1) re-executes under torch.distributed.run; --dry-run swaps the GPU step for a stand-in, so the
    launcher, the env partition, the gather windows (at least one must close inside the 20 timed steps) and the
    one-line JSON contract are checked on the CPU."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run",
                                   "--steps", "20", "--warmup", "5"], stderr=subprocess.DEVNULL, timeout=300).decode()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5
    assert d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["gather_windows_in_region"] >= 1 and d["gather_ok"] is True
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    assert d["config"]["last_rank_envs"] == [256, 512]


def test_gather_window_always_closes_inside_the_region():
    import bench
    for requested, steps in ((64, 20), (64, 400), (8, 20), (64, 1), (64, 2), (1, 20)):
        every = bench.gather_window(requested, steps)
        assert 1 <= every <= max(1, requested)
        for first in range(0, 70):                     # wherever the timed region starts
            assert any(t % every == every - 1 for t in range(first, first + steps)) or steps < every
        assert steps >= every
