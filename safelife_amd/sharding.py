"""
Multi-GPU layout: one process per GPU, envs block-partitioned, no halo exchange.

Boards are independent (the torus wraps inside a board), so the step itself needs no collective.
The only cross-GPU traffic of the path is what a centralised learner on rank 0 needs from the
other ranks each step: ``reward`` (float32) and ``done`` (uint8) per env.  xGMI is point-to-point,
and a 5-byte-per-env message per step is purely latency bound, so the records of ``every``
consecutive steps are written by the step kernel straight into one packed device buffer
(``[every*B] float32 | [every*B] uint8``) and gathered with ONE RCCL ``gather`` per ``every``
steps, issued asynchronously so it overlaps the following steps; two buffers alternate.
"""
import numpy as np


def shard_bounds(total_envs, world, rank):
    """Contiguous block of global env ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RewardGather(object):
    """Per-step (reward, done) records of a SafeLifeVectorEnv -> rank 0, batched.

    Usage per step t:  ``before_step(t); env.step(a); after_step(t)``; ``flush()`` at the end.
    On rank 0, ``latest()`` returns (reward[world, every, B], done[world, every, B]) of the last
    completed window (views into the receive buffer).
    """

    def __init__(self, env, every=32, world=1, rank=0, group=None):
        import torch
        self.torch = torch
        self.env, self.every, self.world, self.rank, self.group = env, int(every), int(world), int(rank), group
        B = env.num_envs
        self.B = B
        n = self.every * B
        self.nbytes = n * 5
        self.buf = [torch.zeros(self.nbytes, dtype=torch.uint8, device=env.device) for _ in range(2)]
        self.recv = None
        if self.world > 1 and self.rank == 0:
            self.recv = [[torch.zeros(self.nbytes, dtype=torch.uint8, device=env.device)
                          for _ in range(self.world)] for _ in range(2)]
        self.work = [None, None]
        self.last = None
        self._n = n

    def before_step(self, t):
        slot, which = t % self.every, (t // self.every) % 2
        if slot == 0 and self.work[which] is not None:
            self.work[which].wait()          # stream-level wait: buffer is free again
            self.work[which] = None
        base = self.buf[which].data_ptr()
        self.env.set_step_outputs(base + 4 * slot * self.B, base + 4 * self._n + slot * self.B)

    def after_step(self, t):
        if t % self.every != self.every - 1:
            return
        which = (t // self.every) % 2
        self.last = which
        if self.world > 1:
            import torch.distributed as dist
            self.work[which] = dist.gather(self.buf[which], self.recv[which] if self.rank == 0 else None,
                                           dst=0, group=self.group, async_op=True)

    def flush(self):
        for k in (0, 1):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        self.env.set_step_outputs(None, None)

    def latest(self):
        if self.last is None:
            return None
        torch = self.torch
        bufs = self.recv[self.last] if self.recv is not None else [self.buf[self.last]]
        n = self._n
        reward = torch.stack([b[:4 * n].view(torch.float32).view(self.every, self.B) for b in bufs])
        done = torch.stack([b[4 * n:].view(self.every, self.B) for b in bufs])
        return reward, done
