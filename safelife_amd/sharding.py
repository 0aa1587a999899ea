"""
Multi-GPU layout: one process per GPU, envs block-partitioned, no halo exchange.

Boards are independent (the torus wraps inside a board), so the step itself needs no collective.
The only cross-GPU traffic of the path is what a centralised learner on rank 0 needs from the
other ranks each step: the per-step output record of every env (``struct sl_step_out``: float32 reward, done / success /
times_up flags, episode reward and length -- 16 bytes).  xGMI is point-to-point, and a message of a
few bytes per env per step is purely latency bound, so the records of ``every`` consecutive steps
are written by the step kernel straight into one device buffer (``sl_step_out[every, B]``) and
gathered with ONE RCCL ``gather`` per ``every`` steps, issued asynchronously so it overlaps the
following steps; two buffers alternate.
"""
import os
import time

import numpy as np


def shard_bounds(total_envs, world, rank):
    """Contiguous block of global env ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RewardGather(object):
    """Per-step output records of a SafeLifeVectorEnv -> rank 0, batched.

    Usage per step t:  ``before_step(t); env.step(a); after_step(t)``; ``flush()`` at the end.
    On rank 0, ``latest()`` returns (reward[world, every, B] float32, done[world, every, B] uint8) of
    the last completed window (views into the receive buffers); ``latest_records()`` the raw records.
    """

    RECORD_BYTES = 16

    def __init__(self, env, every=32, world=1, rank=0, group=None):
        import torch
        self.torch = torch
        self.env, self.every, self.world, self.rank, self.group = env, int(every), int(world), int(rank), group
        self.B = B = env.num_envs
        shape = (self.every, B, 4)
        self.buf = [torch.zeros(shape, dtype=torch.int32, device=env.device) for _ in range(2)]
        self.recv = None
        # SAFELIFE_FORCE_GATHER=1 issues the collective even with one rank (exercises the RCCL path
        # on a single-GPU box)
        self.force = os.environ.get("SAFELIFE_FORCE_GATHER", "0") == "1"
        if (self.world > 1 or self.force) and self.rank == 0:
            self.recv = [[torch.zeros(shape, dtype=torch.int32, device=env.device) for _ in range(self.world)]
                         for _ in range(2)]
        self.collective = self.world > 1 or self.force     # False: one rank, nothing to gather
        self.cuda = torch.device(env.device).type == "cuda"
        self._join_ev = ([torch.cuda.Event() for _ in (getattr(env, "_slice_streams", None) or [None])[1:]]
                         if self.cuda else [])
        self.work = [None, None]
        self.last = None
        self.exposed_s = 0.0          # host time spent issuing / waiting on the collective (what is not overlapped)
        self._slot_ptr = [[b.data_ptr() + self.RECORD_BYTES * slot * B for slot in range(self.every)]
                          for b in self.buf]

    # ------------------------------------------------------------------ stream plumbing
    # The window is written on the env's slice streams.  The collective is issued ON slice 0's stream (made to wait
    # for the other slices first), and when a buffer comes round again every slice stream waits for the collective's
    # own completion directly.  What this avoids -- measured, one rank through RCCL: making the slice streams wait
    # on an event freshly recorded on the caller's stream behind the collective's wait (work.wait() + env.fence())
    # left hipModuleLaunchKernel at 10-45 us instead of 3 for the next dozen steps, ~3 us per step on average.
    def _writer_streams(self):
        if not self.cuda:
            return []
        return getattr(self.env, "_slice_streams", None) or [self.torch.cuda.current_stream()]

    def _issue(self, which):
        import torch.distributed as dist
        torch, dst = self.torch, (self.recv[which] if self.rank == 0 else None)
        streams = self._writer_streams()
        if not streams:                                    # CPU tensors (gloo): nothing to order
            return dist.gather(self.buf[which], dst, dst=0, group=self.group, async_op=True)
        lead = streams[0]
        for st, ev in zip(streams[1:], self._join_ev):
            ev.record(st)
            lead.wait_event(ev)
        with torch.cuda.stream(lead):
            return dist.gather(self.buf[which], dst, dst=0, group=self.group, async_op=True)

    def _wait(self, work, streams):
        """`streams` wait (stream-level) for the collective."""
        if not self.cuda:
            work.wait()
            return
        torch = self.torch
        for st in streams:
            with torch.cuda.stream(st):
                work.wait()

    def prime(self):
        """One throw-away gather of each (empty) window buffer, issued the way ``after_step`` issues it: the
        first asynchronous exchange of a process group sets up its channels and work objects and costs
        300-500 us on the host -- keep that out of the stepping loop."""
        if self.collective:
            for which in (0, 1):
                self._wait(self._issue(which), self._writer_streams())
            if self.cuda:
                self.torch.cuda.synchronize()

    def before_step(self, t):
        slot, which = t % self.every, (t // self.every) % 2
        if slot == 0 and self.work[which] is not None:
            t0 = time.perf_counter()
            self._wait(self.work[which], self._writer_streams())      # the buffer is free again
            self.work[which] = None
            self.exposed_s += time.perf_counter() - t0
        self.env.set_step_outputs(self._slot_ptr[which][slot])

    def after_step(self, t):
        if t % self.every != self.every - 1:
            return
        which = (t // self.every) % 2
        self.last = which
        if self.collective:
            t0 = time.perf_counter()
            self.work[which] = self._issue(which)
            self.exposed_s += time.perf_counter() - t0

    def flush(self):
        """Wait (stream-level, on the caller's current stream and the writers') for outstanding gathers and hand the step outputs
        back to the env's own tensor.  While a gather is active, ``env.reward`` / ``env.done`` / ``env.info`` are
        NOT written -- the records go to the window buffers; read them through ``latest()``."""
        for k in (0, 1):
            if self.work[k] is not None:        # (the writers too: their next window may reuse the buffer)
                streams = self._writer_streams()
                if self.cuda and self.torch.cuda.current_stream() not in streams:
                    streams = streams + [self.torch.cuda.current_stream()]
                self._wait(self.work[k], streams)
                self.work[k] = None
        self.env.set_step_outputs(None)

    def latest_records(self):
        """Records of the last COMPLETED window, int32 [world, every, B, 4] (the gather that filled it is
        waited for on the current stream first)."""
        if self.last is None:
            return None
        if self.work[self.last] is not None:
            self._wait(self.work[self.last], [self.torch.cuda.current_stream()] if self.cuda else [])
        elif self.cuda and getattr(self.env, "slices", 1) > 1:
            self.env.join()                                 # no collective in flight: the window sits on the slice streams
        bufs = self.recv[self.last] if self.recv is not None else [self.buf[self.last]]
        return self.torch.stack(bufs)

    def latest(self):
        rec = self.latest_records()
        if rec is None:
            return None
        torch = self.torch
        reward = rec[..., 0].view(torch.float32)
        done = rec[..., 1:2].contiguous().view(torch.uint8)[..., 0]
        return reward, done
