"""
Multi-GPU layout: one process per GPU, envs block-partitioned, no halo exchange.

Boards are independent (the torus wraps inside a board), so the step itself needs no collective.
The only cross-GPU traffic of the path is what a centralised learner on rank 0 needs from the
other ranks each step: the per-step output record of every env (``struct sl_step_out``: float32 reward, done / success /
times_up flags, episode reward and length -- 16 bytes).  xGMI is point-to-point, and a message of a
few bytes per env per step is purely latency bound, so the records of ``every`` consecutive steps
are written by the step kernel straight into one device buffer (``sl_step_out[every, B]``, a WINDOW) and
moved to rank 0 once per window, asynchronously, so that the exchange overlaps the following steps; two
buffers alternate.

Two transports:

* ``"rccl"`` (device tensors): the library's own ``slhip_gather_*`` entry points -- one RCCL group of
  ``ncclSend`` / ``ncclRecv`` per window on a side stream of this class, ordered against the streams that
  wrote the window by events only; the communicator is bootstrapped from a unique id that rank 0 broadcasts
  over the existing ``torch.distributed`` process group.  The hand-off itself is asynchronous
  (``slhip_gather_window_async``): a worker thread inside the library issues the event records, stream waits and
  the RCCL group, so a window costs the stepping thread a queue push, and it never waits on the host.  (Round 2
  handed every window to ``torch.distributed.gather``: ~100 us of host time per window, 200-350 us for the
  first, against ~2.4 us of host slack per step.)
* ``"torch"``: ``torch.distributed.gather`` -- CPU tensors over gloo (the CPU test-suite, ``bench.py
  --dry-run``), or any backend when ``SAFELIFE_GATHER_BACKEND=torch``.
"""
import ctypes as C
import os
import sys
import time

import numpy as np


def shard_bounds(total_envs, world, rank):
    """Contiguous block of global env ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_DEBUG = os.environ.get("SL_GATHER_DEBUG") == "1"


class RewardGather(object):
    """Per-step output records of a SafeLifeVectorEnv -> rank 0, batched.

    Usage per step t:  ``before_step(t); env.step(a) or env.step_async(a); after_step(t)``; ``flush()`` at the end.
    On rank 0, ``latest()`` returns (reward[world, every, B] float32, done[world, every, B] uint8) of
    the last completed window (views into the receive buffers); ``latest_records()`` the raw records.

    Which streams wrote a window is read off the env when the window closes: its slice streams after
    ``step_async()``, the caller's current stream after ``step()`` (the env orders the two against each other when a
    caller switches, so the last writer is ordered behind every earlier one).
    """

    RECORD_BYTES = 16

    def __init__(self, env, every=32, world=1, rank=0, group=None, backend=None, force=None, record="full"):
        """``record``: "full" -- sl_step_out, 16 bytes per env and step (reward, done / success / times_up, episode reward
        and length) -- or "compact": its first 8 bytes (reward and the three flags: SURVEY 5.8's record), which halves
        what every window carries to rank 0 (seven peers' worth on an 8-GPU node)."""
        import torch
        self.torch = torch
        self.env, self.every, self.world, self.rank, self.group = env, int(every), int(world), int(rank), group
        self.B = B = env.num_envs
        if record not in ("full", "compact"):
            raise ValueError("record must be 'full' or 'compact'")
        self.compact = record == "compact"
        self.RECORD_BYTES = 8 if self.compact else 16
        shape = (self.every, B, self.RECORD_BYTES // 4)
        self.cuda = torch.device(env.device).type == "cuda"
        # SAFELIFE_FORCE_GATHER=1 runs the exchange even with one rank (exercises the RCCL path on a one-GPU box)
        self.force = (os.environ.get("SAFELIFE_FORCE_GATHER", "0") == "1") if force is None else bool(force)
        self.collective = self.world > 1 or self.force     # False: one rank, nothing to gather
        if backend is None:
            backend = os.environ.get("SAFELIFE_GATHER_BACKEND") or ("rccl" if self.cuda else "torch")
        if backend not in ("rccl", "torch"):
            raise ValueError("backend must be 'rccl' or 'torch'")
        if backend == "rccl" and not self.cuda:
            raise ValueError("the rccl transport needs device tensors")
        self.backend = backend
        self.buf = [torch.zeros(shape, dtype=torch.int32, device=env.device) for _ in range(2)]
        self.recv = None
        if self.collective and self.rank == 0:
            self.recv = [torch.zeros((self.world,) + shape, dtype=torch.int32, device=env.device) for _ in range(2)]
        self.work = [None, None]      # torch transport: the outstanding collective of each buffer
        self.busy = [False, False]    # rccl transport: a gather of this buffer has been enqueued and not fenced yet
        self.last = None
        self.windows = 0              # windows handed to the transport so far
        self.exposed_s = 0.0          # host time spent issuing / waiting on the exchange (what is not overlapped)
        self._slot_ptr = [[b.data_ptr() + self.RECORD_BYTES * slot * B for slot in range(self.every)]
                          for b in self.buf]
        self._comm = None
        self.queued = False           # the caller steps through env.step_queues(): the windows are written from the
                                      # library's AQL queues, ordered against the exchange by the HOST (queues_sync /
                                      # a blocking query) instead of by stream waits
        self._join_ev = []
        if self.collective and self.backend == "rccl":
            self._init_rccl()
        elif self.cuda:
            self._join_ev = [torch.cuda.Event() for _ in (getattr(env, "_slice_streams", None) or [None])[1:]]

    # ------------------------------------------------------------------ rccl transport
    def _init_rccl(self):
        from . import _hip
        torch = self.torch
        self._lib = lib = _hip.lib()
        ident = torch.zeros(_hip.SL_GATHER_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0:
            raw = (C.c_ubyte * _hip.SL_GATHER_ID_BYTES)()
            _hip.check(lib.slhip_gather_unique_id(raw))
            ident = torch.tensor(list(raw), dtype=torch.uint8)
        if self.world > 1:
            import torch.distributed as dist
            if dist.get_backend(self.group) == "gloo":
                dist.broadcast(ident, 0, group=self.group)
            else:
                dev_ident = ident.to(self.env.device)
                dist.broadcast(dev_ident, 0, group=self.group)
                ident = dev_ident.cpu()
        raw = (C.c_ubyte * _hip.SL_GATHER_ID_BYTES)(*ident.tolist())
        comm = C.c_void_p()
        _hip.check(lib.slhip_gather_init(raw, self.world, self.rank, C.byref(comm)))
        self._comm = comm
        self._stream = torch.cuda.Stream(device=self.env.device)       # the exchange's own stream
        self._gptr = (C.c_void_p * 1)(self._stream.cuda_stream)
        self._ticket = [-1, -1]                                         # the window each buffer was last handed off as

    def free_queues(self, n_queues=4):
        """The library's step queues 0 .. n_queues - 1 that the exchange does NOT hold up (rccl transport; COLLECTIVE:
        every rank calls it).  RCCL's kernel comes from one of HIP's hardware queues, and the step queue that takes turns
        with it stands still for the length of every exchange while the others step on; a driver that steps through
        queues leaves that one out: ``env.queues_open(queue_ids=gather.free_queues()[:3])``.  Returns all of them where
        nothing is to be exchanged."""
        ids = list(range(int(n_queues)))
        if not (self.collective and self.backend == "rccl"):
            return ids
        from . import _hip
        mask = C.c_int(0)
        _hip.check(self._lib.slhip_gather_stream_shares(self._comm, int(n_queues), self._gptr[0], C.byref(mask)))
        self.shared_queues = mask.value
        return [q for q in ids if not (mask.value >> q) & 1]

    def _order(self, before, after):
        """Streams of `after` wait for what is enqueued on the streams of `before` (events, no host wait)."""
        from . import _hip
        b = (C.c_void_p * len(before))(*[s.cuda_stream for s in before])
        a = (C.c_void_p * len(after))(*[s.cuda_stream for s in after])
        _hip.check(self._lib.slhip_streams_order(b, len(before), a, len(after)))

    def close(self):
        if self._comm is not None:
            self.torch.cuda.synchronize()
            self._lib.slhip_gather_destroy(self._comm)
            self._comm = None

    # ------------------------------------------------------------------ stream plumbing
    def _writer_streams(self):
        """The streams the env's step kernels are currently enqueued on."""
        if not self.cuda:
            return []
        slices = getattr(self.env, "_slice_streams", None)
        if slices and getattr(self.env, "_async_pending", True):
            return list(slices)
        return [self.torch.cuda.current_stream()]

    def _issue(self, which):
        from . import _hip
        torch = self.torch
        self.windows += 1
        if self.queued and self.backend == "rccl" and getattr(self.env, "_queues", None) is not None:
            # the window was written from the library's AQL queues: a marker goes behind its last step now, and the
            # library's worker thread -- not this one -- waits for it before it hands the window to RCCL
            recv = self.recv[which].data_ptr() if self.rank == 0 else None
            ticket = C.c_longlong(-1)
            _hip.check(self._lib.slhip_gather_window_queued(self._comm, self.buf[which].data_ptr(), recv,
                                                            self.buf[which].numel() * 4, self.env._queues,
                                                            self._gptr[0], C.byref(ticket)))
            self._ticket[which] = ticket.value
            self.busy[which] = True
            return None
        if self.queued:
            self.env.queues_sync()     # the window is complete and visible before the exchange reads it
        streams = self._writer_streams()
        if self.backend == "rccl":
            recv = self.recv[which].data_ptr() if self.rank == 0 else None
            writers = (C.c_void_p * len(streams))(*[s_.cuda_stream for s_ in streams])
            ticket = C.c_longlong(-1)
            _hip.check(self._lib.slhip_gather_window_async(self._comm, self.buf[which].data_ptr(), recv,
                                                           self.buf[which].numel() * 4, writers, len(streams),
                                                           self._gptr[0], C.byref(ticket)))
            self._ticket[which] = ticket.value
            self.busy[which] = True
            return None
        import torch.distributed as dist
        dst = list(self.recv[which].unbind(0)) if self.rank == 0 else None
        if not streams:                                    # CPU tensors (gloo): nothing to order
            return dist.gather(self.buf[which], dst, dst=0, group=self.group, async_op=True)
        # torch transport on device tensors: issued ON the first writer stream, made to wait for the others; when the
        # buffer comes round again every writer waits for the collective's own completion
        lead = streams[0]
        for st, ev in zip(streams[1:], self._join_ev):
            ev.record(st)
            lead.wait_event(ev)
        with torch.cuda.stream(lead):
            return dist.gather(self.buf[which], dst, dst=0, group=self.group, async_op=True)

    def _wait(self, which, streams):
        """`streams` wait (stream-level) for the exchange of buffer `which`."""
        if self.backend == "rccl":
            # (a window later the exchange has long finished: one query instead of a stream wait per writer)
            # queued windows, or nobody to make wait: block on the ticket (a buffer whose exchange may still be in
            # flight is never handed back unfenced)
            if self.busy[which] and (self.queued or not streams):
                done = C.c_int(0)
                from . import _hip
                _hip.check(self._lib.slhip_gather_done(self._comm, self._ticket[which], 1, C.byref(done)))
                return
            if self.busy[which] and streams:
                done = C.c_int(0)
                from . import _hip
                _hip.check(self._lib.slhip_gather_done(self._comm, self._ticket[which], 0, C.byref(done)))
                if not done.value:
                    arr = (C.c_void_p * len(streams))(*[s_.cuda_stream for s_ in streams])
                    _hip.check(self._lib.slhip_gather_wait_streams(self._comm, self._ticket[which], arr, len(streams)))
            return
        work = self.work[which]
        if work is None:
            return
        if not self.cuda:
            work.wait()
            return
        torch = self.torch
        for st in streams:
            with torch.cuda.stream(st):
                work.wait()
            if self.queued:
                st.synchronize()

    def prime(self):
        """One throw-away exchange of each (empty) window buffer: the first one of a communicator sets up its
        channels (hundreds of microseconds on the host) -- keep that out of the stepping loop."""
        if self.collective:
            for which in (0, 1):
                self.work[which] = self._issue(which)
                self._wait(which, self._writer_streams())
                self.work[which], self.busy[which] = None, False
            self.windows = 0
            if self.cuda:
                self.torch.cuda.synchronize()

    def before_step(self, t):
        slot, which = t % self.every, (t // self.every) % 2
        if slot == 0 and (self.work[which] is not None or self.busy[which]):
            t0 = time.perf_counter()
            self._wait(which, self._writer_streams())      # the buffer is free again
            self.work[which], self.busy[which] = None, False
            self.exposed_s += time.perf_counter() - t0
        self.env.set_step_outputs(self._slot_ptr[which][slot], compact=self.compact)

    def after_step(self, t):
        if t % self.every != self.every - 1:
            return
        which = (t // self.every) % 2
        self.last = which
        if self.collective:
            t0 = time.perf_counter()
            self.work[which] = self._issue(which)
            self.exposed_s += time.perf_counter() - t0

    def run_queued(self, t0, n, action_ptr, action_stride, shift=0, assume_ordered=False):
        """Steps t0 .. t0+n-1 through ``env.step_queues_many``: whole windows (or what is left of one) per call to the
        library, the records of step t going to slot (t + shift) % every of its window, and a window handed to the
        exchange right behind the call that completes it.  `action_ptr`: device address of step t0's actions, steps
        `action_stride` int32 elements apart.  ``assume_ordered``: as ``SafeLifeVectorEnv.step_queues``."""
        env, B = self.env, self.B
        # The windows of this call are written from the library's AQL queues: the exchange must be ordered against THEM
        # (a marker behind the window's last step, waited for by the library's worker), not against HIP streams that
        # never wrote the window -- so this entry point switches the gather to its queued mode itself instead of
        # trusting the caller to have set the attribute.
        if getattr(env, "_queues", None) is None:
            raise RuntimeError("run_queued: the env has no open step queues (env.queues_open() first)")
        self.queued = True
        if not self.collective:             # one rank: the records stay in the env's own tensor
            env.step_queues_many(action_ptr, n, action_stride, assume_ordered=assume_ordered)
            return
        if self.backend == "rccl":
            self._lib.slhip_gather_poke(self._comm)         # the worker wakes up now, not when the window closes
        t = t0
        while t < t0 + n:
            tt = t + shift
            slot, which = tt % self.every, (tt // self.every) % 2
            seg = min(self.every - slot, t0 + n - t)
            if slot == 0 and (self.work[which] is not None or self.busy[which]):
                w0 = time.perf_counter()
                self._wait(which, [])                       # the buffer is free again (the exchange of two windows ago)
                self.work[which], self.busy[which] = None, False
                self.exposed_s += time.perf_counter() - w0
            env.set_step_outputs(self._slot_ptr[which][slot], compact=self.compact)
            d0 = time.perf_counter()
            env.step_queues_many(action_ptr + 4 * action_stride * (t - t0), seg, action_stride, out_stride=B,
                                 assume_ordered=assume_ordered)
            d1 = time.perf_counter()
            if slot + seg == self.every:
                self.last = which
                w0 = time.perf_counter()
                self.work[which] = self._issue(which)
                self.exposed_s += time.perf_counter() - w0
            if _DEBUG:
                print("run_queued: %d steps enqueued in %.1f us, hand-over %.1f us" % (seg, (d1 - d0) * 1e6,
                      (time.perf_counter() - d1) * 1e6), file=sys.stderr)
            t += seg

    def flush(self):
        """Wait (stream-level, on the caller's current stream and the writers') for outstanding exchanges and hand
        the step outputs back to the env's own tensor.  While a gather is active, ``env.reward`` / ``env.done`` /
        ``env.info`` are NOT written -- the records go to the window buffers; read them through ``latest()``."""
        for k in (0, 1):
            if self.work[k] is not None or self.busy[k]:        # (the writers too: their next window may reuse the buffer)
                streams = self._writer_streams()
                if self.cuda and self.torch.cuda.current_stream() not in streams:
                    streams = streams + [self.torch.cuda.current_stream()]
                self._wait(k, streams)
                self.work[k], self.busy[k] = None, False
        self.env.set_step_outputs(None)

    def latest_records(self):
        """Records of the last COMPLETED window, int32 [world, every, B, 4] ("compact": [..., 2]) (the exchange that
        filled it is waited for on the current stream first)."""
        if self.last is None:
            return None
        if self.work[self.last] is not None or self.busy[self.last]:
            self._wait(self.last, [self.torch.cuda.current_stream()] if self.cuda else [])
        elif self.queued:
            self.env.queues_sync()
        elif self.cuda and getattr(self.env, "slices", 1) > 1:
            self.env.join()                                 # no exchange in flight: the window sits on the slice streams
        if self.recv is not None:
            return self.recv[self.last]
        return self.buf[self.last][None]

    def latest(self):
        rec = self.latest_records()
        if rec is None:
            return None
        torch = self.torch
        reward = rec[..., 0].view(torch.float32)
        done = rec[..., 1:2].contiguous().view(torch.uint8)[..., 0]
        return reward, done
