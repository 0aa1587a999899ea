#!/usr/bin/env python3
"""
bench.py -- env-steps/s of the batched SafeLife step() on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): 8192 envs x 25x25 per GPU,
levels from the reference's `prune-still` procgen (fixture pool, cycled), uniform random actions,
full step = execute_actions + advance_board + exit colours + score/reward/done + episode
accounting + on-device auto-reset.  One "step" = every env of a rank stepped once (four slice
launches of the fused kernel on the library's AQL queues).  Inputs are resident in HBM before the
timed region, which starts with the envs spread over their 1000-step episodes (episode
ends -- about envs/1000 per step -- and their in-kernel resets are INSIDE it; --spread 0: rounds
1-5's region straight behind a reset of all envs).  Weak scaling: every rank owns 8192 envs; the only
cross-GPU traffic is the gather of 8-byte (reward, done, success, times_up) records to rank 0,
batched every --gather-every steps on a side stream.

Output: one JSON line on rank 0 (see README / the driver contract), with
  roofline     -- algorithmic HBM bytes per step / ms_per_step against the 8 TB/s HBM3E peak, plus -- as
                  short numeric fields, measured in the same run behind the timed region -- the same
                  region under the library's default fences, without episode ends, 400 steps long,
                  unstaged, with a forced one-rank exchange, and C5 / the observation variants
  cpu_baseline -- the oracle (plain-C restatement, OpenMP over envs) timed on this host on a bounded
                  sample of the same workload, and the bit-for-bit replay of the timed run on it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# kernel arguments in device memory: read by the runtime when it loads, i.e. before torch is imported
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAIN_CHANNELS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27)   # env_factory.py:301-327


def load_pool(name, counts_fn, n=None):
    from safelife_amd.levels import Level, LevelPool
    path = os.path.join(REPO, "tests", "golden", "pool_%s.npz" % name)
    with np.load(path) as d:
        L = int(d["n_levels"]) if n is None else min(int(n), int(d["n_levels"]))
        levels = [Level(d["board"][k], d["goals"][k], d["agent_locs"][k],
                        spawn_prob=float(d["spawn_prob"][k]), min_performance=float(d["min_performance"][k]),
                        points_table=d["points_table"][k], rng_words=d["rng"][k]) for k in range(L)]
    return LevelPool(levels, counts_fn=counts_fn)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def spread_num_steps(n_envs, first=0, time_limit=1000):
    """The episode phase every env of the timed run starts in (C3 is defined with episode ends and resets
    inside the run): env e has played (e * 997) mod time_limit steps of its first episode, so ~n_envs / time_limit
    envs reach their time limit -- and load their next level inside the step kernel -- at EVERY step, as in a training
    run's steady state, instead of none for the first 1000 steps."""
    return ((np.arange(first, first + n_envs, dtype=np.int64) * 997) % time_limit).astype(np.int32)


def parity_replay(pool, actions, n_envs, device_env, checkpoints, threads, spread=True):
    """Checker use of the oracle inside the cpu_baseline leg ('all boards for B <= 8192, every step
    for the first K steps and at the end'): replay ALL envs of the measured run on the CPU -- same levels, same
    action stream, every step since the reset.  `checkpoints`: {step count: device state snapshot taken after
    that many steps of the measured run} for the first steps; the device's final state is compared at the end."""
    import oracle
    from safelife_amd.levels import empty_env_arrays
    arrays = empty_env_arrays(pool, n_envs)
    arrays["level_idx"][:] = np.arange(n_envs) % len(pool)
    env = oracle.OracleEnv(arrays, time_limit=1000, auto_reset=True, level_stride=1, view_shape=(25, 25),
                           output_channels=TRAIN_CHANNELS, with_obs=False, stream_salt=1)
    env.reset()
    if spread:
        arrays["num_steps"][:] = spread_num_steps(n_envs)
    names = ("board", "goals", "agent_loc", "rng", "num_steps", "episode_idx", "episode_length", "level_idx")
    ok, checked = True, []
    for t, a in enumerate(actions):
        env.step(np.ascontiguousarray(a[:n_envs]), n_threads=threads)
        snap = checkpoints.get(t + 1)
        if snap is not None:
            ok = ok and all(bool(np.array_equal(device_env.numpy(name, snapshot=snap)[:n_envs], arrays[name]))
                            for name in ("board", "rng", "agent_loc", "num_steps", "episode_idx"))
            checked.append(t + 1)
    for name in names:
        ok = ok and bool(np.array_equal(device_env.numpy(name)[:n_envs], arrays[name]))
    return {"envs": n_envs, "steps": len(actions), "every_step_until": max(checked) if checked else 0,
            "and_at_the_end": True, "bit_exact": ok,
            "episodes_ended_in_the_run": int((arrays["episode_idx"] > 0).sum()) if "episode_idx" in arrays else None}


def cpu_baseline(pool, envs, steps, seed):
    """Oracle (CPU checker) timed on the same workload; bounded sample."""
    import oracle
    from safelife_amd.levels import empty_env_arrays
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    arrays = empty_env_arrays(pool, envs)
    arrays["level_idx"][:] = np.arange(envs) % len(pool)
    env = oracle.OracleEnv(arrays, time_limit=1000, auto_reset=True, level_stride=1,
                           view_shape=(25, 25), output_channels=TRAIN_CHANNELS, with_obs=False, stream_salt=1)
    env.reset()
    arrays["num_steps"][:] = spread_num_steps(envs)
    rng = np.random.default_rng(seed)
    acts = rng.integers(0, 9, (steps, envs)).astype(np.int32)
    env.step(acts[0], n_threads=threads)            # warm
    t0 = time.perf_counter()
    for t in range(1, steps):
        env.step(acts[t], n_threads=threads)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    n1 = max(1, (steps - 1) // 8)
    for t in range(n1):
        env.step(acts[t], n_threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": envs * (steps - 1) / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
        "sample": "%d envs x %d steps of the same 25x25 prune-still workload (oracle/sl_oracle.c, OpenMP "
                  "over envs, no observation) on %s; single thread: %.3g env-steps/s" % (
                      envs, steps - 1, cpu_model(), envs * n1 / dt1),
    }


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run (what the
    driver's own command line does) and hand their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class _StandInEnv(object):
    """--dry-run: what RewardGather needs of an env, on the CPU.  A 'step' writes one record per env through the
    pointer the real kernel would be given."""

    def __init__(self, B, rank):
        import ctypes
        import torch
        self.num_envs, self.rank, self.device, self.slices = B, rank, torch.device("cpu"), 1
        self.own = np.zeros((B, 4), np.int32)
        self._ct = ctypes
        self.set_step_outputs(None)

    def set_step_outputs(self, out_ptr, compact=False):
        self.out_ptr = self.own.ctypes.data if out_ptr is None else int(out_ptr)

    def step_async(self, t):
        ct = self._ct
        rec = np.ctypeslib.as_array(ct.cast(self.out_ptr, ct.POINTER(ct.c_int32)), (self.num_envs, 4))
        rec[:, 0] = np.float32(self.rank * 1000 + t).view(np.int32)
        rec[:, 1] = (t % 7 == 0)


def dry_run(args):
    """The multi-process plumbing of the bench without a GPU (CPU test-suite): process group over gloo, env
    partition, windowed gather to rank 0 with a window closing inside the timed region, max-over-ranks timing, one
    JSON line from rank 0.  The step is a stand-in; the line says so and carries no performance claim."""
    import torch
    import torch.distributed as dist
    from safelife_amd.sharding import RewardGather, shard_bounds
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
    B, K, W = min(args.envs, 256), args.steps, args.warmup
    every = gather_window(args.gather_every, K) if world > 1 else args.gather_every
    env = _StandInEnv(B, rank)
    gather = RewardGather(env, every=every, world=world, rank=rank, backend="torch")
    gather.prime()
    for t in range(W):
        gather.before_step(t)
        env.step_async(t)
        gather.after_step(t)
    gather.flush()
    if world > 1:
        dist.barrier()
    w0, x0 = gather.windows, gather.exposed_s
    t_start = time.perf_counter()
    for t in range(W, W + K):
        gather.before_step(t)
        env.step_async(t)
        gather.after_step(t)
    gather.flush()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    windows, exposed = gather.windows - w0, gather.exposed_s - x0
    ok = True
    if rank == 0 and gather.latest() is not None:
        rw, _ = gather.latest()
        last = W + K - 1 - (W + K) % every          # last step of the last complete window
        ok = all(float(rw[r, -1, 0]) == float(r * 1000 + last) for r in range(world))
    per_rank = None
    if world > 1:
        mine = torch.tensor([elapsed, exposed], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "elapsed_ms": float(v[0]) * 1e3, "gather_exposed_ms": float(v[1]) * 1e3}
                    for r, v in enumerate(allr)]
        elapsed = max(float(v[0]) for v in allr)
    if rank == 0:
        lo, hi = shard_bounds(world * B, world, world - 1)
        print(json.dumps({
            "metric": "env steps/sec (whole node), 8192x25x25 boards; bit-exact vs C advance_board",
            "dry_run": True, "value": world * B * K / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "none (stand-in step on the CPU: plumbing check only)",
            "config": {"workload": "dry run: %d stand-in envs per rank" % B, "global_envs": world * B,
                       "last_rank_envs": [lo, hi]},
            "gather_every": every, "gather_windows_in_region": windows, "gather_exposed_ms": exposed * 1e3,
            "gather_ok": ok, "per_rank": per_rank, "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def gather_window(requested, steps):
    """Steps per gather window for N > 1: at least one window must CLOSE inside the timed region, so that the
    exchange is part of what is measured (the driver times 20 steps)."""
    return max(1, min(int(requested), int(steps)))     # (any `every` <= steps consecutive steps hold a step = every-1 mod every)


DEFAULT_QUEUES = "4"       # slices on AQL queues by default (0: HIP-stream slices); see DESIGN 4.1c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--envs", type=int, default=8192, help="envs per GPU")
    ap.add_argument("--pool", default="prune_still_25")
    ap.add_argument("--obs", type=int, default=0,
                    help="1: also write the 25x25x15 uint8 observation; 2: the raw 25x25 uint32 view")
    ap.add_argument("--gather-every", type=int, default=64,
                    help="steps per gather window (one RCCL gather of the window to rank 0 when --gpus > 1).  Handing a "
                         "window to torch.distributed costs the stepping thread ~100 us all told; a step leaves ~2.5 us of "
                         "host slack, so 64 steps hide it and 32 do not (DESIGN.md section 5)")
    ap.add_argument("--slices", type=int, default=2,
                    help="slices of the per-GPU batch, each stepped by its own launch on its own stream "
                         "(1 = one launch per step on one stream)")
    ap.add_argument("--queues", type=int, default=-1,
                    help="N > 0: issue a step as N slices on the library's own AQL queues (slhip_queues_*: the same kernel, "
                         "the same ordering as a stream, without HIP's per-launch host cost); 0: the HIP-stream slices of "
                         "--slices; -1: SAFELIFE_BENCH_QUEUES or the default below, falling back to streams where the "
                         "runtime offers no queue")
    ap.add_argument("--queue-fences", choices=("none", "agent"), default=os.environ.get("SAFELIFE_QUEUE_FENCES") or "none",
                    help="queue stepping: 'agent' = every step with a stream's agent-scope acquire AND release (the "
                         "library's default); 'none' = this bench OPTS IN to release-free stepping "
                         "(SL_QUEUES_RELEASE_FREE: ~0.9 us per step faster; valid only while a workgroup index keeps its "
                         "XCD, which the library probes at open and every step verifies -- on a violation the run is "
                         "repeated with 'agent', on every rank).  config.queue_fences says which mode produced the line")
    ap.add_argument("--spread", type=int, default=1,
                    help="1 (default): the timed region starts with the envs spread evenly over their 1000-step episodes, so "
                         "that ~envs/1000 episodes end -- and reset inside the kernel -- at every step (SURVEY 8d); 0: straight "
                         "behind a reset of all envs (no episode end within the first 1000 steps: rounds 1-5's region)")
    ap.add_argument("--stage", type=int, default=int(os.environ.get("SAFELIFE_BENCH_STAGE", "1")),
                    help="queue stepping on one GPU: 1 = the timed steps (up to 48 of them) are STAGED before the clock starts -- "
                         "argument blocks and packets written, nothing handed to the device (slhip_queues_stage) -- and the timed "
                         "region opens with slhip_queues_go (one doorbell per queue), as a captured hipGraph would be launched; "
                         "0 = everything is enqueued inside the region (roofline.unstaged_us either way)")
    ap.add_argument("--stream-leg", type=int, default=1,
                    help="queue stepping only: 1 = also run K steps of the same kernel through the stream slices under HIP "
                         "events (roofline.stream_leg_*); 0 = leave it out (profiling runs: only the queues' launches in the trace)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=1001)
    ap.add_argument("--rollout", type=int, default=32,
                    help="T>0: additionally time T-step fused rollouts (reported under 'extra')")
    ap.add_argument("--extras", type=int, default=1,
                    help="1: also time the step with observations (reported under 'extra'; 1 GPU only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo process group and a stand-in step (checks the launcher, the partition, the "
                         "gather windows and the JSON line)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.dry_run:
        raise SystemExit(dry_run(args))

    import torch
    import torch.distributed as dist
    from safelife_amd import _hip
    from safelife_amd.levels import _device_counts
    from safelife_amd.vector_env import SafeLifeVectorEnv
    from safelife_amd.sharding import RewardGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dev = _hip.device()
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    B, K, W = args.envs, args.steps, args.warmup
    pool = load_pool(args.pool, _device_counts)
    H, Wd = pool.shape
    env = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25),
                            output_channels=None if args.obs == 2 else TRAIN_CHANNELS,
                            auto_reset=True, level_stride=1, env_offset=rank * B, with_obs=bool(args.obs),
                            slices=args.slices)
    n_queues = args.queues if args.queues >= 0 else int(os.environ.get("SAFELIFE_BENCH_QUEUES", DEFAULT_QUEUES))
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)
    # P checkpointed steps for the parity replay, then W warm-up steps, then the K timed ones
    P = 8 if (args.cpu_baseline and world == 1) else 0
    # (rows beyond P + W + K: the K = 400 repetition of the region reported as roofline.k400_us)
    n_rows = max(P + W + K, 440 if (args.extras and world == 1) else 0)
    actions = torch.randint(0, 9, (n_rows, B), generator=gen, device=dev, dtype=torch.int32)
    act_ptr = [actions[t].data_ptr() for t in range(n_rows)]
    # N > 1: the window is cut so that at least one closes -- one RCCL exchange is issued -- inside the timed steps
    forced = os.environ.get("SAFELIFE_FORCE_GATHER", "0") == "1"         # one rank, exchange on (RCCL to itself)
    every_used = gather_window(args.gather_every, K) if (world > 1 or forced) else args.gather_every
    # (the 8-byte record -- reward, done, success, times_up: what a learner on rank 0 needs of every step -- not the whole
    #  16-byte sl_step_out: half the bytes per window)
    gather = RewardGather(env, every=every_used, world=world, rank=rank, record="compact")
    gather.prime()
    every = gather.every
    # windows are counted from the END of the timed block: its last step closes one (a learner that consumes K-step
    # rollouts exchanges once per rollout), so the hand-off falls where the host is ahead of the device instead of
    # in the middle of the launches
    shift = (every - (P + W + K) % every) % every
    # ... some steps ahead of the end, so that the exchange runs under the last steps: two for stream slices (RCCL's
    # send / receive kernel and its hand-shake take ~25 us on the device); ten (at most half a window) for queue
    # stepping, where the chain behind the window's last step is longer -- the marker's system-scope release, the
    # worker's RCCL group, its kernel next to a full chip, the completion event: ~70-100 us, and the closing
    # synchronize of a stream that has just run something costs another ~40 (round 4: 11-13 us per step with the
    # window closing five steps before the end of a 20-step region, 9.0-9.2 with ten)
    window_ahead = 0
    if K >= 8 and every >= 8:
        window_ahead = int(os.environ.get("SAFELIFE_BENCH_WINDOW_AHEAD", str(min(10, every // 2)) if n_queues > 0 else "2"))
        shift = (shift + window_ahead) % every
    dbg = os.environ.get("SL_BENCH_DEBUG") == "1"
    import gc

    queue_ids = [None]

    def attempt(fences, gather=gather, P=P, shift=shift, spread=bool(args.spread), K=K, W=W, stage=bool(args.stage)):
        """Reset, P checkpointed steps, W warm-up steps, the K timed steps.  Returns the measurements, or None when
        release-free queue stepping was refused by its placement check on ANY rank (the caller repeats with 'agent').
        (`gather`, `P`, `shift`, `spread`, `K`, `W`, `stage`: the variants reported in `roofline` repeat the region
        with another exchange, regime or length, without checkpoints.)"""
        every = gather.every
        res = {"use_queues": False, "queues_why": "switched off", "fences": None, "queue_slices": None,
               "queue_ids": None, "spread": spread, "K": K, "W": W, "staged": 0}
        env.queues_close()
        env.reset()
        if spread:
            env.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = torch.from_numpy(spread_num_steps(B, first=rank * B)).to(dev)
            torch.cuda.synchronize()
        if n_queues > 0:
            try:
                # N > 1 (or the forced one-rank exchange): the step queue that RCCL's kernel would hold up is left
                # out -- three slices on the three queues the exchange does not touch (probed once, collectively)
                if os.environ.get("SAFELIFE_BENCH_QUEUE_IDS"):        # (experiments: slices on exactly these queues)
                    queue_ids[0] = [int(x) for x in os.environ["SAFELIFE_BENCH_QUEUE_IDS"].split(",")]
                if gather.collective and n_queues == 4 and queue_ids[0] is None:
                    try:
                        free = gather.free_queues(4)
                    except _hip.SafeLifeHipError as e:
                        print("bench: gather_stream_shares failed (%s)" % e, file=sys.stderr)
                        free = [0, 1, 2, 3]
                    queue_ids[0] = free[:3] if 3 <= len(free) < 4 else [0, 1, 2, 3]
                ids = queue_ids[0] if ((gather.collective and n_queues == 4) or os.environ.get("SAFELIFE_BENCH_QUEUE_IDS")) else None
                # (recover=False: this bench has its own answer to a refused placement -- the whole run again with a
                #  stream's fences -- and keeps the env's per-sync state copy out of the timed region)
                env.queues_open(len(ids) if ids else n_queues, release_free=(fences == "none"), queue_ids=ids, recover=False)
                res["use_queues"], res["queues_why"] = True, None
                res["fences"] = "none" if env.queue_release_free else "agent"
                # what THIS attempt steps with (the JSON line is built from these, not from whatever a later attempt
                # leaves in `env`)
                res["queue_slices"], res["queue_ids"] = int(env.queue_slices), list(env.queue_ids)
                if fences == "none" and not env.queue_release_free:
                    print("bench: release-free queue stepping not granted (%s): agent-scope fences" % env.queue_mode_note,
                          file=sys.stderr)
            except _hip.SafeLifeHipError as e:
                if args.queues > 0:
                    raise
                res["queues_why"] = str(e)
                print("bench: AQL queues unavailable (%s): stepping through HIP streams" % e, file=sys.stderr)
        use_queues = res["use_queues"]
        gather.queued = use_queues
        refused = [False]

        def guarded_sync():
            try:
                env.queues_sync()
            except _hip.SafeLifeHipError as e:
                if "another XCD" not in str(e):
                    raise
                print("bench: %s" % e, file=sys.stderr)
                refused[0] = True

        def run(t0, n, ordered=True):
            # one step = every env stepped once = one dispatch per slice; the action tensor is complete before the
            # loop starts, so nothing has to be fenced per step
            if use_queues:
                # ALL n steps (and their windows) are enqueued by the library in one call per window: the stepping
                # thread is out of the loop (slhip_queues_steps).  ordered: the device has just been synchronised and
                # no stream holds work on the envs (the warm-up and the timed region); the checkpointed steps are
                # not -- their snapshots are still being copied on a stream -- and wait for the device themselves
                # ("untouched": between the sync that closed the steps before and this call nothing has written the envs,
                #  these actions or the outputs -- the first step needs no more of an acquire than any other)
                gather.run_queued(t0, n, act_ptr[t0], B, shift, assume_ordered="untouched" if ordered else False)
                return
            if not gather.collective:        # one rank: records stay in the env's own tensor, no windows to rotate
                for t in range(t0, t0 + n):
                    env.step_async(act_ptr[t])
                return
            for t in range(t0, t0 + n):
                gather.before_step(t + shift)
                env.step_async(act_ptr[t])
                if (t + shift) % every == every - 1:
                    gather.after_step(t + shift)

        # the first steps since the reset run one at a time, with a snapshot of the device state after each
        # (untimed; the CPU replay of the cpu_baseline leg compares every one of them)
        checkpoints = {}
        torch.cuda.synchronize()
        for t in range(P):
            run(t, 1, ordered=False)
            if use_queues:
                guarded_sync()
            checkpoints[t + 1] = env.snapshot()     # device-side copies: nothing crosses to the host before the timed region
        # (no garbage-collector pass inside the timed region: with ~20 steps in it, one young-generation pass of the
        #  interpreter shows up as +1 us per step; timeit does the same)
        gc.collect()
        gc.disable()
        torch.cuda.synchronize()
        run(P, W)              # the W untimed warm-up steps, issued exactly like the timed ones
        gather.flush()
        if use_queues:
            guarded_sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        # HIP events on the stream the kernels are launched on.  With slices, every slice stream runs the same K
        # launches concurrently; the pair sits on slice 0's stream only: an event record as the LAST command of a
        # stream makes the closing synchronize ~15 us slower for that stream, and one such stream is enough.
        # (queue stepping: nothing of the timed region runs on a HIP stream, so no event pair goes onto one -- a marker
        #  on an otherwise idle stream costs the closing synchronize ~20 us; the pair is used further down, around the
        #  stream-path run of the same kernel)
        streams = env._slice_streams or [torch.cuda.current_stream()]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))]
        if not use_queues:
            evs[0][0].record(streams[0])
        windows0, exposed0 = gather.windows, gather.exposed_s
        # One GPU, queue stepping: the region's steps are STAGED before the clock starts (argument blocks and packets
        # written, nothing handed over: slhip_queues_stage) and the region opens with slhip_queues_go -- host-side
        # encoding ahead of time, every step's execution inside the bracket (a captured graph's instantiate / launch)
        staged = 0
        if stage and use_queues and not gather.collective and hasattr(env, "queues_go"):
            staged = min(K, _hip.QUEUES_STAGE_MAX)
            env.step_queues_many(act_ptr[P + W], staged, B, assume_ordered="untouched", defer=True)
        res["staged"] = staged
        t_start = time.perf_counter()
        if staged:
            env.queues_go()
        if K > staged:
            run(P + W + staged, K - staged)
        t_enqueued = time.perf_counter()
        if not use_queues:
            evs[0][1].record(streams[0])
        # (completion is left to the synchronize below: polling hipStreamQuery / hipEventQuery first was measured
        #  10-25 us slower over the region -- the polls contend with the runtime's own completion handling)
        t_b = time.perf_counter()
        gather.flush()
        t_c = time.perf_counter()
        if use_queues:
            # The closing bracket is BOTH waits: torch.cuda.synchronize() (HIP's streams: idle in queue mode, apart from
            # the exchange's) and the queues' fence (a system-scope release behind the last step, waited for on its
            # completion signals).  The clock stops when both have returned, i.e. every step has completed and is
            # visible; the stream-side wait -- ~5 us of runtime time even when there is nothing to wait for -- comes
            # first so that it passes while the queues are still stepping.
            torch.cuda.synchronize()
            guarded_sync()
        else:
            torch.cuda.synchronize()
        t_d = time.perf_counter()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        gc.enable()
        if world > 1:
            bad = torch.tensor([1 if refused[0] else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            refused[0] = bool(bad.item())
        if refused[0]:
            return None
        if dbg:
            print("timeline us: enqueue %.1f | e1 record %.1f | flush %.1f | synchronize + queues_sync %.1f | barrier %.1f | total %.1f"
                  % ((t_enqueued - t_start) * 1e6, (t_b - t_enqueued) * 1e6, (t_c - t_b) * 1e6, (t_d - t_c) * 1e6,
                     (t_start + elapsed - t_d) * 1e6, elapsed * 1e6), file=sys.stderr)
        res.update(elapsed=elapsed, t_start=t_start, t_enqueued=t_enqueued, checkpoints=checkpoints, evs=evs, streams=streams,
                   gather_windows=gather.windows - windows0, gather_exposed=gather.exposed_s - exposed0)
        return res

    res = attempt(args.queue_fences)
    if res is None:
        print("bench: repeating the run with agent-scope fences", file=sys.stderr)
        res = attempt("agent")
        if res is None:
            raise SystemExit("bench: placement check failed with agent-scope fences -- cannot happen")
    if res["use_queues"]:
        assert res["queue_slices"] == len(res["queue_ids"]), (res["queue_slices"], res["queue_ids"])
    use_queues, queues_why = res["use_queues"], res["queues_why"]
    elapsed, t_start, t_enqueued = res["elapsed"], res["t_start"], res["t_enqueued"]
    checkpoints, evs, streams = res["checkpoints"], res["evs"], res["streams"]
    gather_windows, gather_exposed = res["gather_windows"], res["gather_exposed"]
    # device time per step: every slice stream runs its K launches back to back, all streams concurrently
    if use_queues:      # (events on a HIP stream see nothing of the queues' steps: filled in below)
        kernel_ms = elapsed / K * 1e3
    else:
        slice_ms = [e0.elapsed_time(e1) / K for e0, e1 in evs]
        kernel_ms = max(slice_ms)
    per_rank = None
    if world > 1:
        # per-rank breakdown for the scaling run: wall time of the region, device time per step, host enqueue time
        # per step, and the time the rank spent blocked in the gather (its exposed part)
        mine = torch.tensor([elapsed, kernel_ms, (t_enqueued - t_start) / K * 1e3, gather_exposed], device=dev,
                            dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "elapsed_ms": float(v[0]) * 1e3, "device_ms_per_step": float(v[1]),
                     "host_enqueue_ms_per_step": float(v[2]), "gather_exposed_ms": float(v[3]) * 1e3}
                    for r, v in enumerate(allr)]
        elapsed = max(float(v[0]) for v in allr)

    parity = None
    if args.cpu_baseline and world == 1:
        # the state the timed launches left behind against a CPU replay of the same envs and actions
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
        parity = parity_replay(pool, actions[:P + W + K].cpu().numpy(), B, env, checkpoints, threads, spread=res["spread"])
    stream_wall_ms = None
    if use_queues and args.stream_leg:
        # the timed steps ran on the library's queues, not on a HIP stream.  The same kernel through the stream
        # slices, K steps under HIP events, gives the per-launch device figure (roofline.stream_leg_launch_ms) next to the
        # queues' wall clock.
        torch.cuda.synchronize()
        for t in range(P, P + W):
            env.step_async(act_ptr[t])
        env.join()
        torch.cuda.synchronize()
        evs[0][0].record(streams[0])
        t_s0 = time.perf_counter()
        for t in range(P + W, P + W + K):
            env.step_async(act_ptr[t])
        evs[0][1].record(streams[0])
        env.join()
        torch.cuda.synchronize()
        stream_wall_ms = (time.perf_counter() - t_s0) / K * 1e3
        kernel_ms = evs[0][0].elapsed_time(evs[0][1]) / K

    extra = {}
    variants = {}       # short numeric fields of `roofline` (the driver keeps `roofline` and drops `extra`)
    if use_queues and args.extras and world == 1 and not gather.collective:
        # What the driver's one-GPU line does not show (outside the timed region; same warm-up, no checkpoints; every
        # variant the median of three repetitions of the whole region):
        # (a) the library's DEFAULT fences -- what SafeLifeVectorEnv.queues_open() gives a user;
        # (c) the region straight behind a reset of all envs -- no episode end inside it (rounds 1-5's region);
        # (d) the same region 400 steps long: what the fixed cost of a 20-step region (first doorbell, the queues'
        #     staggered pick-up, the closing fence) hides;
        # (e) the step as every N > 1 run takes it: the RCCL exchange on (one rank, to itself), three slices on the three
        #     queues the exchange does not hold up, one window closing inside the region.
        def median_us(n=3, **kw):
            fences = kw.pop("fences", res["fences"])
            kw.setdefault("spread", res["spread"])
            vals = []
            for _ in range(n):
                r = attempt(fences, P=0, **kw)
                if r is None or r["fences"] != fences:
                    return None
                vals.append(r["elapsed"] / r["K"] * 1e6)
            return sorted(vals)[len(vals) // 2]
        try:
            variants["agent_fences_us"] = median_us(fences="agent") if res["fences"] != "agent" else elapsed / K * 1e6
            extra["queue_fences_agent_us_per_step"] = variants["agent_fences_us"]
            variants["no_reset_us" if res["spread"] else "steady_state_us"] = median_us(spread=not res["spread"])
            variants["k400_us"] = median_us(K=400, W=40)
            if res["staged"]:
                variants["unstaged_us"] = median_us(stage=False)
            variants["k20_median_us"] = median_us(n=5)
            env.queues_close()
            g2 = RewardGather(env, every=gather_window(args.gather_every, K), world=1, rank=0, force=True, record="compact")
            g2.prime()
            shift2 = (g2.every - (W + K) % g2.every) % g2.every
            if K >= 8 and g2.every >= 8:
                shift2 = (shift2 + min(10, g2.every // 2)) % g2.every
            queue_ids[0] = None
            r3 = attempt(res["fences"], gather=g2, P=0, shift=shift2)
            if r3 is None:
                r3 = attempt("agent", gather=g2, P=0, shift=shift2)
            if r3 is not None:
                extra["forced_gather_us_per_step"] = variants["forced_gather_us"] = r3["elapsed"] / K * 1e6
                extra["forced_gather_queue_ids"] = r3["queue_ids"]
                extra["forced_gather_windows_in_region"] = r3["gather_windows"]
                extra["forced_gather_fences"] = r3["fences"]
                extra["forced_gather_note"] = ("the same %d-step region with the RCCL exchange forced on for one rank "
                                               "(send / receive to itself), windows of %d steps, the step queue the "
                                               "exchange's kernel shares a hardware pipe with left out: what --gpus N > 1 "
                                               "runs per rank" % (K, g2.every))
            env.queues_close()
            g2.flush()
            g2.close()
            queue_ids[0] = None
        except Exception as e:          # noqa: BLE001  (an extra: reported, never fatal to the headline line)
            extra["forced_gather_error"] = "%s: %s" % (type(e).__name__, e)
        env.set_step_outputs(None)
    if args.rollout > 0:
        try:
            T = args.rollout
            reps = max(1, K // T)
            a = actions[:T].contiguous()
            env.rollout(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                env.rollout(a)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            extra["rollout_T"] = T
            extra["rollout_env_steps_per_s_per_gpu"] = B * T / (ms * 1e-3)
            extra["rollout_us_per_step"] = ms * 1e3 / T
        except Exception as e:          # noqa: BLE001
            extra["rollout_error"] = "%s: %s" % (type(e).__name__, e)
    if args.extras and world == 1:
        # (an extra that fails must not take the headline line with it: the error is reported in its place)
        try:
            # the same step with the two observation formats of the reference (SafeLifeEnv.output_channels)
            for tag, chans in (("obs_u8_25x25x15", TRAIN_CHANNELS), ("obs_u32_view_25x25", None)):
                env2 = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=chans,
                                         auto_reset=True, level_stride=1, with_obs=True)
                env2.reset()
                for t in range(20):
                    env2.step(actions[t])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = min(K, 200)
                e0.record()
                for t in range(n):
                    env2.step(actions[t])
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                extra[tag + "_us_per_step"] = us
                extra[tag + "_env_steps_per_s"] = B / (us * 1e-6)
                obs_b = H * Wd * (len(chans) if chans else 4)           # step bytes + observation bytes
                extra[tag + "_roofline_frac"] = (3 * H * Wd * 2 + obs_b) * B / (us * 1e-6) / (HBM_PEAK_GBS * 1e9)
                del env2

            # f4: the observation written by the step kernel in the policy network's layout ([B,C,W,H] uint8), no (h,w,c) tensor
            env2 = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=TRAIN_CHANNELS,
                                     auto_reset=True, level_stride=1, with_obs=False, policy_layout="uint8")
            env2.reset()
            for t in range(20):
                env2.step(actions[t])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = min(K, 200)
            e0.record()
            for t in range(n):
                env2.step(actions[t])
            e1.record()
            torch.cuda.synchronize()
            extra["obs_policy_layout_u8_25x25x15_us_per_step"] = e0.elapsed_time(e1) / n * 1e3
            del env2

            # f4 closed loop (training/base_algo.py:152-244, ppo.py:61-73, models.py:80-109): observation (policy layout,
            # uint8, written by the step kernel) -> policy -> one categorical draw per env on the device -> step, nothing on
            # the host.  Serial (VectorRunner: one stream, one launch per step) and pipelined (PipelinedRunner: two groups
            # of 4096 envs, each group's policy / draw / step on the group's own stream, the groups overlapping), with a
            # trivial policy (uniform probabilities, no arithmetic: what is left is the plumbing) and with a network of
            # the reference's SafeLifePolicyNetwork shape (stock torch convolutions, float32, random weights: NOT part of
            # this build's kernels).  Wall clock, device idle before and after.
            from safelife_amd.runner import VectorRunner, PipelinedRunner

            class TrivialPolicy(object):
                def __init__(self):
                    self.cache = {}

                def __call__(self, obs):
                    n = obs.shape[0]
                    if n not in self.cache:
                        self.cache[n] = (torch.zeros(n, device=dev), torch.full((n, 9), 1.0 / 9.0, device=dev))
                    return self.cache[n]

            class RefShapedPolicy(torch.nn.Module):
                def __init__(self, c):
                    super().__init__()
                    nn = torch.nn
                    self.cnn = nn.Sequential(nn.Conv2d(c, 32, 5, 2), nn.ReLU(), nn.Conv2d(32, 64, 3, 2), nn.ReLU(),
                                             nn.Conv2d(64, 64, 3, 1), nn.ReLU())
                    self.dense = nn.Sequential(nn.Linear(64 * 3 * 3, 512), nn.ReLU())
                    self.logits, self.value = nn.Linear(512, 9), nn.Linear(512, 1)

                def forward(self, obs):
                    x = self.dense(self.cnn(obs.to(torch.float32)).flatten(1))
                    return self.value(x)[..., 0], torch.softmax(self.logits(x), dim=-1)

            def wall(fn, n, warm=5):
                for _ in range(warm):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e6

            def loop_env(slices):
                e = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=TRAIN_CHANNELS,
                                      auto_reset=True, level_stride=1, with_obs=False, policy_layout="uint8", slices=slices)
                return e
            cnn = RefShapedPolicy(len(TRAIN_CHANNELS)).to(dev).eval()
            cl = {}
            for pname, pol, n in (("trivial", TrivialPolicy(), 200), ("refshaped_cnn", cnn, 20)):
                r1 = VectorRunner(loop_env(1), pol, copy_obs=False, cast_obs=False)
                cl["serial_%s_us_per_step" % pname] = wall(r1.take_one_step, n)
                r2 = PipelinedRunner(loop_env(2), pol)
                r2.start()
                cl["pipelined_%s_us_per_step" % pname] = wall(lambda: r2.run(1), n)
                r2.finish()
                del r1, r2
            # the parts, each by itself: the two-group step with the policy-layout observation, the draw, the forward pass
            e2 = loop_env(2)
            e2.reset()
            e2.fence()
            fixed = torch.randint(0, 9, (B,), device=dev, dtype=torch.int32)
            torch.cuda.synchronize()

            def both_groups():
                e2.step_slice(0, fixed)
                e2.step_slice(1, fixed)
            cl["parts_step_two_groups_us"] = wall(both_groups, 200)
            e2.join()
            uni = torch.full((B, 9), 1.0 / 9.0, device=dev)
            acts_buf = torch.zeros(B, dtype=torch.int32, device=dev)
            cl["parts_draw_torch_multinomial_us"] = wall(lambda: acts_buf.copy_(torch.multinomial(uni, 1).view(-1)), 200)
            st_ptr = _hip.current_stream_ptr()
            cl["parts_draw_us"] = wall(lambda: _hip.lib().slhip_sample_actions(uni.data_ptr(), B, 9, 1, 2, acts_buf.data_ptr(), st_ptr), 200)
            obs_f = e2.policy_tensor
            with torch.no_grad():
                cl["parts_refshaped_cnn_forward_us"] = wall(lambda: cnn(obs_f), 20)
            cl["note"] = ("8192 envs, 25x25x15 uint8 observation in the policy layout written by the step kernel; serial = "
                          "VectorRunner (one stream), pipelined = PipelinedRunner (two groups of 4096 envs on two streams); the "
                          "network has the reference's SafeLifePolicyNetwork shape (conv 5x5/2-32, 3x3/2-64, 3x3-64, dense 512), "
                          "stock torch float32 kernels, random weights")
            extra["closed_loop"] = cl
            del e2, cnn

            def time_steps(env3, n_envs, n=200):
                acts = torch.randint(0, 9, (n + 20, n_envs), generator=gen, device=dev, dtype=torch.int32)
                env3.reset()
                if res["spread"]:       # (the headline's regime: episode ends at every step)
                    env3.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = torch.from_numpy(spread_num_steps(n_envs)).to(dev)
                torch.cuda.synchronize()
                for t in range(20):
                    env3.step_async(acts[t])
                env3.join()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(20, 20 + n):
                    env3.step_async(acts[t])
                env3.join()
                e1.record()
                torch.cuda.synchronize()
                us_streams = e0.elapsed_time(e1) / n * 1e3
                # the same steps through the library's AQL queues (wall clock around n steps + their fence), where the
                # timed region ran on them
                env3.last_queues_us = None
                if use_queues:
                    try:
                        env3.queues_open(n_queues, release_free=(res["fences"] == "none"))
                        for t in range(20):
                            env3.step_queues(acts[t])
                        env3.queues_sync()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for t in range(20, 20 + n):
                            env3.step_queues(acts[t])
                        env3.queues_sync()
                        env3.last_queues_us = (time.perf_counter() - t0) / n * 1e6
                        env3.queues_close()
                    except _hip.SafeLifeHipError:
                        pass
                return us_streams

            # the observation variants again as a trainer would step them: the batch in stream slices, and through the
            # library's queues (the one-launch-per-step figures above are what SafeLifeVectorEnv.step() costs)
            for tag, okw in (("obs_u8_25x25x15", dict(output_channels=TRAIN_CHANNELS, with_obs=True)),
                             ("obs_u32_view_25x25", dict(output_channels=None, with_obs=True)),
                             ("obs_policy_layout_u8_25x25x15", dict(output_channels=TRAIN_CHANNELS, with_obs=False,
                                                                    policy_layout="uint8"))):
                envo = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), auto_reset=True, level_stride=1,
                                         slices=args.slices, **okw)
                extra[tag + "_slices_us_per_step"] = time_steps(envo, B)
                if envo.last_queues_us:
                    extra[tag + "_queues_us_per_step"] = variants[tag.split("_25x25")[0] + "_queues_us"] = envo.last_queues_us
                del envo

            # multi-agent envs (SafeLifeEnv(single_agent=False), the reference's levels/random/multi-agent specs at 26x26):
            # two agents per board, one fused launch per step of the size-generic family (slhip_env_step_multi)
            try:
                from safelife_amd.levels import Level, LevelPool
                from safelife_amd.multi_env import SafeLifeMultiAgentVectorEnv
                mlv = []
                for nm in ("multi_asym1", "multi_build_coop", "multi_build_compete"):
                    with np.load(os.path.join(REPO, "tests", "golden", "trace_%s.npz" % nm)) as d:
                        for i in range(int(d["n_levels"])):
                            rec = {k[len("level%d_" % i):]: d[k] for k in d.files if k.startswith("level%d_" % i)}
                            rw = rec.pop("rng")
                            lv = Level.from_data(rec)
                            lv.rng_words = np.array(rw, np.uint64)
                            mlv.append(lv)
                envm = SafeLifeMultiAgentVectorEnv(LevelPool(mlv, counts_fn=_device_counts, n_agents=2), B, time_limit=1000,
                                                   view_shape=(25, 25), output_channels=TRAIN_CHANNELS, auto_reset=True)
                envm.reset()
                actm = torch.randint(0, 9, (64, B, 2), generator=gen, device=dev, dtype=torch.int32)
                for t in range(10):
                    envm.step(actm[t])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(50):
                    envm.step(actm[10 + t])
                e1.record()
                torch.cuda.synchronize()
                extra["multi_agent_2x26x26_obs15_us_per_step"] = e0.elapsed_time(e1) / 50 * 1e3
                extra["multi_agent_note"] = ("%d envs x 2 agents, 26x26 levels of the reference's multi-agent specs, uint8 25x25x15 "
                                             "observation per agent, one launch per step (one workgroup per board)" % B)
                del envm
            except Exception as e:          # noqa: BLE001
                extra["multi_agent_error"] = "%s: %s" % (type(e).__name__, e)

            # level-pool refresh while stepping (levels.LevelPool(refreshable=True), pool_stage / pool_commit): C3's batch through
            # the queues in calls of `chunk` steps, with a sixth of the pool's levels replaced every call (staged one call
            # ahead, committed between two calls) against the same calls without any refresh
            if use_queues and args.pool == "prune_still_25":
                try:
                    from safelife_amd.levels import LevelPool
                    lv_all = list(pool.levels)
                    n_half = len(lv_all) // 2
                    pool_r = LevelPool(lv_all[:n_half], counts_fn=_device_counts, refreshable=True)
                    env_r = SafeLifeVectorEnv(pool_r, B, time_limit=1000, view_shape=(25, 25), output_channels=TRAIN_CHANNELS,
                                              auto_reset=True, with_obs=False, slices=args.slices)
                    env_r.reset()
                    chunk, n_calls = 100, 8
                    acts_r = torch.randint(0, 9, (chunk * (n_calls + 1), B), generator=gen, device=dev, dtype=torch.int32)
                    env_r.queues_open(n_queues, release_free=(res["fences"] == "none"), recover=False)
                    rr = np.random.default_rng(5)
                    env_r.pool_stage([0], [lv_all[n_half]])        # (untimed: the first staging pins its host buffers)
                    # new levels come prepared (LevelPool.prepare: checks, cell counts, points, RNG words) -- part of making
                    # a level, which the reference does away from the stepping thread too (level_iterator.py:200-223)
                    ready = pool_r.prepare(lv_all)
                    env_r.pool_commit()
                    for refresh in (False, True):
                        env_r.step_queues_many(acts_r[:chunk])
                        env_r.queues_sync()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        n_committed = 0
                        for c in range(n_calls):
                            free = refresh and env_r.pool_commit(wait=False)    # (not yet there: next call)
                            n_committed += int(free)
                            env_r.step_queues_many(acts_r[chunk * (c + 1):chunk * (c + 2)], assume_ordered=True)
                            if free:            # (behind the call: what staging waits for -- the steps enqueued before the
                                #                  last commit -- completes while the device works through this call)
                                slots = rr.choice(n_half, n_half // 6, replace=False)
                                env_r.pool_stage(slots, ready.take(rr.integers(0, len(ready), len(slots))), background=True)
                        env_r.queues_sync()
                        us = (time.perf_counter() - t0) / (chunk * n_calls) * 1e6
                        extra["pool_refresh_us_per_step" if refresh else "pool_static_us_per_step"] = us
                        if refresh:
                            extra["pool_refresh_commits"] = "%d of %d calls" % (n_committed, n_calls)
                        env_r.pool_commit()
                    extra["pool_refresh_note"] = ("8192 envs, %d-level refreshable pool, %d steps per queue call, %d levels "
                                                  "replaced per call (prepared levels, staged a call ahead by the env's helper thread, committed "
                                                  "between calls; no queue drain for the refresh)" % (n_half, chunk, n_half // 6))
                    env_r.queues_close()
                    del env_r
                except _hip.SafeLifeHipError as e:
                    extra["pool_refresh_error"] = str(e)

            # C2 (BASELINE configs[1]): advance_board alone on 1024 random 25x25 boards (SURVEY 8d palette-like)
            from safelife_amd import speedups
            c2 = np.random.default_rng(1234)
            pal = np.array([0] * 10 + [9] * 4 + [1, 16, 17, 32788, 152, 152 | 0x200, 144, 48, 53, 85, 32884, 272, 9 | 0x200,
                                          9 | 0x400, 9 | 0x800, 122], np.uint16)
            c2_boards = torch.from_numpy(pal[c2.integers(0, len(pal), (1024, 25, 25))].view(np.int16)).to(dev)
            c2_prob = torch.full((1024,), 0.3, dtype=torch.float32, device=dev)
            c2_rng = torch.arange(4096, dtype=torch.int64, device=dev).reshape(1024, 4) * 2 + 1
            c2_out = torch.empty_like(c2_boards)
            for _ in range(10):
                speedups.advance_board_batch(c2_boards, c2_prob, c2_rng, 1, out=c2_out)
            torch.cuda.synchronize()
            # (the launch through the C-ABI with its arguments bound once: the Python shim's per-call work -- pointer
            #  objects, the current-stream lookup -- is ~8 us, more than the kernel)
            c2_fn = _hip.lib().slhip_advance_board
            c2_args = (_hip.ptr(c2_boards), _hip.ptr(c2_out), 1024, 25, 25, _hip.ptr(c2_prob), 1, _hip.ptr(c2_rng),
                       _hip.current_stream_ptr())
            for _ in range(10):
                c2_fn(*c2_args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                c2_fn(*c2_args)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 200 * 1e3
            extra["c2_advance_board_1024x25x25_us_per_launch"] = us
            extra["c2_advance_board_board_steps_per_s"] = 1024 / (us * 1e-6)
            # the floor of that figure: the SAME launch (same kernel, grid, arguments, stream, back to back) with zero CA
            # steps -- rows in, rows out: what one dependent launch of this shape costs on a HIP stream before any rule is
            # evaluated.  1024 boards are 512 one-wave workgroups on 256 CUs and 2.56 MB: the number is launch latency.
            c2_floor = c2_args[:6] + (0,) + c2_args[7:]
            for _ in range(10):
                c2_fn(*c2_floor)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                c2_fn(*c2_floor)
            e1.record()
            torch.cuda.synchronize()
            extra["c2_launch_floor_us_per_launch"] = e0.elapsed_time(e1) / 200 * 1e3
            extra["c2_launch_floor_note"] = ("slhip_advance_board with n_steps = 0 (load, no step, store), 200 launches back to "
                                             "back on one stream under HIP events, as the line above")
            # ... and the reference's own C advance_board (oracle/_ref: its sources compiled by oracle/Makefile) on the
            # same 1024 boards, one host core, next to it (cpu_baseline kind "reference" for C2)
            if args.cpu_baseline:
                import oracle
                ref = oracle.load_ref()
                if ref is not None:
                    host_boards = pal[np.random.default_rng(1234).integers(0, len(pal), (1024, 25, 25))]
                    c2_bg = np.random.PCG64(1234)          # (kept alive: the module holds a borrowed pointer)
                    ref.set_bit_generator(c2_bg)
                    t0 = time.perf_counter()
                    reps = 0
                    while time.perf_counter() - t0 < 1.0:
                        oracle.ref_advance_batch(ref, host_boards, 0.3, 1)
                        reps += 1
                    dt = time.perf_counter() - t0
                    extra["c2_cpu_reference_board_steps_per_s"] = 1024 * reps / dt
                    extra["c2_cpu_reference_note"] = ("safelife/speedups_src advance_board_nstep compiled with gcc -O3 (oracle/_ref), "
                                                      "1 core of %s (the reference draws from one process-wide generator), "
                                                      "called per board from a C loop: no interpreter or wrapper time" % cpu_model())

            # the same step with the training wrappers of env_factory.py:277-283 fused in (float64 shaped reward)
            envw = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=TRAIN_CHANNELS,
                                     auto_reset=True, with_obs=False, slices=args.slices,
                                     wrappers=dict(movement_bonus=0.1, exit_bonus=0.5, penalty_coef=0.3))
            us = time_steps(envw, B)
            extra["training_wrappers_us_per_step"] = us
            if envw.last_queues_us:
                extra["training_wrappers_queues_us_per_step"] = envw.last_queues_us
            del envw
            # ... with SimpleSideEffectPenalty's "inaction" baseline (env_wrappers.py:179-180): a CA step of every env's
            # baseline board per step, a third pass of the step kernel's CA loop
            envi = SafeLifeVectorEnv(pool, B, time_limit=1000, view_shape=(25, 25), output_channels=TRAIN_CHANNELS,
                                     auto_reset=True, with_obs=False, slices=args.slices,
                                     wrappers=dict(movement_bonus=0.1, exit_bonus=0.5, penalty_coef=0.3,
                                                   baseline="inaction", inaction_seed=11))
            extra["training_wrappers_inaction_baseline_us_per_step"] = time_steps(envi, B)
            if envi.last_queues_us:
                extra["training_wrappers_inaction_baseline_queues_us_per_step"] = envi.last_queues_us
            del envi
            # per-GPU shares of the sharded configs of BASELINE.json (C4: append-spawn 25x25, C5: navigation 64x64)
            # (and the shape the reference's own random-level YAMLs use, levels/random/*.yaml: board_shape [26, 26])
            for tag, pname, n_envs in (("c4_append_spawn_25", "append_spawn_25", 8192), ("c5_navigation_64", "navigation_64", 4096),
                                       ("append_still_26x26", "append_still_26", 8192)):
                if not os.path.exists(os.path.join(REPO, "tests", "golden", "pool_%s.npz" % pname)):
                    continue
                p2 = load_pool(pname, _device_counts)
                envc = SafeLifeVectorEnv(p2, n_envs, time_limit=1000, view_shape=(25, 25), slices=args.slices,
                                         output_channels=TRAIN_CHANNELS, auto_reset=True, with_obs=False)
                us = time_steps(envc, n_envs)
                extra[tag + "_us_per_step"] = us
                extra[tag + "_env_steps_per_s_per_gpu"] = n_envs / (us * 1e-6)
                if envc.last_queues_us:
                    extra[tag + "_queues_us_per_step"] = envc.last_queues_us
                    extra[tag + "_queues_env_steps_per_s_per_gpu"] = n_envs / (envc.last_queues_us * 1e-6)
                del envc
                if pname == "navigation_64":
                    # side_effects.py:109-111 runs life_occupancy(board, n_step=1000) twice at every episode end
                    from safelife_amd import speedups
                    nb = 4096
                    boards = env3_boards = torch.from_numpy(
                        np.ascontiguousarray(p2.arrays()["pool_board"][np.arange(nb) % len(p2)]).view(np.int16)).to(dev)
                    probs = torch.full((nb,), 0.3, dtype=torch.float32, device=dev)
                    rngs = torch.arange(nb * 4, dtype=torch.int64, device=dev).reshape(nb, 4) * 2 + 1
                    # (the 537 MB result is allocated and touched BEFORE the first event, the warm-up runs at full size, and
                    #  the figure is the median of three launches: round 5's one-shot timing had a cold allocation of that
                    #  size between its two events)
                    occ_out = torch.zeros((nb, 64, 64, 8), dtype=torch.int32, device=dev)
                    speedups.life_occupancy_batch(boards, probs, rngs.clone(), 100, out=occ_out)
                    torch.cuda.synchronize()
                    occ_ms = []
                    for _ in range(3):
                        r_rep = rngs.clone()
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        speedups.life_occupancy_batch(boards, probs, r_rep, 1000, out=occ_out)
                        e1.record()
                        torch.cuda.synchronize()
                        occ_ms.append(e0.elapsed_time(e1))
                    ms = sorted(occ_ms)[1]
                    del occ_out
                    extra["life_occupancy_64x64_1000steps_boards_per_s"] = nb / (ms * 1e-3)
                    extra["life_occupancy_64x64_board_steps_per_s"] = variants["life_occupancy_64x64_board_steps_per_s"] = nb * 1000 / (ms * 1e-3)
                    extra["life_occupancy_64x64_ms_of_three"] = occ_ms
                    # C5 as BASELINE.json states it: navigation WITH the side-effect score.  Episode ends are spread
                    # evenly (every env starts at a different point of its 1000-step episode), the step kernels queue
                    # the finished episodes, and every `flush_every` steps the episode-end pass of side_effect_score
                    # (roll-forward by the episode's length + 2 x 1000-step occupancy + distributions) runs on the
                    # device for whatever the queue holds -- all inside the timed region; the earth-mover distances
                    # (host, pyemd: parity unpinned) are not.
                    n_c5, flush_every, n_meas = n_envs, 512, 2048
                    env5 = SafeLifeVectorEnv(p2, n_c5, time_limit=1000, view_shape=(25, 25), output_channels=TRAIN_CHANNELS,
                                             auto_reset=True, with_obs=False, slices=args.slices,
                                             side_effects=dict(capacity=2 * (n_c5 * flush_every // 1000 + 64), num_samples=1000))
                    env5.reset()
                    env5.t["scalars"][:, _hip.SCALAR_COLS["num_steps"]] = (
                        torch.arange(n_c5, device=dev, dtype=torch.int32) * 997) % 1000
                    acts5 = torch.randint(0, 9, (n_meas + 20, n_c5), generator=gen, device=dev, dtype=torch.int32)
                    torch.cuda.synchronize()
                    # (warm-up at full size: one whole window of steps and its episode-end pass, so that the pass's work and
                    #  output buffers -- 560 MB -- exist and have been touched before the first event; round 5 measured them
                    #  being allocated inside the region: 102 us per step on the driver's box against 48 here)
                    for t in range(20):
                        env5.step_async(acts5[t])
                    for t in range(20, 20 + flush_every):
                        env5.step_async(acts5[t])
                    env5.side_effects_flush(overlap=True)
                    env5.join()
                    env5.side_effects_join()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    batches = []
                    e0.record()
                    for t in range(20, 20 + n_meas):
                        env5.step_async(acts5[t])
                        if (t - 19) % flush_every == 0:
                            # (round 5) the pass runs on a stream of its own UNDER the steps that follow; the last one of the
                            # region has nothing to hide under and is waited for in full
                            batches.append(env5.side_effects_flush(overlap=True))
                    env5.join()
                    env5.side_effects_join()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1)
                    n_eps = sum(len(b) for b in batches)
                    # (extra only: through HIP streams the figure depends on how the runtime maps the slice and side streams
                    #  onto hardware queues on the box -- 46.7 us on most, 86-102 where two of them share one; the queues'
                    #  figure below is the one `roofline` carries)
                    extra["c5_with_side_effects_us_per_step"] = ms * 1e3 / n_meas
                    extra["c5_with_side_effects_env_steps_per_s_per_gpu"] = n_c5 * n_meas / (ms * 1e-3)
                    extra["c5_with_side_effects_episodes_scored"] = n_eps
                    extra["c5_with_side_effects_note"] = ("%d envs x 64x64 navigation, %d steps, episode-end pass every %d steps "
                                                          "on the device, on a side stream under the following steps, the "
                                                          "region's last pass waited for in full (%d episodes: roll-forward + "
                                                          "2 x 1000-step life_occupancy + distributions); EMD on the host not "
                                                          "included" % (n_c5, n_meas, flush_every, n_eps))
                    # the same through the library's queues (the launcher of the headline line; wall clock: HIP events do
                    # not see the queues), whole windows of flush_every steps per call
                    if use_queues:
                        try:
                            env5.queues_open(4, release_free=(res["fences"] == "none"), recover=False)
                            env5.side_effects_flush()
                            torch.cuda.synchronize()
                            # (one untimed window first, as above)
                            torch.cuda.current_stream().synchronize()
                            env5.step_queues_many(acts5[20:20 + flush_every], assume_ordered=True)
                            env5.side_effects_flush(overlap=True)
                            env5.queues_sync()
                            env5.side_effects_join()
                            torch.cuda.synchronize()
                            batches = []
                            t0 = time.perf_counter()
                            for w0 in range(0, n_meas, flush_every):
                                # (the fresh queue's buffers were zeroed on this stream: wait for THAT, not for the device --
                                #  the pass of the window before is still running on its side stream, and may)
                                torch.cuda.current_stream().synchronize()
                                env5.step_queues_many(acts5[20 + w0:20 + w0 + flush_every], assume_ordered=True)
                                batches.append(env5.side_effects_flush(overlap=True))      # (waits for the queues' fence first)
                            env5.queues_sync()
                            env5.side_effects_join()
                            torch.cuda.synchronize()
                            us = (time.perf_counter() - t0) / n_meas * 1e6
                            extra["c5_with_side_effects_queues_us_per_step"] = variants["c5_with_side_effects_us"] = us
                            extra["c5_with_side_effects_queues_env_steps_per_s_per_gpu"] = n_c5 / (us * 1e-6)
                            extra["c5_with_side_effects_queues_episodes_scored"] = sum(len(b) for b in batches)
                            env5.queues_close()
                        except _hip.SafeLifeHipError as e:
                            extra["c5_with_side_effects_queues_error"] = str(e)
                    del env5, batches
        except Exception as e:          # noqa: BLE001
            import traceback
            extra["extras_error"] = "%s: %s" % (type(e).__name__, e)
            print("bench: an extra failed:\n" + traceback.format_exc(), file=sys.stderr)
            try:
                torch.cuda.synchronize()
            except Exception:           # noqa: BLE001
                pass

    if rank == 0:
        obs_bytes = {0: 0, 1: H * Wd * len(TRAIN_CHANNELS), 2: H * Wd * 4}[args.obs]
        bytes_per_step = 3 * H * Wd * 2 + obs_bytes
        # next to the vendor peak, what plain copies reach on THIS box -- (a) a 1 GiB device-to-device copy
        # (the sustainable HBM rate, read + written bytes), (b) copies that move exactly one step's algorithmic bytes
        # (half read, half written), back to back like the steps.  Untimed by the driver: after the timed region.
        ceiling = None
        if args.extras and world == 1:
            def timed(fn, n):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n * 1e-3           # seconds per call
            big_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            big_b = torch.empty_like(big_a)
            t_big = timed(lambda: big_b.copy_(big_a), 10)
            del big_a, big_b
            half = bytes_per_step * B // 2 // 16 * 16
            src, dst = torch.empty(half, dtype=torch.uint8, device=dev), torch.empty(half, dtype=torch.uint8, device=dev)
            t_one = timed(lambda: dst.copy_(src), 400)
            ceiling = {"hbm_copy_1GiB_GBps": 2 * (1 << 30) / t_big / 1e9,
                       "step_bytes_copy_us": t_one * 1e6,
                       "step_bytes_copy_GBps": 2 * half / t_one / 1e9,
                       "note": "torch copy_ kernels, HIP events over back-to-back launches on one stream; step_bytes_copy "
                               "moves bytes_per_env_step x envs per launch (half read, half written: 15 MB buffers, "
                               "resident in the memory-side cache)"}
            del src, dst
        # the fraction the line is graded on comes from the wall clock of the timed region (ms_per_step); the HIP
        # events over the same region give the device-side figure next to it
        achieved = bytes_per_step * B / (elapsed / K) / 1e9
        achieved_device = bytes_per_step * B / (kernel_ms * 1e-3) / 1e9
        stream_key = "stream_leg" if use_queues else "device"
        traffic, traffic_note, traffic_agent = None, None, None
        try:    # HBM bytes per launch from the committed PMC passes (tools/pmc_run.sh), if they match this run
            with open(os.path.join(REPO, "profiles", "traffic_latest.json")) as f:
                tj = json.load(f)
            # (only a PMC run of the SAME stepping mode counts: launches per step and, for the queues, the fences)
            same = (tj.get("slices", 1) == (res["queue_slices"] if use_queues else env.slices)
                    and tj.get("queue_fences") == (res["fences"] if use_queues else None))
            if tj.get("envs_per_gpu") == B and tj.get("obs") == args.obs:
                if same:
                    traffic = tj.get("hbm_bytes_per_step", tj["hbm_bytes_per_launch"])      # all launches of one step
                else:
                    # the same kernel on the same batch, counted with a stream's fences: per-dispatch counters cannot
                    # attribute a release-free step's traffic (the profiler serialises the dispatches and flushes the
                    # L2s around each: every launch starts cold and its write-back falls outside its window)
                    traffic_agent = tj.get("hbm_bytes_per_step", tj["hbm_bytes_per_launch"])
                    traffic_note = ("null for this stepping mode (per-dispatch counters cannot attribute release-free "
                                    "steps); traffic_agent_fences = PMC bytes per step of the same kernel and batch "
                                    "stepped with agent-scope fences (%s), %.3f x the algorithmic bytes"
                                    % (tj.get("source", "profiles/traffic_latest.json"),
                                       traffic_agent / float((3 * H * Wd * 2 + obs_bytes) * B)))
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "env steps/sec (whole node), 8192x25x25 boards; bit-exact vs C advance_board",
            "value": world * B * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic (reference-procgen %s level pool cycled on device, "
                                    "uniform random actions)" % args.pool.rsplit("_", 1)[0].replace("_", "-"),
            "config": {"workload": "%s%d envs/GPU x %dx%d %s, fused step()+reward+auto-reset%s" % (
                           "C3: " if args.pool == "prune_still_25" else "", B, H, Wd, args.pool.rsplit("_", 1)[0].replace("_", "-"),
                           {0: ", no observation", 1: " + 25x25x15 u8 obs", 2: " + 25x25 u32 view"}[args.obs]),
                       "envs_per_gpu": B, "global_envs": world * B, "board": [H, Wd],
                       "level_pool": len(pool), "parallelism": "envs sharded %d-way, step records gathered "
                                                               "to rank 0 every %d steps (%s); %s" % (
                                                                   world, every_used,
                                                                   "8-byte records, RCCL send/recv on a side stream" if gather.collective
                                                                   else "one rank: nothing to exchange",
                                                                   ("%d slice(s) per GPU, one dispatch each on an AQL queue of "
                                                                    "the library's own (barrier bit; fences: see "
                                                                    "queue_fences), all K steps enqueued by one C call"
                                                                    % res["queue_slices"]) if use_queues
                                                                   else ("%d slice(s) per GPU, one launch and one stream each"
                                                                         % env.slices)),
                       "stepping": "aql-queues" if use_queues else "hip-streams",
                       "gather_window_closes_steps_before_end": (window_ahead if gather.collective else None),
                       "queue_ids": (res["queue_ids"] if use_queues else None),
                       "queue_ids_note": ("the step queue RCCL's exchange kernel would hold up is left out (probed: "
                                          "slhip_gather_stream_shares)" if (use_queues and gather.collective and
                                                                            len(res["queue_ids"]) == 3) else None),
                       "staged_steps": ("%d of the %d timed steps were staged before the clock started (slhip_queues_stage: argument "
                                        "blocks and packets written, nothing handed to the device) and released inside the region "
                                        "by slhip_queues_go; all %d steps execute inside the region" % (res["staged"], K, K)
                                        if res["staged"] else 0) if use_queues else None,
                       "episode_phase": ("spread: env e starts (e x 997) mod 1000 steps into its first episode -- about %d "
                                         "episodes end and reset inside the kernel at every timed step (SURVEY 8d)" % (B // 1000)
                                         if res["spread"] else "all envs straight behind a reset: no episode end in the region"),
                       "queue_fences": ({"none": "none: bench opted in to release-free stepping (SL_QUEUES_RELEASE_FREE) -- "
                                                 "agent-scope acquire, NO release between the steps of a queue; placement "
                                                 "probed at open and verified by every step",
                                         "agent": "agent: agent-scope acquire and release on every step (a stream's "
                                                  "fences; the library's default)"}[res["fences"]]) if use_queues else None,
                       "queue_fences_requested": args.queue_fences if use_queues else None,
                       "queues_unavailable": queues_why},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_agent_fences": traffic_agent,
                         "kernel": "fused env step", "bytes_per_env_step": bytes_per_step,
                         "time_base": "wall clock of the timed region (ms_per_step): first launch from an idle GPU to "
                                      "the return of the closing synchronize",
                         # device time of one step (HIP events on slice 0's stream over the same region): the slice
                         # launches of a step run concurrently, each stream back to back, so a step costs one
                         # stream's launch-to-launch time
                         "launches_per_step": res["queue_slices"] if use_queues else env.slices,
                         # queue stepping: a SEPARATE leg behind the timed region -- K steps of the same kernel through
                         # the HIP-stream slices under HIP events (events cannot see the library's queues).  It describes
                         # another launcher and may well exceed ms_per_step; stream stepping: the events bracket the
                         # timed region itself.
                         stream_key + "_launch_ms": kernel_ms,
                         stream_key + "_note": ("HIP events over %d steps of the SAME kernel issued through the %d stream "
                                                "slice(s) right after the timed region (wall %.5f ms per step)"
                                                % (K, env.slices, stream_wall_ms))
                         if stream_wall_ms is not None else "HIP events over the timed region (slice 0's stream)",
                         stream_key + "_achieved": achieved_device, stream_key + "_frac": achieved_device / HBM_PEAK_GBS,
                         "host_enqueue_ms_per_step": (t_enqueued - t_start) / K * 1e3,
                         # the same region under other launchers / regimes / lengths, measured in this run behind the
                         # timed region (us per step, medians of three repetitions; null: not measured in this run):
                         #   agent_fences_us   the library's DEFAULT queue fences (a stream's acquire and release per step)
                         #   no_reset_us       the region straight behind a reset of all envs: no episode end inside it
                         #                     (rounds 1-5's region; steady_state_us is its counterpart with --spread 0)
                         #   k400_us           the same region 400 steps long (what a 20-step region's fixed cost hides)
                         #   unstaged_us       the same region with every step enqueued INSIDE it (config.staged_steps = 0)
                         #   k20_median_us     this line's own region again, median of five
                         #   forced_gather_us  one rank with the RCCL exchange forced on (what every N > 1 rank runs)
                         #   c5_with_side_effects_us  C5's per-GPU share with the episode-end pass in the region (queues)
                         **{k: variants.get(k) for k in ("agent_fences_us", "no_reset_us", "steady_state_us", "k400_us", "unstaged_us",
                                                         "k20_median_us", "forced_gather_us", "c5_with_side_effects_us",
                                                         "life_occupancy_64x64_board_steps_per_s",
                                                         "obs_u8_queues_us", "obs_u32_view_queues_us", "obs_policy_layout_u8_queues_us")},
                         # (rounds 1-5 timed the region without episode ends: the fraction of THAT regime, for comparison)
                         "no_reset_frac": (bytes_per_step * B / (variants["no_reset_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                                           if variants.get("no_reset_us") else None),
                         "k400_frac": (bytes_per_step * B / (variants["k400_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                                       if variants.get("k400_us") else None),
                         "scaling_curve": "not measured by this build (no multi-GPU node was available to it): the 1-to-N curve "
                                          "is the driver's",
                         "measured_ceiling": ceiling,
                         # the issue side next to the HBM side (C3, four waves per SIMD; counters and timing-only builds
                         # committed under profiles/, not measured in this run)
                         "issue_side": ({"valu_per_wave": 538, "salu_per_wave": 226, "lds_per_wave": 52, "waves_per_simd": 4,
                                         "valu_cycles_each": [2, 4],
                                         "valu_issue_us_per_step": [4 * 538 * 2 / 2.4e3, 4 * 538 * 4 / 2.4e3],
                                         "hbm_floor_us_per_step": bytes_per_step * B / (HBM_PEAK_GBS * 1e3),
                                         "timing_only_us_per_step": {"full": 6.4, "no_ca": 5.7, "no_scores": 6.2,
                                                                     "no_leader_work": 6.1, "skeleton": 4.8},
                                         "binds": "neither roofline: the step is one launch's latency chain (DMA issue, "
                                                  "memory latency, three workgroup barriers, stores) plus the CP's "
                                                  "boundary between launches; the SIMDs are at most half busy",
                                         "source": "profiles/round6_z_pmc.txt (538 / 226 / 52 per wave on this tree), "
                                                   "round5_a_c3_issue_pmc.txt, round5_f_timing_only.txt; DESIGN.md section 6"}
                                        if args.pool == "prune_still_25" and B == 8192 and not args.obs else None),
                         "note": "frac = bytes_per_env_step x envs / ms_per_step / peak.  rocprofv3 serialises the queues' "
                                 "dispatches (its kernel trace gives per-launch durations only); the four queues' overlap "
                                 "is shown by the kernels' own clocks (tools/trace_overlap.py); traffic is null for "
                                 "release-free stepping (per-dispatch counters cannot attribute it, DESIGN.md)"},
        }
        if world > 1 or gather.collective:
            out["gather_every"] = every_used
            out["gather_windows_in_region"] = gather_windows
            out["gather_exposed_ms"] = gather_exposed * 1e3
            out["gather_transport"] = gather.backend
        if per_rank:
            out["per_rank"] = per_rank
        if extra:
            out["extra"] = extra
        if args.cpu_baseline and world == 1:
            import oracle

            def cpu_counts(b, g):
                return oracle.alive_counts_batch(b, g)
            out["cpu_baseline"] = cpu_baseline(load_pool(args.pool, cpu_counts), B, args.cpu_steps, 7)
            out["cpu_baseline"]["parity_check"] = parity
        # (RCCL writes a version banner through C stdio, which would otherwise come out at exit, BEHIND this line)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
