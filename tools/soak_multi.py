#!/usr/bin/env python3
"""Randomised differential soak of the MULTI-agent step: SafeLifeMultiAgentVectorEnv (HIP, slhip_env_step_multi) vs the CPU
oracle (slo_env_step_multi) over synthetic levels with 1-4 agents close to one another and to crates, exits, walls and
spawners -- agents pushing, destroying and blocking each other, leaving one by one -- random per-agent points tables,
shapes, views, time limits, masked resets.
    python tools/soak_multi.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from safelife_amd.levels import Level, LevelPool, _device_counts
from safelife_amd.cell_types import CellTypes as CT

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
PAL = np.array([0] * 16 + [9] * 4 + [1, 16, 17, 32788, 32788, 48, 53, 85, 32884, 9 | 0x200, 9 | 0x400, 9 | 0x800, 4 | 8, 32],
               np.uint16)
SPAWN = np.array([152, 152 | 0x200, 152 | 0x600], np.uint16)
COLORS = [0x200, 0x400, 0x800, 0x600, 0xC00]


def random_level(H, W, A, spawners, n_exits, min_perf):
    b = PAL[rng.integers(0, len(PAL), (H, W))]
    for _ in range(spawners):
        b[rng.integers(0, H), rng.integers(0, W)] = rng.choice(SPAWN)
    for _ in range(n_exits):
        b[rng.integers(0, H), rng.integers(0, W)] = CT.level_exit
    g = ((rng.integers(0, 8, (H, W)) << 9).astype(np.uint16) * (rng.random((H, W)) < 0.3)).astype(np.uint16)
    # the agents in a small neighbourhood, so that they meet
    cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
    locs = []
    while len(locs) < A:
        y, x = (cy + int(rng.integers(-2, 3))) % H, (cx + int(rng.integers(-2, 3))) % W
        if [y, x] not in locs:
            locs.append([y, x])
    for y, x in locs:
        flags = CT.player if rng.random() < 0.7 else (CT.agent | CT.destructible | CT.frozen)
        b[y, x] = flags | int(rng.choice(COLORS)) | (int(rng.integers(0, 4)) << 12)
    tables = rng.integers(-3, 4, (A, 8, 9))
    return Level(b, g, np.array(locs), spawn_prob=float(rng.choice([0.3, 0.9])), min_performance=min_perf, points_table=tables)


t_end, n_cfg, n_steps = time.time() + budget, 0, 0
while time.time() < t_end:
    H, W = [(10, 10), (9, 13), (26, 26), (15, 15), (7, 8), (25, 25)][rng.integers(0, 6)]
    A = int(rng.integers(1, 5))
    L = int(rng.integers(1, 5))
    levels = [random_level(H, W, A, int(rng.choice([0, 0, 3])), int(rng.integers(0, 4)), float(rng.choice([-1, 0.0, 0.3])))
              for _ in range(L)]
    if A == 1:
        continue        # (single-agent pools are the other soak's)
    pool_d = LevelPool(levels, counts_fn=_device_counts, n_agents=A, min_performance_fraction=float(rng.choice([1.0, 0.1])))
    pool_c = pool_d         # (one pool: levels without a seed draw their generators from the pool's SeedSequence)
    B = int(rng.choice([1, 3, 8, 17]))
    chans = [None, (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 25, 26, 27), tuple(range(16)) + (25, 26, 27)][rng.integers(0, 3)]
    kw = dict(first_level=rng.integers(0, L, B), auto_reset=bool(rng.random() < 0.8), level_stride=int(rng.integers(1, 3)),
              time_limit=int(rng.choice([3, 9, 30])), view_shape=(int(rng.integers(1, 27)), int(rng.integers(1, 27))),
              output_channels=chans, remove_white_goals=bool(rng.random() < 0.5))
    desc = dict(shape=(H, W), A=A, B=B, L=L, kw={k: v for k, v in kw.items() if k != "first_level"})
    dev, cpu = util.DeviceMultiBackend(pool_d, B, **kw), util.OracleMultiBackend(pool_c, B, **kw)
    assert np.array_equal(dev.reset(), cpu.reset()), ("reset obs", desc)
    for name in util.MULTI_STATE:
        if name not in ("reward", "done", "success"):
            assert np.array_equal(dev.get(name), cpu.get(name)), ("after reset", name, desc)
    for t in range(int(rng.integers(5, 40))):
        a = rng.integers(0, 9, (B, A)).astype(np.int32)
        if rng.random() < 0.5:
            a[cpu.get("is_active") == 0] = 0
        o1, r1, d1 = dev.step(a)
        o2, r2, d2 = cpu.step(a)
        if not (np.array_equal(r1, r2) and np.array_equal(d1, d2)):
            bad = np.argwhere((r1 != r2) | (d1 != d2))
            for e, k in bad[:4]:
                print("env", e, "agent", k, "action", a[e], "reward dev/cpu", r1[e], r2[e], "done", d1[e], d2[e],
                      "locs dev", dev.get("agent_locs")[e].tolist(), "cpu", cpu.get("agent_locs")[e].tolist(),
                      "old_value", dev.get("old_value")[e], cpu.get("old_value")[e], "level", cpu.get("level_idx")[e],
                      "board equal", np.array_equal(dev.get("board")[e], cpu.get("board")[e]),
                      "cells", [hex(int(dev.get("board")[e][y, x])) for y, x in cpu.get("agent_locs")[e]],
                      [hex(int(cpu.get("board")[e][y, x])) for y, x in cpu.get("agent_locs")[e]])
            e = int(bad[0][0])
            bd, bc = dev.get("board")[e], cpu.get("board")[e]
            diff = np.argwhere(bd != bc)
            print("differing cells:", [(int(y), int(x), hex(int(bd[y, x])), hex(int(bc[y, x]))) for y, x in diff[:12]], "of", len(diff))
            print("goals equal", np.array_equal(dev.get("goals")[e], cpu.get("goals")[e]), "rng equal", np.array_equal(dev.get("rng")[e], cpu.get("rng")[e]),
                  "exit_locs", dev.get("exit_locs")[e], cpu.get("exit_locs")[e], "spawners in level", int((levels[int(cpu.get("level_idx")[e])].board & 0x80).astype(bool).sum()))
            raise AssertionError(("reward/done", t, desc))
        assert np.array_equal(o1, o2), ("obs", t, desc)
        for name in util.MULTI_STATE:
            assert np.array_equal(dev.get(name), cpu.get(name)), (name, t, desc)
        if rng.random() < 0.1:
            mask = (rng.random(B) < 0.4).astype(np.uint8)
            assert np.array_equal(dev.env.reset(mask).cpu().numpy().view(o2.dtype) if chans is None else dev.env.reset(mask).cpu().numpy(),
                                  cpu.env.reset(mask)), ("masked reset", t, desc)
        n_steps += B
    n_cfg += 1
print("multi-agent soak ok: %d configurations, %d env-steps compared, seed %d" % (n_cfg, n_steps, seed))
