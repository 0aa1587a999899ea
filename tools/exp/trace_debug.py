#!/usr/bin/env python3
"""Replay a reference trace on the device and on the oracle side by side; print every state field that differs.
    python tools/exp/trace_debug.py append_still_1_chan15_v25 [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from safelife_amd.levels import _device_counts

FIELDS = ("board", "goals", "agent_loc", "exit_locs", "rng", "num_steps", "old_value", "required_points",
          "initial_points", "goals_static", "is_active", "episode_reward", "episode_length", "level_idx",
          "episode_idx", "success", "times_up", "reward", "done")
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tr = util.load_trace(name)
kw = util.env_kwargs_from_trace(tr)
w = util.wrappers_from_trace(tr)
if w is not None:
    kw["wrappers"] = w
def make(cls, counts):
    pool = util.pool_from_trace(tr, counts)
    return cls(pool, 1, first_level=0, auto_reset=True, level_stride=1, episode_streams=False, **dict(kw))
dev = make(util.DeviceBackend, _device_counts)
cpu = make(util.OracleBackend, util.oracle_counts)
def compare(tag):
    bad = []
    for f in FIELDS:
        try:
            a, b = dev.get(f), cpu.get(f)
        except Exception as e:
            continue
        if not np.array_equal(np.asarray(a), np.asarray(b)):
            bad.append((f, np.asarray(a).ravel()[:8], np.asarray(b).ravel()[:8]))
    print(tag, "OK" if not bad else "MISMATCH")
    for f, a, b in bad:
        print("   ", f, "device", a, "oracle", b)
od, oc = dev.reset(), cpu.reset()
print("reset obs equal:", np.array_equal(od, oc))
compare("after reset")
for t in range(min(steps, len(tr["trace_reward"]))):
    a = np.array([tr["trace_actions"][t]], np.int32)
    od, rd, dd = dev.step(a)
    oc, rc, dc = cpu.step(a)
    print("step", t, "action", int(a[0]), "reward dev/cpu/ref", rd[0], rc[0], tr["trace_reward"][t], "obs equal", np.array_equal(od, oc))
    compare("after step %d" % t)
