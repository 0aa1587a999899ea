import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import util
from safelife_amd.levels import _device_counts
name = sys.argv[1]
tr = util.load_trace(name)
pool = util.pool_from_trace(tr, _device_counts)
kw = util.env_kwargs_from_trace(tr); kw["wrappers"] = util.wrappers_from_trace(tr)
dev = util.DeviceBackend(pool, 1, first_level=0, auto_reset=True, level_stride=1, **kw)
cpu = util.OracleBackend(pool, 1, first_level=0, auto_reset=True, level_stride=1, **kw)
dev.reset(); cpu.reset()
for t in range(len(tr["trace_reward"])):
    a = np.array([tr["trace_actions"][t]], np.int32)
    dev.step(a); cpu.step(a)
    sd, sc, sr = dev.get("shaped_reward")[0], cpu.get("shaped_reward")[0], tr["trace_shaped_reward"][t]
    if sd != sr or sc != sr or tr["trace_done"][t]:
        ws = dev.env.t["wrap_state"].cpu().numpy()[0]
        print(t, "dev", sd, "cpu", sc, "ref", sr, "done", tr["trace_done"][t], "reward", tr["trace_reward"][t],
              "dev state", ws[:2], "cpu last", cpu.env.wa["last_side_effect"][0], "n_prior", cpu.env.wa["n_prior"][0])
