import sys, numpy as np
sys.path.insert(0,'.')
from tests import util
from safelife_amd.levels import _device_counts
pool,_=util.pool_from_fixture("prune_still_25", _device_counts)
B=int(sys.argv[1]) if len(sys.argv)>1 else 8192
kw=dict(auto_reset=True, level_stride=3, time_limit=20, view_shape=(25,25), output_channels=None)
first=np.arange(B)%len(pool)
dev=util.DeviceBackend(pool,B,first_level=first,slices=1,**kw)
cpu=util.OracleBackend(pool,B,first_level=first,**kw)
dev.env.reset(); cpu.env.reset()
rng=np.random.default_rng(4)
for t in range(30):
    a=rng.integers(0,9,B).astype(np.int32)
    dev.env.step(a); cpu.env.step(a,n_threads=8)
    bad=[]
    for name in ("reward","done","board","agent_loc","num_steps","level_idx","goals","rng"):
        d,c=dev.get(name),cpu.get(name)
        if not np.array_equal(d,c):
            idx=np.unique(np.argwhere(d!=c)[:,0])
            bad.append((name,len(idx),idx[:8].tolist()))
    if bad:
        print("step",t,bad)
        e=bad[0][2][0]
        print("env",e,"dev reward",dev.get("reward")[e],"cpu",cpu.get("reward")[e],"done",dev.get("done")[e],cpu.get("done")[e], "steps", dev.get("num_steps")[e], cpu.get("num_steps")[e], "loc", dev.get("agent_loc")[e], cpu.get("agent_loc")[e], "action", a[e])
        db,cb=dev.get("board")[e],cpu.get("board")[e]
        print("board diff cells", np.argwhere(db!=cb)[:10].tolist(), [ (hex(db[tuple(x)]),hex(cb[tuple(x)])) for x in np.argwhere(db!=cb)[:10]])
        break
else:
    print("all ok")
if bad:
    for e in bad[0][2]:
        db,cb=dev.get("board")[e],cpu.get("board")[e]
        dg,cg=dev.get("goals")[e],cpu.get("goals")[e]
        print(e,"level dev/cpu",dev.get("level_idx")[e],cpu.get("level_idx")[e],"rows differing board",sorted(set(np.argwhere(db!=cb)[:,0].tolist())),"goals",sorted(set(np.argwhere(dg!=cg)[:,0].tolist())))
        # does the device row match some other level's row?
        pb=pool.arrays()["pool_board"]
        for r in sorted(set(np.argwhere(db!=cb)[:,0].tolist()))[:3]:
            m=[l for l in range(len(pool)) if np.array_equal(pb[l][r]&0xFFFF, db[r])]
            print("   row",r,"matches pool levels",m[:6])
