"""What the closing synchronize of a K-step region costs, and whether a device-written flag in pinned host memory
(hipStreamWriteValue32 behind the last step of every slice stream, host spins on it) gets the host out earlier."""
import ctypes as C, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from safelife_amd.levels import _device_counts
from safelife_amd.vector_env import SafeLifeVectorEnv
hip = C.CDLL("libamdhip64.so")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pool = bench.load_pool("prune_still_25", _device_counts)
env = SafeLifeVectorEnv(pool, 8192, time_limit=1000, view_shape=(25, 25), output_channels=bench.TRAIN_CHANNELS,
                        auto_reset=True, with_obs=False, slices=2)
env.reset()
acts = torch.randint(0, 9, (K + 8, 8192), device="cuda", dtype=torch.int32)
ptrs = [acts[t].data_ptr() for t in range(K + 8)]
flags = torch.zeros(16, dtype=torch.int32).pin_memory()
fl = flags.numpy()
fptr = flags.data_ptr()
streams = [s.cuda_stream for s in env._slice_streams]
for t in range(8):
    env.step_async(ptrs[t])
torch.cuda.synchronize()
import gc; gc.disable()
def region(mode):
    fl[:] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(8, 8 + K):
        env.step_async(ptrs[t])
    t1 = time.perf_counter()
    if mode == 1:
        for i, s in enumerate(streams):
            rc = hip.hipStreamWriteValue32(C.c_void_p(s), C.c_void_p(fptr + 4 * i), 1, 0)
            assert rc == 0, rc
        while not (fl[0] and fl[1]):
            pass
    if mode in (2, 3):
        for ev, st in zip(evs, env._slice_streams):
            ev.record(st)
        if mode == 2:
            for i, s_ in enumerate(streams):
                hip.hipStreamWriteValue32(C.c_void_p(s_), C.c_void_p(fptr + 4 * i), 1, 0)
            while not (fl[0] and fl[1]):
                pass
    if mode == 4:
        for st in env._slice_streams:
            st.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
evs = [torch.cuda.Event() for _ in streams]
for mode in (0, 2, 3, 4, 0, 2, 3, 4):
    r = np.array([region(mode) for _ in range(25)])
    print("mode", mode, "K", K, "median total %.1f (min %.1f) | enqueue %.1f | spin %.1f | sync %.1f  -> %.2f us/step" % (
        np.median(r[:, 0]), r[:, 0].min(), np.median(r[:, 1]), np.median(r[:, 2]), np.median(r[:, 3]), np.median(r[:, 0]) / K))
