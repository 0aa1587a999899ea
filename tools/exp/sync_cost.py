import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
x = torch.zeros(1 << 20, device="cuda")
s2 = torch.cuda.Stream()
def t(f, n=50):
    r = []
    for _ in range(n):
        a = time.perf_counter(); f(); r.append((time.perf_counter() - a) * 1e6)
    r.sort(); return "%.1f/%.1f" % (r[len(r) // 2], r[int(len(r) * 0.9)])
torch.cuda.synchronize()
print("env", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "ROC_", "HIP_", "AMD_", "GPU_"))})
print("idle torch.cuda.synchronize (median/p90 us):", t(torch.cuda.synchronize))
def k1():
    x.add_(1); torch.cuda.synchronize()
print("1 small kernel + synchronize:", t(k1))
def k2():
    x.add_(1)
    with torch.cuda.stream(s2): x.add_(1)
    torch.cuda.synchronize()
print("2 streams kernel + synchronize:", t(k2))
def k3():
    x.add_(1); e = torch.cuda.Event(); e.record()
    while not e.query(): pass
    a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
r = sorted(k3() for _ in range(50)); print("synchronize after polled completion: %.1f/%.1f" % (r[25], r[45]))
def k4():
    x.add_(1)
    with torch.cuda.stream(s2): x.add_(1); e2 = torch.cuda.Event(); e2.record()
    e = torch.cuda.Event(); e.record()
    while not (e.query() and e2.query()): pass
    a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
r = sorted(k4() for _ in range(50)); print("2 streams: synchronize after polled completion: %.1f/%.1f" % (r[25], r[45]))
def k5():
    x.add_(1)
    with torch.cuda.stream(s2): x.add_(1)
    a = time.perf_counter(); torch.cuda.current_stream().synchronize(); s2.synchronize(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    return ((b - a) * 1e6, (c - b) * 1e6)
r = [k5() for _ in range(50)]; print("2 streams: stream syncs then device sync: %.1f + %.1f" % (sorted(q[0] for q in r)[25], sorted(q[1] for q in r)[25]))
for N in (1, 10, 40, 80):
    def k6():
        for _ in range(N):
            x.add_(1)
        e = torch.cuda.Event(); e.record()
        while not e.query(): pass
        a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
    r = sorted(k6() for _ in range(30)); print("N=%d kernels, polled completion, then synchronize: %.1f/%.1f" % (N, r[15], r[27]))
for N in (40,):
    def k7():
        evs = []
        for i in range(N):
            x.add_(1)
            if i % 8 == 7:
                e = torch.cuda.Event(); e.record(); evs.append(e)
            if len(evs) > 1: evs[-2].query()
        e = torch.cuda.Event(); e.record()
        while not e.query(): pass
        a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
    r = sorted(k7() for _ in range(30)); print("N=%d kernels with periodic event queries: %.1f/%.1f" % (N, r[15], r[27]))
    def k8():
        for i in range(N):
            x.add_(1)
        e = torch.cuda.Event(); e.record()
        while not e.query(): pass
        a = time.perf_counter(); torch.cuda.current_stream().synchronize(); b = time.perf_counter(); torch.cuda.synchronize(); return ((b - a) * 1e6, (time.perf_counter() - b) * 1e6)
    r = [k8() for _ in range(30)]; print("N=%d polled, stream.synchronize %.1f then device synchronize %.1f" % (N, sorted(q[0] for q in r)[15], sorted(q[1] for q in r)[15]))
def k9():
    for i in range(40):
        x.add_(1)
    st = torch.cuda.current_stream()
    while not st.query(): pass
    a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
r = sorted(k9() for _ in range(30)); print("N=40 polled with stream.query(), then device synchronize: %.1f/%.1f" % (r[15], r[27]))
def k10():
    for i in range(40):
        x.add_(1)
    with torch.cuda.stream(s2):
        for i in range(40): x.add_(1)
    st = torch.cuda.current_stream()
    while not (st.query() and s2.query()): pass
    a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
r = sorted(k10() for _ in range(30)); print("N=40 x 2 streams polled with stream.query(), then device synchronize: %.1f/%.1f" % (r[15], r[27]))
def k11(timing):
    e0 = torch.cuda.Event(enable_timing=timing); e1 = torch.cuda.Event(enable_timing=timing)
    st = torch.cuda.current_stream()
    e0.record(st)
    for i in range(40):
        x.add_(1)
    e1.record(st)
    while not st.query(): pass
    a = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e6
for timing in (False, True):
    r = sorted(k11(timing) for _ in range(30)); print("N=40 with e0/e1 events (timing=%s), stream.query() poll, then device synchronize: %.1f/%.1f" % (timing, r[15], r[27]))
