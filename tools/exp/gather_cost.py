"""Host cost of handing one window to RCCL (1 rank): gather(list) vs all_gather_into_tensor vs a copy."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
buf = torch.zeros((16, 8192, 4), dtype=torch.int32, device=dev)
recv = [torch.zeros_like(buf)]
out = torch.zeros((1,) + tuple(buf.shape), dtype=torch.int32, device=dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    xs = []
    for _ in range(n):
        t0 = time.perf_counter(); w = fn(); xs.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    xs.sort(); return xs[len(xs) // 2] * 1e6, xs[-1] * 1e6
print("gather(list) async", t(lambda: dist.gather(buf, recv, dst=0, async_op=True)))
print("all_gather_into_tensor async", t(lambda: dist.all_gather_into_tensor(out, buf, async_op=True)))
print("all_gather(list) async", t(lambda: dist.all_gather(recv, buf, async_op=True)))
print("reduce async", t(lambda: dist.reduce(buf, dst=0, async_op=True)))
print("copy_", t(lambda: out[0].copy_(buf, non_blocking=True)))
s = torch.cuda.Stream(priority=-1)
e = torch.cuda.Event()
def ev():
    e.record(); s.wait_event(e)
print("event record+wait", t(ev))
dist.destroy_process_group()
