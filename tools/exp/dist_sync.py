import os, time, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
def t(f, n=30):
    r = []
    for _ in range(n):
        a = time.perf_counter(); f(); r.append((time.perf_counter() - a) * 1e6)
    r.sort(); return "%.1f/%.1f" % (r[len(r) // 2], r[int(len(r) * 0.9)])
x = torch.zeros(1 << 20, device="cuda")
def k40():
    for _ in range(40): x.add_(1)
    torch.cuda.synchronize()
torch.cuda.synchronize()
print("before init: idle sync", t(torch.cuda.synchronize), " 40 kernels + sync", t(k40))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.synchronize()
print("after init : idle sync", t(torch.cuda.synchronize), " 40 kernels + sync", t(k40))
dist.barrier(); torch.cuda.synchronize()
print("after a barrier: idle sync", t(torch.cuda.synchronize), " 40 kernels + sync", t(k40))
def kb():
    for _ in range(40): x.add_(1)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
print("40 kernels + sync + barrier + sync", t(kb))
dist.destroy_process_group()
