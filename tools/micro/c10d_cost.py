"""Cost of one c10d collective on a one-rank RCCL group, host side and on the stream it is issued from (the GPU time a
pair of tiny kernels around it sees): python tools/micro/c10d_cost.py"""
import os, time, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
t = torch.zeros(6_000_000, device="cuda:0")
small = torch.zeros(8, device="cuda:0")
a = torch.zeros(1024, device="cuda:0")
def loop(fn, n=200):
    for _ in range(20):
        a.add_(1.0); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        a.add_(1.0); fn(); a.add_(1.0)
    e1.record(); h = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return h, e0.elapsed_time(e1) / n * 1e3
base = loop(lambda: None)
print(f"two tiny kernels alone:            host {base[0]:6.1f} us  stream {base[1]:6.1f} us per iteration")
for name, fn in (("all_reduce 24 MB", lambda: dist.all_reduce(t)), ("all_reduce 32 B", lambda: dist.all_reduce(small)),
                 ("reduce_scatter_tensor 24 MB in place", lambda: dist.reduce_scatter_tensor(t[:6_000_000], t)),
                 ("all_gather_into_tensor 24 MB in place", lambda: dist.all_gather_into_tensor(t, t[:6_000_000]))):
    h, g = loop(fn)
    print(f"{name:40s} host +{h - base[0]:6.1f} us  stream +{g - base[1]:6.1f} us per call")
# does a collective hold the HOST until the stream has reached it?  (5 ms of GPU work queued in front)
big = torch.zeros(64 << 20, device="cuda:0")
for name, fn in (("all_reduce", lambda: dist.all_reduce(t)), ("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(t[:6_000_000], t)),
                 ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(t, t[:6_000_000]))):
    torch.cuda.synchronize()
    for _ in range(40):
        big.add_(1.0)
    t0 = time.perf_counter(); fn(); h = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"{name:24s} behind ~{tot*1e3:.1f} ms of queued GPU work: the call returned after {h*1e3:.3f} ms")
dist.destroy_process_group()
