"""HBM write / copy bandwidth seen by plain streaming kernels (torch fill_ / copy_), to put the GEMM epilogues'
store phase (155 MB in ~38 us = 4.1 TB/s) against what a pure store stream reaches on the same box."""
import torch
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mb in (77, 155, 310, 1240):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    us = t(lambda: a.fill_(1.0)); print(f"fill  {mb:5d} MB  {us:8.1f} us  {mb/us*1e-3*1e3:7.2f} GB/ms = {mb*1e6/us/1e6:6.2f} TB/s written")
    us = t(lambda: b.copy_(a)); print(f"copy  {mb:5d} MB  {us:8.1f} us  read+write {2*mb*1e6/us/1e6:6.2f} TB/s")
    us = t(lambda: a.sum()); print(f"sum   {mb:5d} MB  {us:8.1f} us  read {mb*1e6/us/1e6:6.2f} TB/s")
