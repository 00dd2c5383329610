import torch, time
n = 1_000_000_000
a = torch.rand(n, device='cuda'); b = torch.rand(n, device='cuda'); c = torch.empty_like(a)
def t(f, bytes_, name):
    for _ in range(2): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e)/10
    print(f"{name}: {ms:.3f} ms  {bytes_/ms/1e6:.1f} GB/s")
t(lambda: torch.add(a,b,out=c), 12e9, "add 2R1W")
t(lambda: c.copy_(a), 8e9, "copy 1R1W")
t(lambda: a.sum(), 4e9, "sum 1R")
t(lambda: c.fill_(1.0), 4e9, "fill 1W")
