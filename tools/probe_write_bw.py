import torch, time
W = torch.empty(24960, 768, device="cuda")
src = torch.randn(24960, 768, device="cuda")
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mb = W.numel() * 4 / 1e6
us = t(lambda: W.fill_(1.0)); print(f"fill  {us:.1f} us  {mb/us*1e-3*1e3:.0f} GB/s write")
us = t(lambda: W.copy_(src)); print(f"copy  {us:.1f} us  {2*mb/us:.0f} GB/s rw")
us = t(lambda: torch.add(src, 1.0, out=W)); print(f"add   {us:.1f} us  {2*mb/us:.0f} GB/s rw")
s = torch.zeros(1, device="cuda")
us = t(lambda: torch.sum(src, dtype=torch.float32)); print(f"sum   {us:.1f} us  {mb/us:.0f} GB/s read")
