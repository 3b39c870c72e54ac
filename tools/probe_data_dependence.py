"""GPU probe: is a kernel's duration a function of what ran just before it (clocks / power) and of its data?
uce_sattn_fwd at B = 32, L = 4096, dh = 40: back-to-back launches vs launches separated by a memory-bound copy."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
B, L, dh = 32, 4096, 40
C = 8 * dh
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); big2 = torch.empty_like(big)
for name, gen in (("randn", lambda: torch.randn(B, L, C, device="cuda")), ("zeros", lambda: torch.zeros(B, L, C, device="cuda"))):
    q, k, v = gen().bfloat16(), gen().bfloat16(), gen().bfloat16()
    o = torch.empty_like(q)
    for _ in range(3): H.sattn(q, k, v, 8, out=o)
    torch.cuda.synchronize()
    for spacer in (False, True):
        ts = []
        for _ in range(20):
            if spacer:
                big2.copy_(big)                       # ~170 us of HBM-bound work between the attention launches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); H.sattn(q, k, v, 8, out=o); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"{name:6s} spacer={spacer!s:5s}: median {ts[len(ts)//2]:8.1f} us  min {ts[0]:8.1f}  max {ts[-1]:8.1f}", flush=True)
