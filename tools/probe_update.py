import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for d, rows in ((768, 24960), (2048, 166400)):
    W = torch.randn(rows, d, device="cuda") * 0.03
    out = torch.empty_like(W)
    for Ne in (50, 100, 130, 200, 256):
        NEP = (Ne + 63) // 64 * 64
        T = torch.randn(rows, NEP, device="cuda"); T[:, Ne:] = 0
        R = torch.randn(Ne, d, device="cuda") * 0.01
        us = timeit(lambda: H.lowrank_update(W, T, R, out=out))
        want = W.double() + T[:, :Ne].double() @ R.double() if rows < 30000 else None
        err = float((out.double() - want).norm() / want.norm()) if want is not None else -1
        print(f"d={d} Ne={Ne}: {us:.1f} us  relF {err:.1e}")
