"""GPU probe: whole-edit time for several (N_edit, N_preserve) on the SD-1.4 slab."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E, synth
H = E.UceHandle.get("cuda:0")
d, rows = 768, 24960
W = torch.randn(rows, d, device="cuda") * 0.03
out = torch.empty_like(W)
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for ne, np_ in ((1, 0), (2, 3), (10, 0), (32, 0), (50, 0), (60, 4), (64, 36), (100, 0), (200, 56), (300, 100), (500, 200), (700, 0), (1000, 500)):
    n = ne + np_
    C = torch.from_numpy(synth.clip_like_embeddings(n + 1, d, n)).cuda()
    G = C[n:n + 1].repeat(ne, 1).contiguous(); C = C[:n].contiguous(); s = torch.ones(n, device="cuda")
    t = timeit(lambda: H.edit(C, G, s, 0.5, W, out=out), 30)
    H.status()
    print(f"N_e={ne:5d} N_p={np_:4d}: {t:9.1f} us  {n / t * 1e6:12.0f} concepts/s")
