"""GPU probe: low-rank apply kernels vs N_edit (fused and two-kernel forms) + whole-edit time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E, synth

H = E.UceHandle.get("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
rows = 24960 if d == 768 else 166400 // 4
W = torch.randn(rows, d, device="cuda") * 0.03
out = torch.empty_like(W)

def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print("copy_ us", timeit(lambda: out.copy_(W)))
for ne in (1, 16, 32, 48, 50, 64, 128, 256):
    Dm = torch.randn(ne, d, device="cuda"); R = torch.randn(ne, d, device="cuda") * 0.01
    t = timeit(lambda: H.apply_lowrank(W, Dm, R, out=out))
    T = H.lowrank_project(W, Dm)
    tp = timeit(lambda: H.lowrank_project(W, Dm))
    tu = timeit(lambda: H.lowrank_update(W, T, R, out=out))
    print(f"Ne={ne:4d}  fused {t:8.2f} us | project {tp:7.2f} us  update {tu:7.2f} us ({8*rows*d/tu/1e3:7.1f} GB/s)")
C = torch.from_numpy(synth.clip_like_embeddings(50, d, 0)).cuda(); G = C.roll(1, 0).contiguous(); s = torch.ones(50, device="cuda")
for name, fn in [("dual_factors (gram+potrf+trisolve)", lambda: H.dual_factors(C, G, s, 0.5)),
                 ("edit auto", lambda: H.edit(C, G, s, 0.5, W, out=out))]:
    print(name, f"{timeit(fn):.2f} us")
