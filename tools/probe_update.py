"""GPU probe: uce_lowrank_update alone (variant picked by UCE_UPDATE_VARIANT) - time and error vs torch f64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E

H = E.UceHandle.get("cuda:0")
d = int(os.environ.get("D", "768"))
rows = int(os.environ.get("ROWS", "24960"))
nes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "50").split(",")]
torch.manual_seed(0)
W = torch.randn(rows, d, device="cuda") * 0.03
out = torch.empty_like(W)

def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

line = [f"variant {os.environ.get('UCE_UPDATE_VARIANT', 'default'):>7s} d={d} rows={rows}:"]
for ne in nes:
    nep = (ne + 63) // 64 * 64
    T = torch.randn(rows, nep, device="cuda"); T[:, ne:] = float("nan")       # pad columns must be ignored
    R = torch.randn(ne, d, device="cuda") * 0.01
    out.fill_(float("nan"))
    H.lowrank_update(W, T, R, out=out)
    ref = W.double() + T[:, :ne].double() @ R.double()
    err = ((out.double() - ref).norm() / ref.norm()).item()
    tu = timeit(lambda: H.lowrank_update(W, T, R, out=out))
    line.append(f"Ne={ne}: {tu:6.2f} us {(8*rows*d + 4*rows*nep)/tu/1e3:6.0f} GB/s err {err:.1e} |")
print(" ".join(line), flush=True)
