"""GPU probe: dense apply kernel (variant by UCE_APPLY_VARIANT: 0 = f32 MFMA, 1 = 3 x bf16 split) - time, error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
for rows, d in ((24960, 768), (41600, 2048), (1000, 1024)):
    torch.manual_seed(0)
    W = (torch.rand(rows, d, device="cuda") * 2 - 1) / d ** 0.5
    DT = torch.randn(d, d, device="cuda") * 0.02
    out = H.apply(W, DT)
    ref = W.double() + W.double() @ DT.double().T
    err = ((out.double() - ref).norm() / ref.norm()).item()
    errd = ((out.double() - ref).norm() / (W.double() @ DT.double().T).norm()).item()
    for _ in range(3): H.apply(W, DT, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): H.apply(W, DT, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"variant {os.environ.get('UCE_APPLY_VARIANT', 'default')} rows={rows} d={d}: {us:8.1f} us  {2*rows*d*d/us/1e6:7.1f} TF(f32-equivalent)  "
          f"rel err vs f64 {err:.2e} (of the update term {errd:.2e})", flush=True)
