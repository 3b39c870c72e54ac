"""GPU probe: GroupNorm(+SiLU) kernel vs torch (error, time) at U-Net shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, C, Hh, G, dt) in ((2, 32, 8, 8, torch.float16), (2, 32, 8, 8, torch.bfloat16), (32, 320, 64, 32, torch.bfloat16), (32, 640, 32, 32, torch.bfloat16),
                          (32, 1280, 16, 32, torch.bfloat16), (32, 2560, 8, 32, torch.bfloat16), (32, 960, 32, 32, torch.bfloat16)):
    x = (torch.randn(N, C, Hh, Hh) * 1.5 + 0.3).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(G, C).to("cuda", dt)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.normal_(0, 0.2)
        y = H.groupnorm_nhwc(x, gn.weight, gn.bias, G, 1e-5, True)
        ref = F.silu(F.group_norm(x.double(), G, gn.weight.double(), gn.bias.double(), 1e-5))
        tt = F.silu(gn(x))
        err = ((y.double() - ref).norm() / ref.norm()).item(); err_t = ((tt.double() - ref).norm() / ref.norm()).item()
        t_hip = timeit(lambda: H.groupnorm_nhwc(x, gn.weight, gn.bias, G, 1e-5, True))
        t_torch = timeit(lambda: F.silu(gn(x)))
    byts = 3 * x.numel() * 2
    print(f"N={N} C={C} HW={Hh}x{Hh} {str(dt)[6:]}: hip {t_hip:7.1f} us ({byts/t_hip/1e3:6.0f} GB/s) err {err:.2e} | torch {t_torch:7.1f} us err {err_t:.2e}", flush=True)
