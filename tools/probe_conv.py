"""GPU probe: 3x3 stride-1 convolution of the U-Net at the generation batch - MIOpen (F.conv2d, channels_last) vs
im2col (9 shifted NHWC slices concatenated) + one hipBLASLt GEMM (F.linear) with K = 9*Cin."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for Cin, Cout, Hh in ((320, 320, 64), (640, 640, 32), (1280, 1280, 16), (960, 320, 64), (640, 320, 64), (1920, 640, 32), (2560, 1280, 16), (1280, 1280, 8)):
    x = torch.randn(B, Cin, Hh, Hh, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to("cuda", torch.bfloat16).to(memory_format=torch.channels_last)
    wmat = conv.weight.detach().permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    def via_gemm():
        xn = x.permute(0, 2, 3, 1)                                   # NHWC view
        xp = F.pad(xn, (0, 0, 1, 1, 1, 1))
        cols = torch.cat([xp[:, ky:ky + Hh, kx:kx + Hh, :] for ky in range(3) for kx in range(3)], dim=-1)
        return F.linear(cols.reshape(-1, 9 * Cin), wmat, conv.bias).view(B, Hh, Hh, Cout).permute(0, 3, 1, 2)
    cols = torch.randn(B * Hh * Hh, 9 * Cin, device="cuda").bfloat16()
    with torch.no_grad():
        a, b = conv(x), via_gemm()
        err = ((a.float() - b.float()).norm() / a.float().norm()).item()
        t_mi = timeit(lambda: conv(x)); t_g = timeit(via_gemm); t_lin = timeit(lambda: F.linear(cols, wmat, conv.bias))
    fl = 2.0 * B * Hh * Hh * Cin * Cout * 9
    print(f"B={B} {Cin}->{Cout} @{Hh}x{Hh}: MIOpen {t_mi:7.1f} us ({fl/t_mi/1e6:6.0f} TF/s) | cat+GEMM {t_g:7.1f} us | GEMM alone {t_lin:7.1f} us ({fl/t_lin/1e6:6.0f} TF/s) | rel diff {err:.1e}", flush=True)
