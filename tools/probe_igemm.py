"""GPU probe: 3x3 convolutions of the U-Net (generation batch) and the VAE decoder - uce_im2col3x3_nhwc + hipBLASLt GEMM
(conv3x3_nhwc) vs the implicit-GEMM kernel (conv3x3_igemm) vs MIOpen (F.conv2d, channels_last); us and TF/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from uce_amd import edit as E  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


H = E.UceHandle.get("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(B, 320, 320, 64), (B, 640, 640, 32), (B, 1280, 1280, 16), (B, 960, 320, 64), (B, 640, 320, 64), (B, 1920, 640, 32),
          (B, 2560, 1280, 16), (B, 1280, 1280, 8), (B, 320, 640, 32), (B, 640, 1280, 16),
          (B // 2, 512, 512, 64), (B // 2, 512, 512, 128), (B // 2, 512, 256, 256), (B // 2, 256, 256, 256),
          (B // 2, 256, 128, 512), (B // 2, 128, 128, 512)]
for N, Cin, Cout, Hh in shapes:
    x = torch.randn(N, Cin, Hh, Hh, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to("cuda", torch.bfloat16).to(memory_format=torch.channels_last)
    with torch.no_grad():
        a, b = H.conv3x3_nhwc(x, conv.weight, conv.bias), H.conv3x3_igemm(x, conv.weight, conv.bias)
        err = ((a.float() - b.float()).norm() / a.float().norm()).item()
        t_a = timeit(lambda: H.conv3x3_nhwc(x, conv.weight, conv.bias))
        t_b = timeit(lambda: H.conv3x3_igemm(x, conv.weight, conv.bias))
        t_m = timeit(lambda: F.conv2d(x, conv.weight, conv.bias, padding=1)) if N * Hh * Hh <= 32 * 64 * 64 else float("nan")
    fl = 2.0 * N * Hh * Hh * Cin * Cout * 9
    print(f"N={N} {Cin}->{Cout} @{Hh}: im2col+GEMM {t_a:8.1f} us ({fl / t_a / 1e6:6.0f} TF/s) | igemm {t_b:8.1f} us "
          f"({fl / t_b / 1e6:6.0f} TF/s) | MIOpen {t_m:8.1f} us | rel diff {err:.1e}", flush=True)
