"""Round-4 measurements on the GPU box (one process, prints JSON lines):
  gemm   uce_linear_fwd vs torch (hipBLASLt) at the U-Net's linear shapes of the generation batch, per tile form
  conv   the direct-to-LDS convolution per tile form vs im2col + GEMM at the U-Net's 16 x 16 / 8 x 8 / stride-2 layers
  sattn  self-attention with / without the V^T pre-pass at the four attn1 shapes
Usage: python tools/probe_r04.py [gemm] [conv] [sattn]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def handle(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return E.UceHandle("cuda:0")
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


PROBE_B = int(os.environ.get("UCE_PROBE_B", "32"))      # CFG batch (32 = 16 prompts per call; 128 = the bench's 64)


def gemm():
    B = PROBE_B
    shapes = []
    for hw, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
        M = B * hw
        shapes += [(M, C, C, "proj"), (M, 3 * C, C, "qkv"), (M, 8 * C, C, "ff_proj"), (M, C, 4 * C, "ff_out")]
    shapes += [(32, 17920, 1280, "temb_cat"), (B * 77, 320, 768, "ctx_kv")]
    H0 = handle(UCE_GEMM_W1=0)
    for M, N, K, tag in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        r = torch.randn(M, N, device="cuda").bfloat16()
        ent = {"M": M, "N": N, "K": K, "tag": tag, "gflop": 2e-9 * M * N * K}
        ent["torch_us"] = timeit(lambda: F.linear(x, w, b))
        ent["own_us"] = timeit(lambda: H0.linear(x, w, b))
        ent["own_res_us"] = timeit(lambda: H0.linear(x, w, b, r))
        Hw = handle(UCE_GEMM_W1=2)                               # the one-wave-per-SIMD kernel (uce_conv_w1.hip, one tap)
        ent["w1_us"] = timeit(lambda: Hw.linear(x, w, b))
        ent["w1_res_us"] = timeit(lambda: Hw.linear(x, w, b, r))
        ent["own2_us"] = timeit(lambda: H0.linear(x, w, b))
        torch.cuda.synchronize()
        Hw.close()
        if tag == "ff_proj":
            ent["torch_geglu_us"] = timeit(lambda: H0.geglu(F.linear(x, w, b)))
            ent["own_geglu_us"] = timeit(lambda: H0.linear(x, w, b, geglu=True))
        for tile in (() if os.environ.get("UCE_PROBE_FAST") == "1" else (256320, 128320, 256256, 2128320, 3128256, 64256320, 64256256, 64128320)):
            Hv = handle(UCE_GEMM_TILE=tile)
            ent[f"t{tile}_us"] = timeit(lambda: Hv.linear(x, w, b))
            if tag == "ff_proj":
                ent[f"g{tile}_us"] = timeit(lambda: Hv.linear(x, w, b, geglu=True))
            torch.cuda.synchronize()
            Hv.close()
        ent["own_TFs"] = ent["gflop"] / ent["own_us"] * 1e-3
        ent["torch_TFs"] = ent["gflop"] / ent["torch_us"] * 1e-3
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in ent.items()}), flush=True)


def conv():
    B = PROBE_B
    cases = [(B, 320, 320, 64, 64, 1), (B, 640, 640, 32, 32, 1), (B, 1280, 1280, 16, 16, 1), (B, 2560, 1280, 16, 16, 1),
             (B, 1280, 1280, 8, 8, 1), (B, 2560, 1280, 8, 8, 1), (B, 1920, 640, 32, 32, 1), (B, 960, 320, 64, 64, 1),
             (B, 320, 320, 64, 64, 2), (B, 640, 640, 32, 32, 2), (B, 1280, 1280, 16, 16, 2),
             (B, 640, 320, 64, 64, 1), (B, 320, 640, 32, 32, 1), (B, 1280, 640, 32, 32, 1), (B, 640, 1280, 16, 16, 1)]
    if os.environ.get("UCE_PROBE_VAE") == "1":                 # the VAE decoder's layers at a decode batch of 16 images
        cases = [(16, 512, 512, 64, 64, 1), (16, 512, 512, 128, 128, 1), (16, 512, 256, 256, 256, 1), (16, 256, 256, 256, 256, 1),
                 (8, 256, 128, 512, 512, 1)]
    H0 = handle(UCE_CONV_W1=0)                                   # the 8-wave forms: the fixed reference of every row
    for N, Cin, Cout, Hh, Ww, stride in cases:
        x = torch.randn(N, Cin, Hh, Ww, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (9 * Cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device="cuda").bfloat16()
        Ho, Wo = Hh // stride, Ww // stride
        ent = {"N": N, "Cin": Cin, "Cout": Cout, "H": Hh, "W": Ww, "stride": stride, "gflop": 2e-9 * N * Ho * Wo * 9 * Cin * Cout}
        fast = os.environ.get("UCE_PROBE_FAST") == "1"          # own forms only (MIOpen's first call per shape searches for seconds)
        if not fast:
            ent["miopen_us"] = timeit(lambda: F.conv2d(x, w, b, stride=stride, padding=1), 5)
        ent["rule_us"] = timeit(lambda: H0.conv3x3_igemm(x, w, b, stride=stride))
        Hw = handle(UCE_CONV_W1=2)                               # the one-wave-per-SIMD form (uce_conv_w1.hip) wherever it exists
        ent["w1_us"] = timeit(lambda: Hw.conv3x3_igemm(x, w, b, stride=stride))
        ent["rule2_us"] = timeit(lambda: H0.conv3x3_igemm(x, w, b, stride=stride))
        ent["w1_TFs"] = ent["gflop"] / ent["w1_us"] * 1e-3
        torch.cuda.synchronize()
        Hw.close()
        if stride == 1 and not fast:
            wmat = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
            cols = torch.empty(N * Hh * Ww, 9 * Cin, device="cuda", dtype=torch.bfloat16)
            y = torch.empty(N * Hh * Ww, Cout, device="cuda", dtype=torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream

            def im2col_own():
                H0.lib.uce_im2col3x3_nhwc(H0._h, x.data_ptr(), cols.data_ptr(), N, Hh, Ww, Cin, 0, st)
                H0.linear(cols, wmat, b, out=y)

            def im2col_lib():
                H0.lib.uce_im2col3x3_nhwc(H0._h, x.data_ptr(), cols.data_ptr(), N, Hh, Ww, Cin, 0, st)
                torch.addmm(b, cols, wmat.t(), out=y)
            ent["im2col_own_gemm_us"] = timeit(im2col_own)
            ent["im2col_lib_gemm_us"] = timeit(im2col_lib)
        for tile in (() if fast else (256320, 128320, 256256, 128128, 64256320, 64128320, 64256256)):
            if Cout % (tile % 1000):
                continue
            Hv = handle(UCE_CONV_TILE=tile)
            ent[f"t{tile}_us"] = timeit(lambda: Hv.conv3x3_igemm(x, w, b, stride=stride))
            torch.cuda.synchronize()
            Hv.close()
        ent["rule_TFs"] = ent["gflop"] / ent["rule_us"] * 1e-3
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in ent.items()}), flush=True)


def sattn():
    B, heads = 32, 8
    for L, dh in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
        C = heads * dh
        qkv = torch.randn(B, L, 3 * C, device="cuda").bfloat16()
        q, k, v = (qkv[..., i * C:(i + 1) * C].contiguous() for i in range(3))
        ent = {"L": L, "dh": dh, "gflop": 4e-9 * B * heads * L * L * dh}
        sp = lambda t: t.view(B, L, heads, dh).transpose(1, 2)  # noqa: E731
        ent["torch_sdpa_us"] = timeit(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v)), 5)
        for vti in (1, 2):
            Hv = handle(UCE_SATTN_VTI=vti)
            ent[f"vti{vti}_us"] = timeit(lambda: Hv.sattn(q, k, v, heads), 5)
            ent[f"vti{vti}_packed_us"] = timeit(lambda: Hv.sattn_packed(qkv, heads), 5)
            torch.cuda.synchronize()
            Hv.close()
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in ent.items()}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "conv", "sattn"]
    for name in what:
        print(f"## {name}", flush=True)
        {"gemm": gemm, "conv": conv, "sattn": sattn}[name]()
