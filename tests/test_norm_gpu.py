"""GPU parity of the channels-last GroupNorm (+ SiLU) kernel (uce_groupnorm_nhwc_fwd) against torch's GroupNorm
evaluated in fp32/fp64 - the arithmetic diffusers' ResnetBlock2D / Transformer2DModel / VAE decoder perform."""
import pytest
import torch
import torch.nn.functional as F

from oracle import uce_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from uce_amd import edit as E
    return E.UceHandle.get("cuda:0")


@pytest.mark.parametrize("N,C,Hh,Ww,G,eps,silu,dtype", [
    (2, 320, 64, 64, 32, 1e-5, True, torch.bfloat16),      # SD-1.4 U-Net shapes (channels / resolution pairs)
    (2, 640, 32, 32, 32, 1e-5, True, torch.bfloat16),
    (2, 1280, 16, 16, 32, 1e-5, True, torch.bfloat16),
    (2, 2560, 8, 8, 32, 1e-5, True, torch.bfloat16),       # up-block concatenations
    (2, 1920, 16, 16, 32, 1e-5, True, torch.bfloat16),
    (2, 960, 32, 32, 32, 1e-5, True, torch.bfloat16),
    (2, 320, 64, 64, 32, 1e-6, False, torch.bfloat16),     # Transformer2DModel.norm: no activation
    (1, 128, 128, 128, 32, 1e-6, True, torch.bfloat16),    # VAE decoder
    (3, 64, 5, 7, 8, 1e-5, True, torch.bfloat16),          # ragged pixel count, the tiny test model's groups
    (2, 32, 8, 8, 8, 1e-5, True, torch.float16),
    (32, 320, 64, 64, 32, 1e-5, True, torch.bfloat16),     # the generation batch
])
def test_groupnorm_nhwc(H, N, C, Hh, Ww, G, eps, silu, dtype):
    g = torch.Generator().manual_seed(C + Hh)
    x = (torch.randn(N, C, Hh, Ww, generator=g) * 1.5 + 0.3).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.rand(C, generator=g) + 0.5).to(dtype).cuda()
    b = (torch.randn(C, generator=g) * 0.2).to(dtype).cuda()
    y = H.groupnorm_nhwc(x, w, b, G, eps, silu)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    ref = F.group_norm(x.double(), G, w.double(), b.double(), eps)
    if silu:
        ref = F.silu(ref)
    assert torch.isfinite(y.float()).all()
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    # bit-repeatable (fixed summation order)
    assert torch.equal(y, H.groupnorm_nhwc(x, w, b, G, eps, silu))


def test_groupnorm_large_mean_is_stable(H):
    """A channel group with mean >> std: the variance comes from f32 sums folded in f64, not from E[x^2] - E[x]^2
    in f32."""
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(1, 64, 32, 32, generator=g) * 0.05 + 20.0).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w = torch.ones(64, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    y = H.groupnorm_nhwc(x, w, b, 8, 1e-5, False)
    ref = F.group_norm(x.double(), 8, w.double(), b.double(), 1e-5)
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < 2e-2


def test_unet_uses_the_kernel_and_matches_torch_groupnorm():
    """The U-Net forward with the HIP GroupNorm against the same weights through torch's GroupNorm + SiLU."""
    from uce_amd.sd import pipeline as sdp
    from uce_amd.sd import unet as U
    pipe = sdp.load_pipeline("tiny-sd-test", torch.bfloat16, "cuda:0", synthetic=True, vae=False, seed=3)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    ctx = torch.randn(2, 77, 64, generator=g).bfloat16().cuda()
    t = torch.tensor([500], device="cuda")
    calls = {"n": 0}
    from uce_amd import edit as E
    orig = E.UceHandle.groupnorm_nhwc

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)

    E.UceHandle.groupnorm_nhwc = counted
    try:
        a = pipe.unet(x, t, ctx).float()
    finally:
        E.UceHandle.groupnorm_nhwc = orig
    assert calls["n"] == 61                     # 22 resnets x 2 + 16 transformer norms + conv_norm_out
    from tests.torch_twin import torch_ops
    with torch_ops():                           # (every kernel off: a 16-bit GPU convolution has no library path to fall to)
        b = pipe.unet(x, t, ctx).float()
    assert O.rel_fro(a.cpu(), b.cpu()) < 3e-2


@pytest.mark.parametrize("shape,dtype", [((2, 4096, 2560), torch.bfloat16), ((3, 77, 512), torch.float16),
                                         ((1, 5, 16), torch.bfloat16), ((32, 1024, 5120), torch.bfloat16)])
def test_geglu(H, shape, dtype):
    g = torch.Generator().manual_seed(shape[-1])
    x = (torch.randn(*shape, generator=g) * 2).to(dtype).cuda()
    y = H.geglu(x)
    h, gate = x.double().chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    assert y.shape == ref.shape
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < (4e-3 if dtype == torch.bfloat16 else 6e-4)


@pytest.mark.parametrize("N,C,Hh,G,dtype", [(2, 320, 64, 32, torch.bfloat16), (3, 2560, 8, 32, torch.bfloat16),
                                            (2, 64, 5, 8, torch.float16)])
def test_groupnorm_with_per_channel_addend(H, N, C, Hh, G, dtype):
    """norm2 of a ResnetBlock2D: silu(group_norm(conv1(.) + (bias + time embedding)[:, :, None, None]))."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, Hh, Hh, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    ad = (torch.randn(N, C, generator=g) * 0.7).to(dtype).cuda()
    w = (torch.rand(C, generator=g) + 0.5).to(dtype).cuda()
    b = (torch.randn(C, generator=g) * 0.2).to(dtype).cuda()
    y = H.groupnorm_nhwc(x, w, b, G, 1e-5, True, addend=ad)
    ref = F.silu(F.group_norm(x.double() + ad.double()[:, :, None, None], G, w.double(), b.double(), 1e-5))
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < (4e-3 if dtype == torch.bfloat16 else 6e-4)


@pytest.mark.parametrize("shape,dtype", [((2, 320, 64, 64), torch.bfloat16), ((3, 64, 5, 7), torch.float16),
                                         ((32, 1280, 16, 16), torch.bfloat16)])
def test_add_bias_nhwc(H, shape, dtype):
    g = torch.Generator().manual_seed(shape[1])
    mk = lambda: torch.randn(*shape, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    a, b = mk(), mk()
    bias = torch.randn(shape[1], generator=g).to(dtype).cuda()
    want = a.double() + b.double() + bias.double()[None, :, None, None]
    y = H.add_bias_nhwc(a, b, bias)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert O.rel_fro(y.double().cpu(), want.cpu()) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    assert O.rel_fro(H.add_bias_nhwc(a, None, bias).double().cpu(), (a.double() + bias.double()[None, :, None, None]).cpu()) < 4e-3
    assert O.rel_fro(H.add_bias_nhwc(a, b, None).double().cpu(), (a.double() + b.double()).cpu()) < 4e-3


@pytest.mark.parametrize("N,Cin,Cout,Hh,Ww,dtype,bias", [
    (2, 320, 320, 64, 64, torch.bfloat16, True), (3, 64, 32, 5, 7, torch.bfloat16, False),
    (2, 960, 320, 16, 16, torch.bfloat16, True), (2, 32, 64, 8, 8, torch.float16, True),
    (5, 128, 128, 40, 24, torch.bfloat16, True), (2, 1920, 64, 8, 6, torch.bfloat16, True),   # C > 1280: flat kernel
])
def test_conv3x3_im2col_gemm_matches_torch(H, N, Cin, Cout, Hh, Ww, dtype, bias):
    """uce_im2col3x3_nhwc + one GEMM against F.conv2d evaluated in fp32 (borders, ragged sizes, batch chunking)."""
    g = torch.Generator().manual_seed(Cin + Hh)
    x = torch.randn(N, Cin, Hh, Ww, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=bias).to("cuda", dtype).to(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = F.conv2d(x.float(), conv.weight.float(), None if conv.bias is None else conv.bias.float(), padding=1)
        y = H.conv3x3_nhwc(x, conv.weight, conv.bias)
        y_chunked = H.conv3x3_nhwc(x, conv.weight, conv.bias, max_cols_bytes=Hh * Ww * 9 * Cin * 2 * 2)   # 2 images per chunk
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-3
    assert O.rel_fro(y.float().cpu(), ref.cpu()) < tol
    assert O.rel_fro(y_chunked.float().cpu(), ref.cpu()) < tol


@pytest.mark.parametrize("N,Cin,Cout,Hh,Ww,dtype,bias,up", [
    (2, 320, 320, 64, 64, torch.bfloat16, True, False),      # 256 x 64 tiles (Cout % 128 == 64)
    (3, 64, 32, 5, 7, torch.bfloat16, False, False),         # ragged pixels and channels, every border case
    (2, 960, 320, 16, 16, torch.bfloat16, True, False),
    (2, 64, 64, 8, 8, torch.float16, True, False),
    (5, 128, 128, 40, 24, torch.bfloat16, True, False),      # 128 x 128 tiles, tiles that straddle image rows / images
    (2, 1920, 640, 8, 6, torch.bfloat16, True, False),
    (1, 512, 256, 32, 32, torch.bfloat16, True, False),      # VAE-like
    (2, 640, 640, 16, 16, torch.bfloat16, True, True),       # fused 2x nearest upsample (Upsample2D)
    (3, 64, 72, 6, 10, torch.float16, False, True),
    # >= 16 384 pixels and 256 / 320-multiple outputs: the direct-to-LDS 256-pixel form (uce_conv_dma.hip)
    (2, 320, 320, 96, 96, torch.bfloat16, True, False),      # 256 x 320 tiles
    (3, 64, 640, 75, 75, torch.bfloat16, True, False),       # ragged last pixel tile, tiles straddling images, two channel tiles
    (1, 192, 256, 128, 128, torch.float16, True, False),     # 256 x 256 tiles, f16, 6 k-tiles per tap
    (2, 128, 512, 96, 96, torch.bfloat16, False, False),
    (2, 128, 320, 128, 128, torch.bfloat16, True, True),     # fused upsample through the DMA gather
    (1, 64, 256, 130, 126, torch.bfloat16, True, False),     # two k-tiles per tap: the ring spans taps
    (1, 128, 128, 160, 128, torch.bfloat16, True, False),    # 256 x 128 tiles (the VAE's 128-channel layers)
])
def test_conv3x3_implicit_gemm_matches_torch(H, N, Cin, Cout, Hh, Ww, dtype, bias, up):
    """uce_conv3x3_nhwc_fwd (one launch, taps gathered into LDS, no patch matrix) against F.conv2d evaluated in fp32;
    with `up` against conv(interpolate(x, 2, "nearest")) for a half-resolution input."""
    g = torch.Generator().manual_seed(Cin + Hh + Cout)
    hs, ws = (Hh // 2, Ww // 2) if up else (Hh, Ww)
    x = torch.randn(N, Cin, hs, ws, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=bias).to("cuda", dtype).to(memory_format=torch.channels_last)
    with torch.no_grad():
        xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
        ref = F.conv2d(xin, conv.weight.float(), None if conv.bias is None else conv.bias.float(), padding=1)
        y = H.conv3x3_igemm(x, conv.weight, conv.bias, upsample=up)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert O.rel_fro(y.float().cpu(), ref.cpu()) < (6e-3 if dtype == torch.bfloat16 else 1e-3)


def test_narrow_output_conv_goes_through_the_padded_implicit_gemm():
    """The VAE's conv_out (128 -> 3 channels at image resolution): sd.unet.conv2d pads the weight to 8 output channels,
    runs uce_conv3x3_nhwc_fwd and slices - against F.conv2d in fp32; an in-place weight update rebuilds the padded copy."""
    from uce_amd.sd import unet as U
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 128, 256, 256, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(128, 3, 3, padding=1).to("cuda", torch.bfloat16).to(memory_format=torch.channels_last)
    assert U._conv3x3_narrow_ok(conv, x)
    with torch.no_grad():
        for _ in range(2):
            ref = F.conv2d(x.float(), conv.weight.float(), conv.bias.float(), padding=1)
            y = U.conv2d(conv, x)
            assert y.shape == ref.shape and O.rel_fro(y.float().cpu(), ref.cpu()) < 6e-3
            conv.weight.mul_(-0.5)                                  # the cached padded weight must follow
            conv.bias.add_(1.0)


@pytest.mark.parametrize("N,C,Hh,Ww", [(2, 64, 5, 7), (3, 320, 16, 16), (1, 1288, 4, 3), (2, 2560, 3, 5), (1, 8, 1, 1)])
def test_im2col_patch_matrix_is_bit_exact(H, N, C, Hh, Ww):
    """Both patch-matrix kernels (row kernel for C <= 1280, flat kernel above) against nine shifted slices of the
    zero-padded NHWC input: pure data movement, so the comparison is bit-exact."""
    from uce_amd import lib as L
    g = torch.Generator().manual_seed(C + Ww)
    x = torch.randn(N, Hh, Ww, C, generator=g).to(torch.bfloat16).cuda()
    cols = torch.full((N * Hh * Ww, 9 * C), 7.0, dtype=torch.bfloat16, device="cuda:0")
    L.check(H.lib.uce_im2col3x3_nhwc(H._h, x.data_ptr(), cols.data_ptr(), N, Hh, Ww, C, 0,
                                     torch.cuda.current_stream().cuda_stream), "uce_im2col3x3_nhwc")
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    want = torch.cat([xp[:, ky:ky + Hh, kx:kx + Ww, :] for ky in range(3) for kx in range(3)], dim=-1)
    assert torch.equal(cols.view(N, Hh, Ww, 9 * C), want)


@pytest.mark.parametrize("rows_shape,C,dtype", [((2, 4096), 320, torch.bfloat16), ((2, 1024), 640, torch.bfloat16),
                                                ((3, 256), 1280, torch.bfloat16), ((5, 7), 768, torch.float16),
                                                ((1, 1), 8, torch.bfloat16), ((2, 33), 2560, torch.bfloat16),
                                                ((4, 9), 200, torch.float16), ((70000,), 64, torch.bfloat16)])
def test_layernorm_matches_torch_fp64(H, rows_shape, C, dtype):
    """uce_layernorm_fwd against F.layer_norm in fp64, with and without the fused residual join (whose sum must be
    the bit-exact rounding torch's own 16-bit add produces)."""
    g = torch.Generator().manual_seed(C + len(rows_shape))
    x = (torch.randn(*rows_shape, C, generator=g) * 1.5 + 0.3).to(dtype).cuda()
    r = torch.randn(*rows_shape, C, generator=g).to(dtype).cuda()
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    b = (0.1 * torch.randn(C, generator=g)).to(dtype).cuda()
    tol = 4e-3 if dtype == torch.bfloat16 else 6e-4
    y = H.layernorm(x, w, b, 1e-5)
    ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-5)
    assert y.shape == x.shape and y.dtype == dtype
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < tol
    s, y2 = H.layernorm(x, w, b, 1e-5, residual=r)
    assert torch.equal(s, x + r)
    ref2 = F.layer_norm((x + r).double(), (C,), w.double(), b.double(), 1e-5)
    assert O.rel_fro(y2.double().cpu(), ref2.cpu()) < tol
    # no worse than torch's own 16-bit LayerNorm kernel
    err_torch = O.rel_fro(F.layer_norm(x, (C,), w, b, 1e-5).double().cpu(), ref.cpu())
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < max(1.5 * err_torch, 1e-4)


def test_layernorm_rejects_bad_arguments(H):
    from uce_amd import lib as L
    x = torch.zeros(4, 4096, dtype=torch.bfloat16, device="cuda:0")          # C > 2560
    w = torch.ones(4096, dtype=torch.bfloat16, device="cuda:0")
    with pytest.raises(L.UceError):
        H.layernorm(x, w, w, 1e-5)
    x = torch.zeros(4, 12, dtype=torch.bfloat16, device="cuda:0")            # C % 8 != 0
    with pytest.raises(L.UceError):
        H.layernorm(x, w[:12], w[:12], 1e-5)


def test_transformer_block_with_hip_layernorm_matches_torch_ops():
    """BasicTransformerBlock through uce_layernorm_fwd (fused joins) against the same block through nn.LayerNorm."""
    from uce_amd.sd import unet as U
    torch.manual_seed(0)
    blk = U.BasicTransformerBlock(320, 8, 40, 768).to("cuda", torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 256, 320, generator=g).bfloat16().cuda()
    ctx = torch.randn(2, 77, 768, generator=g).bfloat16().cuda()
    a = blk(x, ctx).float()
    from tests.torch_twin import torch_ops
    with torch_ops("_hip_ln_ok"):
        b = blk(x, ctx).float()
    assert O.rel_fro(a.cpu(), b.cpu()) < 1e-2


@pytest.mark.parametrize("N,C,Hs,Ws", [(2, 64, 4, 6), (2, 640, 16, 16), (1, 1920, 3, 2), (1, 128, 40, 72)])
def test_im2col_with_fused_upsample_is_bit_exact(H, N, C, Hs, Ws):
    """upsample = 1 gathers the patches of the 2x nearest-neighbour upsampling straight from the half-resolution
    tensor: must equal the patch matrix of the materialised F.interpolate output, in both kernels."""
    from uce_amd import lib as L
    g = torch.Generator().manual_seed(C + Ws)
    x = torch.randn(N, Hs, Ws, C, generator=g).to(torch.bfloat16).cuda()
    Hh, Ww = 2 * Hs, 2 * Ws
    cols = torch.full((N * Hh * Ww, 9 * C), 7.0, dtype=torch.bfloat16, device="cuda:0")
    L.check(H.lib.uce_im2col3x3_nhwc(H._h, x.data_ptr(), cols.data_ptr(), N, Hh, Ww, C, 1,
                                     torch.cuda.current_stream().cuda_stream), "uce_im2col3x3_nhwc")
    xu = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    xp = F.pad(xu, (0, 0, 1, 1, 1, 1))
    want = torch.cat([xp[:, ky:ky + Hh, kx:kx + Ww, :] for ky in range(3) for kx in range(3)], dim=-1)
    assert torch.equal(cols.view(N, Hh, Ww, 9 * C), want)
    assert H.lib.uce_im2col3x3_nhwc(H._h, x.data_ptr(), cols.data_ptr(), N, 5, 6, C, 1,
                                    torch.cuda.current_stream().cuda_stream) == L.EINVAL     # odd output height


def test_upsample_conv_matches_interpolate_then_conv(H):
    from uce_amd.sd import unet as U
    torch.manual_seed(1)
    up = U.Upsample2D(64).to("cuda", torch.bfloat16).to(memory_format=torch.channels_last)
    x = torch.randn(3, 64, 8, 8).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    y = up(x)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), up.conv.weight.float(), up.conv.bias.float(), padding=1)
    assert y.shape == (3, 64, 16, 16) and y.is_contiguous(memory_format=torch.channels_last)
    assert O.rel_fro(y.float().cpu(), ref.cpu()) < 6e-3


# ------------------------------------------------------------------------------------ round 4: strides, residuals, tiles

def test_groupnorm_addend_as_a_column_slice(H):
    """The addend read in place from a wider tensor (row stride != C): a block's slice of the hoisted time projections."""
    g = torch.Generator().manual_seed(4)
    N, C, Hh = 3, 320, 16
    x = torch.randn(N, C, Hh, Hh, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    wide = torch.randn(N, 3 * C + 64, generator=g).bfloat16().cuda()
    w, b = torch.randn(C, generator=g).bfloat16().cuda(), torch.randn(C, generator=g).bfloat16().cuda()
    ad = wide[:, 64 + C:64 + 2 * C]
    assert not ad.is_contiguous()
    y = H.groupnorm_nhwc(x, w, b, 32, 1e-5, True, ad)
    assert torch.equal(y, H.groupnorm_nhwc(x, w, b, 32, 1e-5, True, ad.contiguous()))


@pytest.mark.parametrize("N,Cin,Cout,Hh,Ww,dtype,res", [
    (2, 320, 320, 64, 64, torch.bfloat16, False),        # Downsample2D of down_blocks.0
    (3, 640, 640, 32, 32, torch.bfloat16, False),
    (2, 1280, 1280, 16, 16, torch.bfloat16, False),
    (1, 64, 128, 10, 6, torch.float16, True),            # small ragged image, stride 2 + residual
])
def test_conv3x3_stride2_matches_torch(H, N, Cin, Cout, Hh, Ww, dtype, res):
    """diffusers' Downsample2D (3x3, stride 2, pad 1) on the direct-to-LDS kernel."""
    g = torch.Generator().manual_seed(Cin + Hh)
    x = torch.randn(N, Cin, Hh, Ww, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, generator=g).to(dtype).cuda()
    r = torch.randn(N, Cout, Hh // 2, Ww // 2, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last) if res else None
    y = H.conv3x3_nhwc(x, w, b, stride=2, residual=r)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    if res:
        ref = ref + r.double()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < (6e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("tile", ["0", "256320", "128320", "256256", "128256", "256128", "128128", "64256320", "64128320", "64256256", "64128256",
                                  "9128064", "9128128"])
@pytest.mark.parametrize("N,Cin,Cout,Hh,Ww", [(2, 64, 1280, 12, 10), (3, 96, 256, 16, 16), (2, 192, 640, 20, 14)])
def test_conv3x3_every_tile_form_with_residual(tile, N, Cin, Cout, Hh, Ww):
    """UCE_CONV_TILE pins the tile of the direct-to-LDS convolution: every form (where it divides Cout), with bias and the
    residual epilogue, ragged pixel tiles."""
    import os
    from uce_amd import edit as E
    if tile != "0" and Cout % (int(tile) % 1000):
        pytest.skip("tile width does not divide Cout")
    if int(tile) >= 1000000 and Cin % 64:
        pytest.skip("128-byte k-tiles need Cin % 64 == 0")
    old = os.environ.get("UCE_CONV_TILE")
    os.environ["UCE_CONV_TILE"] = tile
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_CONV_TILE"]
        else:
            os.environ["UCE_CONV_TILE"] = old
    g = torch.Generator().manual_seed(Cin + Cout + int(tile))
    x = torch.randn(N, Cin, Hh, Ww, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, generator=g).bfloat16().cuda()
    r = torch.randn(N, Cout, Hh, Ww, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    try:
        y = Hv.conv3x3_igemm(x, w, b, residual=r)
        torch.cuda.synchronize()
    finally:
        Hv.close()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < 6e-3


@pytest.mark.parametrize("N,Cout,Hh,Ww,dtype", [(2, 320, 64, 64, torch.bfloat16), (3, 512, 9, 7, torch.float16), (1, 32, 8, 8, torch.bfloat16)])
def test_conv_in_on_four_channels(H, N, Cout, Hh, Ww, dtype):
    """conv_in on the latents: uce_im2col3x3_c4 (bit-exact patch matrix) + uce_linear_fwd, through the module dispatch."""
    from uce_amd import lib as L
    from uce_amd.sd import unet as U
    g = torch.Generator().manual_seed(Cout + Hh)
    conv = torch.nn.Conv2d(4, Cout, 3, padding=1).to("cuda", dtype)
    x = torch.randn(N, 4, Hh, Ww, generator=g).to(dtype).cuda()
    cols = torch.full((N * Hh * Ww, 64), 7.0, dtype=dtype, device="cuda:0")
    xs = x.permute(0, 2, 3, 1).contiguous()
    L.check(H.lib.uce_im2col3x3_c4(H._h, xs.data_ptr(), cols.data_ptr(), N, Hh, Ww, torch.cuda.current_stream().cuda_stream),
            "uce_im2col3x3_c4")
    xp = F.pad(xs, (0, 0, 1, 1, 1, 1))
    want = torch.cat([xp[:, ky:ky + Hh, kx:kx + Ww, :] for ky in range(3) for kx in range(3)], dim=-1)
    assert torch.equal(cols.view(N, Hh, Ww, 64)[..., :36], want) and float(cols[:, 36:].abs().max()) == 0.0
    y = U.conv2d(conv, x)
    ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    assert y.shape == ref.shape
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < (6e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("N,C1,C2,Hh,Ww,G,dtype,addend", [
    (2, 1280, 1280, 8, 8, 32, torch.bfloat16, False),      # the up blocks' concatenations of SD-1.4 (x | skip)
    (2, 1280, 640, 16, 16, 32, torch.bfloat16, False),
    (2, 640, 320, 32, 32, 32, torch.bfloat16, True),
    (3, 320, 320, 64, 64, 32, torch.bfloat16, False),
    (2, 64, 8, 5, 7, 8, torch.float16, True),              # one octet from the second tensor, ragged pixel count
    (1, 2048, 2048, 4, 4, 32, torch.bfloat16, False),      # two octets per thread, the split between them
])
def test_groupnorm_of_a_concatenation_that_is_never_written(H, N, C1, C2, Hh, Ww, G, dtype, addend):
    """uce_groupnorm_cat_nhwc_fwd(x, x2) == uce_groupnorm_nhwc_fwd(torch.cat([x, x2], dim=1)), bit for bit (same thread <->
    channel map, same summation order), and the argument checks of the two-source form."""
    g = torch.Generator().manual_seed(C1 + C2 + Hh)
    cl = lambda t: t.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    x, x2 = cl(torch.randn(N, C1, Hh, Ww, generator=g) * 1.5 + 0.3), cl(torch.randn(N, C2, Hh, Ww, generator=g) - 0.2)
    C = C1 + C2
    w = (torch.rand(C, generator=g) + 0.5).to(dtype).cuda()
    b = (torch.randn(C, generator=g) * 0.2).to(dtype).cuda()
    ad = (torch.randn(N, C, generator=g) * 0.5).to(dtype).cuda() if addend else None
    got = H.groupnorm_nhwc(x, w, b, G, 1e-5, True, ad, x2=x2)
    want = H.groupnorm_nhwc(torch.cat([x, x2], dim=1).contiguous(memory_format=torch.channels_last), w, b, G, 1e-5, True, ad)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    from uce_amd import lib as L
    if C1 > 8:
        with pytest.raises(L.UceError):                    # C1 = 4 is not a whole octet
            H.groupnorm_nhwc(x[:, :4].contiguous(memory_format=torch.channels_last), w[:C2 + 4].contiguous(), b[:C2 + 4].contiguous(),
                             1, 1e-5, True, None, x2=x2)


@pytest.mark.parametrize("N,Cin,Cout,Hh,Ww,dtype,stride,up,bias,res", [
    (2, 320, 320, 64, 64, torch.bfloat16, 1, False, True, True),      # the resnet convolutions of the 64 x 64 level, BN = 320
    (3, 64, 640, 12, 10, torch.bfloat16, 1, False, True, True),       # ragged pixel tiles (360 pixels), one k-tile per tap
    (2, 192, 1280, 20, 14, torch.float16, 1, False, False, True),     # f16, three k-tiles per tap, four channel tiles, no bias
    (2, 128, 256, 16, 16, torch.bfloat16, 1, False, True, False),     # BN = 256 (the VAE's widths): 64 tiles per wave, all in AGPRs
    (1, 512, 512, 24, 24, torch.bfloat16, 1, False, True, True),
    (2, 320, 320, 32, 32, torch.bfloat16, 2, False, True, False),     # Downsample2D: stride 2
    (1, 64, 320, 10, 6, torch.float16, 2, False, True, True),         # stride 2, ragged, residual
    (2, 640, 640, 16, 16, torch.bfloat16, 1, True, True, False),      # Upsample2D: the 2x nearest upsample fused into the taps
])
def test_conv3x3_one_wave_per_simd_form(N, Cin, Cout, Hh, Ww, dtype, stride, up, bias, res):
    """k_conv3x3_w1 (UCE_CONV_W1=2: wherever the shape allows - 4 waves, 128 x 160 / 128 x 128 wave tiles, accumulators pinned in
    AGPRs / VGPRs by inline-asm MFMAs, DMAs and fragment reads issued from inside the MFMA stream) against torch in fp64, twice
    for bit-repeatability, and against the 8-wave form on the same inputs."""
    import os
    from uce_amd import edit as E
    old = os.environ.get("UCE_CONV_W1")
    os.environ["UCE_CONV_W1"] = "2"
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_CONV_W1"]
        else:
            os.environ["UCE_CONV_W1"] = old
    g = torch.Generator().manual_seed(Cin + Cout + Hh + stride)
    cl = lambda t: t.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn(N, Cin, Hh, Ww, generator=g))
    w = cl(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g).to(dtype).cuda() if bias else None
    Ho, Wo = (2 * Hh, 2 * Ww) if up else (Hh // stride, Ww // stride)
    r = cl(torch.randn(N, Cout, Ho, Wo, generator=g)) if res else None
    try:
        y = Hv.conv3x3_igemm(x, w, b, upsample=up, stride=stride, residual=r)
        again = Hv.conv3x3_igemm(x, w, b, upsample=up, stride=stride, residual=r)
        torch.cuda.synchronize()
    finally:
        Hv.close()
    xin = F.interpolate(x.double(), scale_factor=2.0, mode="nearest") if up else x.double()
    ref = F.conv2d(xin, w.double(), None if b is None else b.double(), stride=stride, padding=1)
    if res:
        ref = ref + r.double()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < (6e-3 if dtype == torch.bfloat16 else 1e-3)
    assert torch.equal(y, again)
    other = E.UceHandle.get("cuda:0").conv3x3_igemm(x, w, b, upsample=up, stride=stride, residual=r)      # the 8-wave form
    assert O.rel_fro(y.double().cpu(), other.double().cpu()) < (6e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("N,Cin,Cout,Hh,Ww,stride,up,res", [
    (2, 320, 320, 64, 64, 1, False, True),          # ONE prompt per call (CFG batch 2): the 64 x 64 level, 128 x 64 tiles, no split
    (2, 640, 640, 32, 32, 1, False, True),          # 32 x 32: split 2
    (2, 1280, 1280, 16, 16, 1, False, True),        # 16 x 16: split 4
    (2, 1280, 1280, 8, 8, 1, False, False),         # 8 x 8: one pixel tile, split 13
    (2, 2560, 1280, 8, 8, 1, False, True),          # up block on x | skip
    (2, 320, 320, 64, 64, 2, False, False),         # Downsample2D
    (2, 1280, 1280, 16, 16, 2, False, False),
    (2, 1280, 1280, 8, 8, 1, True, False),          # Upsample2D: taps gathered from the half-resolution tensor
    (1, 512, 512, 64, 64, 1, False, True),          # the VAE decoder's mid block on one image
    (1, 64, 72, 10, 6, 1, False, True),             # ragged pixels, Cout not a multiple of 64
    (32, 1280, 1280, 8, 8, 1, False, True),         # the 8 x 8 level at a generation batch
])
def test_conv3x3_few_tile_forms_match_torch_and_repeat_bit_for_bit(H, N, Cin, Cout, Hh, Ww, stride, up, res):
    """3x3 convolutions without 200 output tiles: 128 x 128 / 128 x 64 tiles and the 9 taps x channel chunks split over S
    workgroups per tile (csrc/uce_splitk.h) - against F.conv2d in fp64; ten runs give the same bits."""
    g = torch.Generator().manual_seed(Cin + Cout + Hh + stride)
    x = torch.randn(N, Cin, Hh, Ww, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, generator=g).bfloat16().cuda()
    Ho, Wo = (2 * Hh, 2 * Ww) if up else (Hh // stride, Ww // stride)
    r = torch.randn(N, Cout, Ho, Wo, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last) if res else None
    y = H.conv3x3_igemm(x, w, b, upsample=up, stride=stride, residual=r)
    xin = F.interpolate(x.double(), scale_factor=2.0, mode="nearest") if up else x.double()
    ref = F.conv2d(xin, w.double(), b.double(), stride=stride, padding=1)
    if res:
        ref = ref + r.double()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert O.rel_fro(y.double().cpu(), ref.cpu()) < 6e-3
    for _ in range(10):
        assert torch.equal(y, H.conv3x3_igemm(x, w, b, upsample=up, stride=stride, residual=r))


@pytest.mark.parametrize("N,C,Hh,Ww,C2,silu,addend", [
    (2, 320, 64, 64, 0, True, True),            # ONE prompt per call (CFG batch 2): every GroupNorm of the U-Net is one launch
    (2, 640, 32, 32, 0, True, False),
    (2, 1280, 16, 16, 0, True, True),
    (2, 1280, 8, 8, 0, False, False),
    (2, 1280, 8, 8, 1280, True, True),          # up block on x | skip (two sources, two channel octets per thread)
    (2, 640, 64, 64, 320, True, False),
    (1, 512, 64, 64, 0, True, False),           # the VAE decoder's mid block on one image
    (3, 96, 7, 5, 0, True, True),               # ragged pixels: a last chunk shorter than the others
    (16, 1280, 8, 8, 0, True, True),            # a generation batch on the 8 x 8 level
])
def test_groupnorm_one_launch_form_matches_the_two_kernel_form(N, C, Hh, Ww, C2, silu, addend):
    """Small activations take k_gn_fused (one launch; the chunk stays in registers across a grid-wide wait inside each sample):
    against fp64, against the stats + apply kernels (UCE_GN_FUSED=0; same arithmetic, another chunking of the f32 partial sums),
    bit-repeatable over twenty runs and replayed from a hipGraph (the counters re-arm themselves)."""
    import os
    from uce_amd import edit as E
    g = torch.Generator().manual_seed(N * C + Hh)
    mk = lambda c: (torch.randn(N, c, Hh, Ww, generator=g) * 1.5 + 0.3).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    x, x2 = mk(C), (mk(C2) if C2 else None)
    Ct = C + C2
    w = (torch.rand(Ct, generator=g) + 0.5).bfloat16().cuda()
    b = (torch.randn(Ct, generator=g) * 0.2).bfloat16().cuda()
    ad = (torch.randn(N, Ct, generator=g) * 0.5).bfloat16().cuda() if addend else None
    Hf = E.UceHandle("cuda:0")
    old = os.environ.get("UCE_GN_FUSED")
    os.environ["UCE_GN_FUSED"] = "0"
    try:
        H2 = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_GN_FUSED"]
        else:
            os.environ["UCE_GN_FUSED"] = old
    try:
        y = Hf.groupnorm_nhwc(x, w, b, 32, 1e-5, silu, ad, x2=x2)
        y2 = H2.groupnorm_nhwc(x, w, b, 32, 1e-5, silu, ad, x2=x2)
        xin = (x if x2 is None else torch.cat([x, x2], 1)).double()
        if ad is not None:
            xin = xin + ad.double()[:, :, None, None]
        ref = F.group_norm(xin, 32, w.double(), b.double(), 1e-5)
        ref = F.silu(ref) if silu else ref
        assert torch.isfinite(y.float()).all()
        assert O.rel_fro(y.double().cpu(), ref.cpu()) < 4e-3
        assert O.rel_fro(y.double().cpu(), y2.double().cpu()) < 1e-3          # (one bf16 ulp here and there: another partition of the sums)
        for _ in range(20):
            assert torch.equal(y, Hf.groupnorm_nhwc(x, w, b, 32, 1e-5, silu, ad, x2=x2))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                yg = Hf.groupnorm_nhwc(x, w, b, 32, 1e-5, silu, ad, x2=x2)
            for _ in range(3):
                yg.zero_()
                gr.replay()
                torch.cuda.synchronize()
                assert torch.equal(yg, y)
    finally:
        torch.cuda.synchronize()
        Hf.close()
        H2.close()
