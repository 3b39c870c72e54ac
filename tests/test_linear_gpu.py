"""GPU parity of the linear-layer kernel with fused epilogues (uce_linear_fwd, csrc/uce_gemm.hip) through the C ABI, against
fp64 evaluations of the same expressions: every tile form, bias / residual / GEGLU epilogues, strided operands, ragged
shapes; and the U-Net pieces that ride on it (packed q|k|v attention, the hoisted time projections, conv_in)."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import uce_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.bfloat16: 4e-3, torch.float16: 6e-4}      # one rounding of the output (2^-9 / 2^-12 relative per element)


@pytest.fixture(scope="module")
def H():
    from uce_amd import edit as E
    return E.UceHandle.get("cuda:0")


def _handle_with(env_name, value):
    from uce_amd import edit as E
    old = os.environ.get(env_name)
    os.environ[env_name] = value
    try:
        return E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ[env_name]
        else:
            os.environ[env_name] = old


def _rand(shape, g, dtype, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


@pytest.mark.parametrize("M,N,K,dtype,bias,res", [
    (4096, 320, 320, torch.bfloat16, True, False),        # to_out at 64 x 64 (B = 1)
    (1000, 640, 640, torch.bfloat16, False, True),        # ragged M, residual
    (8192, 1280, 1280, torch.bfloat16, True, True),       # 128-row tiles (few 256-row tiles)
    (333, 768, 768, torch.float16, True, True),           # 256-wide tiles (text-encoder widths), ragged M
    (77, 1280, 768, torch.bfloat16, False, False),        # to_k on a context
    (32, 1280, 320, torch.bfloat16, True, False),         # the time embedding
    (2048, 512, 512, torch.bfloat16, True, True),         # the VAE attention's projections
    (515, 128, 256, torch.bfloat16, True, False),         # 128-wide tiles
    (300, 36, 64, torch.bfloat16, True, False),           # a ragged last column tile (N not a multiple of any tile)
    (70000, 320, 64, torch.bfloat16, True, False),        # many row tiles, two k-tiles (conv_in's GEMM)
])
def test_linear_matches_fp64(H, M, N, K, dtype, bias, res):
    g = torch.Generator().manual_seed(M + N + K)
    x = _rand((M, K), g, dtype)
    w = _rand((N, K), g, dtype, K ** -0.5)
    w[0, 1] += 2.0                                        # asymmetric: a transposed operand cannot pass
    b = _rand((N,), g, dtype) if bias else None
    r = _rand((M, N), g, dtype) if res else None
    y = H.linear(x, w, b, r)
    want = x.double() @ w.double().T
    if bias:
        want = want + b.double()
    if res:
        want = want + r.double()
    assert y.shape == (M, N) and y.dtype == dtype
    assert O.rel_fro(y.double().cpu(), want.cpu()) < TOL[dtype]
    assert torch.equal(y, H.linear(x, w, b, r))           # bit-repeatable


@pytest.mark.parametrize("tile", ["256320", "256256", "128320", "128256", "256128", "2128320", "3128256", "3256128", "64256320", "64256256", "64128320",
                                  "9128064", "9128128"])
@pytest.mark.parametrize("M,N,K", [(700, 960, 320), (256, 640, 96), (1500, 640, 1280)])
def test_linear_every_tile_form_forced(tile, M, N, K):
    """UCE_GEMM_TILE pins one tile form for every call (read at uce_create): each form - 2... / 3... / 64... are the shallow rings
    that put two workgroups on a CU, 9... the few-tile forms with the split contraction - on shapes with ragged row and column
    tiles, with bias + residual, and with the GEGLU epilogue."""
    if tile[0] in "6789" and len(tile) == 7 and K % 64:
        pytest.skip("the few-tile forms move 128-byte k-tiles")
    Hv = _handle_with("UCE_GEMM_TILE", tile)
    g = torch.Generator().manual_seed(int(tile) + M)
    x, w = _rand((M, K), g, torch.bfloat16), _rand((N, K), g, torch.bfloat16, K ** -0.5)
    b, r = _rand((N,), g, torch.bfloat16), _rand((M, N), g, torch.bfloat16)
    from uce_amd.sd import unet as U
    wi, bi = U.geglu_interleave(w, b)
    try:
        y = Hv.linear(x, w, b, r)
        yg = Hv.linear(x, wi, bi, geglu=True)
        torch.cuda.synchronize()
    finally:
        Hv.close()
    p = x.double() @ w.double().T + b.double()
    assert O.rel_fro(y.double().cpu(), (p + r.double()).cpu()) < TOL[torch.bfloat16]
    assert O.rel_fro(yg.double().cpu(), (p[:, :N // 2] * F.gelu(p[:, N // 2:])).cpu()) < TOL[torch.bfloat16]


@pytest.mark.parametrize("M,C,dtype", [(4096, 320, torch.bfloat16), (1000, 640, torch.bfloat16), (300, 1280, torch.float16),
                                       (77, 64, torch.bfloat16)])
def test_linear_geglu_epilogue(H, M, C, dtype):
    """diffusers GEGLU: hidden, gate = proj(x).chunk(2); hidden * gelu(gate) - formed on the accumulators from the interleaved
    weight rows (sd.unet.geglu_interleave), against the fp64 expression on the ORIGINAL weight."""
    from uce_amd.sd import unet as U
    inner = 4 * C
    g = torch.Generator().manual_seed(M + C)
    x = _rand((M, C), g, dtype)
    w = _rand((2 * inner, C), g, dtype, C ** -0.5)
    b = _rand((2 * inner,), g, dtype)
    wi, bi = U.geglu_interleave(w, b)
    y = H.linear(x, wi, bi, geglu=True)
    p = x.double() @ w.double().T + b.double()
    want = p[:, :inner] * F.gelu(p[:, inner:])
    assert y.shape == (M, inner)
    assert O.rel_fro(y.double().cpu(), want.cpu()) < TOL[dtype]
    # and without a bias
    y0 = H.linear(x, wi, None, geglu=True)
    p0 = x.double() @ w.double().T
    assert O.rel_fro(y0.double().cpu(), (p0[:, :inner] * F.gelu(p0[:, inner:])).cpu()) < TOL[dtype]


def test_linear_strided_operands(H):
    """x, residual and out as column slices of wider tensors (row strides != widths): what the packed projections use."""
    g = torch.Generator().manual_seed(3)
    M, K, N = 900, 320, 640
    big_x = _rand((M, 3 * K), g, torch.bfloat16)
    big_r = _rand((M, 2 * N), g, torch.bfloat16)
    big_y = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device="cuda:0")
    w, b = _rand((N, K), g, torch.bfloat16, K ** -0.5), _rand((N,), g, torch.bfloat16)
    x, r, out = big_x[:, K:2 * K], big_r[:, N:], big_y[:, N:2 * N]
    H.linear(x, w, b, r, out=out)
    want = x.double() @ w.double().T + b.double() + r.double()
    assert O.rel_fro(out.double().cpu(), want.cpu()) < TOL[torch.bfloat16]
    assert float(big_y[:, :N].abs().max()) == 0.0 and float(big_y[:, 2 * N:].abs().max()) == 0.0   # neighbours untouched


@pytest.mark.parametrize("tile", ["0", "64256320", "64256256", "2128320", "256320", "9128064"])
@pytest.mark.parametrize("M", [1000, 257, 31])
def test_whole_row_epilogue_writes_its_rows_and_columns_and_nothing_else(tile, M):
    """The whole-row epilogue stores through a buffer descriptor of the wave's rows clipped to M and masks columns beyond N by the
    offset (uce_epilogue.h): the rows after M and the columns either side of the output slice keep their sentinel -
    with bias + residual (the residual rows of the next chunk are requested ahead: rows beyond M must not fault either) and with
    the GEGLU epilogue, on the wide, shallow-ring and few-tile forms."""
    from uce_amd.sd import unet as U
    Hv = _handle_with("UCE_GEMM_TILE", tile)
    g = torch.Generator().manual_seed(M + int(tile))
    K, N = 320, 328                                                      # N: a ragged last column tile of every form (N % 8 == 0)
    x, w, b = _rand((M, K), g, torch.bfloat16), _rand((N, K), g, torch.bfloat16, K ** -0.5), _rand((N,), g, torch.bfloat16)
    r = _rand((M, N), g, torch.bfloat16)                                 # exactly M rows: nothing readable behind them is promised
    big = torch.full((M + 40, N + 64), 7.0, dtype=torch.bfloat16, device="cuda:0")
    Ng = 640
    wg, bg = _rand((Ng, K), g, torch.bfloat16, K ** -0.5), _rand((Ng,), g, torch.bfloat16)
    wi, bi = U.geglu_interleave(wg, bg)
    bigg = torch.full((M + 40, Ng // 2 + 64), 7.0, dtype=torch.bfloat16, device="cuda:0")
    try:
        Hv.linear(x, w, b, r, out=big[:M, 32:32 + N])
        Hv.linear(x, wi, bi, geglu=True, out=bigg[:M, 32:32 + Ng // 2])
        torch.cuda.synchronize()
    finally:
        Hv.close()
    want = x.double() @ w.double().T + b.double() + r.double()
    assert O.rel_fro(big[:M, 32:32 + N].double().cpu(), want.cpu()) < TOL[torch.bfloat16]
    pg = x.double() @ wg.double().T + bg.double()
    assert O.rel_fro(bigg[:M, 32:32 + Ng // 2].double().cpu(), (pg[:, :Ng // 2] * F.gelu(pg[:, Ng // 2:])).cpu()) < TOL[torch.bfloat16]
    for t, n in ((big, N), (bigg, Ng // 2)):
        assert bool((t[M:] == 7.0).all()) and bool((t[:, :32] == 7.0).all()) and bool((t[:, 32 + n:] == 7.0).all())


def test_linear_rejects_bad_arguments(H):
    from uce_amd import lib as L
    x = torch.zeros(8, 40, dtype=torch.bfloat16, device="cuda:0")          # K = 40 is not a multiple of 32
    w = torch.zeros(64, 40, dtype=torch.bfloat16, device="cuda:0")
    with pytest.raises(L.UceError):
        H.linear(x, w)
    x = torch.zeros(8, 64, dtype=torch.bfloat16, device="cuda:0")
    w = torch.zeros(30, 64, dtype=torch.bfloat16, device="cuda:0")         # N = 30 is not a multiple of 4
    with pytest.raises(L.UceError):
        H.linear(x, w)
    w = torch.zeros(48, 64, dtype=torch.bfloat16, device="cuda:0")         # GEGLU needs whole 32-row groups
    with pytest.raises(L.UceError):
        H.linear(x, w, geglu=True)


# ------------------------------------------------------------------------------------ what rides on it in the U-Net

@pytest.mark.parametrize("B,heads,L,dh,dtype", [(2, 8, 4096, 40, torch.bfloat16), (2, 8, 1024, 80, torch.bfloat16),
                                                (3, 8, 256, 160, torch.bfloat16), (2, 8, 64, 160, torch.bfloat16),
                                                (1, 5, 100, 64, torch.float16), (2, 2, 77, 16, torch.bfloat16)])
def test_packed_self_attention(H, B, heads, L, dh, dtype):
    """uce_sattn_packed_fwd on qkv [B, L, 3C] == uce_sattn_fwd on the three slices made contiguous, and both == fp64."""
    C = heads * dh
    g = torch.Generator().manual_seed(L + dh)
    qkv = _rand((B, L, 3 * C), g, dtype)
    q, k, v = (qkv[..., i * C:(i + 1) * C].contiguous() for i in range(3))
    o = H.sattn_packed(qkv, heads)
    assert torch.equal(o, H.sattn(q, k, v, heads))
    sp = lambda t: t.double().view(B, L, heads, dh).transpose(1, 2)
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5, dim=-1) @ sp(v)).transpose(1, 2).reshape(B, L, C)
    assert O.rel_fro(o.double().cpu(), ref.cpu()) < (8e-3 if dtype == torch.bfloat16 else 1.5e-3)


@pytest.mark.parametrize("vti", ["1", "2", "3"])
@pytest.mark.parametrize("B,heads,Lq,Lk,dh,dtype", [(2, 8, 1024, 1024, 40, torch.bfloat16), (1, 8, 300, 130, 40, torch.bfloat16),
                                                    (2, 8, 512, 200, 80, torch.bfloat16), (1, 4, 200, 130, 128, torch.float16),
                                                    (2, 8, 256, 256, 160, torch.bfloat16), (1, 7, 70, 191, 56, torch.bfloat16),
                                                    (3, 2, 33, 1, 40, torch.bfloat16)])
def test_self_attention_with_and_without_the_vt_prepass(vti, B, heads, Lq, Lk, dh, dtype):
    """UCE_SATTN_VTI = 1: V^T transposed on the way into LDS at every length; 2: always the k_vt pre-pass (the by-rule default
    switches at 1024 keys): both forms on ragged shapes, a head dim whose row of ones falls inside the padded tile (dh = 56),
    and a single key."""
    Hv = _handle_with("UCE_SATTN_VTI", vti)
    C = heads * dh
    g = torch.Generator().manual_seed(Lq + Lk + dh)
    q, k, v = _rand((B, Lq, C), g, dtype), _rand((B, Lk, C), g, dtype), _rand((B, Lk, C), g, dtype)
    try:
        o = Hv.sattn(q, k, v, heads)
        again = Hv.sattn(q, k, v, heads)
        torch.cuda.synchronize()
    finally:
        Hv.close()
    sp = lambda t, n: t.double().view(B, n, heads, dh).transpose(1, 2)
    ref = (torch.softmax(sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) * dh ** -0.5, dim=-1) @ sp(v, Lk)).transpose(1, 2).reshape(B, Lq, C)
    assert torch.isfinite(o.float()).all()
    assert O.rel_fro(o.double().cpu(), ref.cpu()) < (8e-3 if dtype == torch.bfloat16 else 1.5e-3)
    assert torch.equal(o, again)
    if Lk == 1:
        assert torch.equal(o, v.expand(B, Lq, C))


def test_transformer_block_matches_its_torch_twin():
    """BasicTransformerBlock on the GPU (packed q|k|v + uce_sattn_packed_fwd, residual joins in the projections' epilogues,
    GEGLU on the accumulators, uce_xattn_fwd) against the same weights through torch ops (tests/torch_twin.py)."""
    from tests.torch_twin import torch_ops
    from uce_amd.sd import unet as U
    # (batch, tokens): the first two give every CU an output tile (the own kernels: packed q|k|v, fused epilogues), the last
    # is a one-prompt batch that the dispatch rule leaves to the GEMM library
    for C, heads, L, B in ((320, 8, 4096, 8), (640, 8, 1024, 16), (1280, 8, 64, 2)):
        torch.manual_seed(C)
        blk = U.BasicTransformerBlock(C, heads, C // heads, 768).to("cuda", torch.bfloat16)
        g = torch.Generator().manual_seed(1)
        x = _rand((B, L, C), g, torch.bfloat16)
        ctx = _rand((B, 77, 768), g, torch.bfloat16)
        a = blk(x, ctx).float()
        with torch_ops():
            b = blk(x, ctx).float()
        ref = blk.float()(x.float(), ctx.float())
        blk.to(torch.bfloat16)
        ea, eb = O.rel_fro(a.cpu(), ref.cpu()), O.rel_fro(b.cpu(), ref.cpu())
        assert ea < 1.5 * eb + 2e-3, (C, ea, eb)          # no further from fp32 than torch's own bf16 ops


def test_unet_forward_matches_its_torch_twin_and_hoists_time_projections():
    """The tiny U-Net on the GPU against the same weights through torch ops; the 22 time projections are ONE stacked linear
    (whichever GEMM takes it at this size), conv_in goes through the 4-channel patch matrix + uce_linear_fwd."""
    from tests.torch_twin import torch_ops
    from uce_amd import edit as E
    from uce_amd.sd import pipeline as sdp
    from uce_amd.sd import unet as U
    pipe = sdp.load_pipeline("tiny-sd-test", torch.bfloat16, "cuda:0", synthetic=True, vae=False, seed=3)
    g = torch.Generator().manual_seed(0)
    x = _rand((2, 4, 8, 8), g, torch.bfloat16)
    ctx = _rand((2, 77, 64), g, torch.bfloat16)
    t = torch.tensor([500], device="cuda")
    seen, own = [], []
    orig_w, orig_own = U.linear_w, E.UceHandle.linear

    def counted_w(x_, w, *a, **k):
        seen.append((tuple(x_.shape), tuple(w.shape)))
        return orig_w(x_, w, *a, **k)

    def counted_own(self, x_, w, *a, **k):
        own.append((tuple(x_.shape), tuple(w.shape)))
        return orig_own(self, x_, w, *a, **k)

    U.linear_w, E.UceHandle.linear = counted_w, counted_own
    try:
        a = pipe.unet(x, t, ctx).float()
    finally:
        U.linear_w, E.UceHandle.linear = orig_w, orig_own
    n_res = sum(1 for m in pipe.unet.modules() if m.__class__.__name__ == "ResnetBlock2D")
    temb_dim = pipe.unet.cfg.block_out_channels[0] * 4
    hoisted = [s for s in seen if s[0] == (2, temb_dim) and s[1][1] == temb_dim and s[1][0] > temb_dim]
    assert n_res == 22 and len(hoisted) == 1                     # one stacked projection, not 22
    # the slices are per call: a block used on its own afterwards computes its own projection (no stale hidden state)
    assert all(m.temb_addend is None for m in pipe.unet.modules() if m.__class__.__name__ == "ResnetBlock2D")
    assert ((2 * 8 * 8, 64), (32, 64)) in own                    # conv_in: [pixels, 64] patch matrix x [Cout, 64]
    with torch_ops():
        b = pipe.unet(x, t, ctx).float()
    assert O.rel_fro(a.cpu(), b.cpu()) < 3e-2


def test_every_16_bit_gpu_linear_layer_takes_the_kernel_and_nothing_falls_to_the_library():
    """sd/unet.py has no tile-count rule any more: a layer of ANY row count goes to uce_linear_fwd (few output tiles: its split-
    contraction forms), a contraction / output off the kernel's granules runs zero-padded, mixed dtypes raise - F.linear is not
    reachable for a 16-bit GPU tensor."""
    from uce_amd import edit as E
    from uce_amd.sd import unet as U
    lin = torch.nn.Linear(320, 640).to("cuda", torch.bfloat16)
    g = torch.Generator().manual_seed(4)
    own = []
    orig, orig_f = E.UceHandle.linear, F.linear

    def counted(self, x_, w, *a, **k):
        own.append(tuple(x_.shape))
        return orig(self, x_, w, *a, **k)

    def no_library(*a, **k):
        raise AssertionError("F.linear reached from the product path")

    E.UceHandle.linear = counted
    F.linear = no_library
    try:
        small, big = _rand((2, 4096, 320), g, torch.bfloat16), _rand((8, 4096, 320), g, torch.bfloat16)
        tiny = _rand((2, 320), g, torch.bfloat16)
        r = _rand((8, 4096, 640), g, torch.bfloat16)
        ys, yb, yt = U.linear(lin, small), U.linear(lin, big, residual=r), U.linear(lin, tiny)
        # off the granules: K = 4 (the VAE's post_quant_conv as a 1x1 convolution), N = 6
        odd = torch.nn.Linear(4, 6).to("cuda", torch.bfloat16)
        xo = _rand((100, 4), g, torch.bfloat16)
        yo = U.linear(odd, xo)
        with pytest.raises(RuntimeError, match="no HIP kernel"):
            U.linear_w(small, lin.weight.half(), None)
    finally:
        E.UceHandle.linear, F.linear = orig, orig_f
    assert own[:3] == [(2, 4096, 320), (8, 4096, 320), (2, 320)] and len(own) == 4
    ref = lambda m, x_: F.linear(x_.double(), m.weight.double(), m.bias.double())
    assert O.rel_fro(ys.double().cpu(), ref(lin, small).cpu()) < 4e-3
    assert O.rel_fro(yb.double().cpu(), (ref(lin, big) + r.double()).cpu()) < 4e-3
    assert O.rel_fro(yt.double().cpu(), ref(lin, tiny).cpu()) < 4e-3
    assert yo.shape == (100, 6) and O.rel_fro(yo.double().cpu(), ref(odd, xo).cpu()) < 4e-3


def test_vae_attention_on_the_linear_kernel_matches_its_torch_twin(H):
    """The VAE mid-block attention (one head, 512 dims): q k^T (f32 scores) -> uce_softmax_rows -> P v on uce_linear_fwd, against
    the same module through torch's SDPA; and the row softmax alone against fp64."""
    from tests.torch_twin import torch_ops
    from uce_amd.sd import pipeline as sdp
    torch.manual_seed(5)
    att = sdp._VaeAttention(512).to("cuda", torch.bfloat16)
    g = torch.Generator().manual_seed(2)
    x = _rand((2, 512, 32, 32), g, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    a = att(x).float()
    with torch_ops():
        b = att(x).float()
    ref = att.float()(x.float())
    ea, eb = O.rel_fro(a.cpu(), ref.cpu()), O.rel_fro(b.cpu(), ref.cpu())
    assert ea < 1.5 * eb + 2e-3, (ea, eb)
    s = _rand((300, 4096), g, torch.float32, 3.0)
    p = H.softmax_rows(s, 0.25, torch.bfloat16)
    want = torch.softmax(s.double() * 0.25, dim=-1)
    assert O.rel_fro(p.double().cpu(), want.cpu()) < 4e-3
    y = H.linear_f32(x.permute(0, 2, 3, 1).reshape(-1, 512)[:700], att.to_q.weight.to(torch.bfloat16))
    assert y.dtype == torch.float32
    assert O.rel_fro(y.double().cpu(), (x.permute(0, 2, 3, 1).reshape(-1, 512)[:700].double() @ att.to_q.weight.double().T).cpu()) < 1e-5


@pytest.mark.parametrize("tile", ["0", "256320", "128320", "256256", "64256320", "64128320", "2128320", "3128256"])
@pytest.mark.parametrize("M,N,K1,K2,bias,res", [
    (4096, 320, 320, 320, True, False),        # up_blocks.3: x | skip at 64 x 64
    (1100, 640, 1280, 640, True, True),        # ragged rows, residual
    (700, 1280, 1280, 1280, False, False),
    (300, 320, 64, 32, True, False),           # one k-tile from each source (64-wide tiles cannot split at 64 | 32: EINVAL -> skipped)
])
def test_linear_over_a_two_source_contraction(tile, M, N, K1, K2, bias, res):
    """uce_linear_cat_fwd(x, x2) == uce_linear_fwd(torch.cat([x, x2], dim=-1)) bit for bit, in every tile form, the two
    sources with different row strides."""
    if int(tile) >= 64000000 and ((K1 + K2) % 64 or K1 % 64):
        pytest.skip("128-byte k-tiles need K and K1 to be multiples of 64")
    Hv = _handle_with("UCE_GEMM_TILE", tile)
    try:
        g = torch.Generator().manual_seed(M + K1 + K2)
        dtype = torch.bfloat16
        x, x2 = _rand((M, K1), g, dtype), _rand((M, K2), g, dtype)
        w = _rand((N, K1 + K2), g, dtype, (K1 + K2) ** -0.5)
        b = _rand((N,), g, dtype) if bias else None
        r = _rand((M, N), g, dtype) if res else None
        got = Hv.linear(x, w, b, r, x2=x2)
        want = Hv.linear(torch.cat([x, x2], dim=-1), w, b, r)
        assert torch.equal(got, want)
        ref = torch.cat([x, x2], dim=-1).double() @ w.double().T
        if bias:
            ref = ref + b.double()
        if res:
            ref = ref + r.double()
        assert O.rel_fro(got.double().cpu(), ref.cpu()) < TOL[dtype]
        # column slices of wider tensors as the two sources (row strides larger than the widths)
        wide1, wide2 = _rand((M, K1 + 64), g, dtype), _rand((M, K2 + 32), g, dtype)
        got = Hv.linear(wide1[:, :K1], w, b, r, x2=wide2[:, 32:])
        assert torch.equal(got, Hv.linear(torch.cat([wide1[:, :K1], wide2[:, 32:]], dim=-1), w, b, r))
    finally:
        torch.cuda.synchronize()
        Hv.close()


def test_up_block_reads_the_skip_connection_in_place():
    """sd.unet.UpBlock with UCE_CAT_FREE: two-source GroupNorm + two-source shortcut GEMM against the torch.cat path of the
    same block - the same arithmetic in the same order, so bit for bit - and no torch.cat on the way."""
    from uce_amd.sd import unet as U
    torch.manual_seed(5)
    blk = U.ResnetBlock2D(640 + 320, 320, 1280, 32).to("cuda:0", torch.bfloat16).to(memory_format=torch.channels_last)
    cl = lambda t: t.to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    x, skip = cl(torch.randn(8, 640, 64, 64)), cl(torch.randn(8, 320, 64, 64))
    temb = torch.randn(8, 1280).to(torch.bfloat16).cuda()
    old = U.CAT_FREE
    cats = []
    real_cat = torch.cat
    try:
        U.CAT_FREE = False
        want = blk(x, temb, skip=skip)
        U.CAT_FREE = True
        assert blk.cat_free_ok(x, skip)
        torch.cat = lambda *a, **k: (cats.append(1), real_cat(*a, **k))[1]
        got = blk(x, temb, skip=skip)
    finally:
        torch.cat = real_cat
        U.CAT_FREE = old
    assert not cats
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------ the few-tile regime (split contraction)

@pytest.mark.parametrize("M,N,K,bias,res", [
    (8192, 320, 320, True, True),          # to_out / proj_out at 64 x 64, ONE prompt per call (CFG batch 2): 128 x 64 tiles, no split
    (8192, 960, 320, False, False),        # the packed q | k | v projection: 128 x 128 tiles
    (8192, 320, 1280, True, True),         # ff_out at 64 x 64
    (2048, 640, 2560, True, True),         # ff_out at 32 x 32: split 2
    (512, 1280, 5120, True, True),         # ff_out at 16 x 16: split 4
    (128, 1280, 1280, True, True),         # the 8 x 8 mid block: one row tile, split 10
    (2, 1280, 320, True, False),           # the time embedding of one prompt
    (154, 640, 768, False, False),         # to_k on the context of one prompt
    (100, 72, 448, True, True),            # ragged everything: N not a multiple of 64, 7 k-tiles over 3 slabs
])
def test_linear_few_tile_forms_match_fp64_and_repeat_bit_for_bit(H, M, N, K, bias, res):
    """Layers that cannot give every CU a 128 x 320 tile take 128 x 128 / 128 x 64 tiles and, below 200 of those, a split
    contraction whose S slabs are summed in slab order by the last arriver (csrc/uce_splitk.h): against fp64, and twenty
    runs give the same bits (the arrival order changes from run to run, the sum must not)."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = _rand((M, K), g, torch.bfloat16)
    w = _rand((N, K), g, torch.bfloat16, K ** -0.5)
    w[0, 1] += 2.0
    b = _rand((N,), g, torch.bfloat16) if bias else None
    r = _rand((M, N), g, torch.bfloat16) if res else None
    y = H.linear(x, w, b, r)
    want = x.double() @ w.double().T
    if bias:
        want = want + b.double()
    if res:
        want = want + r.double()
    assert O.rel_fro(y.double().cpu(), want.cpu()) < TOL[torch.bfloat16]
    for _ in range(20):
        assert torch.equal(y, H.linear(x, w, b, r))


def test_linear_few_tile_geglu_two_sources_and_graph_replay(H):
    """The split forms with the GEGLU epilogue and with the two-source contraction; and a captured launch replays (the tickets
    re-arm themselves, the slabs live in the handle)."""
    from uce_amd.sd import unet as U
    g = torch.Generator().manual_seed(11)
    M, C = 512, 1280
    x, w, b = _rand((M, C), g, torch.bfloat16), _rand((8 * C, C), g, torch.bfloat16, C ** -0.5), _rand((8 * C,), g, torch.bfloat16)
    wi, bi = U.geglu_interleave(w, b)
    y = H.linear(x, wi, bi, geglu=True)
    p = x.double() @ w.double().T + b.double()
    assert O.rel_fro(y.double().cpu(), (p[:, :4 * C] * F.gelu(p[:, 4 * C:])).cpu()) < TOL[torch.bfloat16]
    # GEGLU at the 8 x 8 level of one prompt: split contraction under the gated epilogue
    x8 = x[:128].contiguous()
    y8 = H.linear(x8, wi, bi, geglu=True)
    assert torch.equal(y8, y[:128]) or O.rel_fro(y8.double().cpu(), y[:128].double().cpu()) < 4e-3
    # two sources (the 1 x 1 shortcut of an up block over x | skip)
    xa, xb = _rand((512, 1280), g, torch.bfloat16), _rand((512, 640), g, torch.bfloat16)
    ws, bs = _rand((1280, 1920), g, torch.bfloat16, 1920 ** -0.5), _rand((1280,), g, torch.bfloat16)
    y2 = H.linear(xa, ws, bs, x2=xb)
    want2 = torch.cat([xa, xb], 1).double() @ ws.double().T + bs.double()
    assert O.rel_fro(y2.double().cpu(), want2.cpu()) < TOL[torch.bfloat16]
    assert torch.equal(y2, H.linear(torch.cat([xa, xb], 1), ws, bs))          # same k-tile order, same slabs: same bits
    # hipGraph replay
    out = torch.empty_like(y2)
    xc = torch.cat([xa, xb], 1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        H.linear(xc, ws, bs, out=out)                                         # (the scratch is allocated outside the capture)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            H.linear(xc, ws, bs, out=out)
        for _ in range(3):
            out.zero_()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, y2)


@pytest.mark.parametrize("M,K,N,cols,dtype", [
    (8192, 320, 960, 320, torch.bfloat16),      # packed q | k | v of the 64 x 64 level, one prompt's CFG pair
    (65536, 320, 960, 320, torch.bfloat16),     # the many-tile forms
    (512, 1280, 3840, 1280, torch.bfloat16),    # few tiles: the split-contraction form (the scale follows the slab reduction)
    (300, 64, 96, 32, torch.float16),           # ragged rows, one scaled MFMA tile
    (1000, 320, 960, 0, torch.bfloat16),        # no scaled columns: the plain projection
])
def test_linear_colscale(H, M, K, N, cols, dtype):
    """uce_linear_colscale_fwd: columns [0, cols) = round(scale * f32 product) - ONE rounding, checked against the f32 product scaled
    and rounded by torch; the other columns are bit-equal to uce_linear_fwd's."""
    g = torch.Generator().manual_seed(M + N)
    x, w = _rand((M, K), g, dtype), _rand((N, K), g, dtype, 0.05)
    scale = 40 ** -0.5 * 1.4426950408889634
    y = H.linear_colscale(x, w, cols, scale)
    plain = H.linear(x, w)
    assert torch.equal(y[:, cols:], plain[:, cols:])
    if cols:
        f = x.double() @ w.double().T
        ref = (f[:, :cols] * scale).to(dtype)
        # f32 accumulation order vs fp64: at most one unit in the last place of the 16-bit result on a few elements
        d = (y[:, :cols].double() - ref.double()).abs()
        ulp = ref.double().abs() * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
        # (plus the f32 summation error itself, which is relative to the sum of |terms|, not to a result that cancels)
        ulp = ulp + 2e-6 * scale * (x.double().abs() @ w.double().abs().T)[:, :cols]
        assert (d <= ulp).all()
        assert (d > 0).double().mean() < 0.02
        # and it is NOT the double rounding round(scale * round(product))
        twice = (plain[:, :cols].float() * scale).to(dtype)
        assert (y[:, :cols] != twice).double().mean() > 0.05
