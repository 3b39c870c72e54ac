"""GPU parity of the cross-attention kernel (through the C ABI) against the oracle and the
golden outputs of torch's CPU scaled_dot_product_attention."""
import pytest
import torch

from oracle import uce_oracle as O
from tests.golden_io import Case, SDPA_CASES

pytestmark = pytest.mark.gpu

# bf16 output rounding (2^-9 relative per element) dominates; P is rounded to bf16 before P.V
TOL_BF16 = 8e-3
TOL_F16 = 1.5e-3


@pytest.fixture(scope="module")
def H():
    from uce_amd import edit as E
    return E.UceHandle.get("cuda:0")


@pytest.mark.parametrize("name", SDPA_CASES)
def test_xattn_golden(H, name):
    c = Case(name)
    m = c.meta
    q, k, v = (c.t(x).view(torch.bfloat16).cuda() for x in ("q", "k", "v"))
    o = H.xattn(q, k, v, m["H"]).cpu()
    ref64 = O.xattn_ref(q.cpu(), k.cpu(), v.cpu(), m["H"])
    assert O.rel_fro(o.double(), ref64) < TOL_BF16
    # and as close to the fp32 SDPA result as torch's own bf16 SDPA is (x1.5)
    sdpa_bf16 = c.t("o_bf16").view(torch.bfloat16).double()
    err_ref = O.rel_fro(sdpa_bf16, c.t("o_f32"))
    assert O.rel_fro(o.double(), c.t("o_f32")) < max(1.5 * err_ref, 4e-3)


@pytest.mark.parametrize("name", SDPA_CASES)
def test_xattn_golden_at_the_generation_batch(H, name):
    """The torch-SDPA fixtures at the shape the generation loop launches (B = 32 = 16 prompts x CFG, full Lq), where
    uce_xattn_fwd takes the 640-byte column-group kernel k_xattn_g by default (dh = 40 / 80) - the fixture's 2 x 64 query
    rows are repeated over the batch and along Lq (query rows and samples are independent), and EVERY replica has to
    reproduce torch's output for the original rows."""
    c = Case(name)
    m = c.meta
    q, k, v = (c.t(x).view(torch.bfloat16) for x in ("q", "k", "v"))
    B0, lq, C = q.shape
    Lq = {"sdpa_Lq4096_dh40": 4096, "sdpa_Lq1024_dh80": 1024, "sdpa_Lq256_dh160": 256, "sdpa_Lq64_dh160": 64}[name]
    rb, rl = 32 // B0, Lq // lq
    Q = q.repeat(rb, rl, 1).cuda()
    K, V = k.repeat(rb, 1, 1).cuda(), v.repeat(rb, 1, 1).cuda()
    o = H.xattn(Q, K, V, m["H"])
    blocks = o.view(rb, B0, rl, lq, C).double()                 # [batch replica, sample, Lq replica, row, channel]
    ref32 = c.t("o_f32").double().cuda()[None, :, None]
    ref64 = O.xattn_ref(q, k, v, m["H"]).cuda()[None, :, None]
    sdpa_bf16 = c.t("o_bf16").view(torch.bfloat16).double()
    err_ref = O.rel_fro(sdpa_bf16, c.t("o_f32"))
    # per replica: norm over (sample, row, channel)
    def rel(x, y):
        num = (x - y).pow(2).sum(dim=(1, 3, 4)).sqrt()
        return float((num / y.pow(2).sum(dim=(1, 3, 4)).sqrt()).max())
    assert rel(blocks, ref64) < TOL_BF16
    assert rel(blocks, ref32) < max(1.5 * err_ref, 4e-3)
    # replicas of the same rows are the same bits (no dependence on the tile position / workgroup)
    assert bool((o.view(rb, B0, rl, lq, C) == o.view(rb, B0, rl, lq, C)[:1, :, :1]).all())


@pytest.mark.parametrize("B,H_,Lq,Lk,dh,dtype", [
    (2, 8, 4096, 77, 40, torch.bfloat16),     # SD-1.4 full shapes
    (2, 8, 1024, 77, 80, torch.bfloat16),
    (2, 8, 256, 77, 160, torch.bfloat16),
    (2, 8, 64, 77, 160, torch.bfloat16),
    (16, 8, 4096, 77, 40, torch.bfloat16),    # the batch the generation loop issues: 4 query tiles per workgroup
    (40, 8, 1000, 77, 40, torch.bfloat16),    # 2 tiles per workgroup, ragged last tile
    (1, 5, 100, 77, 64, torch.bfloat16),      # ragged Lq (not a multiple of 128 or 32)
    (3, 2, 33, 1, 40, torch.bfloat16),        # a single key: softmax == 1, O == V
    (1, 4, 200, 128, 128, torch.bfloat16),    # 4 key tiles
    (2, 10, 130, 97, 64, torch.float16),      # f16 path, SDXL-like head dim
    (1, 20, 64, 77, 64, torch.float16),
])
def test_xattn_shapes(H, B, H_, Lq, Lk, dh, dtype):
    g = torch.Generator().manual_seed(Lq * 7 + dh)
    C = H_ * dh
    q = torch.randn(B, Lq, C, generator=g).to(dtype)
    k = torch.randn(B, Lk, C, generator=g).to(dtype)
    v = torch.randn(B, Lk, C, generator=g).to(dtype)
    o = H.xattn(q.cuda(), k.cuda(), v.cuda(), H_).cpu()
    ref = O.xattn_ref(q, k, v, H_)
    assert O.rel_fro(o.double(), ref) < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16)
    if Lk == 1:
        assert torch.equal(o, v.expand(B, Lq, C).contiguous())


@pytest.mark.parametrize("variant", ["2", "3"])
@pytest.mark.parametrize("B,H_,Lq,Lk,dh,dtype", [
    (2, 8, 4096, 77, 40, torch.bfloat16),     # one tile per workgroup
    (8, 8, 1000, 77, 40, torch.bfloat16),     # XCD remap (B % 8 == 0), ragged last tile
    (3, 16, 333, 77, 40, torch.bfloat16),     # two column groups, ragged, no remap
    (5, 8, 130, 80, 80, torch.bfloat16),      # 80 keys: the last key row of the LDS image
    (8, 8, 1024, 77, 80, torch.bfloat16),
    (2, 8, 256, 77, 160, torch.bfloat16),
    (16, 4, 70, 33, 160, torch.float16),      # f16, two key tiles only
    (3, 8, 64, 1, 40, torch.bfloat16),        # a single key: O == V
    (2, 10, 1000, 77, 64, torch.bfloat16),    # dh = 64 (SD-2.x / SDXL): five heads per column group, ten waves; two groups
    (8, 5, 333, 33, 64, torch.bfloat16),      # one group, XCD remap, two key tiles
    (3, 20, 130, 80, 64, torch.float16),      # four groups, 80 keys, f16
    (2, 10, 64, 1, 64, torch.bfloat16),       # a single key
])
def test_xattn_group_kernel_forced(H, variant, B, H_, Lq, Lk, dh, dtype):
    """The 640-byte column-group kernel (default only at generation-batch sizes) forced at every size: its 16-wave
    (variant 2) and 8-wave (variant 3) dh = 40 forms, dh = 64 / 80 / 160, ragged tiles, the XCD remap, both dtypes."""
    import os
    g = torch.Generator().manual_seed(Lq * 11 + dh + B)
    C = H_ * dh
    q = torch.randn(B, Lq, C, generator=g).to(dtype)
    k = torch.randn(B, Lk, C, generator=g).to(dtype)
    v = torch.randn(B, Lk, C, generator=g).to(dtype)
    # the A/B switches are read when a handle is created: a handle of its own for the forced variant
    from uce_amd import edit as E
    old = os.environ.get("UCE_XATTN_VARIANT")
    os.environ["UCE_XATTN_VARIANT"] = variant
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_XATTN_VARIANT"]
        else:
            os.environ["UCE_XATTN_VARIANT"] = old
    try:
        o = Hv.xattn(q.cuda(), k.cuda(), v.cuda(), H_).cpu()
    finally:
        Hv.close()
    ref = O.xattn_ref(q, k, v, H_)
    assert O.rel_fro(o.double(), ref) < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16)
    if Lk == 1:
        assert torch.equal(o, v.expand(B, Lq, C).contiguous())


def test_xattn_peaked_and_scaled(H):
    """Large logits (one key dominating) and a custom scale: exercises max-subtraction."""
    g = torch.Generator().manual_seed(3)
    B, H_, Lq, Lk, dh = 1, 8, 128, 77, 40
    C = H_ * dh
    q = (torch.randn(B, Lq, C, generator=g) * 6).to(torch.bfloat16)
    k = (torch.randn(B, Lk, C, generator=g) * 6).to(torch.bfloat16)
    v = torch.randn(B, Lk, C, generator=g).to(torch.bfloat16)
    o = H.xattn(q.cuda(), k.cuda(), v.cuda(), H_, scale=0.5).cpu()
    ref = O.xattn_ref(q, k, v, H_, scale=0.5)
    assert torch.isfinite(o.float()).all()
    assert O.rel_fro(o.double(), ref) < TOL_BF16


def test_xattn_rejects_bad_arguments(H):
    from uce_amd import lib as L
    q = torch.zeros(1, 8, 8 * 36, dtype=torch.bfloat16, device="cuda:0")   # dh = 36 is not a multiple of 8
    with pytest.raises(L.UceError):
        H.xattn(q, q, q, 8)
    k = torch.zeros(1, 129, 64, dtype=torch.bfloat16, device="cuda:0")     # too many keys
    with pytest.raises(L.UceError):
        H.xattn(torch.zeros(1, 8, 64, dtype=torch.bfloat16, device="cuda:0"), k, k, 1)
