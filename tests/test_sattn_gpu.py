"""GPU parity of the self-attention (flash-style, any Lk) kernel through the C ABI against an fp64 evaluation
of softmax(Q K^T / sqrt(dh)) V (the oracle's xattn_ref: the arithmetic F.scaled_dot_product_attention performs
inside diffusers' AttnProcessor2_0 for attn1)."""
import pytest
import torch

from oracle import uce_oracle as O

pytestmark = pytest.mark.gpu

# bf16 output rounding (2^-9 relative per element) dominates; P is rounded to bf16 before P.V
TOL_BF16 = 8e-3
TOL_F16 = 1.5e-3


@pytest.fixture(scope="module")
def H():
    from uce_amd import edit as E
    return E.UceHandle.get("cuda:0")


def _ref_gpu(q, k, v, heads, scale=None):
    """fp64 attention on the GPU (the CPU oracle is too slow at Lq = Lk = 4096); checked against the oracle below."""
    B, Lq, C = q.shape
    dh = C // heads
    sp = lambda t: t.double().view(B, t.shape[1], heads, dh).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) * (dh ** -0.5 if scale is None else scale)
    return (torch.softmax(s, dim=-1) @ sp(v)).transpose(1, 2).reshape(B, Lq, C)


@pytest.mark.parametrize("B,H_,Lq,Lk,dh,dtype", [
    (2, 8, 4096, 4096, 40, torch.bfloat16),     # SD-1.4 attn1 shapes (B = 2: one prompt's CFG pair)
    (2, 8, 1024, 1024, 80, torch.bfloat16),
    (2, 8, 256, 256, 160, torch.bfloat16),
    (2, 8, 64, 64, 160, torch.bfloat16),
    (16, 8, 1024, 1024, 40, torch.bfloat16),    # batched prompts
    (1, 5, 100, 77, 64, torch.bfloat16),        # ragged Lq and Lk (neither a multiple of the tile sizes)
    (3, 2, 33, 1, 40, torch.bfloat16),          # a single key: softmax == 1, O == V
    (1, 4, 200, 130, 128, torch.bfloat16),      # three key tiles, the last one nearly empty
    (2, 10, 130, 97, 64, torch.float16),        # f16 path
    (1, 2, 70, 191, 16, torch.bfloat16),        # the tiny test U-Net's head dim
    (1, 3, 50, 65, 96, torch.float16),
    (32, 8, 256, 256, 160, torch.bfloat16),     # 512 workgroups at dh = 160: the one-LDS-image form, two workgroups per CU
    (64, 8, 130, 200, 136, torch.float16),      # the same form, ragged both ways, the denominator in V^T's row of ones (dh < 160)
    (70, 8, 100, 1, 160, torch.bfloat16),       # ... and a single key
])
def test_sattn_shapes(H, B, H_, Lq, Lk, dh, dtype):
    g = torch.Generator().manual_seed(Lq * 7 + dh + Lk)
    C = H_ * dh
    q = torch.randn(B, Lq, C, generator=g).to(dtype)
    k = torch.randn(B, Lk, C, generator=g).to(dtype)
    v = torch.randn(B, Lk, C, generator=g).to(dtype)
    o = H.sattn(q.cuda(), k.cuda(), v.cuda(), H_)
    ref = _ref_gpu(q.cuda(), k.cuda(), v.cuda(), H_)
    assert torch.isfinite(o.float()).all()
    assert O.rel_fro(o.double().cpu(), ref.cpu()) < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16)
    if Lk == 1:
        assert torch.equal(o.cpu(), v.expand(B, Lq, C).contiguous())
    if Lq * Lk <= 256 * 256:                     # small enough for the CPU oracle: pins the GPU reference too
        assert O.rel_fro(ref.cpu(), O.xattn_ref(q, k, v, H_)) < 1e-12


def test_sattn_peaked_logits_and_running_max(H):
    """Keys ordered so that the running maximum keeps rising across the key tiles (every tile rescales the
    accumulators), with large logits and a custom scale."""
    g = torch.Generator().manual_seed(11)
    B, H_, Lq, Lk, dh = 1, 4, 128, 640, 40
    C = H_ * dh
    q = (torch.randn(B, Lq, C, generator=g).abs() * 3).to(torch.bfloat16)
    ramp = torch.linspace(0.1, 3.0, Lk)[None, :, None]
    k = (torch.randn(B, Lk, C, generator=g).abs() * ramp).to(torch.bfloat16)
    v = torch.randn(B, Lk, C, generator=g).to(torch.bfloat16)
    o = H.sattn(q.cuda(), k.cuda(), v.cuda(), H_, scale=0.5)
    ref = _ref_gpu(q.cuda(), k.cuda(), v.cuda(), H_, scale=0.5)
    assert torch.isfinite(o.float()).all()
    assert O.rel_fro(o.double().cpu(), ref.cpu()) < TOL_BF16


@pytest.mark.parametrize("variant", ["0", "1", "2", "3", "4"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sattn_two_peaks_far_above_the_rest(variant, dtype):
    """One key scores `base`, a second `base + J`, every other key -64 (all in powers of two): the spread exceeds the f32 exponent
    range, so a running maximum that misses a key of its half tile overflows.  Key 4 sits in the half of the lane pair whose maximum
    hipcc 7.2.0 dropped from k_sattn_h (round 6: inf / NaN rows on q, k x 5; see lane_pair_max); the second peak comes early (same
    tile) or late (key tile 51).  Every kernel form, forced."""
    import os
    from uce_amd import edit as E
    old = os.environ.get("UCE_SATTN_QT")
    os.environ["UCE_SATTN_QT"] = variant
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_SATTN_QT"]
        else:
            os.environ["UCE_SATTN_QT"] = old
    B, H_, L, dh = 1, 2, 4096, 40
    C = H_ * dh
    c = dh ** -0.5 * 1.4426950408889634
    try:
        for base, J, second in ((100.0, 4.0, 40), (0.0, 100.0, 3325), (60.0, 70.0, 3325), (70.0, 4.0, 40)):
            if dtype == torch.float16 and base + J > 110:
                continue                                     # the key itself would leave f16's range
            g = torch.Generator().manual_seed(3)
            q, k = torch.zeros(B, L, C), torch.zeros(B, L, C)
            v = torch.randn(B, L, C, generator=g)
            q[..., 0::dh] = 16.0
            k[:, :, 0::dh] = -64.0 / (16 * c)
            k[:, 4, 0::dh] = base / (16 * c)
            k[:, second, 0::dh] = (base + J) / (16 * c)
            q, k, v = q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda()
            o = Hv.sattn(q, k, v, H_)
            assert torch.isfinite(o.float()).all(), (base, J, second)
            ref = _ref_gpu(q, k, v, H_)
            assert O.rel_fro(o.double().cpu(), ref.cpu()) < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16), (base, J, second)
    finally:
        torch.cuda.synchronize()
        Hv.close()


def test_sattn_matches_xattn_on_short_contexts(H):
    """Same problem through both kernels (Lk = 77 fits the cross-attention kernel)."""
    g = torch.Generator().manual_seed(5)
    q = torch.randn(2, 300, 320, generator=g).bfloat16().cuda()
    k = torch.randn(2, 77, 320, generator=g).bfloat16().cuda()
    v = torch.randn(2, 77, 320, generator=g).bfloat16().cuda()
    a, b = H.sattn(q, k, v, 8), H.xattn(q, k, v, 8)
    assert O.rel_fro(a.double().cpu(), b.double().cpu()) < 4e-3


def test_sattn_scratch_growth_keeps_old_buffers_valid(H):
    """The V^T scratch grows on demand; a smaller call after a larger one (and vice versa) stays correct."""
    g = torch.Generator().manual_seed(9)
    for B, L in ((1, 64), (4, 512), (1, 64), (8, 1024)):
        q = torch.randn(B, L, 320, generator=g).bfloat16().cuda()
        o = H.sattn(q, q, q, 8)
        assert O.rel_fro(o.double().cpu(), _ref_gpu(q, q, q, 8).cpu()) < TOL_BF16


def test_sattn_rejects_bad_arguments(H):
    from uce_amd import lib as L
    q = torch.zeros(1, 8, 8 * 36, dtype=torch.bfloat16, device="cuda:0")      # dh = 36 is not a multiple of 8
    with pytest.raises(L.UceError):
        H.sattn(q, q, q, 8)
    q = torch.zeros(1, 8, 8 * 168, dtype=torch.bfloat16, device="cuda:0")     # dh > 160
    with pytest.raises(L.UceError):
        H.sattn(q, q, q, 8)


@pytest.mark.parametrize("variant", ["1", "2", "3", "4"])
@pytest.mark.parametrize("B,H_,Lq,Lk,dh,dtype", [
    (2, 8, 1024, 1024, 40, torch.bfloat16),     # dh = 40: one / two query tiles per wave, the pipelined kernel
    (1, 8, 300, 130, 40, torch.bfloat16),       # ragged: a 256-row workgroup with a partial second tile, three key tiles
    (3, 4, 77, 64, 40, torch.float16),          # exactly one key tile (the pipeline's prologue only)
    (2, 8, 512, 200, 80, torch.bfloat16),       # dh = 80: k_sattn and k_sattn_p
    (1, 2, 40, 1, 80, torch.bfloat16),          # a single key
    (2, 4, 100, 33, 40, torch.bfloat16),        # one key tile whose second half holds a single key
    (2, 4, 260, 97, 40, torch.bfloat16),        # two key tiles, the last with one key in its second half
    (1, 4, 257, 128, 40, torch.float16),        # exactly two full key tiles (no padding path in the last tile)
    (1, 8, 256, 4000, 48, torch.bfloat16),      # 63 key tiles, dh = DHP
])
def test_sattn_every_kernel_form_forced(variant, B, H_, Lq, Lk, dh, dtype):
    """k_sattn with one (1) and two (2) query tiles per wave, the software-pipelined k_sattn_p (3) and the half-tile pipeline
    k_sattn_h (4; dh <= 48, elsewhere the by-shape kernel), each forced through a handle of its own (UCE_SATTN_QT is read at uce_create) at sizes the by-shape rule would not send there."""
    import os
    from uce_amd import edit as E
    old = os.environ.get("UCE_SATTN_QT")
    os.environ["UCE_SATTN_QT"] = variant
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_SATTN_QT"]
        else:
            os.environ["UCE_SATTN_QT"] = old
    g = torch.Generator().manual_seed(Lq * 5 + dh + Lk)
    C = H_ * dh
    q = torch.randn(B, Lq, C, generator=g).to(dtype)
    k = torch.randn(B, Lk, C, generator=g).to(dtype)
    v = torch.randn(B, Lk, C, generator=g).to(dtype)
    try:
        o = Hv.sattn(q.cuda(), k.cuda(), v.cuda(), H_)
        again = Hv.sattn(q.cuda(), k.cuda(), v.cuda(), H_)
    finally:
        torch.cuda.synchronize()
        Hv.close()
    ref = _ref_gpu(q.cuda(), k.cuda(), v.cuda(), H_)
    assert O.rel_fro(o.double().cpu(), ref.cpu()) < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16)
    assert torch.equal(o, again)
    if Lk == 1:
        assert torch.equal(o.cpu(), v.expand(B, Lq, C).contiguous())


def test_sattn_half_tile_pipeline_rising_max_and_packed():
    """k_sattn_h forced: keys ordered so that the running maximum rises in every half tile (the deferred P V product is
    rescaled together with the accumulators), and the packed q | k | v entry against three separate tensors."""
    import os
    from uce_amd import edit as E
    old = os.environ.get("UCE_SATTN_QT")
    os.environ["UCE_SATTN_QT"] = "4"
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_SATTN_QT"]
        else:
            os.environ["UCE_SATTN_QT"] = old
    try:
        g = torch.Generator().manual_seed(21)
        B, H_, Lq, Lk, dh = 1, 4, 128, 640, 40
        C = H_ * dh
        q = (torch.randn(B, Lq, C, generator=g).abs() * 3).to(torch.bfloat16).cuda()
        ramp = torch.linspace(0.1, 3.0, Lk)[None, :, None]
        k = (torch.randn(B, Lk, C, generator=g).abs() * ramp).to(torch.bfloat16).cuda()
        v = torch.randn(B, Lk, C, generator=g).to(torch.bfloat16).cuda()
        o = Hv.sattn(q, k, v, H_, scale=0.5)
        assert torch.isfinite(o.float()).all()
        assert O.rel_fro(o.double().cpu(), _ref_gpu(q, k, v, H_, scale=0.5).cpu()) < TOL_BF16
        qkv = torch.randn(3, 700, 3 * 320, generator=g).to(torch.bfloat16).cuda()
        a = Hv.sattn_packed(qkv, 8)
        qq, kk, vv = (t.contiguous() for t in qkv.split(320, dim=-1))
        assert torch.equal(a, Hv.sattn(qq, kk, vv, 8))
        assert O.rel_fro(a.double().cpu(), _ref_gpu(qq, kk, vv, 8).cpu()) < TOL_BF16
    finally:
        torch.cuda.synchronize()
        Hv.close()


@pytest.mark.parametrize("vti", ["1", "3"])
@pytest.mark.parametrize("B,H_,Lq,Lk,dh,dtype", [
    (2, 8, 1024, 1024, 40, torch.bfloat16), (1, 8, 300, 130, 40, torch.bfloat16), (3, 4, 77, 64, 40, torch.float16),
    (2, 4, 100, 33, 40, torch.bfloat16), (2, 4, 260, 97, 40, torch.bfloat16), (1, 8, 256, 4000, 48, torch.bfloat16),
    (1, 2, 40, 1, 40, torch.bfloat16),
])
def test_sattn_half_tile_pipeline_both_inline_v_forms(vti, B, H_, Lq, Lk, dh, dtype):
    """k_sattn_h (UCE_SATTN_QT=4) with V transposed on its way into LDS by 2-byte stores (UCE_SATTN_VTI=1) and with V kept row-major
    and read through ds_read_b64_tr_b16 (3; what the by-shape rule takes): the same shapes, against fp64 - and the two agree with each
    other to the last bit (same fragments, same MFMA order)."""
    import os
    from uce_amd import edit as E
    hs = {}
    for val in ("1", vti):
        old = {k: os.environ.get(k) for k in ("UCE_SATTN_QT", "UCE_SATTN_VTI")}
        os.environ.update(UCE_SATTN_QT="4", UCE_SATTN_VTI=val)
        try:
            hs[val] = E.UceHandle("cuda:0")
        finally:
            for k, v_ in old.items():
                if v_ is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v_
    g = torch.Generator().manual_seed(Lq * 3 + dh + Lk)
    C = H_ * dh
    q = torch.randn(B, Lq, C, generator=g).to(dtype).cuda()
    k = torch.randn(B, Lk, C, generator=g).to(dtype).cuda()
    v = torch.randn(B, Lk, C, generator=g).to(dtype).cuda()
    try:
        o = hs[vti].sattn(q, k, v, H_)
        o1 = hs["1"].sattn(q, k, v, H_)
    finally:
        torch.cuda.synchronize()
        for h_ in set(hs.values()):
            h_.close()
    ref = _ref_gpu(q, k, v, H_)
    assert O.rel_fro(o.double().cpu(), ref.cpu()) < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16)
    assert torch.equal(o, o1)


def _exp2_inputs(B, L, heads, dh, dtype, seed, gain=1.0):
    """x-free stand-in for the projection: f32 q | k | v, the q columns scaled by dh^-0.5 * log2(e) BEFORE the one rounding."""
    g = torch.Generator().manual_seed(seed)
    C = heads * dh
    f = torch.randn(B, L, 3 * C, generator=g) * gain
    c = dh ** -0.5 * 1.4426950408889634
    scaled = f.clone()
    scaled[..., :C] *= c
    return f.to(dtype).cuda(), scaled.to(dtype).cuda(), c


def _ref_exp2(qkv_scaled, heads):
    """fp64 softmax in the exp2 domain on the rounded, pre-scaled q: 2^(q'.k) / sum."""
    B, L, C3 = qkv_scaled.shape
    C = C3 // 3
    dh = C // heads
    sp = lambda t: t.double().view(B, L, heads, dh).transpose(1, 2)
    q, k, v = (sp(t) for t in qkv_scaled.split(C, dim=-1))
    s = q @ k.transpose(-1, -2) * 0.6931471805599453                  # exp2(x) = exp(x ln 2)
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, C)


@pytest.mark.parametrize("B,L,dtype,gain", [
    (16, 2048, torch.bfloat16, 1.0),      # 1024 two-tile workgroups: the exp2-domain kernel by rule
    (4, 2050, torch.bfloat16, 1.0),       # a ragged last key tile (masking) on the same kernel
    (16, 2048, torch.float16, 1.0),
    (32, 1024, torch.bfloat16, 5.0),      # peaked logits (q, k x 5): the reference point keeps moving, sc -= d / rescale path
    (2, 4096, torch.bfloat16, 1.0),       # one prompt's CFG pair at the 64 x 64 level
])
def test_sattn_packed_exp2_matches_fp64(H, B, L, dtype, gain):
    """uce_sattn_packed_exp2_fwd (scores = exp2 arguments straight off the matrix pipe, the running maximum in the head's padding
    dim) against fp64 on the same pre-scaled q, and against uce_sattn_packed_fwd on the unscaled projection (they differ by the
    rounding of q' in place of q only)."""
    heads, dh = 8, 40
    plain, scaled, _ = _exp2_inputs(B, L, heads, dh, dtype, seed=L + B, gain=gain)
    assert H.sattn_exp2_form(B, heads, L, dh)
    o = H.sattn_packed_exp2(scaled, heads)
    tol = TOL_BF16 if dtype == torch.bfloat16 else TOL_F16
    assert torch.isfinite(o.float()).all()
    err = O.rel_fro(o.double().cpu(), _ref_exp2(scaled, heads).cpu())
    assert err < tol, err
    # the same attention from the unscaled projection: equal up to the rounding of q (bf16: 2^-9 per element of q, times the logit spread)
    o2 = H.sattn_packed(plain, heads)
    assert O.rel_fro(o.double().cpu(), o2.double().cpu()) < (4 if gain > 1 else 2) * tol


def test_sattn_packed_exp2_other_shapes_take_the_unit_factor_path(H):
    """Shapes without the exp2-domain form (another head dim, a short layer) run their usual kernels on the pre-scaled q."""
    for B, L, heads, dh in ((2, 256, 8, 160), (2, 64, 8, 160), (2, 1024, 8, 80), (1, 300, 4, 40)):
        assert not H.sattn_exp2_form(B, heads, L, dh)
        _, scaled, _ = _exp2_inputs(B, L, heads, dh, torch.bfloat16, seed=L)
        o = H.sattn_packed_exp2(scaled, heads)
        assert O.rel_fro(o.double().cpu(), _ref_exp2(scaled, heads).cpu()) < TOL_BF16


def test_sattn_packed_exp2_two_tile_kernel_small_and_ragged():
    """The exp2-domain kernel forced at small / ragged shapes (UCE_SATTN_QT=4): one key tile, a single key, Lq off the tile size."""
    import os
    from uce_amd import edit as E
    old = os.environ.get("UCE_SATTN_QT")
    os.environ["UCE_SATTN_QT"] = "4"
    try:
        Hv = E.UceHandle("cuda:0")
    finally:
        if old is None:
            del os.environ["UCE_SATTN_QT"]
        else:
            os.environ["UCE_SATTN_QT"] = old
    try:
        for B, L, dtype, gain in ((1, 1, torch.bfloat16, 1.0), (2, 33, torch.bfloat16, 1.0), (1, 64, torch.float16, 1.0),
                                  (3, 130, torch.bfloat16, 4.0), (1, 700, torch.bfloat16, 1.0), (2, 257, torch.float16, 3.0)):
            assert Hv.sattn_exp2_form(B, 4, L, 40)
            _, scaled, _ = _exp2_inputs(B, L, 4, 40, dtype, seed=L, gain=gain)
            o = Hv.sattn_packed_exp2(scaled, 4)
            assert torch.isfinite(o.float()).all()
            err = O.rel_fro(o.double().cpu(), _ref_exp2(scaled, 4).cpu())
            assert err < (TOL_BF16 if dtype == torch.bfloat16 else TOL_F16), (B, L, err)
    finally:
        torch.cuda.synchronize()
        Hv.close()
