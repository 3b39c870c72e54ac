"""GPU stress of the cross-workgroup hand-offs (SURVEY.md section 5 "race detection": no sanitizer exists for the ticket /
stage-word recipes, so they are exercised instead): random edits back to back on two handles and two streams, the same
edit replayed from a captured hipGraph, and random convolution shapes through the LDS-DMA ring - every result against an
independent fp64 / fp32 evaluation, every repeat bit-identical.  (The long forms: tools/stress_edit.py, stress_conv.py.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import uce_oracle as O
from uce_amd import edit as E

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _exact(C, G, s, W64, n_e, d):
    C64, s64 = C.double(), s.double()
    A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda") + C64.T @ (s64[:, None] * C64)
    Delta = torch.linalg.solve(A, (s64[:n_e, None] * C64[:n_e]).T @ (G - C[:n_e]).double()).T
    return W64 + W64 @ Delta


def _random_job(rng, d, big):
    n_e = int(rng.integers(1, 200 if big else 129))
    n_p = int(rng.integers(0, max(1, 129 - n_e))) if n_e < 128 else 0
    N = n_e + n_p
    Call = O.clip_like_embeddings(N + 1, d, seed=int(rng.integers(1 << 30)))
    return _dev(Call[:N]), _dev(np.repeat(Call[N:N + 1], n_e, axis=0)), _dev((0.5 + rng.random(N)).astype(np.float32)), n_e


@pytest.mark.parametrize("seed", [0, 1])
def test_random_edits_back_to_back_on_two_handles_and_streams(seed):
    """24 random concept sets (1..199 edit concepts: one- and two-block rider systems, the launch chain beyond 128), issued
    alternately on the cached handle / torch's current stream and on a second handle / a side stream WITHOUT any
    synchronisation in between (each handle owns its ticket, stage and counter words), then all checked against fp64."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, rows = 768, 4096
    H1, H2 = E.UceHandle.get("cuda:0"), E.UceHandle("cuda:0")
    side = torch.cuda.Stream()
    W = _dev(O.linear_default_weight(rows, d, rng))
    W64 = W.double()
    jobs, outs = [], []
    torch.cuda.synchronize()
    try:
        for it in range(24):
            C, G, s, n_e = _random_job(rng, d, big=(it % 6 == 0))
            jobs.append((C, G, s, n_e))
            if it % 2 == 0:
                outs.append(H1.edit(C, G, s, 0.5, W))
            else:
                side.wait_stream(torch.cuda.current_stream())          # the inputs were produced on the current stream
                with torch.cuda.stream(side):
                    outs.append(H2.edit(C, G, s, 0.5, W))
        torch.cuda.synchronize()
        H1.status()
        H2.status()
        worst = 0.0
        for (C, G, s, n_e), out in zip(jobs, outs):
            want = _exact(C, G, s, W64, n_e, d)
            err = float((out.double() - want).norm() / want.norm())
            worst = max(worst, err)
            assert err < 1e-5, (n_e, C.shape[0], err)
        assert worst > 0
    finally:
        H2.close()


@pytest.mark.parametrize("n_e,n_p", [(50, 0), (100, 20), (600, 400)])
def test_edit_replays_from_a_captured_graph(n_e, n_p):
    """The rider hand-off keeps no launch-specific value in the kernel arguments (the stage word and its reset live on the
    device), so a captured uce_edit can be replayed: three replays on changed weights, each against fp64.  (600 + 400
    concepts: the primal path - the persistent Cholesky launch with its rider jobs, whose flags the launch clears itself.)"""
    d, rows = 768, 2048
    rng = np.random.Generator(np.random.PCG64(n_e))
    H = E.UceHandle.get("cuda:0")
    N = n_e + n_p
    Call = O.clip_like_embeddings(N + 1, d, seed=7)
    C, G = _dev(Call[:N]), _dev(np.repeat(Call[N:N + 1], n_e, axis=0))
    s = _dev((0.5 + rng.random(N)).astype(np.float32))
    W = _dev(O.linear_default_weight(rows, d, rng))
    out = torch.empty_like(W)
    H.edit(C, G, s, 0.5, W, out=out, check=True)                      # warm-up: workspace, function attributes
    first = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        H.edit(C, G, s, 0.5, W, out=out)
    for rep in range(3):
        if rep:
            W.copy_(_dev(O.linear_default_weight(rows, d, rng)))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        H.status()
        want = _exact(C, G, s, W.double(), n_e, d)
        assert float((out.double() - want).norm() / want.norm()) < 1e-5, rep
        if rep == 0:
            assert torch.equal(out, first)                            # bit-identical to the eager launch
    H.edit(C, G, s, 0.5, W, out=out, check=True)                      # and the handle still works eagerly afterwards


def test_random_convolution_shapes_repeat_bit_identically():
    """uce_conv3x3_nhwc_fwd on random shapes (ragged pixel counts, tiles straddling images, every output-tile width, fused
    upsample, both element types), four launches each: the direct-to-LDS ring must never read a stage before its loads
    have landed - against F.conv2d in fp32."""
    rng = np.random.Generator(np.random.PCG64(3))
    H = E.UceHandle.get("cuda:0")
    for it in range(14):
        Cin = int(rng.choice([64, 128, 192, 320, 640]))
        Cout = int(rng.choice([64, 128, 256, 320, 384, 512, 640]))
        up = bool(rng.integers(0, 2))
        N = int(rng.integers(1, 4))
        Hh, Ww = int(rng.integers(40, 150)), int(rng.integers(40, 150))
        if up:
            Hh, Ww = Hh & ~1, Ww & ~1
        dtype = torch.float16 if rng.integers(0, 4) == 0 else torch.bfloat16
        gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        hs, ws = (Hh // 2, Ww // 2) if up else (Hh, Ww)
        x = torch.randn(N, Cin, hs, ws, generator=gen).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to("cuda", dtype).to(memory_format=torch.channels_last)
        with torch.no_grad():
            xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
            ref = F.conv2d(xin, conv.weight.float(), conv.bias.float(), padding=1)
            outs = [H.conv3x3_igemm(x, conv.weight, conv.bias, upsample=up) for _ in range(4)]
            torch.cuda.synchronize()
        for y in outs:
            err = float((y.float() - ref).norm() / ref.norm())
            assert err < (6e-3 if dtype == torch.bfloat16 else 1e-3), (it, N, Cin, Cout, Hh, Ww, up, dtype, err)
            assert torch.equal(y, outs[0]), (it, "not bit-repeatable")


@pytest.mark.parametrize("d", [768, 1024])
def test_two_persistent_factorisations_side_by_side(d):
    """The primal path's persistent Cholesky (1 walker + tile + L^-1 workgroups with flag hand-offs, 133 / 241 workgroups per
    launch) issued from two handles on two streams without synchronisation in between: both launches must make progress side by
    side on the 256 CUs (each handle owns its flags), every result against fp64."""
    rng = np.random.Generator(np.random.PCG64(d))
    rows, N, n_e = 2048, d + 40, 300                                   # N >= d: primal
    H1, H2 = E.UceHandle.get("cuda:0"), E.UceHandle("cuda:0")
    side = torch.cuda.Stream()
    W = _dev(O.linear_default_weight(rows, d, rng))
    W64 = W.double()
    jobs, outs = [], []
    try:
        for it in range(6):
            Call = O.clip_like_embeddings(N + 1, d, seed=int(rng.integers(1 << 30)))
            C, G = _dev(Call[:N]), _dev(np.repeat(Call[N:N + 1], n_e, axis=0))
            s = _dev((0.5 + rng.random(N)).astype(np.float32))
            jobs.append((C, G, s))
            if it % 2 == 0:
                outs.append(H1.edit(C, G, s, 0.5, W))
            else:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    outs.append(H2.edit(C, G, s, 0.5, W))
        torch.cuda.synchronize()
        H1.status()
        H2.status()
        for (C, G, s), out in zip(jobs, outs):
            want = _exact(C, G, s, W64, n_e, d)
            assert float((out.double() - want).norm() / want.norm()) < 1e-5
    finally:
        H2.close()


ASAN_GPU_CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["UCE_REPO_ROOT"])
import torch
from uce_amd import edit as E, lib as L
assert L.lib_path() == os.environ["UCE_HIP_LIB"]
ref = torch.load(os.environ["UCE_ASAN_REF"])
H = E.UceHandle("cuda:0")
outs = {}
for name, (C, G, s, W) in ref["inputs"].items():
    outs[name] = H.edit(C.cuda(), G.cuda(), s.cuda(), 0.5, W.cuda(), check=True).cpu()
q, k, v = (t.cuda() for t in ref["qkv"])
outs["xattn"] = H.xattn(q, k[:, :77].contiguous(), v[:, :77].contiguous(), 8).cpu()
outs["sattn"] = H.sattn(q, k, v, 8).cpu()
x, w = ref["lin"]
outs["linear"] = H.linear(x.cuda(), w.cuda()).cpu()
torch.cuda.synchronize()
H.close()
for n, t in outs.items():
    assert torch.equal(t, ref["outputs"][n]), n
print("asan gpu child ok")
"""


def test_edit_and_attention_through_the_address_sanitizer_build(tmp_path):
    """The host side under AddressSanitizer on the GPU: the low-rank, the two-block and the primal edit (workspace growth, rider
    hand-off words, the persistent Cholesky's job tables), both attention entries and a linear layer run in a child process that
    loads the ASAN build - results bit-identical to the product library's (same device code), no ASAN report."""
    import os
    import subprocess
    import sys
    from uce_amd import REPO_ROOT, build as B, edit as E
    path = B.asan_lib_path()
    if not os.path.exists(path):
        pytest.skip("the ASAN variant was not built (tests/test_abi_cpu.py builds it: uce_amd.build.build_asan)")
    H = E.UceHandle.get("cuda:0")
    g = torch.Generator().manual_seed(77)
    inputs = {}
    for name, (N, n_e, rows) in {"lowrank": (50, 50, 2048), "two_block": (100, 100, 1024), "primal": (800, 500, 1024)}.items():
        C = torch.randn(N, 768, generator=g)
        C = (C + 2.0 * torch.randn(1, 768, generator=g)) * 0.9
        G = C[:n_e].roll(1, 0).contiguous()
        inputs[name] = (C, G, torch.ones(N), (torch.rand(rows, 768, generator=g) * 2 - 1) * 768 ** -0.5)
    qkv = [torch.randn(2, 300, 320, generator=g).to(torch.bfloat16) for _ in range(3)]
    lin = (torch.randn(4096, 320, generator=g).to(torch.bfloat16), (torch.randn(320, 320, generator=g) * 0.05).to(torch.bfloat16))
    outputs = {n: H.edit(C.cuda(), G.cuda(), s.cuda(), 0.5, W.cuda(), check=True).cpu() for n, (C, G, s, W) in inputs.items()}
    q, k, v = (t.cuda() for t in qkv)
    outputs["xattn"] = H.xattn(q, k[:, :77].contiguous(), v[:, :77].contiguous(), 8).cpu()
    outputs["sattn"] = H.sattn(q, k, v, 8).cpu()
    outputs["linear"] = H.linear(lin[0].cuda(), lin[1].cuda()).cpu()
    ref = tmp_path / "ref.pt"
    torch.save({"inputs": inputs, "qkv": qkv, "lin": lin, "outputs": outputs}, ref)
    # ROCm's ASAN runtime intercepts the HSA pool allocator (its device-ASAN support) and serves it from its own heap, which
    # needs XNACK (retryable page faults) on the GPU: ask for it
    env = dict(os.environ, UCE_HIP_LIB=path, LD_PRELOAD=B.asan_runtime(), UCE_REPO_ROOT=REPO_ROOT, UCE_ASAN_REF=str(ref),
               HSA_XNACK="1", ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86:protect_shadow_gap=0")
    res = subprocess.run([sys.executable, "-c", ASAN_GPU_CHILD], env=env, capture_output=True, text=True, timeout=900)
    if "asan gpu child ok" not in res.stdout and "hsa_amd_memory_pool_allocate" in res.stderr and "libuce_hip" not in res.stderr:
        # the report comes from the runtime's own HSA interceptor inside libamdhip64 (no frame of this library): the box does not
        # grant XNACK, so HIP cannot come up under this ASAN runtime at all - an environment limit, not a finding
        pytest.skip("ROCm's ASAN runtime could not allocate HSA pool memory on this box (XNACK unavailable); the host-only checks "
                    "of the same build run in tests/test_abi_cpu.py")
    assert "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
    assert res.returncode == 0 and "asan gpu child ok" in res.stdout, (res.returncode, res.stdout[-500:], res.stderr[-3000:])


def test_random_self_attention_shapes_through_every_kernel_form():
    """tools/stress_sattn.py (short form): 120 random (batch, heads, head dim, lengths, dtype) through uce_sattn_fwd, the packed and the
    exp2-domain entry points, on the by-rule handle and on one that forces the two-tile kernel - each against fp64 (the buffer-load
    tile descriptors, the row-major V image and the one-LDS-image form see every ragged edge here)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_sattn.py"), "120", "5"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0 and "ok: 120 shapes" in r.stdout, (r.stdout[-600:], r.stderr[-600:])
