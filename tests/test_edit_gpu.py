"""GPU parity tests of the closed-form edit: every call goes through the C ABI (libuce_hip.so).

Acceptance protocol:
  eps_ref   = relF(reference fp32 output, exact64)      - what the reference itself achieves
  eps_build = relF(build, exact64)           <= 1e-5    - ours, against the fp64 evaluation
  relF(build, reference fp32)                <= max(1e-4, 1.5 * eps_ref)
"""
import math

import numpy as np
import pytest
import torch

from oracle import uce_oracle as O
from tests import fakepipe
from tests.golden_io import Case, ERASE_CASES, DEBIAS_CASES, DEBIAS_ALIAS_CASES, CLI_CASES
from uce_amd import lib as L

pytestmark = pytest.mark.gpu

EPS_BUILD = 1e-5


@pytest.fixture(scope="module")
def H():
    from uce_amd import edit as E
    return E.UceHandle.get("cuda:0")


def _dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to(device="cuda:0", dtype=dtype).contiguous()


def _synthetic(N, Ne, d, seed):
    C = O.clip_like_embeddings(N + 1, d, seed)
    Cm = C[:N]
    G = np.repeat(C[N:N + 1], Ne, axis=0) + 0.0
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    s = rng.uniform(0.5, 1.5, size=N).astype(np.float32)
    return Cm, G, s


def _exact(C, G, s, lamb):
    """float64 A, Bt, DeltaT on the host."""
    C64, s64 = C.astype(np.float64), s.astype(np.float64)
    Ne = G.shape[0]
    A = lamb * np.eye(C.shape[1]) + C64.T @ (s64[:, None] * C64)
    Dm = (G.astype(np.float32) - C[:Ne].astype(np.float32)).astype(np.float64)
    Bt = C64[:Ne].T @ (s64[:Ne, None] * Dm)
    DeltaT = np.linalg.solve(A, Bt)
    return A, Bt, DeltaT


# ------------------------------------------------------------------------------------ kernels

@pytest.mark.parametrize("N,Ne,d", [(5, 2, 64), (33, 33, 128), (50, 50, 768), (700, 400, 768), (130, 100, 1024),
                                    (1500, 1000, 768)])
def test_gram_matches_f64(H, N, Ne, d):
    C, G, s = _synthetic(N, Ne, d, seed=N)
    A, Bt = H.gram(_dev(C), _dev(G), _dev(s), 0.5)
    Ae, Bte, _ = _exact(C, G, s, 0.5)
    assert O.rel_fro(A.cpu(), Ae) < 1e-14
    assert O.rel_fro(Bt.cpu(), Bte) < 1e-13
    assert torch.equal(A, A.T)


@pytest.mark.parametrize("N,Ne,d", [(5, 2, 64), (40, 40, 128), (50, 50, 768), (900, 500, 768), (100, 64, 1024)])
def test_solve_delta_matches_f64(H, N, Ne, d):
    C, G, s = _synthetic(N, Ne, d, seed=N + 1)
    Ae, Bte, DTe = _exact(C, G, s, 0.5)
    DT = H.solve_delta(_dev(Ae, torch.float64), _dev(Bte, torch.float64))
    H.status()
    assert O.rel_fro(DT.cpu(), DTe) < 5e-7     # fp64 solve, fp32 store


def test_solve_reports_indefinite_system(H):
    d = 128
    A = -np.eye(d)
    with pytest.raises(L.UceError):
        H.solve_delta(_dev(A, torch.float64), _dev(np.zeros((d, d)), torch.float64))
        H.status()
    # and the handle recovers
    DT = H.solve_delta(_dev(np.eye(d), torch.float64), _dev(np.eye(d), torch.float64))
    H.status()
    assert O.rel_fro(DT.cpu(), np.eye(d)) < 1e-7


@pytest.mark.parametrize("rows_,d", [(96, 64), (128, 128), (1000, 768), (24960, 768), (333, 1024), (700, 2048)])
def test_apply_matches_f64(H, rows_, d):
    rng = np.random.Generator(np.random.PCG64(rows_))
    W = O.linear_default_weight(rows_, d, rng)
    # an asymmetric Delta so a transposed operand or output cannot pass
    DT = (rng.standard_normal((d, d)) * (0.5 / math.sqrt(d))).astype(np.float32)
    DT[0, 1] += 3.0
    out = H.apply(_dev(W), _dev(DT))
    want = W.astype(np.float64) + W.astype(np.float64) @ DT.astype(np.float64).T
    assert O.rel_fro(out.cpu(), want) < 2e-6      # 768-term f32 fmaf chains


def _handle_with(env_name, value):
    """The A/B switches are read when a handle is created: a handle of its own for a forced variant."""
    import os
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


def test_apply_beyond_one_buffer_descriptor_walks_row_chunks(H):
    """A slab of more than 2 GB (the reach of the dense apply's buffer descriptors): uce_apply walks it in row chunks - rows are
    independent, so every block of the big result carries the bits the same rows give as a small slab of their own."""
    d = 768
    rows_ = (1 << 31) // (4 * d) + 5000                                   # 2.16 GB of f32 weights: two chunks
    g = torch.Generator(device="cuda").manual_seed(5)
    W = (torch.rand(rows_, d, device="cuda", generator=g) - 0.5) * (2.0 / math.sqrt(d))
    DT = torch.randn(d, d, device="cuda", generator=g) * (0.5 / math.sqrt(d))
    DT[0, 1] += 3.0
    out = H.apply(W, DT)
    for r0 in (0, rows_ // 2 - 700, rows_ - 1500):                        # first chunk, across the chunk boundary, tail
        blk = W[r0:r0 + 1400].contiguous()
        small = H.apply(blk, DT)
        assert torch.equal(out[r0:r0 + 1400], small)
        want = blk.double() + blk.double() @ DT.double().T
        assert O.rel_fro(small.cpu(), want.cpu()) < 2e-6
    del out, W
    torch.cuda.empty_cache()


def test_apply_f16_split_scales_rows_and_columns(H):
    """The two-way f16 split takes a power-of-two scale per row of W_old and per row of (I + Delta)^T: rows of W 2^40
    apart, an all-zero row, columns of Delta far below the identity - the scaled result must carry fp32-level error on
    every row separately (a Frobenius norm over the whole matrix would hide the small rows)."""
    rng = np.random.Generator(np.random.PCG64(99))
    rows_, d = 640, 768
    W = O.linear_default_weight(rows_, d, rng)
    W *= np.ldexp(1.0, rng.integers(-20, 21, size=(rows_, 1))).astype(np.float32)
    W[17] = 0.0
    DT = (rng.standard_normal((d, d)) * (0.5 / math.sqrt(d))).astype(np.float32)
    DT[:, 5] *= 1e-6
    DT[9] *= 1e-5
    out = H.apply(_dev(W), _dev(DT)).cpu().numpy().astype(np.float64)
    want = W.astype(np.float64) + W.astype(np.float64) @ DT.astype(np.float64).T
    assert np.all(out[17] == 0.0)
    rown = np.linalg.norm(want, axis=1)
    err = np.linalg.norm(out - want, axis=1)
    ok = rown > 0
    assert float((err[ok] / rown[ok]).max()) < 2e-6


def test_primal_edit_riders_against_the_separate_calls(H):
    """uce_edit's primal path hands two jobs to rider workgroups of the persistent Cholesky launch - the f16 split of W_old
    and the Bt half of the Gram (A alone is then split over the concepts) - and uce_gram / uce_solve_delta / uce_apply do
    the same work in launches of their own: the same result up to the summation order of A, and the same bits on a
    second run (every reduction has a fixed order)."""
    C, G, s = _synthetic(900, 400, 768, seed=5)
    rng = np.random.Generator(np.random.PCG64(5))
    W = _dev(O.linear_default_weight(2000, 768, rng))
    Cd, Gd, sd = _dev(C), _dev(G), _dev(s)
    out = H.edit(Cd, Gd, sd, 0.5, W, algo=L.ALGO_PRIMAL, check=True)
    assert torch.equal(out, H.edit(Cd, Gd, sd, 0.5, W, algo=L.ALGO_PRIMAL, check=True))
    A, Bt = H.gram(Cd, Gd, sd, 0.5)
    DT = H.solve_delta(A, Bt)
    H.status()
    assert O.rel_fro(out.cpu(), H.apply(W, DT).cpu().double().numpy()) < 1e-6


@pytest.mark.parametrize("N,Ne,d", [(5, 2, 64), (50, 50, 768), (64, 10, 128), (300, 200, 768), (40, 36, 2048),
                                    (65, 65, 128)])
def test_dual_factors_and_lowrank_apply(H, N, Ne, d):
    C, G, s = _synthetic(N, Ne, d, seed=N + 2)
    Dm, R = H.dual_factors(_dev(C), _dev(G), _dev(s), 0.5)
    H.status()
    C64 = C.astype(np.float64)
    K = np.diag(0.5 / s.astype(np.float64)) + C64 @ C64.T
    Re = np.linalg.solve(K, C64)[:Ne]
    assert O.rel_fro(R.cpu(), Re) < 5e-7
    assert torch.equal(Dm.cpu(), torch.from_numpy(G - C[:Ne]))
    _, _, DTe = _exact(C, G, s, 0.5)
    rng = np.random.Generator(np.random.PCG64(7))
    W = O.linear_default_weight(200, d, rng)
    want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
    DT = H.delta_from_factors(Dm, R)
    assert O.rel_fro(DT.cpu(), DTe) < 5e-6
    if Ne <= 256:
        out = H.apply_lowrank(_dev(W), Dm, R)
        assert O.rel_fro(out.cpu(), want) < 2e-6


def test_dual_rejects_nonpositive_scale(H):
    C, G, s = _synthetic(6, 3, 64, seed=3)
    s[1] = -1.0
    with pytest.raises(L.UceError):
        H.dual_factors(_dev(C), _dev(G), _dev(s), 0.5)
        H.status()


def test_cast_bf16_bit_exact(H):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100003, generator=g) * 3
    x[:4] = torch.tensor([float("inf"), -float("inf"), 0.0, -0.0])
    x[4] = float("nan")
    xd = x.cuda()
    out = torch.empty(x.numel(), dtype=torch.bfloat16, device="cuda:0")
    H.cast_bf16(xd, out)
    want = xd.to(torch.bfloat16)
    a, b = out.view(torch.int16).cpu(), want.view(torch.int16).cpu()
    nan = torch.isnan(want.float().cpu())
    assert torch.equal(a[~nan], b[~nan]) and bool(torch.isnan(out.float().cpu()[nan]).all())


def test_debias_targets(H):
    rng = np.random.Generator(np.random.PCG64(9))
    Ce, Cd = O.clip_like_embeddings(7, 128, 1), O.clip_like_embeddings(3, 128, 2)
    D = rng.uniform(-0.5, 0.5, size=(7, 3))
    G = H.debias_targets(_dev(Ce), _dev(Cd), _dev(D, torch.float64))
    want = (Ce.astype(np.float64) + D @ Cd.astype(np.float64)).astype(np.float32)
    assert torch.equal(G.cpu(), torch.from_numpy(want))


# ------------------------------------------------------------------------------------ goldens

TILE = 11          # 96 fixture rows x 11 = 1056 >= 1024: uce_edit's product two-launch path (uce_api.hip: rows >= 1024)


def _run_case(H, c, algo, tile=1):
    """The edit of a golden case through uce_edit.  `tile` repeats the fixture's rows (rows of W are independent, so
    every replica must reproduce the reference's output for the original rows): with 96 rows uce_edit takes its small-slab
    form (k_apply_lowrank_generic); with 11 x 96 it launches what the bench times - k_lr_project (+ riders / the hosted
    Cholesky) and k_lr_update_s, or the primal chain with its rider jobs."""
    m = c.meta
    Ce, Ge, Cp = c.arr("C_edit"), c.arr("G_edit"), c.arr("C_pres")
    C = np.concatenate([Ce, Cp]) if len(Cp) else Ce
    s = np.array([m["erase_scale"]] * len(Ce) + [m["preserve_scale"]] * len(Cp), dtype=np.float32)
    W = torch.cat(c.w_old()).repeat(tile, 1)
    out = H.edit(_dev(C), _dev(Ge), _dev(s), m["lamb"], W.cuda().contiguous(), algo=algo, check=True)
    return out.cpu()


def _check_replicas(out, ex, ref, tile):
    """Every replica of the fixture's rows against the reference's own fp32 output and the fp64 evaluation."""
    eps_ref = O.rel_fro(ref, ex)
    n = ref.shape[0]
    assert out.shape[0] == n * tile
    worst = (0.0, 0.0)
    for r in range(tile):
        part = out[r * n:(r + 1) * n]
        eps_build, to_ref = O.rel_fro(part, ex), O.rel_fro(part, ref)
        worst = (max(worst[0], eps_build), max(worst[1], to_ref))
        assert eps_build < EPS_BUILD, (r, eps_build, eps_ref)
        assert to_ref < max(1e-4, 1.5 * eps_ref), (r, to_ref, eps_ref)
    print(f"eps_ref={eps_ref:.3e} eps_build={worst[0]:.3e} rel(build,ref32)={worst[1]:.3e} tile={tile}")


@pytest.mark.parametrize("tile", [1, TILE])
@pytest.mark.parametrize("algo", [L.ALGO_PRIMAL, L.ALGO_DUAL, L.ALGO_AUTO])
@pytest.mark.parametrize("name", ERASE_CASES)
def test_erase_golden(H, name, algo, tile):
    c = Case(name)
    N = len(c.arr("C_edit")) + len(c.arr("C_pres"))
    if algo == L.ALGO_DUAL and N > 1024:
        pytest.skip("dual form is for N < d")
    out = _run_case(H, c, algo, tile)
    _check_replicas(out, torch.cat(c.w_exact64()), torch.cat(c.w_ref32()), tile)


@pytest.mark.parametrize("tile", [1, TILE])
@pytest.mark.parametrize("name", DEBIAS_CASES + DEBIAS_ALIAS_CASES)
def test_debias_golden(H, name, tile):
    """DebiasState against the reference's own debias output.  The alias cases list a string twice / in two roles: the
    reference drifts ONE cached tensor per string in place (uce_sd_debias.py:122-127) and DebiasState follows it."""
    from uce_amd import edit as E
    c = Case(name)
    m = c.meta
    names = m["modules"]
    ws = [w.repeat(tile, 1) for w in c.w_old()]        # per module: replica r of module i = rows [r*n_i, (r+1)*n_i)
    rows_ = [w.shape[0] for w in ws]
    offs = [0] + list(np.cumsum(rows_)[:-1])
    slab = E.WeightSlab(names, [int(o) for o in offs], rows_, torch.cat(ws).cuda().contiguous())
    Cp = c.arr("C_pres")
    st = E.DebiasState(H, slab, _dev(c.arr("C_edit")), _dev(c.arr("C_debias")), _dev(Cp) if len(Cp) else None,
                       m["edit_scale"], m["preserve_scale"], m["lamb"], keys=(m["edit"], m["debias"], m["preserve"]))
    assert st.aliased == (name in DEBIAS_ALIAS_CASES)
    for ds in c.arr("direction_scales"):
        if np.abs(ds).max() == 0:          # uce_sd_debias.py:110-112
            break
        st.step(ds)
    out = st.current.data.cpu()
    ex, ref = c.w_exact64(), c.w_ref32()
    n_i = [w.shape[0] for w in ref]
    # regroup: replica r of every module, in module order = one copy of the fixture
    regrouped = torch.cat([out[offs[i] + r * n_i[i]: offs[i] + (r + 1) * n_i[i]] for r in range(tile) for i in range(len(ws))])
    _check_replicas(regrouped, torch.cat(ex), torch.cat(ref), tile)


@pytest.mark.parametrize("name", CLI_CASES)
def test_UCE_dropin_matches_reference_cli(H, name, tmp_path):
    """UCE() on the same fake pipe the reference CLI ran on: same keys, shapes and weights."""
    from safetensors.torch import load_file
    from uce_amd import edit as E
    from uce_amd import cli
    c = Case(name)
    m = c.meta
    seed = {"cli_erase_art_expand": 21, "cli_erase_object_default": 22, "cli_erase_object_expand_guided": 23}[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    unet = fakepipe.build_unet(O.sd14_module_table(), 768, rng)
    pipe = fakepipe.FakePipe(unet, 768)
    args = cli.parse_erase_args(m["argv"] + ["--save_dir", str(tmp_path), "--exp_name", name])
    job = cli.erase_job_from_args(args)
    E.UCE(pipe, job.edit_concepts, job.guide_concepts, job.preserve_concepts, job.erase_scale, job.preserve_scale,
          job.lamb, job.save_dir, job.exp_name, device="cuda:0")
    state = load_file(str(tmp_path / (name + ".safetensors")))
    assert sorted(state) == sorted(m["st_keys"])
    assert pipe.encode_calls == m["encode_calls"]
    # the arbiter of the three-way protocol: the same formula in float64 on the rows the fixture
    # keeps, from the embeddings the (deterministic) fake pipe returns for the job's concept lists
    emb = E.last_token_embeddings(pipe, list(job.edit_concepts) + list(job.guide_concepts) + list(job.preserve_concepts), "cpu")
    row = lambda names: [emb[x][None].cpu() for x in names]
    olds = [c.t(f"W_old_{i}") for i in range(len(m["modules"]))]
    exact = O.uce_edit_exact64(olds, row(job.edit_concepts), row(job.guide_concepts), row(job.preserve_concepts),
                               job.erase_scale, job.preserve_scale, job.lamb)
    for i, (n, shp) in enumerate(zip(m["modules"], m["shapes"])):
        got = state[n + ".weight"]
        assert list(got.shape) == shp and got.dtype == torch.float32
        ref = c.t(f"W_ref32_{i}")
        assert torch.equal(olds[i], unet.get_submodule(n).weight[: olds[i].shape[0]])
        eps_ref = O.rel_fro(ref, exact[i])                     # the reference's own fp32 error on these rows
        assert O.rel_fro(got[: ref.shape[0]], exact[i]) < EPS_BUILD
        assert O.rel_fro(got[: ref.shape[0]], ref) < max(1e-4, 1.5 * eps_ref)


# ------------------------------------------------------------------------------------ full size

@pytest.mark.parametrize("N_e,N_p", [(50, 0), (1000, 500)])
def test_full_size_properties(H, N_e, N_p):
    """BASELINE configs 2 and 3 at SD-1.4's full 24960 x 768 slab: size-independent properties."""
    d, rows_ = 768, 24960
    N = N_e + N_p
    Call = O.clip_like_embeddings(N + 1, d, seed=N)
    C, g = Call[:N], Call[N]
    G = np.repeat(g[None], N_e, axis=0)
    s = np.ones(N, dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(N))
    W = O.linear_default_weight(rows_, d, rng)
    Cd, Gd, sd, Wd = _dev(C), _dev(G), _dev(s), _dev(W)
    out = H.edit(Cd, Gd, sd, 0.5, Wd, check=True)
    # (1) against an fp64 evaluation on the GPU by torch (independent of our kernels)
    C64, W64 = Cd.double(), Wd.double()
    A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda:0") + C64.T @ C64
    Dm = (Gd - Cd[:N_e]).double()
    Delta = torch.linalg.solve(A, C64[:N_e].T @ Dm).T
    want = W64 + W64 @ Delta
    assert O.rel_fro(out.cpu(), want.cpu()) < EPS_BUILD
    # (2) both algorithms agree
    if N < d:
        out_p = H.edit(Cd, Gd, sd, 0.5, Wd, algo=L.ALGO_PRIMAL, check=True)
        assert O.rel_fro(out_p.cpu(), out.cpu()) < 2e-6
    # (3) linearity in W: edit(2 W) = 2 edit(W) exactly-ish, and row independence
    out2 = H.edit(Cd, Gd, sd, 0.5, (2 * Wd).contiguous(), check=True)
    assert O.rel_fro(out2.cpu(), (2 * out).cpu()) < 1e-6
    sub = H.edit(Cd, Gd, sd, 0.5, Wd[1000:1100].contiguous(), check=True)
    assert O.rel_fro(sub.cpu(), out[1000:1100].cpu()) < 1e-6
    # (4) what the edit is FOR: erased concepts map near the guide's output, preserved ones stay
    v_new = out.double() @ C64[:N_e].T
    v_tgt = W64 @ Gd.double().T
    v_old = W64 @ C64[:N_e].T
    # (1000 concepts cannot all be moved exactly in a 768-d space: the bound is looser there)
    assert float((v_new - v_tgt).norm() / (v_old - v_tgt).norm()) < (0.2 if N < d else 0.5)
    if N_p:
        p_new = out.double() @ C64[N_e:].T
        p_old = W64 @ C64[N_e:].T
        assert float((p_new - p_old).norm() / p_old.norm()) < 0.5     # 1500 constraints in 768 dims: soft
    # (5) bit-repeatable
    again = H.edit(Cd, Gd, sd, 0.5, Wd, check=True)
    assert torch.equal(again, out)


def test_edit_edge_cases(H):
    from uce_amd import edit as E
    d = 128
    C, G, s = _synthetic(4, 2, d, seed=11)
    rng = np.random.Generator(np.random.PCG64(3))
    W = _dev(O.linear_default_weight(40, d, rng))
    slab = E.WeightSlab(["m.attn2.to_k"], [0], [40], W)
    # preserve-only edit and all-zero scales leave the weights untouched
    out = E.edit_slab(H, slab, _dev(C), None, _dev(s), 0.5)
    assert torch.equal(out.data, W)
    out = E.edit_slab(H, slab, _dev(C), _dev(G), _dev(np.zeros(4, np.float32)), 0.5)
    assert torch.equal(out.data, W)
    # a single row, a single concept
    one = H.edit(_dev(C[:1]), _dev(G[:1]), _dev(s[:1]), 0.5, W[:1].contiguous(), check=True)
    c64, g64, w64 = C[0].astype(np.float64), G[0].astype(np.float64), W[:1].cpu().double().numpy()
    A = 0.5 * np.eye(d) + s[0] * np.outer(c64, c64)
    want = (0.5 * w64 + s[0] * np.outer(w64 @ g64, c64)) @ np.linalg.inv(A)
    assert O.rel_fro(one.cpu(), want) < 1e-6
    # invalid arguments are rejected, not crashed on
    with pytest.raises(L.UceError):
        H.edit(_dev(C), _dev(G), _dev(s), 0.5, W, out=W)


@pytest.mark.parametrize("Ne,d,rows_", [(50, 768, 24960), (33, 768, 1500), (64, 1024, 2000), (100, 768, 3000),
                                        (36, 2048, 1300), (200, 768, 1111),
                                        (160, 768, 1500), (130, 1024, 1100), (70, 2048, 1200)])   # half-filled last 128-concept batch
def test_lowrank_project_and_update(H, Ne, d, rows_):
    """The two-kernel low-rank apply (projection, then update: what uce_edit launches) against fp64."""
    rng = np.random.Generator(np.random.PCG64(Ne + d))
    W = O.linear_default_weight(rows_, d, rng)
    Dm = rng.standard_normal((Ne, d)).astype(np.float32)
    R = (rng.standard_normal((Ne, d)) * 0.01).astype(np.float32)
    Wd, Dd, Rd = _dev(W), _dev(Dm), _dev(R)
    T = H.lowrank_project(Wd, Dd)
    T64 = W.astype(np.float64) @ Dm.astype(np.float64).T
    assert T.shape == (rows_, (Ne + 63) // 64 * 64)
    assert O.rel_fro(T[:, :Ne].cpu(), T64) < 1e-6
    assert float(T[:, Ne:].abs().max()) == 0.0 if T.shape[1] > Ne else True
    out = H.lowrank_update(Wd, T, Rd)
    want = W.astype(np.float64) + T64 @ R.astype(np.float64)
    assert O.rel_fro(out.cpu(), want) < 1e-6
    # and the fused single-kernel form agrees
    fused = H.apply_lowrank(Wd, Dd, Rd)
    assert O.rel_fro(fused.cpu(), want) < 1e-6


@pytest.mark.parametrize("N_e,N_p,d,rows_", [(40, 10, 1024, 4096), (36, 4, 2048, 2600), (120, 30, 768, 5000),
                                             (65, 0, 768, 1024), (100, 0, 768, 24960), (100, 28, 768, 2048),
                                             (60, 30, 1024, 2048), (70, 50, 2048, 1300), (128, 64, 768, 1500),
                                             (150, 20, 768, 1500), (180, 100, 1024, 1100),
                                             (100, 100, 1024, 2048), (64, 100, 2048, 1300), (20, 500, 768, 1024)])
def test_edit_two_stream_path_other_widths(H, N_e, N_p, d, rows_):
    """uce_edit's project / solve / update path (rows >= 1024) vs torch fp64 on the GPU, run twice back to back
    (the handle's workspace and the riders' ticket word are re-used): N <= 64 and 64 < N <= 128 take the
    Gram + Cholesky rider blocks inside the projection launch (one and three system tiles), larger N (with at most 128 edit
    concepts) the persistent Cholesky hosted in the projection launch's first workgroups; SD-1.x / SD-2.x / SDXL widths;
    non-uniform scales."""
    N = N_e + N_p
    Call = O.clip_like_embeddings(N + 1, d, seed=N + d)
    C, G = Call[:N], np.repeat(Call[N:N + 1], N_e, axis=0)
    s = (0.5 + np.random.Generator(np.random.PCG64(7 * N + d)).random(N)).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(N))
    Cd, Gd, sd = _dev(C), _dev(G), _dev(s)
    for rep in range(2):
        W = O.linear_default_weight(rows_, d, rng)
        Wd = _dev(W)
        out = H.edit(Cd, Gd, sd, 0.5, Wd, check=True)
        C64, W64 = Cd.double(), Wd.double()
        s64 = sd.double()
        A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda:0") + C64.T @ (s64[:, None] * C64)
        Delta = torch.linalg.solve(A, (s64[:N_e, None] * C64[:N_e]).T @ (Gd - Cd[:N_e]).double()).T
        assert O.rel_fro(out.cpu(), (W64 + W64 @ Delta).cpu()) < EPS_BUILD


@pytest.mark.parametrize("N_e,N_p,d,rows_", [
    (900, 700, 192, 700),      # 3 diagonal blocks: the smallest system the persistent Cholesky (and its riders) takes; 1600
                               # concepts against 6 lower tiles: the concept split of A is capped by the slab workspace
    (300, 100, 256, 1300),     # 4 blocks, rows not a multiple of the 320-row apply tile
    (600, 500, 1024, 2500),    # 16 blocks: the launch has no room for riders - the apply splits W_old itself, one Gram launch
    (2200, 100, 192, 700),     # more than 2048 edit concepts: Bt is too long a job for the riders - the Gram launch computes it
    (0, 300, 256, 640),        # nothing to edit: Bt = 0, W_new = W_old (I + 0)
    (70, 10, 2048, 1000),      # primal forced below d: 32 blocks take the launch chain
])
def test_primal_edit_other_widths(H, N_e, N_p, d, rows_):
    """uce_edit's primal path at every way it can be put together: riders (d = 192 ... 960), no room for riders (d = 1024),
    the launch chain (d = 2048), no edit concepts."""
    C, G, s = _synthetic(N_e + N_p, N_e, d, seed=N_e + d)
    rng = np.random.Generator(np.random.PCG64(d))
    W = O.linear_default_weight(rows_, d, rng)
    Cd, sd, Wd = _dev(C), _dev(s), _dev(W)
    Gd = _dev(G) if N_e else None
    out = H.edit(Cd, Gd, sd, 0.5, Wd, algo=L.ALGO_PRIMAL, check=True)
    if N_e == 0:
        # W (I + 0): the two f16 terms of a weight carry 22 of its 24 significand bits - one fp32 rounding, not a copy
        assert O.rel_fro(out.cpu(), W.astype(np.float64)) < 1e-7
        return
    _, _, DTe = _exact(C, G, s, 0.5)
    want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
    assert O.rel_fro(out.cpu(), want) < EPS_BUILD


@pytest.mark.parametrize("N_e,N_p,d,neg", [(40, 20, 768, "scales"), (300, 700, 768, "scales"), (30, 30, 1024, "lamb")])
def test_edit_slab_indefinite_system_matches_the_lu_solve(H, N_e, N_p, d, neg):
    """Negative scales / lamb <= 0 (the reference's `torch.inverse` takes them): edit_slab's general form (uce_gram ->
    uce_solve_general, the library's own f64 LU -> uce_apply) against numpy's LU solve in float64."""
    from uce_amd import edit as E
    C, G, s = _synthetic(N_e + N_p, N_e, d, seed=N_e + 3)
    lamb = 0.5
    if neg == "scales":
        s = s.copy()
        s[::3] *= -0.2                                                  # every third concept pushes the other way
    else:
        lamb = -0.05
    rng = np.random.Generator(np.random.PCG64(d + N_e))
    W = O.linear_default_weight(1500, d, rng)
    A, Bt, DTe = _exact(C, G, s, lamb)
    w = np.linalg.eigvalsh(A)
    assert w[0] < 0 < w[-1]                                             # really indefinite
    cond = float(np.abs(w).max() / np.abs(w).min())
    slab = E.WeightSlab(["m"], [0], [W.shape[0]], _dev(W))
    out = E.edit_slab(H, slab, _dev(C), _dev(G), _dev(s), lamb).data.cpu()
    want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
    assert O.rel_fro(out, want) < max(2e-6, 1e-14 * cond), cond


@pytest.mark.parametrize("n,m", [(64, 64), (200, 72), (768, 768), (1024, 130)])
def test_general_solve_is_the_library_s_own_lu_and_matches_numpy(H, n, m):
    """uce_solve_general (Gaussian elimination with partial pivoting in f64, csrc/uce_lu.hip - no vendor solver on this edge path):
    a symmetric indefinite matrix with a zero on its diagonal (no pivoting = division by zero), against numpy's LU solve; torch's
    linalg is made unreachable for the call; a singular matrix raises EDOM."""
    rng = np.random.Generator(np.random.PCG64(n + m))
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.concatenate([np.linspace(-3.0, -0.05, n // 3), np.linspace(0.02, 5.0, n - n // 3)])
    A = (Q * ev) @ Q.T
    A = 0.5 * (A + A.T)
    P = np.eye(n)[rng.permutation(n)]
    A = P @ A @ P.T
    A[0, :] -= A[0, 0] * np.eye(n)[0]                                    # a zero leading entry: the first pivot MUST come from below
    A[:, 0] = A[0, :]
    B = rng.standard_normal((n, m))
    want = np.linalg.solve(A, B)
    orig = torch.linalg.solve

    def no_library(*a, **k):
        raise AssertionError("torch.linalg.solve reached from the product path")
    torch.linalg.solve = no_library
    try:
        X = H.solve_general(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda())
    finally:
        torch.linalg.solve = orig
    cond = np.linalg.cond(A)
    assert O.rel_fro(X.cpu(), want) < max(3e-7, 1e-13 * cond), cond     # (the result is rounded to f32 once)
    S = np.ones((n, n))                                                  # rank one: singular
    with pytest.raises(L.UceError) as ei:
        H.solve_general(torch.from_numpy(S).cuda(), torch.from_numpy(B).cuda())
    assert ei.value.code == L.EDOM
    X2 = H.solve_general(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda())          # the handle recovers
    assert torch.equal(X, X2)                                            # bit-repeatable (fixed pivot order, no atomics)


@pytest.mark.parametrize("env,value,N_e,N_p", [("UCE_PROJECT_LA", "0", 100, 80), ("UCE_SPLIT_MAX_NE", "256", 200, 60),
                                               ("UCE_POTRF_RIDER_CUS", "0", 300, 600), ("UCE_EDIT_RESIDENT", "0", 50, 0),
                                               ("UCE_EDIT_RESIDENT", "0", 70, 30)])
def test_edit_forms_behind_the_switches(env, value, N_e, N_p):
    """The forms uce_edit no longer takes by default stay correct behind their switches: the dual system's Cholesky in a
    launch of its own in front of the projection, the two-pass project + update form for 129 ... 256 edit concepts, the
    primal path without rider jobs.  (The one-launch forms of the <= 128-concept edit - measured 2-4 us behind the projection +
    update launch pair in round 4 - are retired: tools/ubench/retired/lowrank_fused.hip, HISTORY.md.)"""
    d, rows_ = 768, 2500
    C, G, s = _synthetic(N_e + N_p, N_e, d, seed=N_e)
    rng = np.random.Generator(np.random.PCG64(N_p))
    W = O.linear_default_weight(rows_, d, rng)
    _, _, DTe = _exact(C, G, s, 0.5)
    Hv = _handle_with(env, value)
    try:
        out = Hv.edit(_dev(C), _dev(G), _dev(s), 0.5, _dev(W), check=True).cpu()
    finally:
        Hv.close()
    want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
    assert O.rel_fro(out, want) < EPS_BUILD


def test_resident_edit_on_changing_inputs_matches_fp64_and_the_two_launch_form():
    """The one-launch register-resident edit (d = 768, N <= 128) publishes its D / R operands as MFMA fragments in handle-owned
    buffers that every launch rewrites and other workgroups of the SAME launch read back: consecutive edits with DIFFERENT concepts,
    targets, weights, row counts and system sizes on one handle (a stale fragment, scale or hand-off word of the previous launch
    would surface here, not in a loop over one input), each against fp64 and against the projection + update launch pair."""
    d = 768
    H1 = _handle_with("UCE_EDIT_RESIDENT", "1")
    H0 = _handle_with("UCE_EDIT_RESIDENT", "0")
    try:
        for i, (N_e, N_p, rows_) in enumerate([(50, 0, 24960), (2, 3, 1056), (64, 0, 3000), (100, 0, 24960), (7, 40, 1025),
                                               (50, 0, 24960), (90, 38, 5000), (1, 0, 1024)]):
            C, G, s = _synthetic(N_e + N_p, N_e, d, seed=100 + i)
            W = O.linear_default_weight(rows_, d, np.random.Generator(np.random.PCG64(200 + i)))
            _, _, DTe = _exact(C, G, s, 0.5)
            want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
            a = H1.edit(_dev(C), _dev(G), _dev(s), 0.5, _dev(W), check=True)
            a2 = H1.edit(_dev(C), _dev(G), _dev(s), 0.5, _dev(W), check=True)
            b = H0.edit(_dev(C), _dev(G), _dev(s), 0.5, _dev(W), check=True)
            assert torch.equal(a, a2), (i, "not bit-repeatable")
            assert O.rel_fro(a.cpu(), want) < EPS_BUILD, i
            assert O.rel_fro(a.cpu(), b.cpu().double()) < 2e-6, i
    finally:
        H1.close()
        H0.close()


def test_resident_edit_at_every_sub_tile_boundary_of_the_system():
    """The Gram of the one-launch edit is built by one rider per lower 16 x 16 sub-tile (two per rider for two-block systems), the D / R
    fragments in tiles of 16 concepts: system sizes on both sides of every multiple of 16 up to 128, erase-only and mixed, against fp64."""
    d, rows_ = 768, 1136
    H1 = _handle_with("UCE_EDIT_RESIDENT", "1")
    try:
        W = O.linear_default_weight(rows_, d, np.random.Generator(np.random.PCG64(77)))
        Wd = _dev(W)
        for N in (15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96, 97, 111, 112, 113, 127, 128):
            for N_e in (N, max(1, N // 3)):
                C, G, s = _synthetic(N, N_e, d, seed=1000 + N)
                _, _, DTe = _exact(C, G, s, 0.5)
                want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
                a = H1.edit(_dev(C), _dev(G), _dev(s), 0.5, Wd, check=True)
                assert O.rel_fro(a.cpu(), want) < EPS_BUILD, (N, N_e)
    finally:
        H1.close()


@pytest.mark.parametrize("N_e,N_p,d,rows_", [(400, 300, 768, 3000), (300, 20, 768, 700), (500, 100, 1024, 1500)])
def test_dual_edit_beyond_256_edit_concepts(H, N_e, N_p, d, rows_):
    """N < d with more than 256 edit concepts: dual system -> Cholesky (the persistent launch, carrying the f16 split of
    W_old as a rider job where it has room) -> R -> Delta = Dm^T R -> the dense apply; twice on the same handle."""
    C, G, s = _synthetic(N_e + N_p, N_e, d, seed=N_e + N_p)
    rng = np.random.Generator(np.random.PCG64(N_e))
    _, _, DTe = _exact(C, G, s, 0.5)
    Cd, Gd, sd = _dev(C), _dev(G), _dev(s)
    for rep in range(2):
        W = O.linear_default_weight(rows_, d, rng)
        out = H.edit(Cd, Gd, sd, 0.5, _dev(W), check=True)
        want = W.astype(np.float64) + W.astype(np.float64) @ DTe.T
        assert O.rel_fro(out.cpu(), want) < EPS_BUILD


@pytest.mark.parametrize("N_e,N_p", [(1, 0), (2, 1), (4, 0), (3, 5), (16, 0), (10, 7), (20, 0), (31, 0), (32, 0), (20, 13),
                                     (47, 0), (48, 0), (40, 9), (63, 0), (64, 0), (66, 0), (96, 0), (127, 0)])
def test_edit_every_segment_of_the_64x64_block(H, N_e, N_p):
    """The in-launch Cholesky inverse runs ceil(N / 4) of its 16 elimination iterations in up to three segments (its
    tile map changes at iteration 4 and 8, identity padding beyond N): every boundary of that schedule, for one- and
    two-block systems, against torch fp64; run twice (ticket / sequence words re-used)."""
    N, d, rows_ = N_e + N_p, 768, 1536
    Call = O.clip_like_embeddings(N + 1, d, seed=3 * N + 1)
    C, G = Call[:N], np.repeat(Call[N:N + 1], N_e, axis=0)
    s = (0.5 + np.random.Generator(np.random.PCG64(N)).random(N)).astype(np.float32)
    Cd, Gd, sd = _dev(C), _dev(G), _dev(s)
    Wd = _dev(O.linear_default_weight(rows_, d, np.random.Generator(np.random.PCG64(N + 9))))
    C64, W64, s64 = Cd.double(), Wd.double(), sd.double()
    A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda:0") + C64.T @ (s64[:, None] * C64)
    Delta = torch.linalg.solve(A, (s64[:N_e, None] * C64[:N_e]).T @ (Gd - Cd[:N_e]).double()).T
    want = (W64 + W64 @ Delta).cpu()
    for _ in range(2):
        assert O.rel_fro(H.edit(Cd, Gd, sd, 0.5, Wd, check=True).cpu(), want) < EPS_BUILD


@pytest.mark.parametrize("N,bad", [(50, 0), (50, 33), (64, 63), (9, 8)])
def test_single_block_rider_reports_indefinite_system(H, N, bad):
    """A non-positive scale in a system of one 64-block: the factorising rider block reports the pivot through uce_status."""
    d = 768
    Call = O.clip_like_embeddings(N + 1, d, seed=11)
    C, G = Call[:N], np.repeat(Call[N:N + 1], N, axis=0)
    s = np.ones(N, dtype=np.float32)
    s[bad] = -1.0
    W = _dev(O.linear_default_weight(2048, d, np.random.Generator(np.random.PCG64(2))))
    with pytest.raises(L.UceError):
        H.edit(_dev(C), _dev(G), _dev(s), 0.5, W, check=True)
    s[bad] = 1.0
    assert torch.isfinite(H.edit(_dev(C), _dev(G), _dev(s), 0.5, W, check=True)).all()


def test_rider_path_reports_indefinite_system(H):
    """A non-positive scale in the SECOND 64-block of a 100-concept dual system: the rider blocks' in-launch
    factorisation must flag it through uce_status (the reference would return an LU garbage inverse)."""
    N, d = 100, 768
    Call = O.clip_like_embeddings(N + 1, d, seed=5)
    C, G = Call[:N], np.repeat(Call[N:N + 1], N, axis=0)
    s = np.ones(N, dtype=np.float32)
    s[70] = -1.0
    W = _dev(O.linear_default_weight(2048, d, np.random.Generator(np.random.PCG64(1))))
    with pytest.raises(L.UceError):
        H.edit(_dev(C), _dev(G), _dev(s), 0.5, W, check=True)
    s[70] = 1.0                                                  # and the handle recovers
    out = H.edit(_dev(C), _dev(G), _dev(s), 0.5, W, check=True)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("N,d", [(20, 64), (100, 64), (300, 128)])
def test_flux_bias_direction_both_forms(H, N, d):
    """u = A^-1 sum_i s_i c_i of the biased-Linear variant: dual form (N < d) and primal form (N >= d, through
    uce_gram + uce_solve_delta) against torch fp64."""
    from uce_amd import flux
    C = _dev(O.clip_like_embeddings(N, d, seed=N + d))
    s = _dev((0.5 + np.random.Generator(np.random.PCG64(N)).random(N)).astype(np.float32))
    u = flux.bias_direction(H, C, s, 0.5)
    C64, s64 = C.double(), s.double()
    A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda:0") + C64.T @ (s64[:, None] * C64)
    want = torch.linalg.solve(A, (C64 * s64[:, None]).sum(dim=0))
    assert u.dtype == torch.float32 and O.rel_fro(u.cpu(), want.cpu()) < 1e-5


def test_flux_variant_matches_reference_golden(H, tmp_path):
    """uce_amd.flux.UCE (biased Linear modules, T5 / pooled-CLIP embedding families) on the fakes
    the golden was generated with, against the reference's uce_flux_edit.py output."""
    from safetensors.torch import load_file
    from tests import fakepipe
    from uce_amd import flux
    c = Case("flux_n6p3")
    m = c.meta
    rng = np.random.Generator(np.random.PCG64(21))
    tr = fakepipe.build_flux_transformer(24, rng)
    mods = dict(tr.named_modules())
    for i, n in enumerate(m["modules"]):
        assert torch.equal(mods[n].weight.detach(), c.t(f"W_old_{i}")) and torch.equal(mods[n].bias.detach(), c.t(f"b_{i}"))
    text = fakepipe.FakeFluxTextPipe()
    state, path = flux.UCE("black-forest-labs/FLUX.1-schnell", m["edit"], m["guide"], m["preserve"], m["erase_scale"],
                           m["preserve_scale"], m["lamb"], str(tmp_path), "flux", torch.float32, "cuda:0", 256,
                           load_transformer=lambda: fakepipe.FakeFluxTransformerPipe(tr), load_text=lambda: text)
    assert text.encode_calls == m["encode_calls"]
    saved = load_file(path)
    assert sorted(saved) == sorted(n + ".weight" for n in m["modules"])
    for i, n in enumerate(m["modules"]):
        ref, ex = c.t(f"W_ref32_{i}"), c.t(f"W_exact64_{i}")
        got = saved[n + ".weight"]
        assert O.rel_fro(got, ex) < EPS_BUILD
        assert O.rel_fro(got, ref) < max(1e-4, 1.5 * O.rel_fro(ref, ex))


def test_hidream_variant_matches_reference_golden(H, tmp_path):
    """uce_amd.hidream.UCE (one embedding family per caption projection) on the fakes the golden
    was generated with, against the reference's uce_hidream_edit.py output."""
    from safetensors.torch import load_file
    from tests import fakepipe
    from uce_amd import hidream
    c = Case("hidream_n4p2")
    m = c.meta
    rng = np.random.Generator(np.random.PCG64(22))
    tr = fakepipe.build_hidream_transformer(16, m["llama_layers"], rng)
    mods = dict(tr.named_modules())
    for i, n in enumerate(m["modules"]):
        assert torch.equal(mods[n].weight.detach(), c.t(f"W_old_{i}"))
    text = fakepipe.FakeHiDreamTextPipe(fakepipe.FakeHiDreamTokenizer(131072))
    state, path = hidream.UCE("HiDream-ai/HiDream-I1-Full", m["edit"], m["guide"], m["preserve"], m["erase_scale"],
                              m["preserve_scale"], m["lamb"], str(tmp_path), "hd", torch.float32, "cuda:0", 128,
                              load_transformer=lambda: fakepipe.FakeFluxTransformerPipe(tr), load_llama=lambda: text,
                              load_t5=lambda: text)
    saved = load_file(path)
    assert sorted(saved) == sorted(n + ".weight" for n in m["modules"])
    for i, n in enumerate(m["modules"]):
        ref, ex = c.t(f"W_ref32_{i}"), c.t(f"W_exact64_{i}")
        got = saved[n + ".weight"]
        assert O.rel_fro(got, ex) < EPS_BUILD
        assert O.rel_fro(got, ref) < max(1e-4, 1.5 * O.rel_fro(ref, ex))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_gather_last_token_matches_indexing(H, dtype):
    """uce_gather_last_token == hidden[i, idx[i], :].float() bit for bit (uce_sd_erase.py:41 per string)."""
    g = torch.Generator().manual_seed(5)
    hidden = torch.randn(9, 77, 768, generator=g).to(dtype).cuda()
    idx = torch.tensor([0, 1, 2, 5, 20, 40, 74, 75, 76])
    got = H.gather_last_token(hidden, idx)
    want = hidden[torch.arange(9, device="cuda"), idx.cuda(), :].float()
    assert got.dtype == torch.float32 and torch.equal(got, want)


def test_batched_embeddings_on_gpu_match_the_reference_call_pattern():
    """--embed_batch (one text-encoder forward per batch + the gather kernel) gives the rows the
    reference's one-string-per-call loop gives (uce_sd_erase.py:25-42), on the synthetic CLIP text encoder; includes
    '' (BOS index), a > 75-token string and duplicates."""
    from uce_amd import edit as E
    from uce_amd.sd import pipeline as sdp
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.float32, "cuda:0", synthetic=True, vae=False)
    prompts = ["Van Gogh", "art", "", "Van Gogh", "a photo of a dog", " ".join(f"w{i}" for i in range(90)), "Monet"]
    one = E.last_token_embeddings(pipe, prompts, "cuda:0")
    calls = {"n": 0}
    orig = E.UceHandle.gather_last_token

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)

    E.UceHandle.gather_last_token = counted
    try:
        bat = E.last_token_embeddings(pipe, prompts, "cuda:0", batch_size=4)
    finally:
        E.UceHandle.gather_last_token = orig
    assert calls["n"] == 2                                   # 6 unique strings in batches of 4
    assert list(one) == list(bat) and len(one) == 6
    for k in one:
        assert one[k].dtype == torch.float32 and one[k].shape == (768,)
        assert float((one[k] - bat[k]).norm() / one[k].norm()) < 1e-5, k


@pytest.mark.parametrize("rows_scale,d", [(1.0, 768), (0.013, 768), (0.05, 1000)])
def test_artifact_streamed_from_the_device_has_the_bytes_of_save_file(rows_scale, d, tmp_path, monkeypatch):
    """save_uce_state streams the slab through two pinned buffers under the file writes (edit._stream_to_file): the file must hold
    exactly what safetensors' own save_file writes for the same tensors - at SD-1.4's size (ten chunks), at a slab smaller than one
    chunk, and with a chunk size that does not divide the slab."""
    from safetensors.torch import load_file, save_file
    from uce_amd import edit as E
    table = [(n, max(8, int(o * rows_scale) // 8 * 8)) for n, o in O.sd14_module_table()]
    g = torch.Generator().manual_seed(5)
    data = torch.randn(sum(o for _, o in table), d, generator=g).cuda()
    offs = np.cumsum([0] + [o for _, o in table])[:-1]
    slab = E.WeightSlab([n for n, _ in table], [int(x) for x in offs], [o for _, o in table], data)
    if d == 1000:
        monkeypatch.setattr(E, "SAVE_CHUNK_BYTES", 1_000_003 // 4 * 4)
        E._save_pinned.clear()
    path = E.save_uce_state(slab, str(tmp_path), "a")
    E._save_pinned.clear()
    host = {n + ".weight": data[o:o + r].cpu() for n, o, r in zip(slab.names, slab.offsets, slab.rows)}
    save_file(host, str(tmp_path / "b.safetensors"))
    a, b = load_file(path), load_file(str(tmp_path / "b.safetensors"))
    assert list(a) == list(b) or sorted(a) == sorted(b)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    raw = open(path, "rb").read()
    n = int.from_bytes(raw[:8], "little")
    assert raw[8 + n:] == bytes(memoryview(data.cpu().numpy()).cast("B"))
