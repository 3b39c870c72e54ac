"""CPU: host-side logic of the drop-in (module discovery, slab packing, concept matrices,
artifact format) - no kernels involved."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import uce_oracle as O
from tests import fakepipe
from tests.golden_io import Case, DEBIAS_ALIAS_CASES, DEBIAS_CASES, keyed_embeds
from uce_amd import REPO_ROOT
from uce_amd import edit as E


def _pipe(table, d=64, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    unet = fakepipe.build_unet(table, d, rng)
    return fakepipe.FakePipe(unet, d)


def test_module_discovery_matches_reference_topology():
    c = Case("cli_erase_art_expand")
    pipe = _pipe([(n, s[0]) for n, s in zip(c.meta["modules"], c.meta["shapes"])], d=768)
    mods = E.collect_uce_modules(pipe.unet)
    assert [n for n, _ in mods] == c.meta["modules"]
    assert [list(m.weight.shape) for _, m in mods] == c.meta["shapes"]
    assert len(mods) == 32


def test_slab_roundtrip_and_artifact(tmp_path):
    table = O.sd14_module_table()[:6]
    pipe = _pipe(table, d=64)
    mods = E.collect_uce_modules(pipe.unet)
    slab = E.WeightSlab.from_modules(mods, "cpu")
    assert slab.data.shape == (sum(o for _, o in table), 64)
    for (n, m), v in zip(mods, slab.views()):
        assert torch.equal(v, m.weight)
    path = E.save_uce_state(slab, str(tmp_path), "t")
    raw = open(path, "rb").read()
    hdr = json.loads(raw[8:8 + int.from_bytes(raw[:8], "little")])
    keys = [k for k in hdr if k != "__metadata__"]
    assert sorted(keys) == sorted(n + ".weight" for n, _ in table)
    assert {hdr[k]["dtype"] for k in keys} == {"F32"}
    from safetensors.torch import load_file
    back = load_file(path)
    for (n, m) in mods:
        assert torch.equal(back[n + ".weight"], m.weight)


def test_last_token_embeddings_follow_reference_index():
    pipe = _pipe(O.sd14_module_table()[:2], d=64)
    long_prompt = " ".join(f"w{i}" for i in range(90))
    prompts = ["Van Gogh", "", "art", "Van Gogh", long_prompt]
    emb = E.last_token_embeddings(pipe, prompts, "cpu")
    assert pipe.encode_calls == ["Van Gogh", "", "art", long_prompt]      # unique strings only
    for p in set(prompts):
        assert torch.equal(emb[p], torch.from_numpy(pipe.embedding(p)))  # idx = mask.sum()-2 (0 for '', 75 truncated)


def test_concept_matrices_keep_duplicates_and_order():
    d = 8
    embeds = {k: torch.full((d,), float(i)) for i, k in enumerate(["a", "b", "g", "p"])}
    C, G, s = E.concept_matrices(embeds, ["a", "b", "a"], ["g", "g", "g"], ["p", "a"], 2.0, 0.5, "cpu")
    assert C[:, 0].tolist() == [0.0, 1.0, 0.0, 3.0, 0.0]
    assert G[:, 0].tolist() == [2.0, 2.0, 2.0]
    assert s.tolist() == [2.0, 2.0, 2.0, 0.5, 0.5]
    with pytest.raises(ValueError):
        E.concept_matrices(embeds, ["a"], ["g", "g"], [], 1, 1, "cpu")


def test_drop_zero_scale_rows():
    C = torch.arange(12.0).view(4, 3)
    G = torch.arange(6.0).view(2, 3)
    s = torch.tensor([0.0, 1.0, 0.0, 2.0])
    C2, G2, s2, ne = E.drop_zero_scale_rows(C, G, s, 2)
    assert ne == 1 and C2.shape[0] == 2 and torch.equal(G2, G[1:2]) and s2.tolist() == [1.0, 2.0]


def test_debias_alias_predicate():
    assert not E.debias_keys_alias(["Doctor", "Nurse"], ["male", "female"], ["Monet"])
    assert not E.debias_keys_alias(["Doctor"], ["male", "female"], ["male", "Monet", "Monet"])   # only EDIT concepts are written
    assert E.debias_keys_alias(["Doctor", "Nurse", "Doctor"], ["male", "female"], [])
    assert E.debias_keys_alias(["Doctor", "Nurse"], ["male", "female"], ["Nurse"])
    assert E.debias_keys_alias(["male", "Doctor"], ["male", "female"], [])


@pytest.mark.parametrize("name", DEBIAS_ALIAS_CASES + DEBIAS_CASES)
def test_debias_alias_step_follows_the_keyed_oracle(name):
    """The product's host recursion on coefficient rows (edit.debias_alias_step, what DebiasState runs when the lists alias)
    against the oracle's recursion on the embeddings themselves, iteration by iteration."""
    c = Case(name)
    m = c.meta
    keys = (m["edit"], m["debias"], m["preserve"])
    emb = keyed_embeds(c)
    uniq = []
    for k in keys[0] + keys[1] + keys[2]:
        if k not in uniq:
            uniq.append(k)
    Cu = torch.cat([emb[k] for k in uniq]).double()
    coef = {k: np.eye(len(uniq))[i] for i, k in enumerate(uniq)}
    ds = [x for x in c.arr("direction_scales") if np.abs(x).max() != 0]
    for t, D in enumerate(ds):
        Dm, moved_pres, pure = E.debias_alias_step(keys, uniq, coef, D)
        assert sorted(moved_pres + pure) == list(range(len(keys[2])))
        first = torch.cat([c.t("C_edit")] + [c.t("C_pres")[j:j + 1] for j in moved_pres]).double()
        G = first + torch.from_numpy(Dm) @ Cu
        G_e, G_p = O.debias_keyed_targets(emb, *keys, ds[:t + 1])
        want = torch.cat([G_e] + [G_p[j:j + 1] for j in moved_pres])
        assert float((G - want).abs().max()) < 1e-12 * float(want.abs().max())
        for j in pure:
            assert torch.equal(G_p[j], c.t("C_pres")[j].double())
    if name in DEBIAS_CASES:
        assert not E.debias_keys_alias(*keys) and moved_pres == []


@pytest.mark.parametrize("name", ["cli_erase_art_expand", "cli_erase_object_default", "cli_erase_object_expand_guided"])
def test_cli_concept_lists_match_reference_stdout(name):
    """Our argparse + list handling vs what the reference's __main__ printed and encoded."""
    from uce_amd import cli
    c = Case(name)
    args = cli.parse_erase_args(c.meta["argv"] + ["--save_dir", "/tmp/x", "--exp_name", name, "--device", "cpu"])
    job = cli.erase_job_from_args(args)
    ours = [ln for b in job.banner for ln in b.splitlines() if ln.strip()]
    ref = [ln for ln in c.meta["stdout_lines"] if ln.strip() and not ln.startswith("Erased concepts")]
    assert ours == ref
    seen = []
    for e in job.edit_concepts + job.guide_concepts + job.preserve_concepts:
        if e not in seen:
            seen.append(e)
    assert seen == c.meta["encode_calls"]


def test_cli_defaults_and_errors():
    from uce_amd import cli
    a = cli.parse_erase_args(["--edit_concepts", "x", "--concept_type", "object"])
    assert (a.model_id, a.device, a.erase_scale, a.preserve_scale, a.lamb, a.expand_prompts, a.save_dir, a.exp_name) == \
        ("CompVis/stable-diffusion-v1-4", "cuda:0", 1, 1, 0.5, "false", "../uce_models", None)
    j = cli.erase_job_from_args(a)
    assert j.guide_concepts == [""] and j.exp_name == "uce_test" and j.preserve_concepts == []
    a = cli.parse_erase_args(["--edit_concepts", "x;y;z", "--guide_concepts", "a;b", "--concept_type", "art"])
    with pytest.raises(Exception, match="do not match"):
        cli.erase_job_from_args(a)
    with pytest.raises(SystemExit):
        cli.parse_erase_args(["--edit_concepts", "x", "--concept_type", "unsafe"])     # argparse rejects (README is wrong)
    d = cli.parse_debias_args(["--edit_concepts", "Doctor; Nurse", "--debias_concepts", "male; female"])
    assert (d.desired_ratios, d.max_iterations, d.max_diff, d.step_size, d.num_images_per_prompt,
            d.num_inference_steps, d.guidance_scale) == ([0.5, 0.5], 30, 0.05, 0.1, 10, 20, 7.5)
    assert cli.debias_job_from_args(d).debias_concepts == ["male", "female"]
    d = cli.parse_debias_args(["--edit_concepts", "Doctor", "--debias_concepts", "a;b;c"])
    with pytest.raises(Exception, match="do not match"):
        cli.debias_job_from_args(d)
    g = cli.parse_generate_args(["--prompts_path", "p.csv"])
    assert (g.save_path, g.exp_name, g.guidance_scale, g.till_case, g.from_case, g.num_images_per_prompt,
            g.num_inference_steps, g.uce_model_path) == ("../uce_results/", "test_images", 7.5, 1000000, 0, 1, 50, None)


def test_product_synth_agrees_with_oracle_tables():
    from uce_amd import synth
    assert synth.sd14_module_table() == O.sd14_module_table()
    assert synth.sdxl_module_table() == O.sdxl_module_table()
    assert np.array_equal(synth.clip_like_embeddings(5, 64, 3), O.clip_like_embeddings(5, 64, 3))


def test_batched_embedding_extraction_matches_per_string_path():
    """one batched text-encoder forward + gather == one call per string."""
    from uce_amd.sd import pipeline as sdp
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    long_prompt = " ".join(f"w{i}" for i in range(90))
    prompts = ["Van Gogh", "", "art", "a painting by Picasso", long_prompt, "Van Gogh"]
    one = E.last_token_embeddings(pipe, prompts, "cpu")
    bat = E.last_token_embeddings(pipe, prompts, "cpu", batch_size=4)
    assert list(one) == list(bat) and len(one) == 5
    for k in one:
        assert torch.allclose(one[k], bat[k], atol=1e-5, rtol=1e-5), k


def test_bench_module_is_self_consistent():
    """bench.py imports on a CPU box and every helper its main() calls exists (the GPU legs only run on the GPU box)."""
    import ast
    import importlib.util
    path = os.path.join(REPO_ROOT, "bench.py")
    spec = importlib.util.spec_from_file_location("bench_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tree = ast.parse(open(path).read())
    called = {n.func.id for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)}
    local = {n.name for f in ast.walk(tree) if isinstance(f, ast.FunctionDef) for n in ast.walk(f)
             if isinstance(n, ast.FunctionDef) and n is not f}                       # nested closures
    local |= {a.arg for f in ast.walk(tree) if isinstance(f, (ast.FunctionDef, ast.Lambda)) for a in f.args.args}
    local |= {t.id for n in ast.walk(tree) if isinstance(n, ast.Assign) for t in n.targets if isinstance(t, ast.Name)}
    import builtins
    missing = [c for c in called if not hasattr(mod, c) and not hasattr(builtins, c) and c not in local]
    assert not missing, missing
    assert "sd14_erase50" in mod.WORKLOADS and mod.WORKLOADS["sd14_erase50"][0] == 50


def test_flux_module_predicate_embeddings_and_cli():
    """uce_flux_edit.py:25 name predicate, :44-66 embedding extraction (T5 last token + pooled), :124-171 flags."""
    from uce_amd import cli, flux
    rng = np.random.Generator(np.random.PCG64(0))
    tr = fakepipe.build_flux_transformer(8, rng)
    names = [n for n, _ in flux.collect_flux_modules(tr)]
    assert names == ["context_embedder", "time_text_embed.text_embedder.linear_1"]
    text = fakepipe.FakeFluxTextPipe()
    emb = flux.flux_embeddings(text, ["Van Gogh", "art", "Van Gogh", ""], "cpu", 256)
    assert list(emb) == ["Van Gogh", "art", ""] and text.encode_calls == ["Van Gogh", "art", ""]
    t5, pooled = emb["Van Gogh"]
    assert t5.shape == (4096,) and pooled.shape == (768,)
    assert np.allclose(t5.numpy(), text.t5_embedding("Van Gogh")) and np.allclose(pooled.numpy(), text.pooled_embedding("Van Gogh"))
    a = cli.parse_flux_args(["--edit_concepts", "Van Gogh; Picasso", "--concept_type", "art"])
    assert a.model_id == "black-forest-labs/FLUX.1-schnell" and cli.flux_max_sequence_length(a.model_id) == 256
    assert cli.flux_max_sequence_length("black-forest-labs/FLUX.1-dev") == 512
    assert cli.erase_job_from_args(a).guide_concepts == ["art", "art"]


def test_hidream_module_predicate_embeddings_and_cli():
    """uce_hidream_edit.py:31 name predicate, :59-116 per-layer Llama + T5 last-token states, :181-214 flags."""
    from uce_amd import cli, hidream
    rng = np.random.Generator(np.random.PCG64(0))
    tr = fakepipe.build_hidream_transformer(8, [1, 3, 4], rng)
    names = [n for n, _ in hidream.collect_hidream_modules(tr)]
    assert names == [f"caption_projection.{i}.linear" for i in range(4)]
    text = fakepipe.FakeHiDreamTextPipe(fakepipe.FakeHiDreamTokenizer(131072))
    emb = hidream.hidream_embeddings(text, text, ["Van Gogh", "art", "Van Gogh"], [1, 3, 4], "cpu", 128)
    assert list(emb) == ["Van Gogh", "art"] and text.calls == ["llama:Van Gogh", "llama:art", "t5:Van Gogh", "t5:art"]
    fam = emb["Van Gogh"]
    assert len(fam) == 4 and all(v.shape == (4096,) for v in fam)
    for v, name in zip(fam, ["llama1", "llama3", "llama4", "t5"]):
        assert np.allclose(v.numpy(), text.family_embedding("Van Gogh", name))
    a = cli.parse_hidream_args(["--edit_concepts", "Van Gogh", "--concept_type", "art"])
    assert a.model_id == "HiDream-ai/HiDream-I1-Full" and cli.HIDREAM_MAX_SEQUENCE_LENGTH == 128


def test_even_chunk_respects_cap_and_balances():
    from uce_amd.sd.conv_dispatch import even_chunk
    for n in range(1, 70):
        for cap in (0, 1, 2, 3, 7, 15, 22, 32, 100):
            step = even_chunk(n, cap)
            sizes = [min(step, n - i) for i in range(0, n, step)]
            assert sum(sizes) == n and max(sizes) <= max(1, min(n, cap))
            assert len(sizes) == -(-n // max(1, min(n, cap)))          # no more chunks than the cap forces
            assert max(sizes) - min(sizes) <= len(sizes)                 # evenly sized: no small tail
    assert even_chunk(32, 15) == 11 and even_chunk(32, 22) == 16 and even_chunk(32, 64) == 32


def test_edit_slab_routes_indefinite_systems_to_the_general_form():
    """Negative scales / lamb <= 0 make lamb*I + C^T S C symmetric INDEFINITE: the reference's LU inverse accepts them, the
    Cholesky path cannot.  edit_slab decides on the host from the scalars (no launch) and hands such jobs to the
    general form (uce_gram, the library's own f64 LU solve uce_solve_general, uce_apply); SPD jobs go to uce_edit as before."""
    from uce_amd import edit as E
    assert E.check_spd_inputs([1.0, 0.0, 2.5], 0.5)
    assert not E.check_spd_inputs([1.0, -0.5, 1.0], 0.5)
    assert not E.check_spd_inputs(torch.ones(3), 0.0)
    assert not E.check_spd_inputs(torch.ones(3), -1.0)

    calls = []

    class FakeHandle:
        def edit(self, C, G, s, lamb, W, algo=0, check=False):
            calls.append(("edit", tuple(C.shape)))
            return W.clone()

        def gram(self, C, G, s, lamb):
            calls.append(("gram", float(s.min())))
            d = C.shape[1]
            return torch.eye(d, dtype=torch.float64) * 2.0, torch.zeros(d, d, dtype=torch.float64)

        def solve_general(self, A, Bt):
            calls.append(("solve_general", float(A[0, 0])))
            return torch.zeros(A.shape[0], A.shape[0])

        def status(self):
            calls.append(("status",))

        def apply(self, W, DT):
            calls.append(("apply",))
            return W.clone()

    slab = E.WeightSlab(["m"], [0], [4], torch.zeros(4, 64))
    C, G = torch.randn(3, 64), torch.randn(2, 64)
    E.edit_slab(FakeHandle(), slab, C, G, torch.tensor([1.0, 1.0, 1.0]), 0.5)
    assert [c[0] for c in calls] == ["edit"]
    calls.clear()
    E.edit_slab(FakeHandle(), slab, C, G, torch.tensor([1.0, -0.5, 1.0]), 0.5)
    assert [c[0] for c in calls] == ["gram", "solve_general", "apply"]
    assert calls[0][1] == -0.5
    calls.clear()
    E.edit_slab(FakeHandle(), slab, C, G, torch.ones(3), 0.0)
    assert [c[0] for c in calls] == ["gram", "solve_general", "apply"]


def test_two_way_f16_split_model_carries_fp32_products():
    """The arithmetic of the dense apply (csrc/uce_apply_h2.hip, uce_h2split.h), restated in numpy: a row scaled by a power
    of two into [2^14, 2^15) splits into two f16 terms - 11 + 11 significand bits - with a residual of <= 2^-22 of the element
    (fp32 keeps 24 bits: 2^-24) for elements within 2^16 of the row maximum, and the three products x_l y_h + x_h y_l + x_h y_h under the exact scales
    reproduce the fp64 product of a weight matrix with I + Delta to fp32 level - also with rows 2^40 apart and a zero row."""
    rng = np.random.Generator(np.random.PCG64(11))

    def split(X):
        mx = np.abs(X).max(axis=1)
        E = np.clip((mx.astype(np.float32).view(np.uint32) >> 23).astype(np.int64), 30, 240)
        s = np.ldexp(1.0, 141 - E)[:, None]                         # row maximum -> [2^14, 2^15)
        y = (X.astype(np.float64) * s).astype(np.float32)           # exact: a power of two
        h = y.astype(np.float16)
        lo = (y - h.astype(np.float32)).astype(np.float16)          # the subtraction is exact in fp32
        return h.astype(np.float64), lo.astype(np.float64), 1.0 / s, y.astype(np.float64)

    rows, d = 96, 256
    W = (rng.uniform(-1, 1, (rows, d)) / 16).astype(np.float32)
    W *= np.ldexp(1.0, rng.integers(-20, 21, size=(rows, 1))).astype(np.float32)
    W[5] = 0.0
    h, lo, inv, y = split(W)
    big = np.abs(y) >= np.abs(y).max(axis=1, keepdims=True) * 2.0 ** -16
    resid = np.abs(h + lo - y)
    assert np.all(resid[big] <= np.abs(y[big]) * 2.0 ** -22)         # 22 significand bits
    assert np.all(resid <= np.maximum(np.abs(y) * 2.0 ** -22, 2.0 ** -25))   # below: the f16 denormal grid
    assert np.all(np.abs(h).max(axis=1) <= 2.0 ** 15)                # far inside f16's range (65504)

    DT = (rng.standard_normal((d, d)) * (0.5 / np.sqrt(d))).astype(np.float32)
    B = DT + np.eye(d, dtype=np.float32)                             # rows of (I + Delta)^T
    bh, bl, binv, _ = split(B)
    acc = lo @ bh.T + h @ bl.T + h @ bh.T                            # the MFMAs accumulate exact products
    out = acc * inv * binv.T
    want = W.astype(np.float64) @ B.astype(np.float64).T
    rown = np.linalg.norm(want, axis=1)
    ok = rown > 0
    assert float((np.linalg.norm(out - want, axis=1)[ok] / rown[ok]).max()) < 3e-7
    assert np.all(out[5] == 0.0)


def test_projection_live_tile_wave_map_covers_every_tile_once():
    """csrc/uce_lowrank2.hip:project_dispatch restated: with 1 / 2 / 3 live 16-concept tiles the eight waves of a projection
    workgroup share the live tiles - every (column tile, row tile) pair belongs to exactly one storing wave, no wave takes
    more than four row tiles (the largest instantiation), and no SIMD (waves w, w + 4) carries more than the standing
    map's MT tiles."""
    def wave_map(w, live, MT):
        if live == 1:
            c4, part, parts = 0, w, 8
        elif live == 2:
            c4, part, parts = w & 1, w >> 1, 4
        else:
            c4 = w % 3 if w < 6 else w - 6
            part = w // 3 if w < 6 else 2
            parts = 2 if c4 == 2 else 3
        base, extra = MT // parts, MT % parts
        nm = base + (1 if part < extra else 0)
        m0 = part * base + min(part, extra)
        return c4, m0, nm

    for MT in (5, 6, 7, 8):
        for live in (1, 2, 3):
            owner = {}
            per_simd = [0, 0, 0, 0]
            for w in range(8):
                c4, m0, nm = wave_map(w, live, MT)
                assert 0 <= c4 < live and 0 <= nm <= 4
                per_simd[w & 3] += nm
                for m in range(m0, m0 + nm):
                    assert (c4, m) not in owner, (MT, live, w, c4, m)
                    owner[(c4, m)] = w
            assert set(owner) == {(c, m) for c in range(live) for m in range(MT)}
            assert max(per_simd) <= MT and max(per_simd) <= -(-live * MT // 4) + 1, (MT, live, per_simd)


def test_conv_dispatch_rule_and_padded_narrow_weights():
    """Host logic of the convolution dispatch: which layers the implicit-GEMM kernels take (shape rule only - no batch-size
    rule, no library path), and the zero-padded copy of a narrow-output weight (VAE conv_out) follows in-place updates of the parameters."""
    from uce_amd.sd import conv_dispatch as E
    from uce_amd.sd import unet as U
    # every layer of SD-1.x / SDXL / their VAEs has an implicit-GEMM kernel, whatever the batch (few tiles: the split-contraction forms)
    for cin, cout in ((320, 320), (320, 640), (1920, 640), (1280, 1280), (2560, 1280), (960, 320), (128, 128), (512, 256), (128, 8)):
        assert E.conv_takes_igemm(cin, cout) and E.conv_takes_igemm(cin, cout, stride=2, residual=True)
    assert E.conv_takes_igemm(96, 256) and not E.conv_takes_igemm(96, 72)      # 64-byte k-tiles: only outputs a wide tile divides
    assert not E.conv_takes_igemm(64, 4)                             # (sd.unet pads a 3- / 4-channel output to 8)
    assert not E.conv_takes_igemm(4, 320)                            # conv_in: 4 channels (its own patch-matrix path)
    conv = torch.nn.Conv2d(128, 3, 3, padding=1)
    w8, b8 = U._padded_out_channels(conv)
    assert w8.shape == (8, 128, 3, 3) and torch.equal(w8[:3], conv.weight) and not w8[3:].any()
    assert torch.equal(b8[:3], conv.bias) and not b8[3:].any()
    assert U._padded_out_channels(conv)[0] is w8                     # cached
    # a by-name patch that writes through .data (patch_unet: no version bump) must drop the derived copy
    from uce_amd.sd import pipeline as sdp
    holder = type("P", (), {})()
    holder.unet = torch.nn.Sequential(conv)
    holder._graphs = {"k": object()}
    sdp.patch_unet(holder, {"0.weight": torch.ones_like(conv.weight), "0.bias": torch.zeros_like(conv.bias)})
    assert not hasattr(conv, "_uce_pad8") and not holder._graphs
    assert bool((U._padded_out_channels(conv)[0][:3] == 1).all())
    with torch.no_grad():
        conv.weight.mul_(2.0)
    w8b, _ = U._padded_out_channels(conv)
    assert w8b is not w8 and torch.equal(w8b[:3], conv.weight)


@pytest.mark.parametrize("TNW", [8, 10])
def test_one_wave_convolution_issue_schedule_and_lds_budget(TNW):
    """CPU restatement of k_conv3x3_w1's hand-placed issue slots (csrc/uce_conv_w1.hip): in the first k-step of a tile every
    fragment of the second k-step and every DMA of the outgoing k-tile goes out behind exactly one MFMA, no two of them behind the
    same one; the barrier of the second k-step leaves exactly one MFMA per fragment of the next tile behind it; ring, epilogue
    slabs and register budget fit the CU."""
    BN = 32 * TNW
    NMF, NFR = TNW * 8, 8 + TNW                       # MFMAs / fragment reads per k-step of 32
    NA, NB = 256 // 8 // 4, BN // 8 // 4              # DMA instructions per wave and k-tile: A image, B image
    PER = NA + NB
    frag_slots = [((f + 1) * NMF) // NFR - 1 for f in range(NFR)]
    dma_slots = [((g + 1) * NMF) // PER - 2 for g in range(PER)]
    assert sorted(set(frag_slots)) == frag_slots and 0 <= frag_slots[0] and frag_slots[-1] == NMF - 1
    assert sorted(set(dma_slots)) == dma_slots and 0 <= dma_slots[0] and dma_slots[-1] < NMF
    assert not set(frag_slots) & set(dma_slots)
    # the X fragments of the next step (needed by its first MFMAs) are requested before the W fragments
    assert frag_slots[:8] == sorted(frag_slots[:8]) and frag_slots[7] < frag_slots[8]
    # second k-step: barrier behind MFMA NMF - HOLD - 1, then one read of the next tile per remaining MFMA
    HOLD = NFR
    assert 0 < NMF - HOLD - 1 and NMF - (NMF - HOLD) == NFR
    # LDS: two stages of (256 + BN) rows x 128 B; four 64-pixel slabs of BN / 2 channels (+ 16 B per row) inside the freed ring
    stage = (256 + BN) * 128
    assert 2 * stage <= 160 * 1024
    assert 4 * 64 * (BN + 16) <= 2 * stage
    assert (64 * (BN // 16)) % 64 == 0                # whole wave passes over a slab's 16-byte pieces
    # registers: 8 x TNW accumulator tiles of 4; 64 tiles fill the AGPR file, the rest + double-buffered fragments are VGPRs
    acc_vgpr = max(0, TNW * 8 - 64) * 4
    # (+ staging coordinates: 2 per A DMA, 1 per B DMA) - at least 16 VGPRs stay for addresses and epilogue temporaries
    assert min(TNW * 8, 64) * 4 == 256 and acc_vgpr + 2 * (8 + TNW) * 4 + 2 * NA + NB <= 256 - 16


@pytest.mark.parametrize("TNW,stride,up", [(10, 1, 0), (8, 1, 0), (10, 2, 0), (8, 1, 1)])
def test_one_wave_convolution_index_arithmetic_restated(TNW, stride, up):
    """CPU restatement of k_conv3x3_w1's data movement (csrc/uce_conv_w1.hip), index for index: the LDS image an `buffer_load ... lds`
    instruction leaves (lane = (row, 16-byte piece), bank swizzle on the source piece, taps outside the image as zeros), the fragment
    a lane reads for a k-step (ONE piece offset per step, tiles at 16-row distances), the operand / result layout of
    v_mfma_f32_16x16x32 with the weight fragment as A and the pixel fragment as B, and the accumulator -> (pixel, channel) map of the
    epilogue - assembled into an output and compared with torch's convolution."""
    import numpy as np
    rng = np.random.default_rng(TNW + stride + up)
    BN, Cin, cch = 32 * TNW, 128, 2                   # two 64-channel chunks: the k-tiles walk them CHUNK-major (nine taps each)
    Ho, Wo = 16, 16                                    # output image: 256 pixels = one 256-pixel tile
    Hi, Wi = Ho * stride, Wo * stride                  # the image the taps index
    Hs, Ws = Hi >> up, Wi >> up                        # the stored image (before the fused 2x upsample)
    X = rng.standard_normal((Hs, Ws, Cin)).astype(np.float32)
    Wt = (rng.standard_normal((BN, 9 * Cin)) * 0.05).astype(np.float32)          # [Cout, tap * Cin + c]
    lanes = np.arange(64)
    acc = np.zeros((4, TNW, 8, 64, 4), np.float64)     # [wave][channel tile a][pixel tile b][lane][register]
    for kt in range(9 * cch):
        chunk, tap = kt // 9, kt % 9                   # conv_tap_inner (csrc/uce_common.h): the taps of a chunk back to back
        c0 = chunk * 64
        dy, dx = tap // 3 - 1, tap % 3 - 1
        # ---- the LDS image of this k-tile: A rows 0..255 (pixels), B rows 0..BN-1 (channels), 8 slots of 8 elements
        A = np.zeros((256, 8, 8), np.float32)
        Bm = np.zeros((BN, 8, 8), np.float32)
        r, p = lanes >> 3, lanes & 7
        for w in range(4):
            for j in range(8):                         # A DMA instructions of wave w
                R = 8 * (4 * j + w) + r
                c = p ^ ((R >> 1) & 7)
                y, x = (R // Wo) * stride + dy, (R % Wo) * stride + dx
                ok = (y >= 0) & (y < Hi) & (x >= 0) & (x < Wi)
                for ln in lanes:
                    if ok[ln]:
                        A[R[ln], p[ln]] = X[y[ln] >> up, x[ln] >> up, c0 + 8 * c[ln]:c0 + 8 * c[ln] + 8]
            for j in range(BN // 8 // 4):              # B DMA instructions
                R = 8 * (4 * j + w) + r
                c = p ^ ((R >> 1) & 7)
                for ln in lanes:
                    Bm[R[ln], p[ln]] = Wt[R[ln], tap * Cin + c0 + 8 * c[ln]:tap * Cin + c0 + 8 * c[ln] + 8]
        # ---- two k-steps of 32: fragments and MFMAs
        l16, lq = lanes & 15, lanes >> 4
        key = (l16 >> 1) & 7
        for w in range(4):
            wm, wn = w & 1, w >> 1
            for s in range(2):
                po = (4 * s + lq) ^ key                # the step's ONE piece slot per lane
                xf = np.stack([A[wm * 128 + 16 * f + l16, po] for f in range(8)])             # [8][lane][8]
                wf = np.stack([Bm[wn * (BN // 2) + 16 * f + l16, po] for f in range(TNW)])    # [TNW][lane][8]
                for a in range(TNW):
                    # A operand (wf[a]): lane -> row i = lane % 16, k = 8 (lane // 16) ..; B operand (xf[b]): lane -> column j = lane % 16
                    Am = np.zeros((16, 32))
                    Am[l16[:, None], (8 * lq)[:, None] + np.arange(8)] = wf[a]
                    for b in range(8):
                        Bk = np.zeros((32, 16))
                        Bk[(8 * lq)[:, None] + np.arange(8), l16[:, None]] = xf[b]
                        D = Am @ Bk                     # [channel i][pixel j]
                        # D layout: lane -> column j = lane % 16, rows i = 4 (lane // 16) + {0..3}
                        acc[w, a, b] += D[(4 * lq)[:, None] + np.arange(4), l16[:, None]]
    # ---- epilogue map: acc[a][b][i] = channel wn BN/2 + 16 a + 4 lq + i of pixel wm 128 + 16 b + l16
    Y = np.zeros((256, BN))
    l16, lq = lanes & 15, lanes >> 4
    for w in range(4):
        wm, wn = w & 1, w >> 1
        for a in range(TNW):
            for b in range(8):
                Y[(wm * 128 + 16 * b + l16)[:, None], (wn * (BN // 2) + 16 * a + 4 * lq)[:, None] + np.arange(4)] = acc[w, a, b]
    xin = torch.from_numpy(X).double().permute(2, 0, 1)[None]
    if up:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest")
    wref = torch.from_numpy(Wt).double().view(BN, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin, wref, None, stride=stride, padding=1)[0].permute(1, 2, 0).reshape(256, BN).numpy()
    assert np.abs(Y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_ranks_of_one_node_get_disjoint_core_blocks():
    """generate.rank_core_block: eight ranks split the cores the job may use into disjoint contiguous blocks (the NUMA-node form
    needs the GPU box's sysfs; with an unknown node the split is even over everything) and the PNG workers fit the block."""
    from uce_amd import generate as G
    allowed = list(range(0, 256))
    blocks = [G.rank_core_block(r, 8, allowed=allowed, node=-1) for r in range(8)]
    assert all(len(b) == 32 for b in blocks)
    assert sorted(c for b in blocks for c in b) == allowed
    assert all(b == list(range(b[0], b[0] + 32)) for b in blocks)
    odd = [G.rank_core_block(r, 3, allowed=range(10), node=-1) for r in range(3)]
    assert all(len(b) == 3 for b in odd) and len({c for b in odd for c in b}) == 9
    assert G.rank_core_block(0, 1, allowed=range(4), node=-1) == [0, 1, 2, 3]
    assert G.pin_rank_to_cores(0, 1) == 0                                # a single rank is left alone
    assert 0 <= G.png_worker_count(8, 8) <= 8 and G.png_worker_count(0, 1) == 0
    auto = G.png_worker_count(G.PNG_WORKERS_AUTO, 1)                     # a quarter of the cores of the rank, 1 .. 32
    assert 1 <= auto <= 32 and auto <= max(1, len(os.sched_getaffinity(0)) - 1)


def test_prefix_encoding_gives_the_states_of_the_full_causal_forward():
    """edit.last_token_embeddings (batched, own pipeline) runs CLIP's text encoder on token positions 0 .. max(idx) only: the encoder
    is causal, so those states are the full 77-position forward's - checked on transformers' CLIPTextModel at SD-1.4's REAL widths
    (12 layers x 768, seeded-random weights, fp32 on the CPU) and through last_token_embeddings on the tiny pipeline incl. the
    cases that cannot be cut ('' -> BOS at index 0 is fine; a > 75-token string needs all positions)."""
    from uce_amd.sd import pipeline as sdp
    torch.manual_seed(0)
    te = sdp.build_text_encoder(sdp.TextConfig()).eval()
    ids = torch.randint(256, 49000, (3, sdp.MAX_LEN))
    ids[:, 0] = sdp.BOS
    with torch.no_grad():
        full = te(input_ids=ids)[0]
        for n_pos in (1, 4, 9, 30):
            part = te(input_ids=ids[:, :n_pos])[0]
            assert part.shape == (3, n_pos, 768)
            assert float((part - full[:, :n_pos]).norm() / full[:, :n_pos].norm()) < 2e-6, n_pos
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    seen = []
    real = pipe.encode_prompt_prefix
    pipe.encode_prompt_prefix = lambda prompts, device, n_pos, input_ids=None: (seen.append(n_pos), real(prompts, device, n_pos, input_ids))[1]
    prompts = ["Van Gogh", "", "art", "a painting by Picasso"]
    one = E.last_token_embeddings(pipe, prompts, "cpu")                       # per string: the reference's full-length calls
    bat = E.last_token_embeddings(pipe, prompts, "cpu", batch_size=4)
    assert seen == [5]                                                         # BOS + 4 words -> index 4
    for k in one:
        assert torch.allclose(one[k], bat[k], atol=1e-5, rtol=1e-5), k
    seen.clear()
    long_prompt = " ".join(f"w{i}" for i in range(90))
    bat = E.last_token_embeddings(pipe, ["art", long_prompt], "cpu", batch_size=2)
    assert seen == [76]                                                        # truncated to 77 tokens: index 75 -> 76 positions
    assert torch.allclose(bat[long_prompt], E.last_token_embeddings(pipe, [long_prompt], "cpu")[long_prompt], atol=1e-5, rtol=1e-5)
