"""GPU: the drop-in scripts end to end on synthetic weights (no checkpoints exist on these
machines): erase CLI -> artifact -> generation with the patched U-Net (cross-attention through the
HIP kernel), and the debias driver against its golden with a scripted sampling step."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import uce_oracle as O
from tests.golden_io import Case
from uce_amd import REPO_ROOT

pytestmark = pytest.mark.gpu


def test_erase_cli_then_generate_cli(tmp_path):
    from safetensors.torch import load_file
    env = dict(os.environ)
    erase = os.path.join(REPO_ROOT, "trainscripts", "uce_sd_erase.py")
    gen = os.path.join(REPO_ROOT, "evalscripts", "generate-images-sd.py")
    r = subprocess.run([sys.executable, erase, "--model_id", "tiny-sd-test", "--synthetic_model",
                        "--edit_concepts", "Van Gogh; Picasso", "--concept_type", "art",
                        "--preserve_concepts", "Monet", "--save_dir", str(tmp_path), "--exp_name", "tiny_edit"],
                       check=True, env=env, capture_output=True, text=True, timeout=600)
    assert "Erasing: ['Van Gogh', 'Picasso']" in r.stdout and "Guiding: ['art', 'art']" in r.stdout
    assert "Model edited in" in r.stdout
    state = load_file(str(tmp_path / "tiny_edit.safetensors"))
    assert len(state) == 32 and all(k.endswith(".weight") and "attn2" in k for k in state)
    prompts = tmp_path / "p.csv"
    pd.DataFrame({"case_number": [0, 1], "prompt": ["a painting by Van Gogh", "a dog"],
                  "evaluation_seed": [7, 8]}).to_csv(prompts, index=False)
    subprocess.run([sys.executable, gen, "--model_id", "tiny-sd-test", "--synthetic_model", "--prompts_path",
                    str(prompts), "--save_path", str(tmp_path), "--exp_name", "imgs", "--uce_model_path",
                    str(tmp_path / "tiny_edit.safetensors"), "--num_inference_steps", "3"],
                   check=True, env=env, timeout=600)
    assert sorted(os.listdir(tmp_path / "imgs")) == ["0_0.png", "1_0.png"]


def test_unet_cross_attention_runs_through_hip_kernel_and_matches_fp32():
    """bf16 U-Net forward on the GPU (attn2 -> uce_xattn_fwd) vs the same weights in fp32 on the CPU."""
    from uce_amd.sd import pipeline as sdp
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False, seed=3)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g)
    ctx = torch.randn(2, 77, 64, generator=g)
    t = torch.tensor([500])
    ref = pipe.unet(x, t, ctx)
    pipe.to("cuda:0", torch.bfloat16)
    calls = {"n": 0}
    from uce_amd import edit as E
    orig = E.UceHandle.xattn

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)

    E.UceHandle.xattn = counted
    try:
        out = pipe.unet(x.cuda().bfloat16(), t.cuda(), ctx.cuda().bfloat16()).float().cpu()
    finally:
        E.UceHandle.xattn = orig
    assert calls["n"] == 16                                   # one per attn2 (16 transformer blocks)
    assert O.rel_fro(out, ref) < 5e-2                         # bf16 network vs fp32


def test_patch_unet_bf16_cast_matches_torch():
    from uce_amd import edit as E
    from uce_amd.sd import pipeline as sdp
    pipe = sdp.load_pipeline("tiny-sd-test", torch.bfloat16, "cuda:0", synthetic=True, vae=False)
    mods = E.collect_uce_modules(pipe.unet)
    g = torch.Generator().manual_seed(1)
    state = {n + ".weight": torch.randn(m.weight.shape, generator=g) for n, m in mods}
    sdp.patch_unet(pipe, state)
    for n, m in mods:
        assert torch.equal(m.weight.cpu(), state[n + ".weight"].to(torch.bfloat16))


def test_debias_driver_scripted_matches_golden(tmp_path):
    """debias.UCE with the sampling step scripted (the reference's is unseeded) on the golden's
    embeddings and weights: final artifact vs the reference's."""
    from safetensors.torch import load_file
    from tests import fakepipe
    from uce_amd import debias
    c = Case("debias_n4x2_d768")
    m = c.meta
    rng = np.random.Generator(np.random.PCG64(12))
    from tests.tools_shim import FIXTURE_TABLE
    unet = fakepipe.build_unet(FIXTURE_TABLE, 768, rng)
    pipe = fakepipe.FakePipe(unet, 768)
    scripted = iter(list(c.arr("direction_scales")))
    slab, path = debias.UCE(pipe, None, m["edit"], m["debias"], m["preserve"], m["edit_scale"], m["preserve_scale"],
                            m["lamb"], str(tmp_path), "deb", 0.05, 0.1, 10, 20, 7.5, desired_ratios=[0.5, 0.5],
                            max_iterations=5, device="cuda:0", ratios_fn=lambda **kw: next(scripted))
    state = load_file(path)
    for i, n in enumerate(m["modules"]):
        ref, ex = c.t(f"W_ref32_{i}"), c.t(f"W_exact64_{i}")
        assert torch.equal(c.t(f"W_old_{i}"), unet.get_submodule(n).weight)
        assert O.rel_fro(state[n + ".weight"], ex) < 1e-5
        assert O.rel_fro(state[n + ".weight"], ref) < max(1e-4, 1.5 * O.rel_fro(ref, ex))


def test_debias_ratio_arithmetic():
    from uce_amd import debias
    r = debias.ratios_from_labels(["male"] * 7 + ["female"] * 3, ["male", "female"], [0.5, 0.5], 0.05)
    assert np.allclose(r, [-0.2, 0.2])
    r = debias.ratios_from_labels(["male"] * 5 + ["female"] * 5, ["male", "female"], [0.52, 0.48], 0.05)
    assert np.all(r == 0)


def _tiny_gpu_pipe(vae=False, seed=3):
    from uce_amd.sd import pipeline as sdp
    return sdp.load_pipeline("tiny-sd-test", torch.bfloat16, "cuda:0", synthetic=True, vae=vae, seed=seed)


def test_hipgraph_step_matches_eager_and_sees_weight_patches():
    """The denoising evaluation replayed from a hipGraph (uce_xattn_fwd captured on the capture stream) vs the
    eager launches; an in-place patch of the attn2 weights must be visible to the captured graph."""
    from uce_amd import edit as E
    from uce_amd.sd import pipeline as sdp
    pipe = _tiny_gpu_pipe()
    g = lambda: torch.Generator().manual_seed(5)
    kw = dict(num_inference_steps=3, output_type="latent")
    pipe.use_graph = False
    eager = pipe("a photo of a dog", generator=g(), **kw).latents.float()
    pipe.use_graph = True
    graph = pipe("a photo of a dog", generator=g(), **kw).latents.float()
    assert len(pipe._graphs) == 1
    assert O.rel_fro(graph, eager) < 2e-2
    again = pipe("a photo of a dog", generator=g(), **kw).latents.float()       # replay of the cached graph
    assert len(pipe._graphs) == 1 and O.rel_fro(again, eager) < 2e-2
    other = pipe("a cat", generator=g(), **kw).latents.float()                    # new context, same graph
    assert len(pipe._graphs) == 1 and O.rel_fro(other, eager) > 1e-3
    mods = E.collect_uce_modules(pipe.unet)
    sdp.patch_unet(pipe, {n + ".weight": m.weight.float() * -2.0 for n, m in mods})
    patched_graph = pipe("a photo of a dog", generator=g(), **kw).latents.float()
    pipe.use_graph = False
    patched_eager = pipe("a photo of a dog", generator=g(), **kw).latents.float()
    assert O.rel_fro(patched_graph, patched_eager) < 2e-2
    assert O.rel_fro(patched_graph, eager) > 1e-3


def test_batched_prompts_match_single_prompts_on_gpu():
    pipe = _tiny_gpu_pipe()
    g = lambda s: torch.Generator().manual_seed(s)
    kw = dict(num_inference_steps=3, output_type="latent")
    single = torch.cat([pipe(p, generator=g(20 + i), **kw).latents for i, p in enumerate(["a dog", "a cat", "a tree"])])
    batch = pipe(["a dog", "a cat", "a tree"], generator=[g(20), g(21), g(22)], **kw).latents
    assert batch.shape == single.shape
    assert O.rel_fro(batch.float(), single.float()) < 2e-2


def test_debias_loop_with_real_sampling_on_gpu(tmp_path):
    """debias.UCE end to end (sampling through the tiny pipeline, a deterministic stand-in classifier):
    get_ratios -> cumulative drift -> re-solve, until balanced or max_iterations; artifact written."""
    from safetensors.torch import load_file
    from uce_amd import debias
    from uce_amd.sd import pipeline as sdp
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cuda:0", synthetic=True, vae=True)
    seen = []

    def classify(images, labels):
        seen.append(len(images))
        return [labels[int(np.asarray(im)[..., 0].mean() > np.asarray(im)[..., 1].mean())] for im in images]

    slab, path = debias.UCE(pipe, classify, ["doctor", "nurse"], ["male", "female"], ["a dog"], 1.0, 1.0, 0.5,
                            str(tmp_path), "deb_real", 0.05, 0.1, 4, 2, 7.5, desired_ratios=[0.5, 0.5],
                            max_iterations=2, device="cuda:0")
    assert seen and all(n == 4 for n in seen) and len(seen) % 2 == 0 and len(seen) <= 4
    state = load_file(path)
    assert len(state) == 32 and all(torch.isfinite(v).all() for v in state.values())


def test_fused_cfg_pndm_step_matches_the_torch_scheduler():
    """uce_cfg_pndm_step (guidance combine + PLMS step, one launch) against PNDMScheduler.step on the same sequence of
    U-Net outputs: every branch of the multistep formula (first step, the repeated timestep, 2 / 3 / 4 stored outputs)."""
    from uce_amd import edit as E
    from uce_amd.sd.scheduler import PNDMScheduler
    H = E.UceHandle.get("cuda:0")
    g = torch.Generator().manual_seed(0)
    a, b = PNDMScheduler(), PNDMScheduler()
    a.set_timesteps(8)
    b.set_timesteps(8)
    xa = torch.randn(2, 4, 16, 16, generator=g).cuda()
    xb = xa.to(torch.bfloat16)
    for t in a.timesteps.tolist():
        raw = torch.randn(4, 4, 16, 16, generator=g).cuda()
        raw16 = raw.to(torch.bfloat16)
        eu, ec = raw16.float().chunk(2)
        xa = a.step(eu + 7.5 * (ec - eu), t, xa)                       # fp32 reference on the bf16-rounded outputs
        xb = b.step_fused(raw16, True, 7.5, t, xb, H)
        assert xb.dtype == torch.bfloat16 and O.rel_fro(xb.float().cpu(), xa.cpu()) < 2e-2, t
    assert len(b.ets) == len(a.ets) and b.counter == a.counter


def test_fused_step_latents_match_the_unfused_path():
    pipe = _tiny_gpu_pipe()
    g = lambda: torch.Generator().manual_seed(11)
    kw = dict(num_inference_steps=6, output_type="latent")
    pipe.fused_step = True
    fused = pipe("a photo of a dog", generator=g(), **kw).latents.float()
    pipe.fused_step = False
    plain = pipe("a photo of a dog", generator=g(), **kw).latents.float()
    assert O.rel_fro(fused, plain) < 2e-2
