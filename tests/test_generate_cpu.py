"""CPU: the generation driver's host logic - scheduler schedule, row sharding, artifact patching,
and the world_size-2 (gloo) path with the rank-0 broadcast of the edited weights."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

from uce_amd import REPO_ROOT, generate
from uce_amd import edit as E
from uce_amd.sd import pipeline as sdp
from uce_amd.sd.scheduler import PNDMScheduler


def test_pndm_schedule_is_51_unet_calls_for_50_steps():
    s = PNDMScheduler()
    s.set_timesteps(50)
    t = s.timesteps.tolist()
    assert len(t) == 51 and t[0] == 981 and t[1] == 961 and t[2] == 961 and t[-1] == 1
    s.set_timesteps(20)
    assert len(s.timesteps) == 21
    # a constant-eps rollout stays finite and deterministic
    x = torch.ones(1, 4, 2, 2)
    for tt in s.timesteps.tolist():
        x = s.step(torch.full_like(x, 0.1), tt, x)
    assert torch.isfinite(x).all()


def test_select_rows_matches_reference_filter_and_shards_disjointly():
    df = pd.DataFrame({"case_number": [5, 6, 7, 8, 9, 10], "prompt": list("abcdef"), "evaluation_seed": range(6)})
    all_rows = [r.case_number for _, r in generate.select_rows(df, 6, 9, 0, 1)]
    assert all_rows == [6, 7, 8, 9]                       # from_case <= case <= till_case, inclusive
    a = [r.case_number for _, r in generate.select_rows(df, 6, 9, 0, 2)]
    b = [r.case_number for _, r in generate.select_rows(df, 6, 9, 1, 2)]
    assert sorted(a + b) == all_rows and not set(a) & set(b)


def _tiny_prompts(tmp_path, n=4):
    p = tmp_path / "prompts.csv"
    pd.DataFrame({"case_number": list(range(n)), "prompt": [f"a photo of thing {i}" for i in range(n)],
                  "evaluation_seed": [100 + i for i in range(n)]}).to_csv(p, index=False)
    return str(p)


def _tiny_artifact(tmp_path, scale=1.5):
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    slab = E.WeightSlab.from_modules(E.collect_uce_modules(pipe.unet), "cpu")
    path = E.save_uce_state(slab.like(slab.data * scale), str(tmp_path), "tiny_uce")
    return path, len(slab.names)


def test_single_process_generation_names_and_seeding(tmp_path):
    prompts = _tiny_prompts(tmp_path, 3)
    art, n = _tiny_artifact(tmp_path)
    assert n == 32
    kw = dict(model_id="tiny-sd-test", prompts_path=prompts, save_path=str(tmp_path), device="cpu",
              torch_dtype=torch.float32, num_inference_steps=3, num_images_per_prompt=2, synthetic=True)
    generate.generate_images(uce_model_path=None, exp_name="orig", **kw)
    generate.generate_images(uce_model_path=art, exp_name="edited", from_case=1, till_case=2, **kw)
    assert sorted(os.listdir(tmp_path / "orig")) == [f"{c}_{i}.png" for c in range(3) for i in range(2)]
    assert sorted(os.listdir(tmp_path / "edited")) == [f"{c}_{i}.png" for c in (1, 2) for i in range(2)]
    from PIL import Image
    a = np.asarray(Image.open(tmp_path / "orig" / "1_0.png"))
    b = np.asarray(Image.open(tmp_path / "edited" / "1_0.png"))
    assert a.shape == (64, 64, 3) and (a != b).any()        # the patch changed the model
    # same seed, same model -> same image
    generate.generate_images(uce_model_path=None, exp_name="orig2", from_case=1, till_case=1, **kw)
    assert (np.asarray(Image.open(tmp_path / "orig2" / "1_0.png")) == a).all()


def test_batched_rows_equal_row_by_row(tmp_path):
    """--batch_prompts: B rows per U-Net call, every row with its own CPU-seeded latents ->
    the same files as the row-by-row loop of generate-images-sd.py:29-46."""
    prompts = _tiny_prompts(tmp_path, 5)
    kw = dict(model_id="tiny-sd-test", uce_model_path=None, prompts_path=prompts, save_path=str(tmp_path), device="cpu",
              torch_dtype=torch.float32, num_inference_steps=3, num_images_per_prompt=2, synthetic=True, latents_only=True)
    generate.generate_images(exp_name="rows", batch_prompts=1, **kw)
    st = generate.generate_images(exp_name="batched", batch_prompts=3, **kw)     # 3 + a ragged batch of 2
    assert st["images"] == 10
    assert sorted(os.listdir(tmp_path / "batched")) == sorted(os.listdir(tmp_path / "rows")) == [f"{c}.pt" for c in range(5)]
    for c in range(5):
        a, b = torch.load(tmp_path / "rows" / f"{c}.pt"), torch.load(tmp_path / "batched" / f"{c}.pt")
        assert a.shape == (2, 4, 8, 8)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), c
    # PNG path through the worker threads: same names
    kw["latents_only"] = False
    generate.generate_images(exp_name="png", batch_prompts=4, **kw)
    assert sorted(os.listdir(tmp_path / "png")) == [f"{c}_{i}.png" for c in range(5) for i in range(2)]


def test_pipeline_generator_list_forms():
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    g = lambda s: torch.Generator().manual_seed(s)
    one = [pipe(p, num_inference_steps=2, num_images_per_prompt=2, generator=g(7 + i), output_type="latent").latents
           for i, p in enumerate(["a", "b"])]
    both = pipe(["a", "b"], num_inference_steps=2, num_images_per_prompt=2, generator=[g(7), g(8)], output_type="latent").latents
    assert torch.allclose(torch.cat(one), both, rtol=1e-4, atol=1e-5)
    per_image = pipe(["a", "b"], num_inference_steps=2, generator=[g(1), g(2)], output_type="latent").latents
    assert per_image.shape[0] == 2
    with pytest.raises(ValueError):
        pipe(["a", "b"], num_inference_steps=2, num_images_per_prompt=2, generator=[g(1), g(2), g(3)], output_type="latent")


def test_patch_unet_ignores_unknown_keys_like_strict_false(tmp_path):
    """generate-images-sd.py:19 loads with strict=False: unknown keys are ignored, shape mismatches still raise."""
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    name = next(n for n, _ in pipe.unet.named_parameters() if "attn2.to_k" in n)
    w = dict(pipe.unet.named_parameters())[name]
    loaded = sdp.patch_unet(pipe, {"nope.weight": torch.zeros(1), name: torch.ones_like(w)})
    assert loaded == [name] and bool((dict(pipe.unet.named_parameters())[name] == 1).all())
    with pytest.raises(ValueError):
        sdp.patch_unet(pipe, {name: torch.zeros(3, 3)})


def test_vae_deprecated_attention_keys_are_renamed():
    """SD-1.x VAE checkpoints: query/key/value/proj_attn -> to_q/to_k/to_v/to_out.0 (diffusers renames at load)."""
    f = sdp.convert_deprecated_vae_key
    assert f("decoder.mid_block.attentions.0.query.weight") == "decoder.mid_block.attentions.0.to_q.weight"
    assert f("decoder.mid_block.attentions.0.key.bias") == "decoder.mid_block.attentions.0.to_k.bias"
    assert f("decoder.mid_block.attentions.0.value.weight") == "decoder.mid_block.attentions.0.to_v.weight"
    assert f("decoder.mid_block.attentions.0.proj_attn.bias") == "decoder.mid_block.attentions.0.to_out.0.bias"
    assert f("decoder.mid_block.attentions.0.group_norm.weight") == "decoder.mid_block.attentions.0.group_norm.weight"
    assert f("decoder.mid_block.resnets.0.conv1.weight") == "decoder.mid_block.resnets.0.conv1.weight"
    vae = sdp.VaeDecoder((32, 32, 64, 64))
    sd = {}
    for k, v in vae.state_dict().items():            # a checkpoint in the deprecated naming, conv-shaped projections
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f".attentions.0.{new}." in k:
                k = k.replace(f".{new}.", f".{old}.")
                v = v[:, :, None, None] if v.dim() == 2 else v
        sd[k] = v
    assert any(".query." in k for k in sd)
    conv = {f(k): (v[:, :, 0, 0] if (".attentions." in k and v.dim() == 4) else v) for k, v in sd.items()}
    vae.load_state_dict(conv, strict=True)


def test_two_rank_gloo_generation_with_broadcast(tmp_path):
    """torch.distributed.run, 2 processes on CPU (gloo): rank 0 broadcasts the artifact, rows are
    sharded, and the union equals the single-process result file for file."""
    prompts = _tiny_prompts(tmp_path, 5)
    art, _ = _tiny_artifact(tmp_path)
    common = ["--model_id", "tiny-sd-test", "--synthetic_model", "--device", "cpu", "--prompts_path", prompts,
              "--save_path", str(tmp_path), "--uce_model_path", art, "--num_inference_steps", "2",
              "--latents_only"]
    script = os.path.join(REPO_ROOT, "evalscripts", "generate-images-sd.py")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, script] + common + ["--exp_name", "one"], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29533", script] + common + ["--exp_name", "two"],
                   check=True, env=env, timeout=900)
    one, two = sorted(os.listdir(tmp_path / "one")), sorted(os.listdir(tmp_path / "two"))
    assert one == two == [f"{c}.pt" for c in range(5)]
    for f in one:
        assert torch.equal(torch.load(tmp_path / "one" / f), torch.load(tmp_path / "two" / f))


def test_debias_sampling_is_sharded_over_ranks(tmp_path):
    """the get_ratios sampling loop (uce_sd_debias.py:14-35) round-robins the edit concepts
    over the ranks and all-reduces the [N_edit, N_debias] matrix: both ranks end with the single-process matrix."""
    worker = os.path.join(REPO_ROOT, "tests", "debias_ratios_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, worker, str(tmp_path)], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29541", worker, str(tmp_path)],
                   check=True, env=env, timeout=900)
    one = np.load(tmp_path / "ratios_w1_r0.npy")
    assert one.shape == (5, 2) and np.allclose(one.sum(axis=1), 0.0)      # desired sums to 1, observed sums to 1
    for r in (0, 1):
        assert np.array_equal(np.load(tmp_path / f"ratios_w2_r{r}.npy"), one)
    c0 = open(tmp_path / "calls_w2_r0.txt").read().split(";")
    c1 = open(tmp_path / "calls_w2_r1.txt").read().split(";")
    assert c0 == ["doctor", "teacher", "chef"] and c1 == ["nurse", "pilot"]   # disjoint, round-robin


def test_sdxl_topology_tokenless_pipeline_on_cpu():
    """The SDXL runtime at test widths: module discovery (2x(1+... ) transformer depths), dual-encoder context, pooled
    conditioning, Euler loop; unknown model ids are an error, not a silent SD-1.x build."""
    from uce_amd import edit as E
    pipe = sdp.load_pipeline("tiny-sdxl-test", torch.float32, "cpu", synthetic=True, vae=True)
    names = [n for n, _ in E.collect_uce_modules(pipe.unet)]
    # down_blocks.1 (2 attn x 1 block), down_blocks.2 (2 x 2), up_blocks.0 (3 x 2), up_blocks.1 (3 x 1), mid (1 x 2)
    assert len(names) == 2 * (2 * 1 + 2 * 2 + 3 * 2 + 3 * 1 + 1 * 2)
    assert names[0].startswith("down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k")
    assert any(n.startswith("up_blocks.0.attentions.2.transformer_blocks.1.attn2") for n in names)
    pe, ne, pp, npool = pipe.encode_prompt("a doctor", "cpu", 2, True)
    assert pe.shape == (2, 77, 64) and pp.shape == (2, 32) and float(ne.abs().max()) == 0.0 and float(npool.abs().max()) == 0.0
    g = lambda: torch.Generator().manual_seed(1)
    a = pipe("a doctor", num_inference_steps=3, num_images_per_prompt=2, guidance_scale=7.5, generator=g())
    b = pipe("a doctor", num_inference_steps=3, num_images_per_prompt=2, guidance_scale=7.5, generator=g())
    assert a.latents.shape == (2, 4, 8, 8) and torch.equal(a.latents, b.latents) and len(a.images) == 2
    c = pipe("a nurse", num_inference_steps=3, num_images_per_prompt=2, guidance_scale=7.5, generator=g())
    assert not torch.equal(a.latents, c.latents)
    with pytest.raises(ValueError, match="unknown architecture"):
        sdp.load_pipeline("stabilityai/some-other-model", torch.float32, "cpu", synthetic=True)


def test_euler_scheduler_matches_its_closed_form():
    """EulerDiscreteScheduler ("leading" spacing, offset 1): sigma table, init sigma, input scaling, one step."""
    from uce_amd.sd.scheduler import EulerDiscreteScheduler
    s = EulerDiscreteScheduler()
    s.set_timesteps(20)
    assert s.timesteps.tolist()[:3] == [951.0, 901.0, 851.0] and s.timesteps.tolist()[-1] == 1.0
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2
    ac = torch.cumprod(1 - betas, 0).double()
    sig = ((1 - ac) / ac) ** 0.5
    assert abs(float(s.sigmas[0]) - float(sig[951])) < 1e-4 and float(s.sigmas[-1]) == 0.0
    assert abs(s.init_noise_sigma - float((sig[951] ** 2 + 1) ** 0.5)) < 1e-4
    x, eps = torch.ones(1, 4, 2, 2), torch.full((1, 4, 2, 2), 0.5)
    assert torch.allclose(s.scale_model_input(x), x / (float(s.sigmas[0]) ** 2 + 1) ** 0.5)
    y = s.step(eps, 951.0, x)
    assert torch.allclose(y, x + 0.5 * (float(s.sigmas[1]) - float(s.sigmas[0])))


# --------------------------------------------------------------------------------------------------------------
# PNDMScheduler.step against the PUBLISHED algorithm, not against its own code: the scheduler config SD-1.4 ships
# (scaled-linear betas 0.00085 -> 0.012, 1000 train steps, steps_offset 1, skip_prk_steps: the PLMS branch only) with
#  * the transfer x_t -> x_{t-d} written as the DDIM deterministic update (Song et al. 2021, eq. 12 with sigma = 0), which
#    PNDM's eq. 9 (Liu et al. 2022) - the form diffusers' `_get_prev_sample` codes - is an algebraic rearrangement of;
#  * the linear-multistep weights DERIVED here as the Adams-Bashforth weights of order k (the unique weights that
#    integrate polynomials of degree < k exactly over one step from the k latest equidistant nodes), not typed in;
#  * the start-up rule of diffusers' `step_plms`: call 0 is an Euler step, call 1 repeats the timestep and redoes the
#    first step with the average of the two outputs (its output is NOT stored), then orders 2, 3, 4, 4, ...
# Everything in float64 on the host; generate-images-sd.py:13-15,37-42 reaches this through pipe(...).
# --------------------------------------------------------------------------------------------------------------

def _adams_bashforth(k: int) -> np.ndarray:
    """w with  sum_i w_i p(-i) = integral_0^1 p  for every polynomial p of degree < k  (nodes 0, -1, ..., -(k-1))."""
    nodes = -np.arange(k, dtype=np.float64)
    V = np.vander(nodes, k, increasing=True).T              # V[j, i] = nodes_i ** j
    rhs = 1.0 / np.arange(1, k + 1, dtype=np.float64)       # integral of t^j over [0, 1]
    return np.linalg.solve(V, rhs)


def _ddim_transfer(x, eps, a_t, a_prev):
    x0 = (x - np.sqrt(1.0 - a_t) * eps) / np.sqrt(a_t)
    return np.sqrt(a_prev) * x0 + np.sqrt(1.0 - a_prev) * eps


def _plms_reference(x, eps_seq, n_steps: int):
    """float64 rollout of the published PLMS schedule for scripted model outputs eps_seq[call]."""
    T = 1000
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, T, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    ratio = T // n_steps
    ts = [i * ratio + 1 for i in range(n_steps)][::-1]      # "leading" spacing + steps_offset 1: 981, 961, ..., 1
    alpha = lambda t: ac[t] if t >= 0 else ac[0]            # set_alpha_to_one = False -> final alpha = alphas_cumprod[0]
    hist, call = [], 0
    # call 0: Euler step from ts[0]; call 1: same timestep pair, trapezoid with the new output (not stored)
    x_start = x
    e0 = eps_seq[call]; call += 1
    hist.append(e0)
    x1 = _ddim_transfer(x_start, e0, alpha(ts[0]), alpha(ts[0] - ratio))
    e1 = eps_seq[call]; call += 1
    x = _ddim_transfer(x_start, 0.5 * (e0 + e1), alpha(ts[0]), alpha(ts[0] - ratio))
    outs = [x1, x]
    for t in ts[1:]:
        hist.append(eps_seq[call]); call += 1
        k = min(len(hist), 4)
        w = _adams_bashforth(k)
        e = sum(w[i] * hist[-1 - i] for i in range(k))
        x = _ddim_transfer(x, e, alpha(t), alpha(t - ratio))
        outs.append(x)
    return outs, ac


def test_adams_bashforth_weights_are_the_plms_coefficients():
    assert np.allclose(_adams_bashforth(2) * 2, [3, -1])
    assert np.allclose(_adams_bashforth(3) * 12, [23, -16, 5])
    assert np.allclose(_adams_bashforth(4) * 24, [55, -59, 37, -9])


@pytest.mark.parametrize("n_steps", [50, 20])
def test_pndm_step_matches_the_published_plms_algorithm(n_steps):
    s = PNDMScheduler()
    s.set_timesteps(n_steps)
    ts = s.timesteps.tolist()
    assert len(ts) == n_steps + 1
    rng = np.random.Generator(np.random.PCG64(n_steps))
    x0 = rng.standard_normal((1, 4, 8, 8))
    eps_seq = [rng.standard_normal((1, 4, 8, 8)) for _ in ts]
    want, ac = _plms_reference(x0, eps_seq, n_steps)
    # the noise schedule itself: scaled-linear betas, cumulative product (float32 in the scheduler)
    assert np.allclose(s.alphas_cumprod.double().numpy(), ac, rtol=2e-6, atol=0)
    assert abs(ac[0] - (1 - 0.00085)) < 1e-12 and abs(ac[-1] - 0.00466) < 5e-5     # SD's well-known end points
    x = torch.from_numpy(x0).double()
    for call, t in enumerate(ts):
        x = s.step(torch.from_numpy(eps_seq[call]).double(), t, x)
        err = float((x - torch.from_numpy(want[call])).norm() / np.linalg.norm(want[call]))
        assert err < 2e-6, (call, t, err)     # float64 tensors, float32 schedule constants
    # the schedule: the second timestep is the repeated one, the last is steps_offset
    ratio = 1000 // n_steps
    assert ts[0] == (n_steps - 1) * ratio + 1 and ts[1] == ts[2] == ts[0] - ratio and ts[-1] == 1


# --------------------------------------------------------------------------------------------------------------
# The reference's REAL prompt table (data/coco_30k.csv: BASELINE config 5): tests/golden/coco30k_rows.csv holds 71 of
# its records - the first rows, every multi-line (quoted newline) caption among the first 5000, quoted commas / doubled
# quotes, the extreme seeds, the last record - and coco30k_rows.json what the reference's own generate_images() did on
# that file with a recording stand-in for the pipeline (tools/make_coco_fixture.py, run in the build container).
# --------------------------------------------------------------------------------------------------------------

class _RecordingPipe:
    """Stands where pipe(...) stands in generate.generate_images: records the call, returns 8 x 8 images."""

    def __init__(self):
        self.calls = []
        self.unet = torch.nn.Linear(1, 1)

    def __call__(self, prompt, num_inference_steps, guidance_scale, num_images_per_prompt, generator):
        from PIL import Image
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        gens = [generator] if isinstance(generator, torch.Generator) else list(generator)
        assert len(prompts) == len(gens)
        for p, g in zip(prompts, gens):
            self.calls.append({"prompt": p, "seed": int(g.initial_seed()), "n": int(num_images_per_prompt),
                               "generator_device": str(g.device), "steps": int(num_inference_steps),
                               "guidance": float(guidance_scale)})
        return type("Out", (), {"images": [Image.new("RGB", (8, 8)) for _ in range(len(prompts) * num_images_per_prompt)]})()


def _coco_fixture():
    import json
    root = os.path.join(REPO_ROOT, "tests", "golden")
    return os.path.join(root, "coco30k_rows.csv"), json.load(open(os.path.join(root, "coco30k_rows.json")))


@pytest.mark.parametrize("batch_prompts", [1, 4])
def test_generation_loop_on_real_coco30k_rows_matches_the_reference(tmp_path, batch_prompts):
    csv_path, meta = _coco_fixture()
    df = pd.read_csv(csv_path)
    assert len(df) == meta["records"] and list(df.columns) == ["case_number", "source", "prompt", "evaluation_seed", "coco_id"]
    assert int(df.prompt.str.contains("\n").sum()) == meta["multi_line_prompts"] >= 10     # quoted newlines survive parsing
    assert int(df.evaluation_seed.max()) == 99998
    for win in meta["windows"]:
        pipe = _RecordingPipe()
        out = tmp_path / f"w{win['from_case']}_{batch_prompts}"
        stats = generate.generate_images("CompVis/stable-diffusion-v1-4", None, csv_path, str(out), exp_name="coco", device="cpu",
                                         guidance_scale=7.5, num_inference_steps=50,
                                         num_images_per_prompt=win["num_images_per_prompt"], from_case=win["from_case"],
                                         till_case=win["till_case"], pipe=pipe, batch_prompts=batch_prompts, png_workers=2)
        # the same pipe(...) arguments row by row: prompt text (incl. embedded newlines), CPU generator, seed, n
        assert pipe.calls == win["calls"]
        assert sorted(os.listdir(out / "coco")) == win["files"]
        assert stats["images"] == len(win["files"])


@pytest.mark.parametrize("world", [2, 8])
def test_real_coco30k_rows_shard_disjointly_and_completely(world):
    csv_path, meta = _coco_fixture()
    df = pd.read_csv(csv_path)
    win = meta["windows"][1]
    shards = [[int(r.case_number) for _, r in generate.select_rows(df, win["from_case"], win["till_case"], rank, world)]
              for rank in range(world)]
    flat = sorted(c for s in shards for c in s)
    want = sorted({int(f.split("_")[0]) for f in win["files"]})
    assert flat == want and len(flat) == len(set(flat))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1            # round-robin balance


def test_eight_rank_gloo_generation_covers_the_reference_file_list(tmp_path):
    """The 8-GPU run of BASELINE config 5 without the hardware: 8 gloo ranks of generate_images (tiny model, one step) over the
    real coco_30k records of the fixture - the union of the PNG names the ranks write equals the list the reference's own loop
    wrote for the same window, no file is written twice, and the per-rank image counts differ by at most one."""
    import json
    csv_path, meta = _coco_fixture()
    win = meta["windows"][0]
    worker = os.path.join(REPO_ROOT, "tests", "gen_world_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                    "--master-addr", "127.0.0.1", "--master-port", "29577", worker, str(tmp_path), csv_path],
                   check=True, env=env, timeout=1200)
    files = sorted(f for f in os.listdir(tmp_path / "coco") if f.endswith(".png"))
    assert files == sorted(win["files"])
    stats = [json.load(open(tmp_path / f"stats_r{r}.json")) for r in range(8)]
    counts = [int(s["images"]) for s in stats]
    assert sum(counts) == len(win["files"]) and max(counts) - min(counts) <= 1
    assert all(int(s["world"]) == 8 and int(s["images_total"]) == len(win["files"]) for s in stats)


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_bench_generation_leg_two_ranks_gloo(tmp_path, fail_rank):
    """bench.py's generation leg as the N > 1 driver line runs it: weight broadcast, barrier, timed loop, barrier,
    MAX-reduce of (seconds, failure flag), all-gather of the per-rank seconds - on two gloo ranks with the tiny model.
    With a failure injected on rank 1 BOTH ranks must still pass every collective (no hang) and report the error."""
    import json
    worker = os.path.join(REPO_ROOT, "tests", "bench_gen_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="2", UCE_BENCH_FAIL_RANK=str(fail_rank))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", str(29551 + (fail_rank > 0)), worker, str(tmp_path)],
                   check=True, env=env, timeout=900)
    res = [json.load(open(tmp_path / f"gen_w2_r{r}.json")) for r in (0, 1)]
    for r in res:
        assert r["n_gpus"] == 2 and r["metric"].startswith("images/sec")
    if fail_rank < 0:
        for r in res:
            assert r["value"] > 0 and r["images_per_rank"] == 2 and r["weight_broadcast_ms"] >= 0
            assert len(r["per_rank_images_per_s"]) == 2 and all(v > 0 for v in r["per_rank_images_per_s"])
        assert res[0]["value"] == res[1]["value"]                      # MAX-reduced seconds: one figure for the job
    else:
        assert res[0]["value"] is None and res[1]["value"] is None
        assert "injected failure" in res[1]["error"] and res[0]["error"] == "another rank failed"


def test_decoded_images_become_the_same_bytes_as_diffusers_numpy_to_pil():
    """sd.pipeline.images_from_decoded does diffusers' post-processing where the tensor lives: `(image / 2 + 0.5).clamp(0, 1)`,
    NHWC float32, then numpy_to_pil's `(images * 255).round().astype("uint8")` - the uint8 conversion in torch (float32 multiply,
    round-half-to-even) must give numpy's bytes, exact halves and out-of-range values included."""
    import numpy as np
    import torch
    from uce_amd.sd import pipeline as sdp
    g = torch.Generator().manual_seed(5)
    dec = (torch.randn((3, 3, 16, 24), generator=g) * 1.3).to(torch.bfloat16)
    dec[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, 0.0, 3.0, -3.0, 1 / 255, 3 / 255, 0.5]).to(torch.bfloat16)
    halves = torch.tensor([(k + 0.5) / 255 * 2 - 1 for k in range(0, 48)], dtype=torch.float32)   # (x / 2 + 0.5) * 255 = k + 0.5
    dec32 = dec.float()
    dec32[1, 1, 2, :] = halves[:24]
    dec32[1, 1, 3, :] = halves[24:]
    for d in (dec, dec32):
        want = (d.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
        want = [(im * 255).round().astype("uint8") for im in want]
        got = sdp.images_from_decoded(d, "pil")
        assert len(got) == 3 and all(im.size == (24, 16) for im in got)
        assert all(np.array_equal(np.asarray(a), b) for a, b in zip(got, want))
        arr = sdp.images_from_decoded(d, "np")
        assert all(np.array_equal(a, (d.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()[i]) for i, a in enumerate(arr))


def test_automatic_batch_sits_on_a_fixed_ladder(monkeypatch):
    # the batch an image is denoised in decides tile forms / contraction splits, i.e. its bf16 bits: the automatic size must not follow
    # the free HBM byte for byte
    pipe = object.__new__(sdp.StableDiffusionPipeline)
    pipe.unet = type("U", (), {"cfg": type("C", (), {"sample_size": 64})()})()
    dev = torch.device("cuda", 0)
    per = generate.AUTO_BATCH_BYTES_PER_IMAGE
    for free_images, want in ((300, 128), (127.9, 64), (64, 64), (47, 32), (3, 2), (0.2, 1)):
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d, f=free_images: (int(2 * f * per), int(4 * 300 * per)))
        assert generate.auto_batch_prompts(pipe, dev, 1, 10 ** 6) == want
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d: (int(2 * 300 * per), int(4 * 300 * per)))
    assert generate.auto_batch_prompts(pipe, dev, 10, 10 ** 6) == 8            # 128 images / 10 per row = 12 rows -> 8
    assert generate.auto_batch_prompts(pipe, dev, 1, 5) == 5                   # never more than the rows there are
    assert generate.auto_batch_prompts(pipe, torch.device("cpu"), 1, 99) == 1  # CPU / foreign pipelines: the reference's loop


def test_automatic_batch_halves_on_out_of_memory_and_an_explicit_one_raises(tmp_path, monkeypatch):
    prompts = _tiny_prompts(tmp_path, 7)
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    calls = []
    real = type(pipe).__call__

    def flaky(self, prompt, **kw):
        n = 1 if isinstance(prompt, str) else len(prompt)
        calls.append(n)
        if n > 2:
            raise torch.cuda.OutOfMemoryError("injected")
        return real(self, prompt, **kw)

    monkeypatch.setattr(type(pipe), "__call__", flaky)
    kw = dict(model_id="tiny-sd-test", uce_model_path=None, prompts_path=prompts, save_path=str(tmp_path), device="cpu",
              torch_dtype=torch.float32, num_inference_steps=2, num_images_per_prompt=1, pipe=pipe, latents_only=True)
    monkeypatch.setattr(generate, "auto_batch_prompts", lambda *a: 8)
    st = generate.generate_images(exp_name="auto", batch_prompts=0, **kw)
    assert calls == [7, 4, 2, 2, 2, 1] and st["images"] == 7.0 and st["batch_prompts"] == 2.0
    calls.clear()
    monkeypatch.setattr(type(pipe), "__call__", real)
    generate.generate_images(exp_name="rows", batch_prompts=1, **kw)
    for i in range(7):                                   # the retried rows were re-seeded: same latents as the row-by-row loop
        a, b = torch.load(tmp_path / "auto" / f"{i}.pt"), torch.load(tmp_path / "rows" / f"{i}.pt")
        assert torch.allclose(a, b, atol=1e-5), i
    monkeypatch.setattr(type(pipe), "__call__", flaky)
    with pytest.raises(torch.cuda.OutOfMemoryError):
        generate.generate_images(exp_name="explicit", batch_prompts=4, **kw)
