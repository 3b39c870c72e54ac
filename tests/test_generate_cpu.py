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


def test_patch_unet_rejects_unknown_keys(tmp_path):
    pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
    with pytest.raises(KeyError):
        sdp.patch_unet(pipe, {"nope.weight": torch.zeros(1)})


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
