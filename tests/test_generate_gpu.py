"""GPU: BASELINE config 5 at SD-1.4 size on one GPU - rows of a coco_30k-schema prompt table
(case_number,source,prompt,evaluation_seed,coco_id; evalscripts/generate-images-sd.py:21-46) through
generate.generate_images with the edited weights patched in, and the RCCL broadcast of those weights."""
import os
import socket
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

from uce_amd import REPO_ROOT

pytestmark = pytest.mark.gpu


def _edited_artifact(tmp_path, pipe):
    """A real edit of the pipeline's 32 attn2 projections (2 erase + 1 preserve) -> safetensors artifact."""
    from uce_amd import edit as E
    E.UCE(pipe, ["Van Gogh", "Picasso"], ["art", "art"], ["Monet"], 1.0, 1.0, 0.5, str(tmp_path), "edit", device="cuda:0")
    return str(tmp_path / "edit.safetensors")


def test_coco_schema_rows_at_sd14_size(tmp_path):
    from PIL import Image
    from safetensors.torch import load_file
    from uce_amd import generate, synth
    from uce_amd.sd import pipeline as sdp
    csv_path = synth.write_prompts_csv(str(tmp_path / "coco_synth.csv"), 12, seed=0)
    df = pd.read_csv(csv_path)
    assert list(df.columns) == ["case_number", "source", "prompt", "evaluation_seed", "coco_id"]
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
    art = _edited_artifact(tmp_path, pipe)
    before = {n: p.detach().clone() for n, p in pipe.unet.named_parameters() if "attn2.to_k" in n}
    kw = dict(model_id="CompVis/stable-diffusion-v1-4", uce_model_path=art, prompts_path=csv_path, save_path=str(tmp_path),
              device="cuda:0", torch_dtype=torch.bfloat16, guidance_scale=7.5, num_inference_steps=50,
              num_images_per_prompt=1, synthetic=True, pipe=pipe)
    stats = generate.generate_images(exp_name="imgs", from_case=2, till_case=7, batch_prompts=3, **kw)
    assert stats["images"] == 6
    assert sorted(os.listdir(tmp_path / "imgs")) == [f"{c}_0.png" for c in range(2, 8)]
    im = Image.open(tmp_path / "imgs" / "2_0.png")
    assert im.size == (512, 512) and np.asarray(im).std() > 0
    # the edited weights really are in the U-Net (bf16 cast of the artifact), and differ from the unedited ones
    state = load_file(art)
    name = next(iter(before))
    assert torch.equal(dict(pipe.unet.named_parameters())[name].cpu(), state[name].to(torch.bfloat16))
    assert not torch.equal(before[name].cpu(), state[name].to(torch.bfloat16))
    # resume: nothing is regenerated for rows whose first PNG exists
    stats = generate.generate_images(exp_name="imgs", from_case=2, till_case=8, batch_prompts=3, skip_existing=True, **kw)
    assert stats["images"] == 1 and os.path.exists(tmp_path / "imgs" / "8_0.png")
    # a row generated alone equals the same row generated inside a batch (latents, fixed CPU seed)
    kw["num_inference_steps"] = 10
    generate.generate_images(exp_name="row", from_case=3, till_case=3, batch_prompts=1, latents_only=True, **kw)
    generate.generate_images(exp_name="bat", from_case=2, till_case=4, batch_prompts=3, latents_only=True, **kw)
    a, b = torch.load(tmp_path / "row" / "3.pt").float(), torch.load(tmp_path / "bat" / "3.pt").float()
    assert a.shape == (1, 4, 64, 64)
    assert float((a - b).norm() / b.norm()) < 3e-2


def test_rccl_broadcast_of_the_edited_weights_single_rank(tmp_path):
    """The exchange step of the sharded generation path through a real RCCL communicator (backend nccl): one rank on
    the one GPU of the test box."""
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(0)
    state = {f"down_blocks.{i}.attentions.0.transformer_blocks.0.attn2.to_k.weight": torch.randn(320, 768, generator=g)
             for i in range(3)}
    path = str(tmp_path / "w.safetensors")
    save_file(state, path)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, UCE_TEST_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO_ROOT, "tests", "rccl_bcast_worker.py"), path], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL_BCAST_OK nccl 3 uce_bcast ok" in r.stdout


def test_real_coco30k_rows_at_sd14_size(tmp_path):
    """Rows of the reference's REAL prompt table (the committed fixture: tests/golden/coco30k_rows.csv, incl. a caption with a
    quoted newline) through generate_images at SD-1.4 size: the PNG names the reference wrote for that window, and each image
    = what pipe(prompt, generator=CPU generator seeded with the row's evaluation_seed) gives for the prompt and seed the
    reference passed (coco30k_rows.json)."""
    import json
    from uce_amd import generate
    from uce_amd.sd import pipeline as sdp
    csv_path = os.path.join(REPO_ROOT, "tests", "golden", "coco30k_rows.csv")
    meta = json.load(open(os.path.join(REPO_ROOT, "tests", "golden", "coco30k_rows.json")))
    win = meta["windows"][1]                                           # from_case 85 ... till_case 2539, 2 images per prompt
    calls = [c for c in win["calls"]][:4]                              # the first four rows of that window
    assert "\n" in calls[0]["prompt"]                                  # case 85: the multi-line caption
    cases = sorted({int(f.split("_")[0]) for f in win["files"]})[:4]
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
    kw = dict(model_id="CompVis/stable-diffusion-v1-4", uce_model_path=None, prompts_path=csv_path, save_path=str(tmp_path),
              device="cuda:0", torch_dtype=torch.bfloat16, guidance_scale=7.5, num_inference_steps=8, num_images_per_prompt=2,
              synthetic=True, pipe=pipe)
    stats = generate.generate_images(exp_name="png", from_case=cases[0], till_case=cases[-1], batch_prompts=3, **kw)
    want_files = [f for f in win["files"] if int(f.split("_")[0]) in cases]
    assert sorted(os.listdir(tmp_path / "png")) == sorted(want_files) and stats["images"] == len(want_files) == 8
    generate.generate_images(exp_name="lat", from_case=cases[0], till_case=cases[-1], batch_prompts=3, latents_only=True, **kw)
    for case, call in zip(cases, calls):
        got = torch.load(tmp_path / "lat" / f"{case}.pt").float()
        alone = pipe(call["prompt"], num_inference_steps=8, guidance_scale=7.5, num_images_per_prompt=2,
                     generator=torch.Generator().manual_seed(call["seed"]), output_type="latent").latents.float().cpu()
        assert got.shape == alone.shape == (2, 4, 64, 64)
        assert float((got - alone).norm() / alone.norm()) < 3e-2, case   # batched vs alone: different GEMM shapes, same draw


def test_default_automatic_batch_on_130_rows_matches_rows_generated_alone(tmp_path):
    """The CLI default (--batch_prompts 0): 130 rows at SD-1.4 size are denoised 64 + 64 + 2 per pipe() call on a whole 288 GB MI355X
    (half of the free HBM at 1.5 GB per image = 96, rounded down on the ladder; 128 + 2 where more is free); every file of the
    row-by-row loop is there, and the latents of three rows - the first of a full batch, one in the middle of a later one, the one
    that lands in the ragged tail batch - are those of the row generated ALONE within the stated distance (same CPU-seeded draw,
    other tile forms)."""
    from uce_amd import generate, synth
    from uce_amd.sd import pipeline as sdp
    csv_path = synth.write_prompts_csv(str(tmp_path / "coco_synth.csv"), 130, seed=1)
    df = pd.read_csv(csv_path)
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=False)
    kw = dict(model_id="CompVis/stable-diffusion-v1-4", uce_model_path=None, prompts_path=csv_path, save_path=str(tmp_path),
              device="cuda:0", torch_dtype=torch.bfloat16, guidance_scale=7.5, num_inference_steps=6, num_images_per_prompt=1,
              synthetic=True, pipe=pipe, latents_only=True)
    stats = generate.generate_images(exp_name="auto", batch_prompts=0, **kw)
    assert stats["images"] == 130 and stats["batch_prompts"] in (64.0, 128.0)
    cases = [int(c) for c in df.case_number]
    assert sorted(os.listdir(tmp_path / "auto")) == sorted(f"{c}.pt" for c in cases)
    for idx in (0, 77, 129):
        row = df.iloc[idx]
        got = torch.load(tmp_path / "auto" / f"{int(row.case_number)}.pt").float()
        alone = pipe(str(row.prompt), num_inference_steps=6, guidance_scale=7.5, num_images_per_prompt=1,
                     generator=torch.Generator().manual_seed(int(row.evaluation_seed)), output_type="latent").latents.float().cpu()
        assert got.shape == alone.shape == (1, 4, 64, 64)
        assert torch.isfinite(got).all()
        assert float((got - alone).norm() / alone.norm()) < 3e-2, idx
