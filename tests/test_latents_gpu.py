"""GPU: latent parity of the generation path at SD-1.4 size (north star: "generated latents within
a stated fp16 tolerance at fixed seed"; reference call site evalscripts/generate-images-sd.py:37-42).

diffusers, checkpoints and tokenizer vocabularies do not exist on these machines, so the pin is the one that can be
built here: the SAME seeded weights and the SAME CPU-seeded initial latents run through
  (a) the fp32 torch path   - the bf16-representable weights held in fp32 (the reference runs its pipeline in bf16,
                              generate-images-sd.py:13, so the MODEL is the bf16-rounded one; with fp32-only weights a
                              50-step trajectory on random weights decorrelates completely - measured relF 1.35 for
                              torch's bf16 ops and for ours alike), fp32 arithmetic, every hand-written kernel off
                              (16-bit-only kernels fall back to torch ops), eager launches: the comparison standard;
  (b) the product path      - bf16, hipGraph-replayed step, uce_xattn_fwd / uce_sattn_fwd / GroupNorm / LayerNorm /
                              GEGLU / im2col kernels, hoisted K/V;
  (c) the torch bf16 path   - bf16 weights, every hand-written kernel off: what the reference's own bf16 pipeline
                              does to the same network (its torch ops in bf16).
Stated tolerances (measured values in profiles/r02/latent_parity.json):
  per U-Net call (same input latents):  relF(eps_product, eps_fp32) <= PER_CALL_TOL
  after the full 50-step PNDM loop:     relF(lat_product, lat_fp32) <= FINAL_TOL
  and the product path is no further from fp32 than torch's own bf16 ops: <= 1.25 x relF(torch bf16, fp32) + 0.01.
"""
import json
import os

import pytest
import torch

from oracle import uce_oracle as O
from uce_amd import REPO_ROOT

pytestmark = pytest.mark.gpu

PER_CALL_TOL = 6e-2          # measured 4.1e-2 .. 4.7e-2 (torch bf16 ops on the same model: 4.9e-2 .. 6.0e-2)
FINAL_TOL = 3e-2             # measured 1.3e-2 after 51 U-Net calls (torch bf16 ops incl. their bf16 scheduler arithmetic: 2.9e-2)
PROMPT = "a photo of an astronaut riding a horse"
SEED = 1234
STEPS = 50
PROBE_STEPS = (0, 1, 12, 25, 38, 50)          # U-Net calls whose input / output the fp32 run records


class _Hip:
    """_set_hip(False): every hand-written kernel off (tests/torch_twin.py patches the product's dispatch predicate)."""
    ctx = None


def _set_hip(on: bool):
    from tests.torch_twin import torch_ops
    if _Hip.ctx is not None:
        _Hip.ctx.__exit__(None, None, None)
        _Hip.ctx = None
    if not on:
        _Hip.ctx = torch_ops()
        _Hip.ctx.__enter__()


def _guided_eps(pipe, latents, t, ctx, scale=7.5):
    x = torch.cat([latents] * 2)
    eu, ec = pipe.unet(x, torch.tensor([t], device=pipe.device), ctx).chunk(2)
    return eu + scale * (ec - eu)


def test_latents_at_sd14_size_match_fp32_within_stated_tolerance():
    from uce_amd.sd import pipeline as sdp
    dev = "cuda:0"
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.float32, dev, synthetic=True, vae=False, seed=0)
    for m in (pipe.unet, pipe.text_encoder):                       # the model under test = the bf16-rounded weights
        for p in m.parameters():
            p.data.copy_(p.data.to(torch.bfloat16).float())
    # initial noise as the reference's bf16 pipeline draws it (generate-images-sd.py:41: a CPU generator; diffusers'
    # randn_tensor draws in the pipeline dtype), the SAME tensor for all three runs
    lat0 = torch.randn((1, 4, 64, 64), generator=torch.Generator().manual_seed(SEED), dtype=torch.bfloat16)
    probes = {}

    def record(i, t, lat, eps):
        if i in PROBE_STEPS:
            probes[i] = (t, lat.detach().float().clone(), eps.detach().float().clone())

    try:
        # ---- (a) fp32 torch path
        _set_hip(False)
        pipe.use_graph = False
        ref = pipe(PROMPT, num_inference_steps=STEPS, latents=lat0, output_type="latent", callback=record).latents.float()
        assert ref.shape == (1, 4, 64, 64) and bool(torch.isfinite(ref).all())
        assert sorted(probes) == list(PROBE_STEPS)
        # ---- (b) product path: bf16, all kernels, hipGraph replay
        pipe.to(dev, torch.bfloat16)
        _set_hip(True)
        pipe.use_graph = True
        pipe.fused_step = True                                      # guidance combine + PLMS step as one HIP launch
        prod = pipe(PROMPT, num_inference_steps=STEPS, latents=lat0, output_type="latent").latents.float()
        assert len(pipe._graphs) == 1                               # the step really was replayed from a hipGraph
        pe, ne = pipe.encode_prompt(PROMPT, dev, 1, True)
        ctx = torch.cat([ne, pe])
        per_call = {}
        pipe.unet.cache_context(ctx)
        for i, (t, lat, eps) in probes.items():
            got = _guided_eps(pipe, lat.to(torch.bfloat16), t, ctx).float()
            per_call[i] = O.rel_fro(got, eps)
        pipe.unet.cache_context(None)
        # ---- (c) torch bf16 path (no hand-written kernels)
        _set_hip(False)
        pipe.use_graph = False
        pipe.fused_step = False
        tb = pipe(PROMPT, num_inference_steps=STEPS, latents=lat0, output_type="latent").latents.float()
        per_call_torch = {i: O.rel_fro(_guided_eps(pipe, lat.to(torch.bfloat16), t, ctx).float(), eps)
                          for i, (t, lat, eps) in probes.items()}
    finally:
        _set_hip(True)
    final_prod, final_torch = O.rel_fro(prod, ref), O.rel_fro(tb, ref)
    report = dict(prompt=PROMPT, seed=SEED, steps=STEPS, size="SD-1.4, 512x512 (latents 1x4x64x64), guidance 7.5, PNDM",
                  weights="seeded-random (no checkpoint on this machine)",
                  final_relF=dict(product_bf16_vs_fp32=final_prod, torch_bf16_vs_fp32=final_torch,
                                  product_vs_torch_bf16=O.rel_fro(prod, tb)),
                  per_call_relF=dict(product_bf16_vs_fp32={str(k): v for k, v in per_call.items()},
                                     torch_bf16_vs_fp32={str(k): v for k, v in per_call_torch.items()}),
                  tolerances=dict(per_call=PER_CALL_TOL, final=FINAL_TOL))
    out = os.path.join(REPO_ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "latent_parity.json"), "w"), indent=1)
    print(json.dumps(report))
    assert max(per_call.values()) <= PER_CALL_TOL, per_call
    assert final_prod <= FINAL_TOL, (final_prod, final_torch)
    assert max(per_call.values()) <= 1.25 * max(per_call_torch.values()) + 0.01, (per_call, per_call_torch)
    assert final_prod <= 1.25 * final_torch + 0.01, (final_prod, final_torch)


@pytest.mark.parametrize("n", [16, 128])
def test_unet_call_at_the_generation_batch_matches_fp32_within_the_per_call_tolerance(n):
    """The tile rules of the kernels follow the batch: at one prompt per call (the test above) every layer takes the few-tile forms
    with the split contraction, at 16 prompts the wide forms of most layers, at the bench's 128 prompts per call (CFG batch 256)
    the 256 x 256 GEGLU tiles, the 128-byte k-tiles from M >= 196 608 and k_conv3x3_w1 on every convolution.  ONE U-Net evaluation at
    SD-1.4 size per batch - the same bf16-rounded weights through fp32 torch ops, through the product path and through torch's own
    bf16 ops - under the per-call tolerance stated above."""
    from uce_amd.sd import pipeline as sdp
    dev = "cuda:0"
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.float32, dev, synthetic=True, vae=False, seed=0)
    for p in pipe.unet.parameters():
        p.data.copy_(p.data.to(torch.bfloat16).float())
    g = torch.Generator().manual_seed(7)
    lat = torch.randn((n, 4, 64, 64), generator=g).to(torch.bfloat16)
    ctx = (torch.randn((2 * n, 77, 768), generator=g) * 0.5).to(torch.bfloat16)
    t = torch.tensor([481], device=dev)
    x = torch.cat([lat] * 2).to(dev)
    try:
        _set_hip(False)
        ref = pipe.unet(x.float(), t, ctx.to(dev).float()).float()
        pipe.unet.to(torch.bfloat16)
        tb = pipe.unet(x, t, ctx.to(dev)).float()
        _set_hip(True)
        from uce_amd import edit as E
        calls = {"linear": 0, "conv": 0, "packed": 0}
        orig = (E.UceHandle.linear, E.UceHandle.conv3x3_igemm, E.UceHandle.sattn_packed)
        orig_exp2 = E.UceHandle.sattn_packed_exp2           # the 64 x 64 level: q pre-scaled by its projection, scores in the exp2 domain

        def wrap(name, fn):
            def inner(self, *a, **k):
                calls[name] += 1
                return fn(self, *a, **k)
            return inner
        E.UceHandle.linear, E.UceHandle.conv3x3_igemm, E.UceHandle.sattn_packed = (wrap("linear", orig[0]), wrap("conv", orig[1]),
                                                                                   wrap("packed", orig[2]))
        E.UceHandle.sattn_packed_exp2 = wrap("packed", orig_exp2)
        try:
            got = pipe.unet(x.contiguous(memory_format=torch.channels_last), t, ctx.to(dev)).float()
        finally:
            E.UceHandle.linear, E.UceHandle.conv3x3_igemm, E.UceHandle.sattn_packed = orig
            E.UceHandle.sattn_packed_exp2 = orig_exp2
    finally:
        _set_hip(True)
    assert calls["packed"] >= 15 and calls["linear"] >= 100 and calls["conv"] >= 20, calls     # the own kernels really ran
    e_prod, e_torch = O.rel_fro(got, ref), O.rel_fro(tb, ref)
    print(json.dumps(dict(batch=2 * n, product_bf16_vs_fp32=e_prod, torch_bf16_vs_fp32=e_torch, calls=calls)))
    assert e_prod <= PER_CALL_TOL, (e_prod, e_torch)
    assert e_prod <= 1.25 * e_torch + 0.01, (e_prod, e_torch)
