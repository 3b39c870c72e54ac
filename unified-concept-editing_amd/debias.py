"""Debias driver: the drop-in for `get_ratios` + `UCE` of the reference's
trainscripts/uce_sd_debias.py:14-35, 37-149.

Per iteration (uce_sd_debias.py:95-141): patch the U-Net with the current weights, generate
`num_images_per_prompt` images per edit concept, zero-shot classify them against the debias
concepts, turn the observed ratios into `direction_scale = desired - observed` (zeroed when every
|diff| < max_diff), stop when everything is balanced, otherwise re-solve the closed form with the
CUMULATIVE drift (the reference adds the drift in place to its cached guide outputs, so it
accumulates; `step_size` is parsed but never used there - kept that way).

Multi-GPU: launched with torch.distributed.run, the sampling +
classification of the edit concepts is sharded round-robin over the ranks and the ratio matrix is
all-reduced; the closed-form solve is replicated (it is < 1 ms; "replicas only").
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import edit as E
from .sd import pipeline as sdp


def ratios_from_labels(labels: Sequence[str], debias_concepts: Sequence[str], desired_ratios: Sequence[float],
                       max_diff: float) -> np.ndarray:
    """uce_sd_debias.py:28-32."""
    results = np.array(list(labels))
    ratios = np.array([desired - (sum(results == c) / len(results)) for c, desired in zip(debias_concepts, desired_ratios)])
    if max(ratios) < max_diff and abs(min(ratios)) < max_diff:
        ratios = 0 * ratios
    return ratios


def get_ratios(pipe, classify: Callable, slab: E.WeightSlab, edit_concepts, debias_concepts, desired_ratios, max_diff,
               num_images_per_prompt=10, num_inference_steps=20, guidance_scale=7.5, rank: int = 0, world: int = 1,
               reduce_device=None) -> np.ndarray:
    """uce_sd_debias.py:14-35 (images are sampled UNSEEDED there too).

    Sharded over the ranks of one node: edit concept i is sampled and classified on rank i % world (the build's own pipeline
    takes several concepts' `num_images_per_prompt` images per U-Net batch, any other pipeline one concept per call), every rank fills its own rows of the
    [N_edit, N_debias] float64 `direction_scale` matrix, and ONE all-reduce (sum; the rows are disjoint)
    gives every rank the whole matrix - a few hundred bytes over RCCL/xGMI (gloo on CPU)."""
    state = slab.state_dict()
    if hasattr(pipe.unet, "cfg"):
        sdp.patch_unet(pipe, state)
    else:
        pipe.unet.load_state_dict(state, strict=False)
    direction_scale = np.zeros((len(edit_concepts), len(debias_concepts)), dtype=np.float64)
    mine = [i for i in range(len(edit_concepts)) if i % world == rank]
    # the build's own pipeline on a GPU samples SEVERAL concepts per pipe() call (their images are independent draws, unseeded in
    # the reference as well): one concept per call is a CFG batch of 2 x num_images_per_prompt = 20, where the U-Net's layers have too
    # few output tiles to fill the chip; a foreign pipeline object keeps the reference's one call per concept
    from .generate import auto_batch_prompts
    dev = torch.device(getattr(pipe, "device", "cpu"))
    group = auto_batch_prompts(pipe, dev, num_images_per_prompt, len(mine))
    for lo in range(0, len(mine), group):
        idx = mine[lo:lo + group]
        prompts = [edit_concepts[i] for i in idx]
        images = pipe(prompts[0] if len(prompts) == 1 else prompts, num_inference_steps=num_inference_steps,
                      num_images_per_prompt=num_images_per_prompt, guidance_scale=guidance_scale).images
        for j, i in enumerate(idx):
            labels = classify(images[j * num_images_per_prompt:(j + 1) * num_images_per_prompt], debias_concepts)
            direction_scale[i] = ratios_from_labels(labels, debias_concepts, desired_ratios, max_diff)
    if world > 1:
        import torch.distributed as dist
        t = torch.from_numpy(direction_scale).to(reduce_device if reduce_device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        direction_scale = t.cpu().numpy()
    return direction_scale


def clip_zero_shot_classifier(device):
    """The reference's classifier (uce_sd_debias.py:245-250): CLIP ViT-B/32 zero-shot, top-1 label."""
    from transformers import pipeline as hf_pipeline
    clf = hf_pipeline(task="zero-shot-image-classification", model="openai/clip-vit-base-patch32",
                      torch_dtype=torch.bfloat16, device=device)

    def classify(images, labels):
        return [r[0]["label"] for r in clf(images, candidate_labels=list(labels))]
    return classify


def rank_device(device) -> str:
    """The device this process works on: `device` as given for a single process; under torch.distributed.run (one
    process per GPU) a CUDA device becomes cuda:LOCAL_RANK.  Call it BEFORE loading the pipeline / classifier."""
    from .generate import dist_env
    _, world, local = dist_env()
    if world > 1 and torch.device(device).type == "cuda":
        torch.cuda.set_device(local)
        return f"cuda:{local}"
    return str(device)


def UCE(pipe, classify, edit_concepts, debias_concepts, preserve_concepts, edit_scale, preserve_scale, lamb, save_dir,
        exp_name, max_diff, step_size, num_images_per_prompt, num_inference_steps, guidance_scale,
        desired_ratios=(0.5, 0.5), max_iterations=30, device="cuda:0", ratios_fn=None, algo: int = 0,
        embed_batch: Optional[int] = None):
    """Same positional signature as the reference's debias UCE() (:37); `desired_ratios`,
    `max_iterations`, `device` replace the module globals it reads; `ratios_fn` lets a test script
    the (unseeded, irreproducible) sampling step; `algo` / `embed_batch` as in edit.UCE."""
    from .generate import dist_env, init_distributed
    rank, world, local = dist_env()
    device = rank_device(device)
    handle = E.UceHandle.get(device)
    pdev = getattr(pipe, "device", None)
    if pdev is not None and torch.device(pdev).type == "cuda" and torch.device(pdev) != handle.device:
        raise RuntimeError(f"the pipeline lives on {pdev} but this rank edits on {handle.device}: load it on "
                           "debias.rank_device(device) (one process per GPU)")
    init_distributed(handle.device)
    modules = E.collect_uce_modules(pipe.unet)
    slab = E.WeightSlab.from_modules(modules, handle.device)
    embeds = E.last_token_embeddings(pipe, list(edit_concepts) + list(debias_concepts) + list(preserve_concepts),
                                     handle.device, batch_size=embed_batch)
    C_edit = torch.stack([embeds[e] for e in edit_concepts]).contiguous()
    C_deb = torch.stack([embeds[c] for c in debias_concepts]).contiguous()
    C_pres = torch.stack([embeds[p] for p in preserve_concepts]).contiguous() if preserve_concepts else None
    state = E.DebiasState(handle, slab, C_edit, C_deb, C_pres, edit_scale, preserve_scale, lamb, algo=algo,
                          keys=(list(edit_concepts), list(debias_concepts), list(preserve_concepts)))
    if hasattr(pipe, "to"):
        pipe = pipe.to(torch.bfloat16)                         # :90
    start_time = time.time()
    for iteration in range(max_iterations):
        if ratios_fn is not None:
            direction_scale = ratios_fn(iteration=iteration, slab=state.current)
        else:
            direction_scale = get_ratios(pipe, classify, state.current, edit_concepts, debias_concepts, desired_ratios,
                                         max_diff, num_images_per_prompt, num_inference_steps, guidance_scale,
                                         rank=rank, world=world, reduce_device=handle.device)
        if np.abs(direction_scale).max() == 0:                 # :110-112
            if rank == 0:
                print("All concepts are debiased")
            break
        state.step(direction_scale)
    end_time = time.time()
    # every rank holds the same weights (same direction_scale history, bit-repeatable solve): rank 0 writes
    path = E.save_uce_state(state.current, save_dir, exp_name) if rank == 0 else None
    if rank == 0:
        print(f"\n\nDebiased concepts using UCE\nModel edited in {end_time - start_time} seconds\n")
    return state.current, path
