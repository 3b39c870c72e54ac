"""FLUX variant of the closed-form edit: the drop-in for `UCE` of the reference's
trainscripts/uce_flux_edit.py:12-122.

What differs from the Stable-Diffusion scripts:
  * the edited modules are `transformer.context_embedder` (Linear 4096 -> 3072) and
    `transformer.time_text_embed.text_embedder.linear_1` (Linear 768 -> 3072), found by the name predicate of
    uce_flux_edit.py:25, and they HAVE A BIAS;
  * each module has its own embedding family: the T5 state at the last real token (`attention_mask.sum() - 2` of
    tokenizer_2, :52-63) for context_embedder, the pooled CLIP vector for the text embedder (:94-96: the family
    whose width matches the module's in_features);
  * the guide outputs are `module(t_emb)` = W g + b (:80-83), so the closed form picks up a rank-1 term:
        W_new = (lamb W + sum_i s_i (W g_i + b) c_i^T) A^-1 = W + W Delta + b u^T ,   u = A^-1 sum_i s_i c_i
    (sum over edit AND preserve concepts; the bias itself is not edited).  With R = K^-1 C the dual factor of ALL
    rows, u = R^T 1 (push-through identity), so the same HIP path serves: `uce_edit` for W + W Delta and one extra
    `uce_dual_factors` call (or, for N >= d, `uce_gram` + a d x d solve) for u.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import edit as E
from . import lib as _lib


def is_flux_uce_module(name: str) -> bool:
    """uce_flux_edit.py:25."""
    return "context_embedder" in name or "text_embedder.linear_1" in name


def collect_flux_modules(transformer: torch.nn.Module) -> List[Tuple[str, torch.nn.Module]]:
    return [(n, m) for n, m in transformer.named_modules() if is_flux_uce_module(n)]


def flux_embeddings(pipe, prompts: Sequence[str], device, max_sequence_length: int
                    ) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """Per unique string: (T5 last-token state [d_t5], pooled CLIP [d_pool]) - uce_flux_edit.py:44-66."""
    out: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    for e in prompts:
        if e in out:
            continue
        t_emb = pipe.encode_prompt(prompt=e, prompt_2=None, device=device, num_images_per_prompt=1,
                                   max_sequence_length=max_sequence_length)
        mask = pipe.tokenizer_2(e, padding="max_length", max_length=max_sequence_length,
                                return_overflowing_tokens=False, truncation=True, return_length=False,
                                return_tensors="pt")["attention_mask"]
        idx = int(mask.sum()) - 2
        out[e] = (t_emb[0][0, idx, :].to(device=device, dtype=torch.float32),
                  t_emb[1][0].to(device=device, dtype=torch.float32))
    return out


def bias_direction(handle: E.UceHandle, C: torch.Tensor, s: torch.Tensor, lamb: float) -> torch.Tensor:
    """u = A^-1 sum_i s_i c_i, A = lamb I + C^T S C, as fp32 [d]."""
    N, d = C.shape
    if (N + 63) // 64 * 64 < d:
        _, R = handle.dual_factors(C, C, s, lamb)              # every row as an "edit" row: R = K^-1 C, all N rows
        return R.double().sum(dim=0).float()
    # N >= d: the primal system through the library's own Cholesky + triangular solves with a NARROW right-hand side
    # (uce_solve_rhs: 64 columns, the vector rides in column 0 - not a dense d x d solve for one column)
    A, _ = handle.gram(C, None, s, lamb)                       # f64 [d, d]
    B = torch.zeros(d, 64, dtype=torch.float64, device=handle.device)
    B[:, 0] = (C.double() * s.double()[:, None]).sum(dim=0)
    u = handle.solve_rhs(A, B)[:, 0].contiguous()
    handle.status()
    return u


def edit_linear_with_bias(handle: E.UceHandle, W: torch.Tensor, b: Optional[torch.Tensor], C: torch.Tensor,
                          G: torch.Tensor, s: torch.Tensor, lamb: float, algo: int = _lib.ALGO_AUTO) -> torch.Tensor:
    """W_new for one Linear(d -> o) with optional bias; C [N, d] (edit rows first), G [N_e, d], s [N]."""
    C, G, s, n_edit = E.drop_zero_scale_rows(C, G, s, G.shape[0])
    if C.shape[0] == 0:
        return W.clone()
    Wd = W.to(device=handle.device, dtype=torch.float32).contiguous()
    if n_edit > 0:
        out = handle.edit(C, G, s, lamb, Wd, algo=algo, check=True)
    else:
        out = Wd.clone()                                       # preserve-only: Delta = 0
    if b is not None:
        u = bias_direction(handle, C, s, lamb)
        out.addr_(b.to(device=handle.device, dtype=torch.float32), u)      # + b u^T (rank 1)
    return out


def UCE(model_id, edit_concepts, guide_concepts, preserve_concepts, erase_scale, preserve_scale, lamb, save_dir,
        exp_name, torch_dtype=torch.float32, device="cuda:0", max_sequence_length=512,
        load_transformer: Optional[Callable] = None, load_text: Optional[Callable] = None, algo: int = _lib.ALGO_AUTO):
    """Same positional signature as the reference's UCE (uce_flux_edit.py:12).  `load_transformer` / `load_text`
    replace the two `DiffusionPipeline.from_pretrained` calls (defaults: diffusers, when it is installed)."""
    from safetensors.torch import save_file
    if load_transformer is None or load_text is None:
        try:
            from diffusers import DiffusionPipeline  # type: ignore
        except ImportError as err:
            raise RuntimeError("uce_flux_edit needs diffusers + the FLUX checkpoint (or explicit loaders)") from err
        load_transformer = load_transformer or (lambda: DiffusionPipeline.from_pretrained(
            model_id, vae=None, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
            torch_dtype=torch_dtype, safety_checker=None))
        load_text = load_text or (lambda: DiffusionPipeline.from_pretrained(
            model_id, vae=None, transformer=None, torch_dtype=torch_dtype, safety_checker=None).to(device))
    handle = E.UceHandle.get(device)
    pipe = load_transformer()
    modules = [(n, m.weight.detach().to(handle.device, torch.float32),
                None if m.bias is None else m.bias.detach().to(handle.device, torch.float32))
               for n, m in collect_flux_modules(pipe.transformer)]
    pipe = None
    pipe = load_text()
    start_time = time.time()
    embeds = flux_embeddings(pipe, list(edit_concepts) + list(guide_concepts) + list(preserve_concepts), handle.device,
                             max_sequence_length)
    pipe = None
    state = {}
    for name, W, b in modules:
        d = W.shape[1]
        fam = None
        for k in (0, 1):                                       # the family whose width fits the module (:80-83, :94-96)
            if next(iter(embeds.values()))[k].shape[0] == d:
                fam = k
        if fam is None:
            raise ValueError(f"{name}: no embedding family of width {d}")
        table = {p: v[fam] for p, v in embeds.items()}
        C, G, s = E.concept_matrices(table, edit_concepts, guide_concepts, preserve_concepts, erase_scale,
                                     preserve_scale, handle.device)
        state[name + ".weight"] = edit_linear_with_bias(handle, W, b, C, G, s, lamb, algo).to(torch_dtype).cpu()
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, exp_name + ".safetensors")
    save_file(state, path)
    end_time = time.time()
    print(f"\n\nErased concepts using UCE\nModel edited in {end_time - start_time} seconds\n")
    return state, path
