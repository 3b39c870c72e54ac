"""HiDream variant of the closed-form edit: the drop-in for `UCE` of the reference's
trainscripts/uce_hidream_edit.py:14-178.

The edited modules are `transformer.caption_projection.<i>.linear` (name predicate :31: 'caption_projection' and
'linear' in the name).  Module i has ITS OWN embedding family: the hidden state of Llama layer
`transformer.config.llama_layers[i]` at the last real token (`attention_mask.sum() - 2` of tokenizer_4, :66-79), the
last projection the T5 state (tokenizer_3, :100-115).  (The reference walks `modules + modules[-1:]` and resets the
index of the extra pass to the last module (:135-137): that pass recomputes the same weight from the same inputs, so
one pass per module is the whole result.)  Per module the arithmetic is the shared closed form - `uce_edit` through
`flux.edit_linear_with_bias`, which also covers a bias should a checkpoint carry one.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import edit as E
from . import lib as _lib
from .flux import edit_linear_with_bias


def is_hidream_uce_module(name: str) -> bool:
    """uce_hidream_edit.py:31."""
    return "caption_projection" in name and "linear" in name


def collect_hidream_modules(transformer: torch.nn.Module) -> List[Tuple[str, torch.nn.Module]]:
    return [(n, m) for n, m in transformer.named_modules() if is_hidream_uce_module(n)]


def _last_token(tokenizer, text: str, max_sequence_length: int) -> int:
    mask = tokenizer(text, padding="max_length", max_length=min(max_sequence_length, tokenizer.model_max_length),
                     truncation=True, add_special_tokens=True, return_tensors="pt")["attention_mask"]
    return int(mask.sum()) - 2


def hidream_embeddings(llama_pipe, t5_pipe, prompts: Sequence[str], llama_layers: Sequence[int], device,
                       max_sequence_length: int, torch_dtype=torch.float32) -> Dict[str, List[torch.Tensor]]:
    """Per unique string: [state of Llama layer l at the last token for l in llama_layers] + [T5 state] - :59-116."""
    out: Dict[str, List[torch.Tensor]] = {}
    for e in prompts:
        if e in out:
            continue
        t = llama_pipe._get_llama3_prompt_embeds(e, max_sequence_length, device, torch_dtype)   # [layers, 1, L, d]
        idx = _last_token(llama_pipe.tokenizer_4, e, max_sequence_length)
        out[e] = [t[layer][0, idx, :].to(device=device, dtype=torch.float32) for layer in llama_layers]
    for e in out:
        t = t5_pipe._get_t5_prompt_embeds(e, max_sequence_length, device, torch_dtype)           # [1, L, d]
        idx = _last_token(t5_pipe.tokenizer_3, e, max_sequence_length)
        out[e].append(t[0, idx, :].to(device=device, dtype=torch.float32))
    return out


def UCE(model_id, edit_concepts, guide_concepts, preserve_concepts, erase_scale, preserve_scale, lamb, save_dir,
        exp_name, torch_dtype=torch.float32, device="cuda:0", max_sequence_length=128,
        load_transformer: Optional[Callable] = None, load_llama: Optional[Callable] = None,
        load_t5: Optional[Callable] = None, algo: int = _lib.ALGO_AUTO):
    """Same positional signature as the reference's UCE (uce_hidream_edit.py:14).  The three loaders replace its
    three `HiDreamImagePipeline.from_pretrained` calls (defaults need diffusers + transformers + the checkpoints)."""
    from safetensors.torch import save_file
    if load_transformer is None or load_llama is None or load_t5 is None:
        try:
            from diffusers import HiDreamImagePipeline  # type: ignore
            from transformers import LlamaForCausalLM, PreTrainedTokenizerFast
        except ImportError as err:
            raise RuntimeError("uce_hidream_edit needs diffusers + the HiDream / Llama checkpoints (or explicit loaders)") from err
        none_text = dict(tokenizer_4=None, text_encoder_4=None, tokenizer_3=None, text_encoder_3=None, tokenizer_2=None,
                         text_encoder_2=None, tokenizer=None, text_encoder=None)
        load_transformer = load_transformer or (lambda: HiDreamImagePipeline.from_pretrained(
            model_id, vae=None, torch_dtype=torch_dtype, **none_text))

        def _llama():
            tok = PreTrainedTokenizerFast.from_pretrained("meta-llama/Meta-Llama-3.1-8B-Instruct")
            enc = LlamaForCausalLM.from_pretrained("meta-llama/Meta-Llama-3.1-8B-Instruct", output_hidden_states=True,
                                                   output_attentions=True, torch_dtype=torch_dtype)
            kw = dict(none_text, tokenizer_4=tok, text_encoder_4=enc)
            return HiDreamImagePipeline.from_pretrained(model_id, transformer=None, vae=None, torch_dtype=torch_dtype,
                                                        **kw).to(device)

        def _t5():
            kw = {k: v for k, v in none_text.items() if k not in ("tokenizer_3", "text_encoder_3")}
            return HiDreamImagePipeline.from_pretrained(model_id, transformer=None, vae=None, torch_dtype=torch_dtype,
                                                        **kw).to(device)
        load_llama, load_t5 = load_llama or _llama, load_t5 or _t5
    handle = E.UceHandle.get(device)
    pipe = load_transformer()
    modules = [(n, m.weight.detach().to(handle.device, torch.float32),
                None if m.bias is None else m.bias.detach().to(handle.device, torch.float32))
               for n, m in collect_hidream_modules(pipe.transformer)]
    llama_layers = list(pipe.transformer.config.llama_layers)
    pipe = None
    if len(modules) != len(llama_layers) + 1:
        raise ValueError(f"{len(modules)} caption projections for {len(llama_layers)} Llama layers + T5")
    embeds = hidream_embeddings(load_llama(), load_t5(), list(edit_concepts) + list(guide_concepts) +
                                list(preserve_concepts), llama_layers, handle.device, max_sequence_length, torch_dtype)
    start_time = time.time()
    state = {}
    for i, (name, W, b) in enumerate(modules):
        table = {p: v[i] for p, v in embeds.items()}
        C, G, s = E.concept_matrices(table, edit_concepts, guide_concepts, preserve_concepts, erase_scale,
                                     preserve_scale, handle.device)
        state[name + ".weight"] = edit_linear_with_bias(handle, W, b, C, G, s, lamb, algo).to(torch_dtype).cpu()
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, exp_name + ".safetensors")
    save_file(state, path)
    end_time = time.time()
    print(f"\n\nErased concepts using UCE\nModel edited in {end_time - start_time} seconds\n")
    return state, path
