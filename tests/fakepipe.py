"""A weight-free stand-in for a diffusers StableDiffusionPipeline, for driving UCE() code.

Used (a) by tools/make_golden.py to run the REAL reference scripts in the build container
and (b) by the tests to run this repo's drop-in scripts on identical inputs.  It offers
exactly the surface the reference touches:
  pipe.unet.named_modules()            uce_sd_erase.py:17
  pipe.encode_prompt(prompt=..., ...)  uce_sd_erase.py:29-32   -> (Tensor[1,77,d], None)
  pipe.tokenizer(e, ...)               uce_sd_erase.py:34-39   -> {'attention_mask': [1,77]}
  pipe.tokenizer.model_max_length      uce_sd_erase.py:36
  pipe.to(...)                         uce_sd_erase.py:200, uce_sd_debias.py:90
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch import nn

MAX_LEN = 77


def prompt_embedding(prompt: str, d: int, norm: float = 28.0, cosine: float = 0.64) -> np.ndarray:
    """Deterministic CLIP-like last-token embedding for an arbitrary string: a shared
    direction (fixed seed) + a per-string direction seeded by crc32(prompt)."""
    ru = np.random.Generator(np.random.PCG64(20230828 + d))
    u = ru.standard_normal(d)
    u /= np.linalg.norm(u)
    rz = np.random.Generator(np.random.PCG64(zlib.crc32(prompt.encode("utf-8"))))
    z = rz.standard_normal(d)
    z -= (z @ u) * u
    z /= np.linalg.norm(z)
    return (norm * (math.sqrt(cosine) * u + math.sqrt(1 - cosine) * z)).astype(np.float32)


def token_count(prompt: str) -> int:
    """Fake tokenisation: one token per whitespace word + BOS + EOS, truncated to 77."""
    return min(len(prompt.split()) + 2, MAX_LEN)


class FakeTokenizer:
    model_max_length = MAX_LEN

    def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
        n = token_count(text)
        mask = torch.zeros(1, MAX_LEN, dtype=torch.long)
        mask[0, :n] = 1
        return {"attention_mask": mask, "input_ids": mask.clone()}


class _Node(nn.Module):
    pass


def build_unet(table: Sequence[Tuple[str, int]], d: int, rng: np.random.Generator,
               decoys: bool = True) -> nn.Module:
    """An nn.Module tree whose named_modules() yields `table`'s paths as bias-free
    nn.Linear(d, o) with nn.Linear-default-range weights, plus decoy modules the UCE name
    predicate must skip (attn1.*, attn2.to_q, attn2.to_out.0)."""
    root = _Node()

    def ensure(path: List[str]) -> nn.Module:
        node = root
        for p in path:
            if not hasattr(node, p):
                node.add_module(p, _Node())
            node = getattr(node, p)
        return node

    def linear(o: int, i: int) -> nn.Linear:
        lin = nn.Linear(i, o, bias=False)
        bound = 1.0 / math.sqrt(i)
        lin.weight.data = torch.from_numpy(rng.uniform(-bound, bound, size=(o, i)).astype(np.float32))
        return lin

    seen_blocks = set()
    for name, o in table:
        parts = name.split(".")
        parent = ensure(parts[:-1])
        parent.add_module(parts[-1], linear(o, d))
        blk = ".".join(parts[:-2])
        if decoys and blk not in seen_blocks:
            seen_blocks.add(blk)
            tb = ensure(parts[:-2])
            a1 = ensure(parts[:-2] + ["attn1"])
            a1.add_module("to_k", linear(8, 8))
            a1.add_module("to_v", linear(8, 8))
            a2 = ensure(parts[:-1])
            a2.add_module("to_q", linear(8, 8))
            a2.add_module("to_out", nn.ModuleList([linear(8, 8)]))
    return root


class FakePipe:
    """encode_prompt places the string's embedding at the reference's last-token index and
    DIFFERENT vectors at every other position, so an off-by-one index changes the result."""

    def __init__(self, unet: nn.Module, d: int, embeddings: Dict[str, np.ndarray] | None = None):
        self.unet = unet
        self.d = d
        self.tokenizer = FakeTokenizer()
        self.embeddings = dict(embeddings or {})
        rp = np.random.Generator(np.random.PCG64(77))
        self._pos = rp.standard_normal(d).astype(np.float32)
        self.encode_calls: List[str] = []

    def embedding(self, prompt: str) -> np.ndarray:
        if prompt not in self.embeddings:
            self.embeddings[prompt] = prompt_embedding(prompt, self.d)
        return self.embeddings[prompt]

    def encode_prompt(self, prompt=None, device=None, num_images_per_prompt=1,
                      do_classifier_free_guidance=False, **kw):
        self.encode_calls.append(prompt)
        e = torch.from_numpy(self.embedding(prompt))
        idx = token_count(prompt) - 2
        pos = torch.arange(MAX_LEN, dtype=torch.float32) - idx
        t = e[None, :] + 0.5 * pos[:, None] * torch.from_numpy(self._pos)[None, :]
        return t[None].contiguous(), None

    def to(self, *a, **k):
        return self

    def set_progress_bar_config(self, **k):
        pass


def uce_weights(unet: nn.Module) -> List[Tuple[str, torch.Tensor]]:
    out = []
    for name, m in unet.named_modules():
        if "attn2" in name and (name.endswith("to_v") or name.endswith("to_k")):
            out.append((name, m.weight.detach().clone()))
    return out


# ----------------------------------------------------------------------------------------------------
# FLUX stand-ins (uce_flux_edit.py:12-66): the reference calls DiffusionPipeline.from_pretrained twice -
# once for the transformer (text encoders = None), once for the text side (transformer = None).
# ----------------------------------------------------------------------------------------------------
FLUX_T5_DIM, FLUX_POOL_DIM = 4096, 768


def build_flux_transformer(out_rows: int, rng: np.random.Generator) -> nn.Module:
    """named_modules() yields `context_embedder` (Linear 4096 -> out_rows, bias) and
    `time_text_embed.text_embedder.linear_1` (Linear 768 -> out_rows, bias) plus decoys the name predicate
    (uce_flux_edit.py:25) must skip.  out_rows stands in for 3072 (rows are independent)."""
    root = _Node()

    def linear(o, i):
        lin = nn.Linear(i, o, bias=True)
        bound = 1.0 / math.sqrt(i)
        lin.weight.data = torch.from_numpy(rng.uniform(-bound, bound, size=(o, i)).astype(np.float32))
        lin.bias.data = torch.from_numpy(rng.uniform(-bound, bound, size=(o,)).astype(np.float32))
        return lin

    root.add_module("x_embedder", linear(8, 8))
    root.add_module("context_embedder", linear(out_rows, FLUX_T5_DIM))
    tte = _Node()
    te = _Node()
    te.add_module("linear_1", linear(out_rows, FLUX_POOL_DIM))
    te.add_module("linear_2", linear(8, 8))
    tte.add_module("text_embedder", te)
    tte.add_module("guidance_embedder", _Node())
    root.add_module("time_text_embed", tte)
    return root


class FakeFluxTransformerPipe:
    def __init__(self, transformer: nn.Module):
        self.transformer = transformer

    def to(self, *a, **k):
        return self


class FakeFluxTextPipe:
    """encode_prompt -> (T5 states [1, L, 4096] with the string's embedding at the reference's last-token index
    and different vectors elsewhere, pooled CLIP [1, 768], text ids)."""

    def __init__(self):
        rp = np.random.Generator(np.random.PCG64(78))
        self._pos = rp.standard_normal(FLUX_T5_DIM).astype(np.float32)
        self.encode_calls: List[str] = []
        outer = self

        class _Tok2:
            def __call__(self, text, padding=None, max_length=None, return_overflowing_tokens=False, truncation=True,
                         return_length=False, return_tensors=None):
                n = min(len(text.split()) + 2, max_length)
                mask = torch.zeros(1, max_length, dtype=torch.long)
                mask[0, :n] = 1
                return {"attention_mask": mask, "input_ids": mask.clone()}

        self.tokenizer_2 = _Tok2()

    @staticmethod
    def t5_embedding(prompt: str) -> np.ndarray:
        return prompt_embedding("t5:" + prompt, FLUX_T5_DIM, norm=12.0, cosine=0.5)

    @staticmethod
    def pooled_embedding(prompt: str) -> np.ndarray:
        return prompt_embedding("pool:" + prompt, FLUX_POOL_DIM, norm=28.0, cosine=0.64)

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt=1, max_sequence_length=512, **kw):
        self.encode_calls.append(prompt)
        e = torch.from_numpy(self.t5_embedding(prompt))
        idx = min(len(prompt.split()) + 2, max_sequence_length) - 2
        pos = torch.arange(max_sequence_length, dtype=torch.float32) - idx
        t = e[None, :] + 0.25 * pos[:, None] * torch.from_numpy(self._pos)[None, :]
        pooled = torch.from_numpy(self.pooled_embedding(prompt))[None]
        return t[None].contiguous(), pooled, None

    def to(self, *a, **k):
        return self


# ----------------------------------------------------------------------------------------------------
# HiDream stand-ins (uce_hidream_edit.py:14-122): three from_pretrained calls - transformer only, Llama text
# side, T5 text side.  caption_projection.<i>.linear is edited with the Llama layer llama_layers[i]'s hidden
# state at the last token; the last projection with the T5 state.
# ----------------------------------------------------------------------------------------------------
HIDREAM_DIM = 4096


class _Cfg:
    def __init__(self, llama_layers):
        self.llama_layers = list(llama_layers)


def build_hidream_transformer(out_rows: int, llama_layers: Sequence[int], rng: np.random.Generator, bias: bool = False
                              ) -> nn.Module:
    root = _Node()
    cp = nn.ModuleList()
    bound = 1.0 / math.sqrt(HIDREAM_DIM)
    for _ in range(len(llama_layers) + 1):                      # + the T5 projection
        blk = _Node()
        lin = nn.Linear(HIDREAM_DIM, out_rows, bias=bias)
        lin.weight.data = torch.from_numpy(rng.uniform(-bound, bound, size=(out_rows, HIDREAM_DIM)).astype(np.float32))
        if bias:
            lin.bias.data = torch.from_numpy(rng.uniform(-bound, bound, size=(out_rows,)).astype(np.float32))
        blk.add_module("linear", lin)
        cp.append(blk)
    root.add_module("caption_projection", cp)
    root.add_module("x_embedder", nn.Linear(8, 8))
    root.config = _Cfg(llama_layers)
    return root


class FakeHiDreamTokenizer:
    def __init__(self, model_max_length: int):
        self.model_max_length = model_max_length

    def __call__(self, text, padding=None, max_length=None, truncation=True, add_special_tokens=True, return_tensors=None):
        n = min(len(text.split()) + 2, max_length)
        mask = torch.zeros(1, max_length, dtype=torch.long)
        mask[0, :n] = 1
        return {"attention_mask": mask, "input_ids": mask.clone()}


class FakeHiDreamTextPipe:
    """`_get_llama3_prompt_embeds` -> [n_layers, 1, L, 4096]; `_get_t5_prompt_embeds` -> [1, L, 4096]; each with the
    string's (family-specific) embedding at the last-token index and different vectors elsewhere."""
    N_LAYERS = 6

    def __init__(self, tokenizer_4=None):
        self.tokenizer_4 = tokenizer_4
        self.tokenizer_3 = FakeHiDreamTokenizer(512)
        rp = np.random.Generator(np.random.PCG64(79))
        self._pos = torch.from_numpy(rp.standard_normal(HIDREAM_DIM).astype(np.float32))
        self.calls: List[str] = []

    @staticmethod
    def family_embedding(prompt: str, family: str) -> np.ndarray:
        return prompt_embedding(f"{family}:" + prompt, HIDREAM_DIM, norm=10.0, cosine=0.55)

    def _seq(self, e: np.ndarray, idx: int, L: int) -> torch.Tensor:
        pos = torch.arange(L, dtype=torch.float32) - idx
        return torch.from_numpy(e)[None, :] + 0.2 * pos[:, None] * self._pos[None, :]

    def _get_llama3_prompt_embeds(self, prompt, max_sequence_length, device, dtype):
        self.calls.append("llama:" + prompt)
        L = min(max_sequence_length, self.tokenizer_4.model_max_length)
        idx = min(len(prompt.split()) + 2, L) - 2
        return torch.stack([self._seq(self.family_embedding(prompt, f"llama{layer}"), idx, L)[None]
                            for layer in range(self.N_LAYERS)])

    def _get_t5_prompt_embeds(self, prompt, max_sequence_length, device, dtype):
        self.calls.append("t5:" + prompt)
        L = min(max_sequence_length, self.tokenizer_3.model_max_length)
        idx = min(len(prompt.split()) + 2, L) - 2
        return self._seq(self.family_embedding(prompt, "t5"), idx, L)[None]

    def to(self, *a, **k):
        return self
