"""Synthetic inputs of the right shape and geometry for benchmarks and self-checks: there are
no model weights, tokenizer vocabularies or datasets on the machines this runs on."""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np


def sd14_module_table() -> List[Tuple[str, int]]:
    """(module path, out_features) of SD-1.x's 32 cross-attention K/V projections
    (in_features 768), in `unet.named_modules()` order: down, up, mid."""
    spec = [("down_blocks.0", 2, 320), ("down_blocks.1", 2, 640), ("down_blocks.2", 2, 1280),
            ("up_blocks.1", 3, 1280), ("up_blocks.2", 3, 640), ("up_blocks.3", 3, 320), ("mid_block", 1, 1280)]
    rows = []
    for prefix, n_attn, width in spec:
        for a in range(n_attn):
            for proj in ("to_k", "to_v"):
                rows.append((f"{prefix}.attentions.{a}.transformer_blocks.0.attn2.{proj}", width))
    return rows


def sdxl_module_table() -> List[Tuple[str, int]]:
    """SDXL-base: 70 transformer blocks -> 140 projections (in_features 2048)."""
    spec = [("down_blocks.1", 2, 2, 640), ("down_blocks.2", 2, 10, 1280), ("up_blocks.0", 3, 10, 1280),
            ("up_blocks.1", 3, 2, 640), ("mid_block", 1, 10, 1280)]
    rows = []
    for prefix, n_attn, depth, width in spec:
        for a in range(n_attn):
            for t in range(depth):
                for proj in ("to_k", "to_v"):
                    rows.append((f"{prefix}.attentions.{a}.transformer_blocks.{t}.attn2.{proj}", width))
    return rows


def clip_like_embeddings(n: int, d: int, seed: int, norm: float = 28.0, cosine: float = 0.64) -> np.ndarray:
    """Last-token text embeddings with CLIP-like geometry: a shared direction plus isotropic
    noise, all rows of norm `norm`, mean pairwise cosine ~ `cosine`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.standard_normal(d)
    u /= np.linalg.norm(u)
    z = rng.standard_normal((n, d))
    z -= np.outer(z @ u, u)
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    return (norm * (math.sqrt(cosine) * u[None, :] + math.sqrt(1.0 - cosine) * z)).astype(np.float32)


def linear_default_weight(o: int, d: int, rng: np.random.Generator) -> np.ndarray:
    """nn.Linear's default init range U(-1/sqrt(d), 1/sqrt(d))."""
    bound = 1.0 / math.sqrt(d)
    return rng.uniform(-bound, bound, size=(o, d)).astype(np.float32)


# ---- prompt tables in the schema of the reference's data/coco_30k.csv -------------------------------------------------
# The real table (30 000 COCO captions) ships with the reference repository but does not travel to the GPU box; 71 of its
# records are committed as the data fixture tests/golden/coco30k_rows.csv (tools/make_coco_fixture.py), which pins the
# row walk of generate-images-sd.py:21-46 on real rows.  The generator below makes tables of any LENGTH for throughput runs.
_SUBJECTS = ["a man", "a woman", "a child", "two people", "a dog", "a cat", "a horse", "a bird", "a giraffe", "an elephant",
             "a bicycle", "a motorcycle", "a bus", "a train", "an airplane", "a boat", "a pizza", "a sandwich", "a cake",
             "a laptop", "a clock", "a vase", "a bench", "a kite", "a skateboard", "a surfboard", "a tennis racket",
             "a teddy bear", "a fire hydrant", "a stop sign"]
_VERBS = ["sitting on", "standing next to", "riding", "holding", "looking at", "parked in front of", "lying on",
          "walking past", "flying over", "placed on"]
_PLACES = ["a wooden table", "a city street", "a grassy field", "the beach", "a kitchen counter", "a snowy hill",
           "a parking lot", "a living room couch", "a river bank", "a brick wall", "a train station platform",
           "a park bench"]
_TAILS = ["", " at sunset", " on a cloudy day", " in black and white", " with mountains in the background",
          " next to a window", " under a blue sky", " at night"]


def coco_like_rows(n: int, seed: int = 0, first_case: int = 0):
    """`n` rows (case_number, source, prompt, evaluation_seed, coco_id) in the schema of the reference's
    data/coco_30k.csv (read by evalscripts/generate-images-sd.py:21-36): caption-like prompts from a small grammar,
    5-digit evaluation seeds, deterministic in `seed`.  For throughput runs of any length (captions of the same shape as
    the table's; the real records used for parity are tests/golden/coco30k_rows.csv)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = []
    for i in range(n):
        s, v, p, t = (int(rng.integers(len(x))) for x in (_SUBJECTS, _VERBS, _PLACES, _TAILS))
        prompt = f"{_SUBJECTS[s]} {_VERBS[v]} {_PLACES[p]}{_TAILS[t]}."
        rows.append((first_case + i, "coco-30k-synthetic", prompt[0].upper() + prompt[1:], int(rng.integers(10000, 100000)),
                     int(rng.integers(1, 600000))))
    return rows


def write_prompts_csv(path: str, n: int, seed: int = 0) -> str:
    import csv
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["case_number", "source", "prompt", "evaluation_seed", "coco_id"])
        w.writerows(coco_like_rows(n, seed))
    return path
