"""CPU oracle for the UCE hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (uce_amd.*) never does and fails loudly without the HIP library.

Pinning status
--------------
* closed-form edit (erase / debias): PINNED.  `tools/make_golden.py` imports the real
  reference scripts from /root/reference in the build container (stubbed `diffusers`, fake
  pipe), and the fixtures it writes under tests/golden/ hold the reference's own outputs.
  tests/test_oracle.py checks every function here against those fixtures.
* cross-attention: the reference delegates to diffusers' AttnProcessor2_0 ->
  torch.nn.functional.scaled_dot_product_attention (diffusers==0.33.0, requirements.txt:1;
  not vendored, not installed).  The restatement below is pinned against torch's own CPU
  SDPA (fixtures in tests/golden/sdpa_*.npz); the diffusers wrapper around it (reshape to
  [B,H,L,dh], scale 1/sqrt(dh), no mask, no dropout) is restated from the published
  source and is otherwise "parity unpinned" (the reference has no test for it).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# module discovery  (trainscripts/uce_sd_erase.py:15-22, uce_sd_debias.py:39-46)
# --------------------------------------------------------------------------------------

def is_uce_module(name: str) -> bool:
    """Name predicate of uce_sd_erase.py:18."""
    return "attn2" in name and (name.endswith("to_v") or name.endswith("to_k"))


def sd14_module_table() -> List[Tuple[str, int]]:
    """(module path, out_features) of the 32 SD-1.4 cross-attention K/V projections in
    `unet.named_modules()` order (down, up, mid: registration order of diffusers'
    UNet2DConditionModel).  in_features = 768 for all of them."""
    rows: List[Tuple[str, int]] = []

    def block(prefix: str, n_attn: int, width: int) -> None:
        for a in range(n_attn):
            base = f"{prefix}.attentions.{a}.transformer_blocks.0.attn2"
            rows.append((base + ".to_k", width))
            rows.append((base + ".to_v", width))

    block("down_blocks.0", 2, 320)
    block("down_blocks.1", 2, 640)
    block("down_blocks.2", 2, 1280)
    block("up_blocks.1", 3, 1280)
    block("up_blocks.2", 3, 640)
    block("up_blocks.3", 3, 320)
    block("mid_block", 1, 1280)
    return rows


def sdxl_module_table() -> List[Tuple[str, int]]:
    """SDXL-base: 70 transformer blocks -> 140 projections, in_features 2048
    (640 x 20 modules, 1280 x 120 modules)."""
    rows: List[Tuple[str, int]] = []

    def block(prefix: str, n_attn: int, depth: int, width: int) -> None:
        for a in range(n_attn):
            for t in range(depth):
                base = f"{prefix}.attentions.{a}.transformer_blocks.{t}.attn2"
                rows.append((base + ".to_k", width))
                rows.append((base + ".to_v", width))

    block("down_blocks.1", 2, 2, 640)
    block("down_blocks.2", 2, 10, 1280)
    block("up_blocks.0", 3, 10, 1280)
    block("up_blocks.1", 3, 2, 640)
    block("mid_block", 1, 10, 1280)
    return rows


# --------------------------------------------------------------------------------------
# last-token index  (uce_sd_erase.py:34-42)
# --------------------------------------------------------------------------------------

def last_token_index(attention_mask_sum: int) -> int:
    """`attention_mask.sum() - 2` (uce_sd_erase.py:34-39): the last real token before EOS;
    '' (BOS+EOS only) -> 0 = the BOS position; >75 tokens -> truncation gives 77-2 = 75."""
    return int(attention_mask_sum) - 2


# --------------------------------------------------------------------------------------
# closed-form edit, reference op order  (uce_sd_erase.py:45-82)
# --------------------------------------------------------------------------------------

def uce_edit_ref(
    weights: Sequence[torch.Tensor],
    edit: Sequence[torch.Tensor],
    guide: Sequence[torch.Tensor],
    preserve: Sequence[torch.Tensor],
    erase_scale: float,
    preserve_scale: float,
    lamb: float,
    dtype: torch.dtype = torch.float32,
) -> List[torch.Tensor]:
    """Restatement of the hot loop of UCE() in the reference's own op order.

    weights  : per-module W_old [o, d]
    edit     : per edit concept c_i   [1, d]   (list order = CLI order; duplicates count twice)
    guide    : per edit concept g_i   [1, d]
    preserve : per preserve concept p [1, d]
    Returns the per-module W_new.
    """
    out: List[torch.Tensor] = []
    for w_old in weights:
        w_old = w_old.to(dtype)
        d = w_old.shape[1]
        # uce_sd_erase.py:52-53  v* = module(t_emb) = F.linear(t_emb, W)
        v_guide = [torch.nn.functional.linear(g.to(dtype), w_old) for g in guide]
        v_pres = [torch.nn.functional.linear(p.to(dtype), w_old) for p in preserve]
        mat1 = lamb * w_old                                    # :61
        mat2 = lamb * torch.eye(d, dtype=dtype)                # :63
        for c, v in zip(edit, v_guide):                        # :66-71
            c_i = c.to(dtype).T
            v_i_star = v.T
            mat1 += erase_scale * (v_i_star @ c_i.T)
            mat2 += erase_scale * (c_i @ c_i.T)
        for p, v in zip(preserve, v_pres):                     # :74-79
            c_i = p.to(dtype).T
            v_i_star = v.T
            mat1 += preserve_scale * (v_i_star @ c_i.T)
            mat2 += preserve_scale * (c_i @ c_i.T)
        out.append(mat1 @ torch.inverse(mat2.float()).to(dtype))   # :82
    return out


def uce_edit_exact64(
    weights: Sequence[torch.Tensor],
    edit: Sequence[torch.Tensor],
    guide: Sequence[torch.Tensor],
    preserve: Sequence[torch.Tensor],
    erase_scale: float,
    preserve_scale: float,
    lamb: float,
) -> List[torch.Tensor]:
    """The same formula (uce_sd_erase.py:56-82) evaluated in float64 with a solve instead
    of an explicit inverse: the arbiter for the three-way acceptance protocol of
    SURVEY.md section 7 (eps_ref, eps_build, relF(build, ref))."""
    C = torch.cat([e.double() for e in edit] + [p.double() for p in preserve], 0)
    G = torch.cat([g.double() for g in guide] + [p.double() for p in preserve], 0)
    s = torch.tensor([erase_scale] * len(edit) + [preserve_scale] * len(preserve),
                     dtype=torch.float64)
    d = C.shape[1]
    A = lamb * torch.eye(d, dtype=torch.float64) + C.T @ (s[:, None] * C)
    out = []
    for w in weights:
        w = w.double()
        mat1 = lamb * w + (w @ G.T) @ (s[:, None] * C)
        out.append(torch.linalg.solve(A, mat1.T).T)
    return out


# --------------------------------------------------------------------------------------
# FLUX variant (uce_flux_edit.py:71-113): the edited nn.Linear modules HAVE a bias, so the guide
# outputs are v* = W g + b and W_new = W + W Delta + b u^T with u = A^-1 sum_i s_i c_i (the bias
# itself is not edited).  One embedding family per module (T5 last token for context_embedder,
# pooled CLIP for time_text_embed.text_embedder.linear_1).
# --------------------------------------------------------------------------------------

def uce_edit_bias_ref(w_old: torch.Tensor, bias: torch.Tensor, edit: Sequence[torch.Tensor],
                      guide: Sequence[torch.Tensor], preserve: Sequence[torch.Tensor], erase_scale: float,
                      preserve_scale: float, lamb: float, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """One module in the reference's op order (uce_flux_edit.py:86-113)."""
    w_old, bias = w_old.to(dtype), bias.to(dtype)
    d = w_old.shape[1]
    lin = lambda t: torch.nn.functional.linear(t.to(dtype), w_old, bias)      # module(t_emb), :81
    mat1 = lamb * w_old
    mat2 = lamb * torch.eye(d, dtype=dtype)
    for c, g in zip(edit, guide):
        c_i, v = c.to(dtype).T, lin(g).T
        mat1 += erase_scale * (v @ c_i.T)
        mat2 += erase_scale * (c_i @ c_i.T)
    for p_ in preserve:
        c_i, v = p_.to(dtype).T, lin(p_).T
        mat1 += preserve_scale * (v @ c_i.T)
        mat2 += preserve_scale * (c_i @ c_i.T)
    return mat1 @ torch.inverse(mat2.float()).to(dtype)


def uce_edit_bias_exact64(w_old: torch.Tensor, bias: torch.Tensor, edit, guide, preserve, erase_scale: float,
                          preserve_scale: float, lamb: float) -> torch.Tensor:
    C = torch.cat([e.double() for e in edit] + [p_.double() for p_ in preserve], 0)
    G = torch.cat([g.double() for g in guide] + [p_.double() for p_ in preserve], 0)
    s = torch.tensor([erase_scale] * len(edit) + [preserve_scale] * len(preserve), dtype=torch.float64)
    d = C.shape[1]
    w, b = w_old.double(), bias.double()
    A = lamb * torch.eye(d, dtype=torch.float64) + C.T @ (s[:, None] * C)
    V = G @ w.T + b[None, :]                                   # [N, o] guide outputs incl. bias
    mat1 = lamb * w + V.T @ (s[:, None] * C)
    return torch.linalg.solve(A, mat1.T).T


# --------------------------------------------------------------------------------------
# debias  (uce_sd_debias.py:95-141)
# --------------------------------------------------------------------------------------

def uce_debias_ref(
    weights: Sequence[torch.Tensor],
    edit: Sequence[torch.Tensor],
    debias: Sequence[torch.Tensor],
    preserve: Sequence[torch.Tensor],
    direction_scales: Sequence[np.ndarray],
    edit_scale: float,
    preserve_scale: float,
    lamb: float,
    dtype: torch.dtype = torch.float32,
) -> List[torch.Tensor]:
    """Restatement of the iterative loop of uce_sd_debias.py:95-141 given a scripted
    sequence of `direction_scale` matrices ([N_edit, N_debias] float64 each, what
    get_ratios :14-35 returns).  Keeps the reference's quirks: the drift is added IN PLACE
    to the cached guide output (:124-126) so it accumulates over iterations; every
    iteration re-solves from W_old; an all-zero direction_scale stops the loop (:110-112)."""
    weights = [w.to(dtype) for w in weights]
    # :68-88 guide outputs for edit + debias + preserve, per module
    v_edit = [[torch.nn.functional.linear(e.to(dtype), w) for w in weights] for e in edit]
    v_deb = [[torch.nn.functional.linear(b.to(dtype), w) for w in weights] for b in debias]
    v_pres = [[torch.nn.functional.linear(p.to(dtype), w) for w in weights] for p in preserve]
    out = [w.clone() for w in weights]
    for direction_scale in direction_scales:
        if np.abs(direction_scale).max() == 0:                      # :110-112
            break
        for m, w_old in enumerate(weights):
            d = w_old.shape[1]
            mat1 = lamb * w_old
            mat2 = lamb * torch.eye(d, dtype=dtype)
            for idx, e in enumerate(edit):
                c_i = e.to(dtype).T
                v_i_star = v_edit[idx][m]
                for i in range(len(debias)):
                    v_i_star += direction_scale[idx][i] * v_deb[i][m]   # in place, :126
                v_i_star = v_i_star.T
                mat1 += edit_scale * (v_i_star @ c_i.T)
                mat2 += edit_scale * (c_i @ c_i.T)
            for k, p in enumerate(preserve):
                c_i = p.to(dtype).T
                v_i_star = v_pres[k][m].T
                mat1 += preserve_scale * (v_i_star @ c_i.T)
                mat2 += preserve_scale * (c_i @ c_i.T)
            out[m] = mat1 @ torch.inverse(mat2.float()).to(dtype)
    return out


def uce_debias_ref_keyed(
    weights: Sequence[torch.Tensor],
    embeds: Dict[str, torch.Tensor],
    edit: Sequence[str],
    debias: Sequence[str],
    preserve: Sequence[str],
    direction_scales: Sequence[np.ndarray],
    edit_scale: float,
    preserve_scale: float,
    lamb: float,
    dtype: torch.dtype = torch.float32,
) -> List[torch.Tensor]:
    """uce_sd_debias.py:68-140 with the reference's STRING-KEYED caches: `uce_guide_outputs` holds ONE tensor per unique
    string and per module (:69-88), and :124-126 adds the drift to it IN PLACE.  So a string listed twice among the edit
    concepts drifts twice per iteration (the second visit starts from what the first left), a string that is both an edit
    and a preserve concept preserves the DRIFTED output (:132), and a debias concept that is also an edit concept is itself
    rescaled / shifted and then used as the direction of the later edit concepts.  `embeds`: {string: [1, d]}."""
    weights = [w.to(dtype) for w in weights]
    v: Dict[str, List[torch.Tensor]] = {}
    for g in list(edit) + list(debias) + list(preserve):          # :69-88
        if g in v:
            continue
        v[g] = [torch.nn.functional.linear(embeds[g].to(dtype), w) for w in weights]
    out = [w.clone() for w in weights]
    for direction_scale in direction_scales:
        if np.abs(direction_scale).max() == 0:                      # :110-112
            break
        for m, w_old in enumerate(weights):
            d = w_old.shape[1]
            mat1 = lamb * w_old
            mat2 = lamb * torch.eye(d, dtype=dtype)
            for idx, e in enumerate(edit):                          # :120-130
                c_i = embeds[e].to(dtype).T
                v_i_star = v[e][m]
                for i, concept in enumerate(debias):
                    v_i_star += direction_scale[idx][i] * v[concept][m]   # in place on the cached tensor, :126
                v_i_star = v_i_star.T
                mat1 += edit_scale * (v_i_star @ c_i.T)
                mat2 += edit_scale * (c_i @ c_i.T)
            for p in preserve:                                      # :133-138
                c_i = embeds[p].to(dtype).T
                v_i_star = v[p][m].T
                mat1 += preserve_scale * (v_i_star @ c_i.T)
                mat2 += preserve_scale * (c_i @ c_i.T)
            out[m] = mat1 @ torch.inverse(mat2.float()).to(dtype)
    return out


def debias_keyed_targets(embeds: Dict[str, torch.Tensor], edit: Sequence[str], debias: Sequence[str],
                         preserve: Sequence[str], direction_scales: Sequence[np.ndarray]
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Closed form of the keyed loop above.  Every cached guide output is W g_x for an EFFECTIVE embedding g_x that does
    not depend on the module (the in-place updates are linear combinations with scalar coefficients), so the loop is a
    recursion on g_x in embedding space, float64.  Returns (G_edit [N_e, d], G_pres [N_p, d]) of the LAST executed
    iteration: the target each row of that iteration's sums was given."""
    g = {}
    for x in list(edit) + list(debias) + list(preserve):
        if x not in g:
            g[x] = embeds[x].double().reshape(-1).clone()
    G_e = torch.stack([g[e] for e in edit]) if len(edit) else torch.zeros(0, 0, dtype=torch.float64)
    G_p = torch.stack([g[p] for p in preserve]) if len(preserve) else torch.zeros(0, G_e.shape[1] if len(edit) else 0,
                                                                                 dtype=torch.float64)
    for ds in direction_scales:
        if np.abs(ds).max() == 0:
            break
        rows_e = []
        for idx, e in enumerate(edit):
            for i, concept in enumerate(debias):
                g[e] += float(ds[idx][i]) * g[concept]          # the right-hand side is evaluated first, like torch's
            rows_e.append(g[e].clone())
        G_e = torch.stack(rows_e)
        if len(preserve):
            G_p = torch.stack([g[p].clone() for p in preserve])
    return G_e, G_p


def uce_exact64_rows(weights: Sequence[torch.Tensor], C: torch.Tensor, G: torch.Tensor, s: torch.Tensor,
                     lamb: float) -> List[torch.Tensor]:
    """W_new = (lamb W + sum_i s_i (W g_i) c_i^T)(lamb I + sum_i s_i c_i c_i^T)^-1 for arbitrary rows (c_i, g_i, s_i),
    float64 (uce_sd_debias.py:114-140 with the targets given)."""
    C, G, s = C.double(), G.double(), s.double()
    d = C.shape[1]
    A = lamb * torch.eye(d, dtype=torch.float64) + C.T @ (s[:, None] * C)
    out = []
    for w in weights:
        w = w.double()
        mat1 = lamb * w + (w @ G.T) @ (s[:, None] * C)
        out.append(torch.linalg.solve(A, mat1.T).T)
    return out


def debias_targets(edit: torch.Tensor, debias: torch.Tensor,
                   direction_scales: Sequence[np.ndarray]) -> torch.Tensor:
    """Closed form of the cumulative drift: after t iterations the target of edit concept e
    is g_e = c_e + (sum_t D_t)[e,:] @ C_debias.  float64."""
    D = np.zeros_like(np.asarray(direction_scales[0], dtype=np.float64))
    for ds in direction_scales:
        if np.abs(ds).max() == 0:
            break
        D = D + np.asarray(ds, dtype=np.float64)
    return edit.double() + torch.from_numpy(D) @ debias.double()


# --------------------------------------------------------------------------------------
# cross-attention  (diffusers AttnProcessor2_0 -> F.scaled_dot_product_attention;
# reached from evalscripts/generate-images-sd.py:37-42)
# --------------------------------------------------------------------------------------

def xattn_ref(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
              scale: Optional[float] = None) -> torch.Tensor:
    """softmax(Q K^T * scale) V per head, float64 internally.

    q: [B, Lq, C], k/v: [B, Lk, C] with C = heads * dh (diffusers' [B, L, C] layout; the
    processor views it as [B, H, L, dh]).  Returns [B, Lq, C] float64."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    dh = C // heads
    scale = (1.0 / math.sqrt(dh)) if scale is None else scale
    qh = q.double().view(B, Lq, heads, dh).transpose(1, 2)
    kh = k.double().view(B, Lk, heads, dh).transpose(1, 2)
    vh = v.double().view(B, Lk, heads, dh).transpose(1, 2)
    p = torch.softmax((qh @ kh.transpose(-1, -2)) * scale, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, C)


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------

def rel_fro(a, b) -> float:
    """||a - b||_F / ||b||_F in float64 (b is the yardstick)."""
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.double()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.double()
    return float((a - b).norm() / b.norm())


def clip_like_embeddings(n: int, d: int, seed: int, norm: float = 28.0,
                         cosine: float = 0.64) -> np.ndarray:
    """Synthetic last-token embeddings with CLIP-text-like geometry:
    one shared direction plus isotropic noise, every row of norm `norm`, mean pairwise
    cosine ~ `cosine`.  numpy PCG64 so the stream is version-stable; fixtures still store
    the arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.standard_normal(d)
    u /= np.linalg.norm(u)
    z = rng.standard_normal((n, d))
    z -= np.outer(z @ u, u)
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    a = math.sqrt(cosine)
    b = math.sqrt(1.0 - cosine)
    return (norm * (a * u[None, :] + b * z)).astype(np.float32)


def linear_default_weight(o: int, d: int, rng: np.random.Generator) -> np.ndarray:
    """nn.Linear's default init range U(-1/sqrt(d), 1/sqrt(d))."""
    bound = 1.0 / math.sqrt(d)
    return rng.uniform(-bound, bound, size=(o, d)).astype(np.float32)
