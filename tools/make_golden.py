#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference scripts from /root/reference.

Runs only in the build container (the reference never travels to the GPU box).  Two shims,
as found by the survey:
  * `diffusers` is not installed -> a stub module with `DiffusionPipeline` whose
    `from_pretrained` hands back tests/fakepipe.FakePipe;
  * UCE() reads module globals that only `__main__` sets -> set them on the imported module.
Library-level cases call the reference's UCE() directly; CLI-level cases run the script's
`__main__` block with runpy (argument parsing, guide broadcast, prompt expansion, prints,
safetensors artifact).

Each fixture stores INPUT arrays and the reference's OUTPUT arrays (+ a float64 evaluation of
the same formula); no reference source text is stored.
"""
from __future__ import annotations

import argparse
import contextlib
import importlib.util
import io
import json
import os
import runpy
import sys
import tempfile
import types
from typing import Dict, List, Sequence

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from tests import fakepipe  # noqa: E402
from oracle import uce_oracle  # noqa: E402


# ---------------------------------------------------------------- reference import shims

_CURRENT_PIPE = {"pipe": None}


def _install_diffusers_stub() -> None:
    stub = types.ModuleType("diffusers")

    class DiffusionPipeline:  # noqa: D401 - stand-in
        @staticmethod
        def from_pretrained(model_id, **kw):
            if _CURRENT_PIPE.get("flux") is not None:      # uce_flux_edit.py loads the model in two halves
                half = "text" if "transformer" in kw and kw["transformer"] is None else "transformer"
                return _CURRENT_PIPE["flux"][half]
            return _CURRENT_PIPE["pipe"]

    class HiDreamImagePipeline:  # noqa: D401 - stand-in: uce_hidream_edit.py loads the model in three pieces
        @staticmethod
        def from_pretrained(model_id, **kw):
            parts = _CURRENT_PIPE["hidream"]
            if "transformer" not in kw:
                return parts["transformer"]
            if kw.get("tokenizer_4") is not None:
                parts["llama"].tokenizer_4 = kw["tokenizer_4"]
                return parts["llama"]
            return parts["t5"]

    stub.DiffusionPipeline = DiffusionPipeline
    stub.HiDreamImagePipeline = HiDreamImagePipeline
    stub.UniPCMultistepScheduler = object
    sys.modules["diffusers"] = stub


def _load_ref_module(relpath: str, alias: str):
    spec = importlib.util.spec_from_file_location(alias, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------- helpers

FIXTURE_TABLE = [  # three slabs (stand-ins for the three SD-1.4 width classes), 32 rows each: rows are independent
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k", 32),
    ("up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_v", 32),
    ("mid_block.attentions.0.transformer_blocks.0.attn2.to_k", 32),
]


def _load_state(path: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return load_file(path)


def _emb_rows(pipe: fakepipe.FakePipe, prompts: Sequence[str]) -> np.ndarray:
    if len(prompts) == 0:
        return np.zeros((0, pipe.d), dtype=np.float32)
    return np.stack([pipe.embedding(p) for p in prompts]).astype(np.float32)


def _save(case: str, out_dir: str, **arrays) -> None:
    path = os.path.join(out_dir, case + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


def _artists(n: int) -> List[str]:
    p = os.path.join(REF, "data", "info",
                     f"erased-{n}artists-towards_art-preserve_true-sd_1_4-method_replace.txt")
    with open(p) as f:
        return json.load(f)


def _other_artists(exclude: Sequence[str], n: int) -> List[str]:
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "data", "artists1734_prompts.csv"))
    ex = set(exclude)
    out: List[str] = []
    for a in df.artist.tolist():
        if a not in ex and a not in out:
            out.append(a)
        if len(out) == n:
            break
    assert len(out) == n
    return out


# ---------------------------------------------------------------- library-level erase cases

def run_erase_case(erase_mod, case: str, out_dir: str, d: int, edit: List[str], guide: List[str],
                   preserve: List[str], erase_scale: float, preserve_scale: float, lamb: float,
                   seed: int) -> None:
    print(f"[erase] {case}: d={d} N_e={len(edit)} N_p={len(preserve)}")
    rng = np.random.Generator(np.random.PCG64(seed))
    unet = fakepipe.build_unet(FIXTURE_TABLE, d, rng)
    pipe = fakepipe.FakePipe(unet, d)
    w_old = fakepipe.uce_weights(unet)
    erase_mod.device = "cpu"
    erase_mod.torch_dtype = torch.float32
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        erase_mod.UCE(pipe, list(edit), list(guide), list(preserve), erase_scale, preserve_scale,
                      lamb, tmp, case)
        state = _load_state(os.path.join(tmp, case + ".safetensors"))
    assert sorted(state) == sorted(n + ".weight" for n, _ in w_old), "name predicate mismatch"

    C_edit, G_edit, C_pres = _emb_rows(pipe, edit), _emb_rows(pipe, guide), _emb_rows(pipe, preserve)
    te = [torch.from_numpy(r[None]) for r in C_edit]
    tg = [torch.from_numpy(r[None]) for r in G_edit]
    tp = [torch.from_numpy(r[None]) for r in C_pres]
    ws = [w for _, w in w_old]
    exact = uce_oracle.uce_edit_exact64(ws, te, tg, tp, erase_scale, preserve_scale, lamb)
    arrays = dict(
        meta=np.array(json.dumps(dict(
            kind="erase", d=d, lamb=lamb, erase_scale=erase_scale, preserve_scale=preserve_scale,
            edit=edit, guide=guide, preserve=preserve, modules=[n for n, _ in w_old],
            encode_calls=pipe.encode_calls))),
        C_edit=C_edit, G_edit=G_edit, C_pres=C_pres,
    )
    for i, (n, w) in enumerate(w_old):
        ref = state[n + ".weight"]
        arrays[f"W_old_{i}"] = w.numpy()
        arrays[f"W_ref32_{i}"] = ref.numpy()
        arrays[f"W_exact64_{i}"] = exact[i].numpy()
        print(f"    {n}: eps_ref = relF(ref32, exact64) = {uce_oracle.rel_fro(ref, exact[i]):.3e}")
    _save(case, out_dir, **arrays)


# ---------------------------------------------------------------- FLUX variant (modules with bias)

def run_flux_case(flux_mod, case: str, out_dir: str, out_rows: int, edit: List[str], guide: List[str],
                  preserve: List[str], erase_scale: float, preserve_scale: float, lamb: float, seed: int) -> None:
    print(f"[flux] {case}: N_e={len(edit)} N_p={len(preserve)} rows={out_rows}")
    rng = np.random.Generator(np.random.PCG64(seed))
    tr = fakepipe.build_flux_transformer(out_rows, rng)
    text = fakepipe.FakeFluxTextPipe()
    names = ["context_embedder", "time_text_embed.text_embedder.linear_1"]
    mods = dict(tr.named_modules())
    w_old = {n: (mods[n].weight.detach().clone(), mods[n].bias.detach().clone()) for n in names}
    _CURRENT_PIPE["flux"] = {"transformer": fakepipe.FakeFluxTransformerPipe(tr), "text": text}
    try:
        with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
            flux_mod.UCE("black-forest-labs/FLUX.1-schnell", list(edit), list(guide), list(preserve), erase_scale,
                         preserve_scale, lamb, tmp, case, torch.float32, "cpu", 256)
            state = _load_state(os.path.join(tmp, case + ".safetensors"))
    finally:
        _CURRENT_PIPE["flux"] = None
    assert sorted(state) == sorted(n + ".weight" for n in names), sorted(state)
    arrays = dict(meta=np.array(json.dumps(dict(
        kind="flux", lamb=lamb, erase_scale=erase_scale, preserve_scale=preserve_scale, edit=edit, guide=guide,
        preserve=preserve, modules=names, max_sequence_length=256, encode_calls=text.encode_calls))))
    for i, n in enumerate(names):
        emb = text.t5_embedding if i == 0 else text.pooled_embedding
        Ce = np.stack([emb(p) for p in edit]).astype(np.float32)
        Ge = np.stack([emb(p) for p in guide]).astype(np.float32)
        Cp = np.stack([emb(p) for p in preserve]).astype(np.float32) if preserve else np.zeros((0, Ce.shape[1]), np.float32)
        w, b = w_old[n]
        te = [torch.from_numpy(r[None]) for r in Ce]
        tg = [torch.from_numpy(r[None]) for r in Ge]
        tp = [torch.from_numpy(r[None]) for r in Cp]
        exact = uce_oracle.uce_edit_bias_exact64(w, b, te, tg, tp, erase_scale, preserve_scale, lamb)
        ref = state[n + ".weight"]
        arrays.update({f"C_edit_{i}": Ce, f"G_edit_{i}": Ge, f"C_pres_{i}": Cp, f"W_old_{i}": w.numpy(),
                       f"b_{i}": b.numpy(), f"W_ref32_{i}": ref.numpy(), f"W_exact64_{i}": exact.numpy()})
        print(f"    {n}: eps_ref = relF(ref32, exact64) = {uce_oracle.rel_fro(ref, exact):.3e}")
    _save(case, out_dir, **arrays)


# ---------------------------------------------------------------- HiDream variant (one embedding family per module)

def run_hidream_case(hd_mod, case: str, out_dir: str, out_rows: int, llama_layers: List[int], edit: List[str],
                     guide: List[str], preserve: List[str], erase_scale: float, preserve_scale: float, lamb: float,
                     seed: int) -> None:
    print(f"[hidream] {case}: N_e={len(edit)} N_p={len(preserve)} rows={out_rows} layers={llama_layers}")
    rng = np.random.Generator(np.random.PCG64(seed))
    tr = fakepipe.build_hidream_transformer(out_rows, llama_layers, rng)
    text = fakepipe.FakeHiDreamTextPipe()
    names = [n for n, _ in tr.named_modules() if "caption_projection" in n and "linear" in n]
    mods = dict(tr.named_modules())
    w_old = {n: mods[n].weight.detach().clone() for n in names}

    class _Tok:
        @staticmethod
        def from_pretrained(*a, **k):
            return fakepipe.FakeHiDreamTokenizer(131072)

    class _Llama:
        @staticmethod
        def from_pretrained(*a, **k):
            return object()

    hd_mod.PreTrainedTokenizerFast, hd_mod.LlamaForCausalLM = _Tok, _Llama
    _CURRENT_PIPE["hidream"] = {"transformer": fakepipe.FakeFluxTransformerPipe(tr), "llama": text, "t5": text}
    try:
        with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
            hd_mod.UCE("HiDream-ai/HiDream-I1-Full", list(edit), list(guide), list(preserve), erase_scale,
                       preserve_scale, lamb, tmp, case, torch.float32, "cpu", 128)
            state = _load_state(os.path.join(tmp, case + ".safetensors"))
    finally:
        _CURRENT_PIPE["hidream"] = None
    assert sorted(state) == sorted(n + ".weight" for n in names), sorted(state)
    fams = [f"llama{layer}" for layer in llama_layers] + ["t5"]
    arrays = dict(meta=np.array(json.dumps(dict(
        kind="hidream", lamb=lamb, erase_scale=erase_scale, preserve_scale=preserve_scale, edit=edit, guide=guide,
        preserve=preserve, modules=names, llama_layers=llama_layers, families=fams, max_sequence_length=128,
        calls=text.calls))))
    for i, n in enumerate(names):
        emb = lambda p: text.family_embedding(p, fams[i])
        Ce = np.stack([emb(p) for p in edit]).astype(np.float32)
        Ge = np.stack([emb(p) for p in guide]).astype(np.float32)
        Cp = np.stack([emb(p) for p in preserve]).astype(np.float32) if preserve else np.zeros((0, Ce.shape[1]), np.float32)
        w = w_old[n]
        te = [torch.from_numpy(r[None]) for r in Ce]
        tg = [torch.from_numpy(r[None]) for r in Ge]
        tp = [torch.from_numpy(r[None]) for r in Cp]
        exact = uce_oracle.uce_edit_exact64([w], te, tg, tp, erase_scale, preserve_scale, lamb)[0]
        ref = state[n + ".weight"]
        arrays.update({f"C_edit_{i}": Ce, f"G_edit_{i}": Ge, f"C_pres_{i}": Cp, f"W_old_{i}": w.numpy(),
                       f"W_ref32_{i}": ref.numpy(), f"W_exact64_{i}": exact.numpy()})
        print(f"    {n} ({fams[i]}): eps_ref = relF(ref32, exact64) = {uce_oracle.rel_fro(ref, exact):.3e}")
    _save(case, out_dir, **arrays)


# ---------------------------------------------------------------- library-level debias cases

def run_debias_case(deb_mod, case: str, out_dir: str, d: int, edit: List[str], debias: List[str],
                    preserve: List[str], scripted: List[np.ndarray], edit_scale: float,
                    preserve_scale: float, lamb: float, seed: int) -> None:
    print(f"[debias] {case}: d={d} N_e={len(edit)} N_debias={len(debias)} iters={len(scripted)}")
    rng = np.random.Generator(np.random.PCG64(seed))
    unet = fakepipe.build_unet(FIXTURE_TABLE, d, rng)
    pipe = fakepipe.FakePipe(unet, d)
    w_old = fakepipe.uce_weights(unet)
    deb_mod.device = "cpu"
    deb_mod.torch_dtype = torch.float32
    deb_mod.max_iterations = len(scripted)
    deb_mod.desired_ratios = [1.0 / len(debias)] * len(debias)
    it = iter(scripted)
    deb_mod.get_ratios = lambda **kw: next(it)
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()), \
            contextlib.redirect_stderr(io.StringIO()):
        deb_mod.UCE(pipe, None, list(edit), list(debias), list(preserve), edit_scale, preserve_scale,
                    lamb, tmp, case, 0.05, 0.1, 10, 20, 7.5)
        state = _load_state(os.path.join(tmp, case + ".safetensors"))
    C_edit, C_deb, C_pres = _emb_rows(pipe, edit), _emb_rows(pipe, debias), _emb_rows(pipe, preserve)
    ws = [w for _, w in w_old]
    if len(set(edit)) == len(edit) and not (set(edit) & (set(debias) | set(preserve))):
        G = uce_oracle.debias_targets(torch.from_numpy(C_edit), torch.from_numpy(C_deb), scripted)
        te = [torch.from_numpy(r[None]) for r in C_edit]
        tg = [g[None] for g in G]
        tp = [torch.from_numpy(r[None]) for r in C_pres]
        exact = uce_oracle.uce_edit_exact64(ws, te, tg, tp, edit_scale, preserve_scale, lamb)
    else:
        # a string cached once and drifted in place (uce_sd_debias.py:122-127): the keyed closed form is the arbiter
        embeds = {}
        for names, rows_ in ((edit, C_edit), (debias, C_deb), (preserve, C_pres)):
            for n, r in zip(names, rows_):
                embeds[n] = torch.from_numpy(r[None])
        G_e, G_p = uce_oracle.debias_keyed_targets(embeds, edit, debias, preserve, scripted)
        Call = torch.from_numpy(np.concatenate([C_edit, C_pres]) if len(C_pres) else C_edit)
        Gall = torch.cat([G_e, G_p]) if len(C_pres) else G_e
        sall = torch.tensor([edit_scale] * len(edit) + [preserve_scale] * len(preserve))
        exact = uce_oracle.uce_exact64_rows(ws, Call, Gall, sall, lamb)
    arrays = dict(
        meta=np.array(json.dumps(dict(
            kind="debias", d=d, lamb=lamb, edit_scale=edit_scale, preserve_scale=preserve_scale,
            edit=edit, debias=debias, preserve=preserve, modules=[n for n, _ in w_old]))),
        C_edit=C_edit, C_debias=C_deb, C_pres=C_pres,
        direction_scales=np.stack(scripted).astype(np.float64),
    )
    for i, (n, w) in enumerate(w_old):
        ref = state[n + ".weight"]
        arrays[f"W_old_{i}"] = w.numpy()
        arrays[f"W_ref32_{i}"] = ref.numpy()
        arrays[f"W_exact64_{i}"] = exact[i].numpy()
        print(f"    {n}: eps_ref = {uce_oracle.rel_fro(ref, exact[i]):.3e}")
    _save(case, out_dir, **arrays)


# ---------------------------------------------------------------- CLI-level cases (runpy)

def run_cli_case(case: str, out_dir: str, script: str, argv: List[str], d: int, table, seed: int,
                 rows_kept: int = 4) -> None:
    print(f"[cli] {case}: {script} {' '.join(argv)}")
    rng = np.random.Generator(np.random.PCG64(seed))
    unet = fakepipe.build_unet(table, d, rng)
    pipe = fakepipe.FakePipe(unet, d)
    w_old = fakepipe.uce_weights(unet)
    _CURRENT_PIPE["pipe"] = pipe
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as tmp:
        old_argv = sys.argv
        sys.argv = [script] + argv + ["--save_dir", tmp, "--exp_name", case, "--device", "cpu"]
        try:
            with contextlib.redirect_stdout(buf):
                runpy.run_path(os.path.join(REF, script), run_name="__main__")
        finally:
            sys.argv = old_argv
        with open(os.path.join(tmp, case + ".safetensors"), "rb") as f:
            raw = f.read()
        state = _load_state(os.path.join(tmp, case + ".safetensors"))
    hdr_len = int.from_bytes(raw[:8], "little")
    header = json.loads(raw[8:8 + hdr_len])
    stdout = buf.getvalue()
    # stdout minus the timing number (uce_sd_erase.py:91)
    lines = [ln for ln in stdout.splitlines() if not ln.startswith("Model edited in")]
    arrays = dict(
        meta=np.array(json.dumps(dict(
            kind="cli", script=script, argv=argv, d=d, modules=[n for n, _ in w_old],
            shapes=[list(w.shape) for _, w in w_old], stdout_lines=lines,
            encode_calls=pipe.encode_calls,
            st_keys=[k for k in header if k != "__metadata__"],
            st_dtypes=sorted({v["dtype"] for k, v in header.items() if k != "__metadata__"}),
            st_has_metadata="__metadata__" in header))),
    )
    for i, (n, w) in enumerate(w_old):
        arrays[f"W_old_{i}"] = w[:rows_kept].numpy()
        arrays[f"W_ref32_{i}"] = state[n + ".weight"][:rows_kept].numpy()
    _save(case, out_dir, **arrays)


# ---------------------------------------------------------------- SDPA goldens

def run_sdpa_cases(out_dir: str) -> None:
    """torch's CPU scaled_dot_product_attention at the four SD-1.4 cross-attention shapes
    (SURVEY.md section 8a row a10), bf16 like the reference's generate path
    (generate-images-sd.py:76).  Lq subsampled (query rows are independent)."""
    shapes = [(4096, 40, 320), (1024, 80, 640), (256, 160, 1280), (64, 160, 1280)]
    B, H, Lk = 2, 8, 77
    for Lq, dh, C in shapes:
        g = torch.Generator().manual_seed(1000 + Lq)
        lq = min(Lq, 64)
        q = torch.randn(B, lq, C, generator=g).to(torch.bfloat16)
        k = torch.randn(B, Lk, C, generator=g).to(torch.bfloat16)
        v = torch.randn(B, Lk, C, generator=g).to(torch.bfloat16)

        def heads(x, L):
            return x.view(B, L, H, dh).transpose(1, 2)

        o = torch.nn.functional.scaled_dot_product_attention(
            heads(q, lq), heads(k, Lk), heads(v, Lk), attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, lq, C)
        o32 = torch.nn.functional.scaled_dot_product_attention(
            heads(q.float(), lq), heads(k.float(), Lk), heads(v.float(), Lk))
        o32 = o32.transpose(1, 2).reshape(B, lq, C)
        case = f"sdpa_Lq{Lq}_dh{dh}"
        print(f"[sdpa] {case}")
        _save(case, out_dir,
              meta=np.array(json.dumps(dict(kind="sdpa", B=B, H=H, Lq_full=Lq, Lq=lq, Lk=Lk, dh=dh, C=C))),
              q=q.view(torch.int16).numpy(), k=k.view(torch.int16).numpy(), v=v.view(torch.int16).numpy(),
              o_bf16=o.view(torch.int16).numpy(), o_f32=o32.numpy())


# ---------------------------------------------------------------- main

def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    _install_diffusers_stub()
    erase_mod = _load_ref_module("trainscripts/uce_sd_erase.py", "ref_uce_sd_erase")
    deb_mod = _load_ref_module("trainscripts/uce_sd_debias.py", "ref_uce_sd_debias")

    def want(name: str) -> bool:
        return args.only is None or args.only in name

    # BASELINE config 1: 2 erase -> 'art', README's 3 preserves
    if want("erase_n2p3_d768"):
        run_erase_case(erase_mod, "erase_n2p3_d768", args.out, 768, ["Van Gogh", "Picasso"],
                       ["art", "art"], ["Monet", "Rembrandt", "Warhol"], 1.0, 1.0, 0.5, seed=1)
    # BASELINE config 2: 50 artists -> 'art'
    if want("erase_n50_d768"):
        a50 = _artists(50)
        run_erase_case(erase_mod, "erase_n50_d768", args.out, 768, a50, ["art"] * 50, [], 1.0, 1.0, 0.5, seed=2)
    # BASELINE config 3: 1000 artists + 500 preserves
    if want("erase_n1000p500_d768"):
        a1000 = _artists(1000)
        keep = _other_artists(a1000, 500)
        run_erase_case(erase_mod, "erase_n1000p500_d768", args.out, 768, a1000, ["art"] * 1000, keep,
                       1.0, 1.0, 0.5, seed=3)
    # quirks: duplicate edit, '' guide (BOS index), string in edit AND preserve, >75-token prompt,
    # non-default scales and lambda
    if want("erase_quirks_d768"):
        long_prompt = " ".join(f"w{i}" for i in range(90))
        edit = ["grumpy cat", "grumpy cat", "chewbacca", long_prompt, "ugly"]
        guide = ["", "", "bear", "", ""]
        preserve = ["cat", "chewbacca", "dog"]
        run_erase_case(erase_mod, "erase_quirks_d768", args.out, 768, edit, guide, preserve, 1.5, 0.7, 0.1, seed=4)
    # SD-2.x width (d=1024) and SDXL width (d=2048)
    if want("erase_n12p4_d1024"):
        a = _artists(50)
        run_erase_case(erase_mod, "erase_n12p4_d1024", args.out, 1024, a[:12], ["art"] * 12, a[12:16],
                       1.0, 1.0, 0.5, seed=5)
    if want("erase_n36p4_d2048"):
        a = _artists(50)
        run_erase_case(erase_mod, "erase_n36p4_d2048", args.out, 2048, a[:36], ["art"] * 36, a[36:40],
                       1.0, 1.0, 0.5, seed=6)
    # an intermediate size that exercises d/4 < N < d
    if want("erase_n300p100_d768"):
        a500 = _artists(500)
        run_erase_case(erase_mod, "erase_n300p100_d768", args.out, 768, a500[:300], ["art"] * 300,
                       a500[300:400], 1.0, 1.0, 0.5, seed=7)

    # FLUX variant: two biased Linear modules, T5 (4096) / pooled CLIP (768) embeddings
    if want("flux_n6p3"):
        flux_mod = _load_ref_module("trainscripts/uce_flux_edit.py", "ref_uce_flux_edit")
        a = _artists(50)
        run_flux_case(flux_mod, "flux_n6p3", args.out, 24, a[:6], ["art"] * 6, a[6:9], 1.0, 1.0, 0.5, seed=21)

    # HiDream variant: caption_projection.<i>.linear <- Llama layer llama_layers[i] (last one: T5)
    if want("hidream_n4p2"):
        hd_mod = _load_ref_module("trainscripts/uce_hidream_edit.py", "ref_uce_hidream_edit")
        a = _artists(50)
        run_hidream_case(hd_mod, "hidream_n4p2", args.out, 16, [1, 3, 4], a[:4], ["art"] * 4, a[4:6], 1.0, 1.0, 0.5,
                         seed=22)

    # debias: scripted direction_scale sequences (get_ratios is unseeded in the reference)
    if want("debias_n4x2_d768"):
        rs = np.random.Generator(np.random.PCG64(11))
        scripted = [np.round(rs.uniform(-0.5, 0.5, size=(4, 2)), 1) for _ in range(3)]
        scripted.append(np.zeros((4, 2)))                       # stops the loop (:110-112)
        scripted.append(np.full((4, 2), 0.3))                   # must never be consumed
        run_debias_case(deb_mod, "debias_n4x2_d768", args.out, 768,
                        ["Doctor", "Nurse", "Carpenter", "Teacher"], ["male", "female"], ["Monet"],
                        scripted, 1.0, 1.0, 0.5, seed=12)
    # the string-keyed caches of the reference (uce_sd_debias.py:69-88) drifted in place (:122-127):
    # (1) an edit concept listed twice, (2) a string that is both an edit and a preserve concept, (3) debias concepts
    # that are edit concepts themselves (their cached output is rescaled, then used as the others' direction)
    for case, ed, db, pr, sd in (
            ("debias_alias_dupedit_d768", ["Doctor", "Nurse", "Doctor"], ["male", "female"], ["Monet"], 31),
            ("debias_alias_editpres_d768", ["Doctor", "Nurse"], ["male", "female"], ["Nurse", "Monet"], 32),
            ("debias_alias_editisdebias_d768", ["male", "Doctor", "female"], ["male", "female"], [], 33)):
        if want(case):
            rs = np.random.Generator(np.random.PCG64(sd))
            scripted = [np.round(rs.uniform(-0.5, 0.5, size=(len(ed), len(db))), 1) for _ in range(3)]
            run_debias_case(deb_mod, case, args.out, 768, ed, db, pr, scripted, 1.0, 0.8, 0.5, seed=sd + 100)
    # BASELINE config 4: SDXL-shaped, 36 professions x 2, 3 scripted iterations
    if want("debias_n36x2_d2048"):
        import pandas as pd
        prof = []
        for p in pd.read_csv(os.path.join(REF, "data", "profession_prompts.csv")).profession.tolist():
            if p not in prof:
                prof.append(p)
        rs = np.random.Generator(np.random.PCG64(13))
        scripted = []
        for _ in range(3):
            x = np.round(rs.uniform(-0.5, 0.5, size=(36, 1)), 1)
            scripted.append(np.concatenate([x, -x], axis=1))
        run_debias_case(deb_mod, "debias_n36x2_d2048", args.out, 2048, prof[:36], ["male", "female"], [],
                        scripted, 1.0, 1.0, 0.5, seed=14)

    # CLI level: full SD-1.4 topology; prompt expansion; default guides; artifact format
    full = uce_oracle.sd14_module_table()
    if want("cli_erase_art_expand"):
        run_cli_case("cli_erase_art_expand", args.out, "trainscripts/uce_sd_erase.py",
                     ["--edit_concepts", "Van Gogh; Picasso", "--concept_type", "art",
                      "--preserve_concepts", "Monet; Rembrandt; Warhol", "--expand_prompts", "true"],
                     768, full, seed=21)
    if want("cli_erase_object_default"):
        run_cli_case("cli_erase_object_default", args.out, "trainscripts/uce_sd_erase.py",
                     ["--edit_concepts", "grumpy cat;chewbacca", "--concept_type", "object",
                      "--erase_scale", "2", "--lamb", "0.25"],
                     768, full, seed=22)
    if want("cli_erase_object_expand_guided"):
        run_cli_case("cli_erase_object_expand_guided", args.out, "trainscripts/uce_sd_erase.py",
                     ["--edit_concepts", "chewbacca ; hermione granger", "--guide_concepts", "bear; girl",
                      "--concept_type", "object", "--expand_prompts", "true", "--preserve_scale", "0.5",
                      "--preserve_concepts", "cat"],
                     768, full, seed=23)

    if want("sdpa"):
        run_sdpa_cases(args.out)


if __name__ == "__main__":
    main()
