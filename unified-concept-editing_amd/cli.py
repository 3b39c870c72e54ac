"""Command-line surface of the drop-in scripts: flag names, types and defaults are those of
the reference (trainscripts/uce_sd_erase.py:97-112, trainscripts/uce_sd_debias.py:155-195,
evalscripts/generate-images-sd.py:52-62 - the code, not the README, is the truth), expressed as
tables.  Extra flags (all optional, all off by default) are listed under `EXTRA_*`.
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

SD14 = "CompVis/stable-diffusion-v1-4"

# (flag, kwargs) tables ---------------------------------------------------------------------
ERASE_FLAGS = [
    ("--edit_concepts", dict(type=str, required=True, help="prompts of the concepts to erase, separated by ;")),
    ("--guide_concepts", dict(type=str, default=None, help="concepts to guide the erased concepts towards, separated by ;")),
    ("--preserve_concepts", dict(type=str, default=None, help="concepts to preserve, separated by ;")),
    ("--concept_type", dict(type=str, required=True, choices=["art", "object"], help="type of concept being erased")),
    ("--model_id", dict(type=str, default=SD14, help="model to edit")),
    ("--device", dict(type=str, default="cuda:0", help="GPU to run on")),
    ("--erase_scale", dict(type=float, default=1, help="weight of the erase terms")),
    ("--preserve_scale", dict(type=float, default=1, help="weight of the preserve terms")),
    ("--lamb", dict(type=float, default=0.5, help="regulariser lambda")),
    ("--expand_prompts", dict(type=str, default="false", choices=["true", "false"], help="add 5 templated variants per concept")),
    ("--save_dir", dict(type=str, default="../uce_models", help="where the edited weights are written")),
    ("--exp_name", dict(type=str, default=None, help="file name of the artifact (default uce_test)")),
]

DEBIAS_FLAGS = [
    ("--edit_concepts", dict(type=str, required=True, help="prompts of the concepts to edit, separated by ;")),
    ("--debias_concepts", dict(type=str, default=None, help="attributes to balance, separated by ;")),
    ("--preserve_concepts", dict(type=str, default=None, help="concepts to preserve, separated by ;")),
    ("--model_id", dict(type=str, default=SD14, help="model to edit")),
    ("--device", dict(type=str, default="cuda:0", help="GPU to run on")),
    ("--edit_scale", dict(type=float, default=1, help="weight of the edit terms")),
    ("--preserve_scale", dict(type=float, default=1, help="weight of the preserve terms")),
    ("--lamb", dict(type=float, default=0.5, help="regulariser lambda")),
    ("--save_dir", dict(type=str, default="../uce_models", help="where the edited weights are written")),
    ("--exp_name", dict(type=str, default=None, help="file name of the artifact (default uce_test)")),
    ("--desired_ratios", dict(type=float, nargs="+", default=[0.5, 0.5], help="target ratio per debias concept")),
    ("--max_iterations", dict(type=int, default=30, help="iteration cap of the debias loop")),
    ("--max_diff", dict(type=float, default=0.05, help="ratio error below which a concept counts as balanced")),
    ("--step_size", dict(type=float, default=0.1, help="accepted for compatibility; the reference never uses it")),
    ("--num_images_per_prompt", dict(type=int, default=10, help="images per concept per iteration")),
    ("--num_inference_steps", dict(type=int, default=20, help="denoising steps per image")),
    ("--guidance_scale", dict(type=float, default=7.5, help="classifier-free guidance scale")),
]

GENERATE_FLAGS = [
    ("--model_id", dict(type=str, default=SD14, help="model to sample from")),
    ("--uce_model_path", dict(type=str, default=None, help="safetensors file with edited attn2 weights")),
    ("--prompts_path", dict(type=str, required=True, help="csv with columns prompt, evaluation_seed, case_number")),
    ("--save_path", dict(type=str, default="../uce_results/", help="output folder")),
    ("--device", dict(type=str, default="cuda:0", help="GPU to run on")),
    ("--exp_name", dict(type=str, default="test_images", help="sub-folder of save_path")),
    ("--guidance_scale", dict(type=float, default=7.5, help="classifier-free guidance scale")),
    ("--till_case", dict(type=int, default=1000000, help="last case_number to generate")),
    ("--from_case", dict(type=int, default=0, help="first case_number to generate")),
    ("--num_images_per_prompt", dict(type=int, default=1, help="samples per prompt")),
    ("--num_inference_steps", dict(type=int, default=50, help="denoising steps")),
]

# additions of this implementation (never required)
EXTRA_RUNTIME_FLAGS = [
    ("--synthetic_model", dict(action="store_true", help="random-initialised weights of the named architecture "
                                                         "(no checkpoint on disk; benchmarking / tests)")),
    ("--model_dir", dict(type=str, default=None, help="local diffusers-format directory to load instead of the hub id")),
]
EXTRA_EDIT_FLAGS = [
    ("--algo", dict(type=str, default="auto", choices=["auto", "primal", "dual"], help="solver formulation")),
    ("--embed_batch", dict(type=int, default=-1, help="concept strings per text-encoder forward: -1 (default) = 64 for this "
                                                       "build's own pipeline on a GPU, one per call for any other pipeline object; "
                                                       "0 = one string per call, as the reference does")),
]
EXTRA_GENERATE_FLAGS = [
    ("--latents_only", dict(action="store_true", help="skip VAE decode / PNG encode, save latents (.pt)")),
    ("--skip_existing", dict(action="store_true", help="resume: skip rows whose first PNG exists")),
    ("--batch_prompts", dict(type=int, default=0, help="CSV rows denoised per U-Net call: 0 = as many as half of the free HBM "
                                                       "holds, at most 128 images (rows are independent: each keeps its own "
                                                       "seeded latents); 1 = row by row, as the reference does")),
]

ART_TEMPLATES = ["painting by {}", "art by {}", "artwork by {}", "picture by {}", "style of {}"]
OBJECT_TEMPLATES = ["image of {}", "photo of {}", "portrait of {}", "picture of {}", "painting of {}"]


def _parser(prog: str, description: str, tables) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog=prog, description=description)
    for table in tables:
        for flag, kw in table:
            p.add_argument(flag, **kw)
    return p


def erase_parser() -> argparse.ArgumentParser:
    return _parser("TrainUCE", "UCE for erasing concepts in Stable Diffusion",
                   [ERASE_FLAGS, EXTRA_RUNTIME_FLAGS, EXTRA_EDIT_FLAGS])


def debias_parser() -> argparse.ArgumentParser:
    return _parser("TrainUCE", "UCE for debiasing concepts in Stable Diffusion",
                   [DEBIAS_FLAGS, EXTRA_RUNTIME_FLAGS, EXTRA_EDIT_FLAGS])


def generate_parser() -> argparse.ArgumentParser:
    return _parser("generateImages", "Generate images with an (edited) Stable Diffusion model",
                   [GENERATE_FLAGS, EXTRA_RUNTIME_FLAGS, EXTRA_GENERATE_FLAGS])


def parse_erase_args(argv: Optional[Sequence[str]] = None):
    return erase_parser().parse_args(argv)


def parse_debias_args(argv: Optional[Sequence[str]] = None):
    return debias_parser().parse_args(argv)


def parse_generate_args(argv: Optional[Sequence[str]] = None):
    return generate_parser().parse_args(argv)


FLUX_SCHNELL = "black-forest-labs/FLUX.1-schnell"
FLUX_FLAGS = [(f, dict(kw, default=FLUX_SCHNELL) if f == "--model_id" else kw) for f, kw in ERASE_FLAGS]


def flux_parser() -> argparse.ArgumentParser:
    """trainscripts/uce_flux_edit.py:124-146: the erase flags with FLUX.1-schnell as the default model."""
    return _parser("TrainUCE", "UCE for erasing concepts in FLUX", [FLUX_FLAGS, EXTRA_EDIT_FLAGS[:1]])


def parse_flux_args(argv: Optional[Sequence[str]] = None):
    return flux_parser().parse_args(argv)


def flux_max_sequence_length(model_id: str) -> int:
    """uce_flux_edit.py:169-171."""
    return 256 if "schnell" in model_id else 512


HIDREAM_FULL = "HiDream-ai/HiDream-I1-Full"
HIDREAM_FLAGS = [(f, dict(kw, default=HIDREAM_FULL) if f == "--model_id" else kw) for f, kw in ERASE_FLAGS]
HIDREAM_MAX_SEQUENCE_LENGTH = 128                   # uce_hidream_edit.py:214


def hidream_parser() -> argparse.ArgumentParser:
    """trainscripts/uce_hidream_edit.py:181-197: the erase flags with HiDream-I1-Full as the default model."""
    return _parser("TrainUCE", "UCE for erasing concepts in HiDream", [HIDREAM_FLAGS, EXTRA_EDIT_FLAGS[:1]])


def parse_hidream_args(argv: Optional[Sequence[str]] = None):
    return hidream_parser().parse_args(argv)


def split_concepts(text: Optional[str]) -> List[str]:
    """';'-separated, each entry stripped (uce_sd_erase.py:134)."""
    return [] if text is None else [c.strip() for c in text.split(";")]


@dataclass
class EraseJob:
    edit_concepts: List[str]
    guide_concepts: List[str]
    preserve_concepts: List[str]
    erase_scale: float
    preserve_scale: float
    lamb: float
    save_dir: str
    exp_name: str
    model_id: str = SD14
    device: str = "cuda:0"
    banner: List[str] = field(default_factory=list)


def erase_job_from_args(args) -> EraseJob:
    """Concept-list semantics of uce_sd_erase.py:130-190."""
    edit = split_concepts(args.edit_concepts)
    guide_text = args.guide_concepts
    if guide_text is None:                                   # :136-141 default target
        guide_text = "art" if args.concept_type == "art" else ""
    guide = split_concepts(guide_text)
    if len(guide) == 1:                                      # :142-143 one guide for all
        guide = guide * len(edit)
    if len(guide) != len(edit):                              # :144-145
        raise Exception("Error! The length of erase concepts and their corresponding guide concepts do not match. "
                        "Please make sure they are seperated by ; and are of equal sizes")
    preserve = split_concepts(args.preserve_concepts)
    if args.expand_prompts == "true":                        # :155-190
        templates = ART_TEMPLATES if args.concept_type == "art" else OBJECT_TEMPLATES
        for concept, target in list(zip(edit, guide)):
            edit.extend(t.format(concept) for t in templates)
            guide.extend(t.format(target) for t in templates)
    return EraseJob(edit, guide, preserve, args.erase_scale, args.preserve_scale, args.lamb, args.save_dir,
                    args.exp_name if args.exp_name is not None else "uce_test", args.model_id, args.device,
                    banner=[f"\n\nErasing: {edit}\n", f"Guiding: {guide}\n", f"Preserving: {preserve}\n"])


@dataclass
class DebiasJob:
    edit_concepts: List[str]
    debias_concepts: List[str]
    preserve_concepts: List[str]
    desired_ratios: List[float]
    edit_scale: float
    preserve_scale: float
    lamb: float
    save_dir: str
    exp_name: str
    max_iterations: int
    max_diff: float
    step_size: float
    num_images_per_prompt: int
    num_inference_steps: int
    guidance_scale: float
    model_id: str = SD14
    device: str = "cuda:0"
    banner: List[str] = field(default_factory=list)


def debias_job_from_args(args) -> DebiasJob:
    """uce_sd_debias.py:199-237."""
    edit = split_concepts(args.edit_concepts)
    if args.debias_concepts is None:
        raise Exception("Error! --debias_concepts is required for debiasing")
    debias = split_concepts(args.debias_concepts)
    if len(debias) != len(args.desired_ratios):              # :226-227
        raise Exception("Error! The length of debias concepts and their corresponding desired ratios concepts do not match.")
    preserve = split_concepts(args.preserve_concepts)
    return DebiasJob(edit, debias, preserve, list(args.desired_ratios), args.edit_scale, args.preserve_scale,
                     args.lamb, args.save_dir, args.exp_name if args.exp_name is not None else "uce_test",
                     args.max_iterations, args.max_diff, args.step_size, args.num_images_per_prompt,
                     args.num_inference_steps, args.guidance_scale, args.model_id, args.device,
                     banner=[f"\n\nEditing: {edit}\n", f"Debias Across: {debias}\n", f"Preserving: {preserve}\n"])


ALGO_IDS = {"auto": 0, "primal": 1, "dual": 2}
