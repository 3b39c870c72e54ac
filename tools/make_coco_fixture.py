#!/usr/bin/env python3
"""Builds the DATA fixture tests/golden/coco30k_rows.{csv,json} in the build container (needs /root/reference):

  * coco30k_rows.csv  - real records of the reference's prompt table data/coco_30k.csv (the table BASELINE config 5 and
                        evalscripts/generate-images-sd.py:21-46 walk): the first rows, EVERY record among the first 5000
                        whose caption spans several lines (quoted newlines), records with quoted commas / doubled
                        quotes, the largest evaluation_seed and the last record.  Same five columns, same quoting rules.
  * coco30k_rows.json - what the REFERENCE's own generate_images() (imported from /root/reference with a stub
                        `diffusers` module whose pipeline only records its arguments) does on that file: the
                        (prompt, seed, num_images_per_prompt) of every pipe(...) call in order and the PNG names it
                        writes, for two (from_case, till_case) windows.

Nothing of the reference travels: the fixture is table rows + recorded call arguments."""
import csv
import importlib.util
import json
import os
import sys
import tempfile
import types

import pandas as pd
import torch

REF = "/root/reference"
SRC = f"{REF}/data/coco_30k.csv"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_CSV = os.path.join(ROOT, "tests", "golden", "coco30k_rows.csv")
OUT_JSON = os.path.join(ROOT, "tests", "golden", "coco30k_rows.json")


def pick_records():
    with open(SRC, newline="") as fh:
        recs = list(csv.reader(fh))
    header, rows = recs[0], recs[1:]
    assert header == ["case_number", "source", "prompt", "evaluation_seed", "coco_id"], header
    keep = set(range(40))
    keep |= {i for i, r in enumerate(rows[:5000]) if "\n" in r[2]}
    keep |= {i for i, r in enumerate(rows[:700]) if '"' in r[2]}
    keep |= set([i for i, r in enumerate(rows[:300]) if "," in r[2]][:12])
    seeds = [int(r[3]) for r in rows]
    keep |= {seeds.index(max(seeds)), seeds.index(min(seeds)), len(rows) - 1}
    idx = sorted(keep)
    return header, [rows[i] for i in idx], idx


def load_reference_generate(calls):
    class _Out:
        def __init__(self, images):
            self.images = images

    class _Pipe:
        unet = torch.nn.Linear(1, 1)

        def to(self, device):
            return self

        def __call__(self, prompt, num_inference_steps, guidance_scale, num_images_per_prompt, generator):
            from PIL import Image
            calls.append({"prompt": prompt, "seed": int(generator.initial_seed()), "n": int(num_images_per_prompt),
                          "generator_device": str(generator.device), "steps": int(num_inference_steps),
                          "guidance": float(guidance_scale)})
            return _Out([Image.new("RGB", (8, 8)) for _ in range(num_images_per_prompt)])

    class _DP:
        @staticmethod
        def from_pretrained(model_id, torch_dtype=None, safety_checker=None):
            return _Pipe()

    stub = types.ModuleType("diffusers")
    stub.DiffusionPipeline = _DP
    sys.modules["diffusers"] = stub
    spec = importlib.util.spec_from_file_location("ref_generate_images_sd", f"{REF}/evalscripts/generate-images-sd.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.generate_images


def main():
    header, recs, idx = pick_records()
    os.makedirs(os.path.dirname(OUT_CSV), exist_ok=True)
    with open(OUT_CSV, "w", newline="") as fh:
        w = csv.writer(fh)                                   # QUOTE_MINIMAL, as the table itself is written
        w.writerow(header)
        w.writerows(recs)
    a, b = pd.read_csv(SRC).iloc[idx].reset_index(drop=True), pd.read_csv(OUT_CSV)
    assert a.equals(b), "the fixture does not parse to the same rows as the reference table"
    windows = [(0, 1000000, 1), (85, 2539, 2)]
    expected = []
    for lo, hi, n in windows:
        calls = []
        gen = load_reference_generate(calls)
        with tempfile.TemporaryDirectory() as tmp:
            gen("CompVis/stable-diffusion-v1-4", None, OUT_CSV, tmp, exp_name="coco", device="cpu", guidance_scale=7.5,
                num_inference_steps=50, num_images_per_prompt=n, from_case=lo, till_case=hi)
            files = sorted(os.listdir(os.path.join(tmp, "coco")))
        expected.append({"from_case": lo, "till_case": hi, "num_images_per_prompt": n, "calls": calls, "files": files})
    meta = {"source": "data/coco_30k.csv of rohitgandikota/unified-concept-editing", "records": len(recs),
            "multi_line_prompts": sum("\n" in r[2] for r in recs), "source_row_index": idx, "windows": expected}
    with open(OUT_JSON, "w") as fh:
        json.dump(meta, fh, indent=0)
    print(OUT_CSV, len(recs), "records;", meta["multi_line_prompts"], "multi-line;", [len(e["calls"]) for e in expected], "calls")


if __name__ == "__main__":
    main()
