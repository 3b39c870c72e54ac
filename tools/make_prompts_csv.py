#!/usr/bin/env python3
"""Writes a prompt table in the schema of the reference's data/coco_30k.csv
(case_number,source,prompt,evaluation_seed,coco_id; read by evalscripts/generate-images-sd.py --prompts_path):

    python tools/make_prompts_csv.py data/coco_30k_synth.csv 30000

The real table is data/coco_30k.csv of the reference repository (it does not travel to the GPU box; 71 of its records are
the fixture tests/golden/coco30k_rows.csv); this tool writes tables of any length: caption-like prompts from a small
grammar with deterministic 5-digit seeds (uce_amd.synth.coco_like_rows).  data/coco_1k_synth.csv (committed) is the
first 1000 rows; BASELINE config 5 = the full 30 000."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import synth  # noqa: E402

if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "data/coco_30k_synth.csv"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    print(synth.write_prompts_csv(path, n, seed=0), n, "rows")
