"""uce_wall_s alone (bench.uce_wall_leg on the fp32 pipeline the CLI loads), twice in one process (the second pass shows what of the
first is one-time library set-up for GEMM shapes not seen before): python tools/probe_wall.py"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for rep in range(2):
    with tempfile.TemporaryDirectory() as tmp:
        print(json.dumps(bench.uce_wall_leg(None, torch.device("cuda:0"), tmp)))
