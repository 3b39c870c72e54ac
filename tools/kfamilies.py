"""Kernel families of the denoising loop out of a rocprofv3 kernel-stats CSV (tools/prof_generate.sh: steady_kernel_stats.csv):
time share per family - conv (3x3 convolutions + their patch matrices), linear (uce_linear_fwd), sattn, xattn, norm (GroupNorm /
LayerNorm), other own kernels, library (Tensile / MIOpen / CK: must be zero), torch elementwise.  bench.py reads the JSON
(profiles/r05/generate_families_b*.json) and prices each family's counted FLOPs against its share of the measured image time.
Usage: kfamilies.py steady_kernel_stats.csv out.json [label]"""
import csv
import json
import sys

FAMILIES = (
    ("conv", ("k_conv3x3", "k_im2col3x3")),
    ("linear", ("k_gemm_dma",)),
    ("sattn", ("k_sattn", "k_vt", "k_softmax_rows")),
    ("xattn", ("k_xattn",)),
    ("norm", ("k_gn_", "k_layernorm")),
    ("library", ("Cijk_", "igemm_fwd", "ck::", "_ZN2ck", "miopen", "MIOpen", "attn_fwd")),
    ("own_other", ("k_add_bias", "k_geglu", "k_cfg_pndm", "k_cast", "k_gather")),
)


def family(name: str) -> str:
    for fam, pats in FAMILIES:
        if any(p in name for p in pats):
            return fam
    return "torch_elementwise"


def main():
    path, out = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    agg, calls, tot = {}, {}, 0.0
    with open(path) as f:
        rd = csv.DictReader(f)
        for r in rd:
            d = float(r.get("TotalDurationNs") or r.get("TotalDuration(ns)") or 0)
            c = int(float(r.get("Calls") or 0))
            fam = family(r["Name"])
            agg[fam] = agg.get(fam, 0.0) + d
            calls[fam] = calls.get(fam, 0) + c
            tot += d
    res = {"source": path, "label": label, "kernel_time_ms": round(tot / 1e6, 3),
           "families": {k: {"share": round(v / tot, 5), "ms": round(v / 1e6, 3), "launches": calls[k]}
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1])}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
