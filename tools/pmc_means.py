"""Per-kernel means of the rocprofv3 --pmc passes of one workload (the per-dispatch CSVs are several MB each):
    python tools/pmc_means.py <dir> <workload> <outdir>
reads <dir>/<workload>_pmc_{fetch,write,sq}_counter_collection.csv, writes <outdir>/<workload>_pmc_{fetch,write,sq}_means.csv
(kernel, counter, mean over launches, launches)."""
import csv
import os
import re
import sys
from collections import defaultdict

d, wl, out = sys.argv[1:4]
for p in ("fetch", "write", "sq"):
    src = os.path.join(d, f"{wl}_pmc_{p}_counter_collection.csv")
    if not os.path.exists(src):
        continue
    per = defaultdict(lambda: defaultdict(float))                      # (kernel, dispatch) -> counter -> sum over instances
    for r in csv.DictReader(open(src)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = name.split("(")[0][:100]
        per[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    acc = defaultdict(list)
    for (name, _), c in per.items():
        for k, v in c.items():
            acc[(name, k)].append(v)
    with open(os.path.join(out, f"{wl}_pmc_{p}_means.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "mean", "launches"])
        for (name, k), v in sorted(acc.items()):
            w.writerow([name, k, sum(v) / len(v), len(v)])
