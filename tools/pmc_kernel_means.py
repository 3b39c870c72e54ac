"""Per-(kernel, grid) means of one rocprofv3 --pmc counter_collection.csv: python tools/pmc_kernel_means.py file.csv [substr]"""
import csv
import re
import sys
from collections import defaultdict

per = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r["Kernel_Name"])).split("(")[0][:60]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    per[(name, r.get("Grid_Size", ""), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
acc = defaultdict(lambda: defaultdict(list))
for (name, grid, _), c in per.items():
    for k, v in c.items():
        acc[(name, grid)][k].append(v)
for (name, grid), c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    line = f"{name} grid={grid} n={len(next(iter(c.values())))}"
    if "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
        line += f" mfma_util={m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32 * m['SQ_BUSY_CYCLES']):.3f}"
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        wc = m["SQ_WAVE_CYCLES"]
        line += " wait_any=%.2f wait_inst=%.2f active=%.2f" % (m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc)
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        line += " lds_conflict=%.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"])
    for k in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
        if k in m:
            line += f" {k}={m[k]:.0f}"
    print(line)
