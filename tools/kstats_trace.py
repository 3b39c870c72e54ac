"""Steady-state kernel shares out of a rocprofv3 *_kernel_trace.csv: aggregates only the last `frac` of the launches
(in start order), so first-call effects (fresh allocations, library tuning kernels, weight initialisation) do not
pollute the averages.  Also prints min / median / max for the kernels named in UCE_KSTATS_WATCH (comma separated
substrings).  Usage: kstats_trace.py trace.csv [frac=0.5] [top=40] [out.csv]"""
import os
import csv, sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = sys.argv[4] if len(sys.argv) > 4 else None
rows = []
with open(path) as f:
    rd = csv.DictReader(f)
    key = {k.lower(): k for k in rd.fieldnames}
    ks, ke, kn = key["start_timestamp"], key["end_timestamp"], key["kernel_name"]
    kg = [key.get(k) for k in ("grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x")]
    for r in rd:
        rows.append((int(r[ks]), int(r[ke]), r[kn], tuple(int(r[k]) if k else 0 for k in kg)))
rows.sort()
t1 = max(r[1] for r in rows)
cut = rows[int(len(rows) * (1.0 - frac))][0]
agg = defaultdict(lambda: [0, 0])
for s, e, n, _ in rows:
    if s >= cut:
        a = agg[n]
        a[0] += 1
        a[1] += e - s
tot = sum(a[1] for a in agg.values())
span = t1 - cut
print(f"steady window {span / 1e6:.1f} ms, kernel time {tot / 1e6:.1f} ms ({100 * tot / span:.1f} % busy), "
      f"{sum(a[0] for a in agg.values())} launches")
order = sorted(agg.items(), key=lambda kv: -kv[1][1])
for n, (c, d) in order[:top]:
    print("%-100s %6d %9.2f ms %9.1f us %6.2f%%" % (n[:100], c, d / 1e6, d / c / 1e3, 100 * d / tot))
if out:
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for n, (c, d) in order:
            w.writerow([n, c, d, d / c, 100 * d / tot])

for pat in filter(None, os.environ.get("UCE_KSTATS_WATCH", "").split(",")):
    d = sorted(e - s for s, e, n, _ in rows if s >= cut and pat in n)
    if d:
        by_grid = defaultdict(list)
        for s, e, n, g in rows:
            if s >= cut and pat in n:
                by_grid[g].append(e - s)
        for g, v in sorted(by_grid.items()):
            v.sort()
            print(f"   {pat} grid {g}: n {len(v)} min {v[0] / 1e3:.1f} median {v[len(v) // 2] / 1e3:.1f} max {v[-1] / 1e3:.1f} us")
        print(f"{pat}: n {len(d)} min {d[0] / 1e3:.1f} us  p25 {d[len(d) // 4] / 1e3:.1f}  median {d[len(d) // 2] / 1e3:.1f}  "
              f"p75 {d[3 * len(d) // 4] / 1e3:.1f}  max {d[-1] / 1e3:.1f}  total {sum(d) / 1e6:.1f} ms")
