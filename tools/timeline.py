"""Per-dispatch timeline of the LAST occurrence of a kernel sequence in a rocprofv3 --kernel-trace CSV:
    python tools/timeline.py <kernel_trace.csv> <first kernel substring> [count]
prints name, workgroups, duration and the gap to the previous dispatch (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]][-1]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prev = None
for r in rows[idx:idx + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:34]
    wgs = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    print("%-34s grid %8s dur %8.2f us gap %6.2f" % (name, wgs, (e - s) / 1e3, gap))
    prev = e
