"""Print the top rows of a rocprofv3 *_kernel_stats.csv (name shortened)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    print("%-100s %6s %9.2f ms %9.1f us %6s%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, r["Percentage"]))
