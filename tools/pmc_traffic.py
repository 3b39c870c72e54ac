"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json (HBM bytes per launch).

    python tools/pmc_traffic.py <workload> <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel ...]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes (value * 1024 bytes), and on gfx950 FETCH_SIZE counts a
wide coalesced streaming read at exactly half its bytes, so it is doubled.  The per-launch value is the mean over
the launches of that kernel in the pass."""
import csv, json, os, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    workload, fetch_csv, write_csv = sys.argv[1:4]
    wanted = sys.argv[4:] or ["k_lr_update_s", "k_lr_project", "k_lr_update", "k_apply", "k_trisolve"]
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    entry = data.setdefault(workload, {})
    for key in wanted:
        fk = [k for k in f if key + "<" in k or key + "(" in k]
        wk = [k for k in w if key + "<" in k or key + "(" in k]
        if not fk or not wk:
            continue
        fv = [v for k in fk for v in f[k]]
        wv = [v for k in wk for v in w[k]]
        fetch = 2.0 * 1024.0 * sum(fv) / len(fv)          # KB -> B, x2 gfx950 correction
        write = 1024.0 * sum(wv) / len(wv)
        entry[key] = {"fetch_bytes": fetch, "write_bytes": write, "total_bytes": fetch + write, "launches": len(fv),
                      "source": [os.path.basename(fetch_csv), os.path.basename(write_csv)]}
        print(f"{workload} {key}: fetch {fetch/1e6:.1f} MB (x2-corrected) + write {write/1e6:.1f} MB = {(fetch+write)/1e6:.1f} MB over {len(fv)} launches")
    json.dump(data, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
