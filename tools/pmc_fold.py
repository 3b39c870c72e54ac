"""Fold rocprofv3 --pmc passes into profiles/traffic.json: HBM bytes per launch and MFMA utilisation of EVERY
kernel bench.py reports (edit kernels, k_xattn, k_sattn).

    python tools/pmc_fold.py edit  <workload> <dir>          # dir holds <workload>_pmc_{fetch,write,sq}_counter_collection.csv
    python tools/pmc_fold.py xattn <dir>                     # passes of `bench.py --only xattn`: the shapes and the launches
    python tools/pmc_fold.py sattn <dir>                     #   per shape are read from the bench line in <mode>_pmc_<pass>.log

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE
come from separate passes, rocprofv3 reports them in kilobytes (value * 1024 bytes), and on gfx950 FETCH_SIZE counts
a wide coalesced streaming read at exactly half its bytes, so it is doubled.  MFMA utilisation =
SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (1024 * kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32
(one count per shader engine).  Values are means over the launches of that kernel in the pass.
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAFFIC = os.path.join(ROOT, "profiles", "traffic.json")
EDIT_KERNELS = ["k_lr_resident", "k_lr_project", "k_lr_update_s", "k_lr_update", "k_lr_fused", "k_trisolve", "k_potrf_la", "k_potrf_first", "k_potrf_step",
                "k_potrf_panel", "k_potrf_diag", "k_trsm", "k_gram_primal", "k_gram_dual", "k_apply_b3", "k_split3", "k_apply_h2", "k_split_h2d", "k_split_h2", "k_apply",
                "k_delta_factors", "k_apply_lowrank_generic", "k_reduce_slabs", "k_trinv_merge", "k_trinv_fwd", "k_trinv_bwd"]
# bench.py's launch-chain scopes: per-step bytes = all launches of the members / launches of the FIRST member
CHAINS = {"potrf": ["k_potrf_la", "k_potrf_first", "k_potrf_step", "k_potrf_panel", "k_potrf_diag"],
          "k_trisolve": ["k_trinv_fwd", "k_trinv_merge", "k_trinv_bwd"],
          "gram_primal": ["k_gram_primal", "k_reduce_slabs"]}     # uce_edit's primal path: A split over the concepts + its reduction
# ONE launch of one of these per call (a self-attention of at most 128 keys runs k_xattn: uce_sattn.hip's short-key rule)
MAIN_KERNELS = {"xattn": ("k_xattn_g", "k_xattn"), "sattn": ("k_sattn_h", "k_sattn_p", "k_sattn", "k_xattn")}
CANONICAL = {"xattn": "k_xattn", "sattn": "k_sattn"}                                                  # the name the folded entry carries
AUX_KERNELS = {"xattn": (), "sattn": ("k_vt",)}                                                      # helpers a call may launch BEFORE its main kernel


def short(name: str, wanted):
    for k in sorted(wanted, key=len, reverse=True):
        if name.startswith(k + "<") or name.startswith(k + "(") or name == k or ("::" + k + "<") in name \
                or ("::" + k + "(") in name or (" " + k + "<") in name or (" " + k + "(") in name:
            return k
    return None


def dispatches(path):
    """[(dispatch id, kernel name, {counter: value}, grid size)] in dispatch order."""
    acc = {}
    for r in csv.DictReader(open(path)):
        did = int(r["Dispatch_Id"])
        ent = acc.setdefault(did, (r["Kernel_Name"], {}, r.get("Grid_Size", "")))
        ent[1][r["Counter_Name"]] = ent[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [(d, n, c, g) for d, (n, c, g) in sorted(acc.items())]


def mean(v):
    return sum(v) / len(v) if v else None


def fold_groups(groups_fetch, groups_write, groups_sq):
    """groups_*: {key: [counter dict per launch]} -> {key: entry}."""
    out = {}
    for key in groups_fetch.keys() | groups_write.keys() | groups_sq.keys():
        f = [c["FETCH_SIZE"] for c in groups_fetch.get(key, []) if "FETCH_SIZE" in c]
        w = [c["WRITE_SIZE"] for c in groups_write.get(key, []) if "WRITE_SIZE" in c]
        ent = {}
        if f and w:
            ent["fetch_bytes"] = 2.0 * 1024.0 * mean(f)             # KB -> B, x2 gfx950 correction
            ent["write_bytes"] = 1024.0 * mean(w)
            ent["total_bytes"] = ent["fetch_bytes"] + ent["write_bytes"]
            ent["launches"] = len(f)
        sq = groups_sq.get(key, [])
        busy = [c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in sq if "SQ_VALU_MFMA_BUSY_CYCLES" in c]
        cyc = [c["SQ_BUSY_CYCLES"] for c in sq if "SQ_BUSY_CYCLES" in c]
        if busy and cyc and mean(cyc) > 0:
            ent["mfma_util"] = round(mean(busy) / (32.0 * mean(cyc)), 4)
            for name in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
                         "SQ_LDS_IDX_ACTIVE"):
                v = [c[name] for c in sq if name in c]
                if v:
                    ent.setdefault("sq", {})[name] = mean(v)
        if ent:
            out[key] = ent
    return out


def by_kernel(path, wanted):
    g = defaultdict(list)
    if path and os.path.exists(path):
        for _, name, c, _g in dispatches(path):
            k = short(name, wanted)
            if k:
                g[k].append(c)
    return g




class FoldError(RuntimeError):
    pass


def bench_line(log_path):
    """The JSON line `bench.py --only xattn|sattn` printed into the pass's log: the shapes IN LAUNCH ORDER, each with the number
    of calls bench.time_kernel made for it (`launches` = untimed burst + timed; bench.kernel_launches is the one place that
    number comes from)."""
    last = None
    with open(log_path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith("{") and '"shapes"' in line:
                last = line
    if last is None:
        raise FoldError(f"{log_path}: no bench.py JSON line with `shapes`")
    shapes = json.loads(last)["shapes"]
    if any("launches" not in e for e in shapes):
        raise FoldError(f"{log_path}: the bench line carries no `launches` per shape (bench.py older than the fold)")
    return shapes


def shape_key(mode, e):
    return f"B{e['B']}_Lq{e['Lq']}_dh{e['dh']}" if mode == "xattn" else f"B{e['B']}_L{e['L']}_dh{e['dh']}"


def by_manifest(path, mode, shapes):
    """Attribute the dispatches of one pass to the shapes of the bench line.  A call launches exactly ONE main kernel
    (MAIN_KERNELS[mode], whatever its template form) after its helpers (AUX_KERNELS): walking the pass in dispatch order, the
    n-th main launch belongs to the shape whose cumulative `launches` range holds n, helpers to the main launch that follows
    them.  Refuses (FoldError) when the pass does not hold exactly the launches the line announces, or when the main launches
    attributed to one shape are not ONE kernel instantiation at ONE grid size - a launch-order split that drifted by one
    call mixes shapes silently (round 5: 5 per shape assumed, 8 issued)."""
    g = defaultdict(list)
    if not (path and os.path.exists(path)):
        return g
    main, aux = MAIN_KERNELS[mode], AUX_KERNELS[mode]
    bounds, tot = [], 0
    for e in shapes:
        tot += int(e["launches"])
        bounds.append(tot)
    n_main, pending, sig = 0, [], {}
    for _, name, c, grid in dispatches(path):
        k = short(name, list(main) + list(aux))
        if k is None:
            continue
        if k in aux:
            pending.append((k, c))
            continue
        idx = next((i for i, b in enumerate(bounds) if n_main < b), None)
        n_main += 1
        if idx is None:
            continue                                   # counted; the total check below reports it
        key = shape_key(mode, shapes[idx])
        sig.setdefault(key, set()).add((name.split("(")[0], grid))
        g[(key, CANONICAL[mode])].append(c)            # every form of the main kernel under one name
        for ak, ac in pending:
            g[(key, ak)].append(ac)
        pending = []
    if n_main != tot:
        raise FoldError(f"{path}: {n_main} launches of {'/'.join(main)} in the pass, the bench line announces {tot} "
                        f"({[int(e['launches']) for e in shapes]} per shape)")
    for key, sg in sig.items():
        if len(sg) != 1:
            raise FoldError(f"{path}: the launches attributed to {key} are not one kernel at one grid: {sorted(sg)}")
    return g


def load():
    return json.load(open(TRAFFIC)) if os.path.exists(TRAFFIC) else {}


def lib_abi():
    """Hash of the kernel sources in the tree (the state the passes were run on, when folded right after them)."""
    try:
        sys.path.insert(0, ROOT)
        from uce_amd import build
        return build.source_hash()
    except Exception:  # noqa: BLE001
        return None


def save(data):
    data["_comment"] = ("HBM bytes per launch and MFMA utilisation from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE / SQ passes, "
                        "tools/prof_round.sh -> tools/pmc_fold.py); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide "
                        "coalesced reads on gfx950; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES); `source` names the "
                        "directory of the CSVs (copies under profiles/)")
    json.dump(data, open(TRAFFIC, "w"), indent=1, sort_keys=True)


def paths(d, prefix):
    return [os.path.join(d, f"{prefix}_pmc_{p}_counter_collection.csv") for p in ("fetch", "write", "sq")]


def main():
    mode = sys.argv[1]
    data = load()
    if mode == "edit":
        workload, d = sys.argv[2], sys.argv[3]
        pf, pw, ps = paths(d, workload)
        ent = fold_groups(by_kernel(pf, EDIT_KERNELS), by_kernel(pw, EDIT_KERNELS), by_kernel(ps, EDIT_KERNELS))
        for chain, members in CHAINS.items():
            have = [m for m in members if m in ent and "total_bytes" in ent[m]]
            if have and chain not in ent:
                firsts = ent[have[0]]["launches"]
                tot = sum(ent[m]["total_bytes"] * ent[m]["launches"] for m in have) / max(firsts, 1)
                ent[chain] = {"total_bytes": tot, "launches": firsts, "members": have}
                if chain == "gram_primal" and len(have) > 1:       # the scope of that name in bench.py covers both launches
                    ent["k_gram_primal"]["chain_bytes"] = tot
        for e in ent.values():
            e["source"] = os.path.basename(os.path.normpath(d))
            e["src"] = lib_abi()
        data[workload] = ent
        for k, e in sorted(ent.items()):
            print(f"{workload} {k}: " + (f"{e['total_bytes'] / 1e6:.2f} MB/launch " if "total_bytes" in e else "")
                  + (f"mfma_util {e['mfma_util']}" if "mfma_util" in e else ""))
    elif mode in ("xattn", "sattn"):
        d = sys.argv[2]
        pf, pw, ps = paths(d, mode)
        names = [CANONICAL[mode]] + list(AUX_KERNELS[mode])
        groups = []
        for p, tag in zip((pf, pw, ps), ("fetch", "write", "sq")):
            log = os.path.join(d, f"{mode}_pmc_{tag}.log")
            groups.append(by_manifest(p, mode, bench_line(log)) if os.path.exists(p) else defaultdict(list))
        ent = fold_groups(*groups)
        merged = {}
        for (key, kern), e in ent.items():
            m = merged.setdefault(key, {"kernels": {}})
            m["kernels"][kern] = e
        for key, m in merged.items():
            m["total_bytes"] = sum(e.get("total_bytes", 0.0) for e in m["kernels"].values())
            main_k = m["kernels"].get(names[0], {})
            if "mfma_util" in main_k:
                m["mfma_util"] = main_k["mfma_util"]
            m["source"] = os.path.basename(os.path.normpath(d))
            m["src"] = lib_abi()
            print(f"{mode} {key}: {m['total_bytes'] / 1e6:.2f} MB/launch mfma_util {m.get('mfma_util')}")
        data[mode] = merged
    else:
        raise SystemExit(__doc__)
    save(data)


if __name__ == "__main__":
    main()
