#!/bin/bash
# SQ counters of the edit kernels (one pass, counters only + kernel trace): tools/pmc_edit.sh <outdir> [workload]
wl=${2:-sd14_erase50}
out=$PWD/gpurun_out/${1:-pmc_edit}; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $out -o pmc --output-format csv -- python $repo/bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --gen-images 0 > $out/pmc.log 2>&1
cd $repo
python - "$out/pmc_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    key = next((k for k in ("k_lr_update_s", "k_lr_project", "k_trisolve", "k_apply_b3", "k_potrf_step", "k_gram_primal") if k in name), None)
    if key:
        acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:16s} {c:28s} {sum(v)/len(v):14.4g}  (n={len(v)})")
PY
find $out -name "*kernel_trace.csv" -delete
