#!/bin/bash
# SQ counters of the dense-apply kernel (one pass, counters only + kernel trace)
out=$PWD/gpurun_out/${1:-pmc_apply}; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $out -o pmc --output-format csv -- python $repo/tools/probe_apply.py > $out/pmc.log 2>&1
cd $repo
python - "$out/pmc_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_apply" in r["Kernel_Name"] and r["Grid_Size"] == str(1170 * 256):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):14.4g}  (n={len(v)})")
PY
find $out -name "*kernel_trace.csv" -delete
