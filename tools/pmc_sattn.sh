#!/bin/bash
# SQ counters of the self-attention forms at the generation shape (tools/probe_sattn_pmc.py): tools/pmc_sattn.sh <tag>
out=$PWD/gpurun_out/${1:-sattn_pmc}; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $out/sq_counters.txt; wc -l $out/sq_counters.txt
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace -d $out -o s1 --output-format csv -- python $repo/tools/probe_sattn_pmc.py > $out/s1.log 2>&1
python $repo/tools/pmc_kernel_means.py $out/s1_counter_collection.csv sattn
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $SQ2 --kernel-trace -d $out -o s2 --output-format csv -- python $repo/tools/probe_sattn_pmc.py > $out/s2.log 2>&1
tail -3 $out/s2.log
python - <<PY
import csv,collections
try:
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$out/s2_counter_collection.csv")):
        if 'sattn' in r["Kernel_Name"]: per[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in per.items(): print(k,{n:sum(v)/len(v) for n,v in c.items()})
except Exception as e: print("s2 failed",e)
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
