#!/bin/bash
# rocprofv3 SQ counters of the GEMM probe for one shape: tools/ubench/pmc_gemm.sh <src.hip> M N K
repo=$(cd "$(dirname "$0")/../.." && pwd); src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -mllvm -amdgpu-mfma-vgpr-form $repo/$src -o /tmp/gemm_probe || exit 1
mkdir -p $repo/gpurun_out/gp; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace -d $repo/gpurun_out/gp -o gp --output-format csv -- /tmp/gemm_probe "$@" > /dev/null 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$repo/gpurun_out/gp/gp_counter_collection.csv")):
    if "gemm" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
d=list(acc.values())[-1]
print({k: round(v) for k,v in d.items()})
print("mfma busy %.3f | lds busy (per CU) %.3f | lds conflict / active %.3f | wait_inst/wave %.3f" % (
  d["SQ_VALU_MFMA_BUSY_CYCLES"]/(32*d["SQ_BUSY_CYCLES"]), d["SQ_LDS_IDX_ACTIVE"]/(8*d["SQ_BUSY_CYCLES"]),
  d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1), d["SQ_WAIT_INST_ANY"]/d["SQ_WAVE_CYCLES"]))
PY
