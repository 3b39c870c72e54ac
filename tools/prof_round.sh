#!/bin/bash
# rocprofv3 passes of one round, run ON THE GPU BOX (gpurun); the CSVs come back under gpurun_out/<tag>/ and are folded
# locally:  python tools/pmc_fold.py edit <workload> gpurun_out/<tag> ; python tools/pmc_fold.py xattn gpurun_out/<tag> 2,32 ; ...
#   usage: tools/prof_round.sh <tag> [workloads...]     (default workloads: sd14_erase50 sd14_erase1000p500; add xattn / sattn)
# Per workload: --kernel-trace --stats (per-kernel durations), then FETCH_SIZE, WRITE_SIZE and the SQ counters in their OWN
# passes (counters are never combined with the runtime/sys traces, as the pool requires).
tag=${1:-prof}; shift
wls=${@:-sd14_erase50 sd14_erase1000p500 xattn sattn}
out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for wl in $wls; do
  case $wl in
    xattn|sattn) args="--only $wl"; sargs="--only $wl" ;;
    *) args="--only edit --workload $wl --steps 20 --warmup 2"; sargs="--only edit --workload $wl --steps 200 --warmup 20" ;;
  esac
  rocprofv3 --kernel-trace --stats -d $out -o ${wl} --output-format csv -- python $repo/bench.py $sargs > $out/${wl}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o ${wl}_pmc_fetch --output-format csv -- python $repo/bench.py $args > $out/${wl}_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out -o ${wl}_pmc_write --output-format csv -- python $repo/bench.py $args > $out/${wl}_pmc_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace -d $out -o ${wl}_pmc_sq --output-format csv -- python $repo/bench.py $args > $out/${wl}_pmc_sq.log 2>&1
  python $repo/tools/kstats.py $out/${wl}_kernel_stats.csv 14
done
cd $repo
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
ls -la $out | head -40
