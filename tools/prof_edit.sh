#!/bin/bash
# rocprofv3 passes for one edit workload: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs.
# usage: tools/prof_edit.sh <outdir under gpurun_out> <workload> [steps]
wl=${2:-sd14_erase50}; steps=${3:-200}
out=$PWD/gpurun_out/$1; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o ${wl} --output-format csv -- python $repo/bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline --gen-images 0 > $out/${wl}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o ${wl}_pmc_fetch --output-format csv -- python $repo/bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --gen-images 0 > $out/${wl}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out -o ${wl}_pmc_write --output-format csv -- python $repo/bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --gen-images 0 > $out/${wl}_pmc_write.log 2>&1
cd $repo
grep -h "timed region" $out/${wl}_bench.log
python tools/kstats.py $out/${wl}_kernel_stats.csv 12
python tools/pmc_traffic.py $wl $out/${wl}_pmc_fetch_counter_collection.csv $out/${wl}_pmc_write_counter_collection.csv
find $out -name "*kernel_trace.csv" -delete
