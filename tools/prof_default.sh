#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the line the driver records), summary -> gpurun_out/<tag>/
tag=${1:-r03_default}; out=$PWD/gpurun_out/$tag; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o bench_default --output-format csv -- python $repo/bench.py > $out/bench_default.json 2> $out/bench_default.log
cd $repo
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
python tools/kstats.py $out/bench_default_kernel_stats.csv 12
tail -c 600 $out/bench_default.json | head -c 300; echo
