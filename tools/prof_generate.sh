#!/bin/bash
# rocprofv3 kernel stats of the steady-state denoising loop (eager launches).  Run on the GPU box from the repo root.
out=$PWD/gpurun_out/${1:-gen}
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $out -o gen --output-format csv -- python $repo/tools/probe_generate3.py ${2:-20} ${3:-8} > $out/probe.log 2>&1
cd $repo
tail -3 $out/probe.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python tools/kstats.py "$f" 60
find $out -name "*kernel_trace.csv" -delete
