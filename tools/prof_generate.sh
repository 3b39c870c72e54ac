#!/bin/bash
# rocprofv3 kernel stats of the steady-state denoising loop (eager launches).  Run on the GPU box from the repo root.
out=$PWD/gpurun_out/${1:-gen}
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $out -o gen --output-format csv -- python $repo/tools/prof_generate_loop.py ${2:-20} ${3:-8} > $out/probe.log 2>&1
cd $repo
tail -3 $out/probe.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python tools/kstats.py "$f" 12
# steady-state shares (second half of the run only), then drop the bulky trace
t=$(find $out -name "*kernel_trace.csv" | head -1)
UCE_KSTATS_WATCH=${UCE_KSTATS_WATCH:-k_im2col3x3} python tools/kstats_trace.py "$t" 0.5 30 $out/steady_kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
