#!/bin/bash
# generation-leg A/B on one box: default | small-L self-attention on the HIP kernel | one query tile per wave
out=gpurun_out/r3j; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 50 --warmup 5 > $out/$tag.json 2> $out/$tag.log
  python -c "import json; d=json.load(open('$out/$tag.json')); print('$tag', d['generate']['value'], d['generate']['seconds'])"; }
run default_1 A=1
run smallL_hip_1 UCE_SATTN_MIN_KEYS=0
run qt1_1 UCE_SATTN_QT=1
run default_2 A=1
run smallL_hip_2 UCE_SATTN_MIN_KEYS=0
run qt1_2 UCE_SATTN_QT=1
