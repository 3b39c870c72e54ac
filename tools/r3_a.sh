#!/bin/bash
# round-3 GPU pass A: full -m gpu suite, edit benches of every workload, chain timelines (debug library)
out=gpurun_out/r3a; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
for wl in sd14_erase50 sd14_erase100 sd14_erase2p3 sd14_erase1000p500 sdxl_debias36x2; do
  timeout 300 python bench.py --only edit --workload $wl --steps 100 --warmup 10 > $out/e_$wl.json 2> $out/e_$wl.log
  python - <<PY
import json
try:
    d=json.load(open("$out/e_$wl.json")); r=d["roofline"]
    print("$wl", d["ms_per_step_events"], [(r["kernel"], r["avg_ms"])]+[(k["kernel"],k["avg_ms"]) for k in r["kernels"]], r.get("gemm_alone"))
except Exception as e: print("$wl failed", e)
PY
done
for wl in sd14_erase100 sd14_erase50; do
  UCE_CHAIN_DEBUG=1 timeout 300 python tools/dbg_chain.py $wl > $out/chain_$wl.txt 2>&1
done
head -45 $out/chain_sd14_erase100.txt
