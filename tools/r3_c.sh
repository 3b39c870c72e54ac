#!/bin/bash
# round-3 GPU pass C: edit tests, edit benches (x2), chain timelines
out=gpurun_out/r3c; mkdir -p $out
timeout 600 python -m pytest tests/test_edit_gpu.py tests/test_sdxl_gpu.py -m gpu -q --timeout 300 -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); r=d["roofline"]
    print("$1", d["ms_per_step_events"], [(r["kernel"], r["avg_ms"])]+[(k["kernel"],k["avg_ms"]) for k in r["kernels"]], r.get("gemm_alone"))
except Exception as e: print("$1 failed", e)
PY
}
for rep in 1 2; do
for wl in sd14_erase50 sd14_erase100 sd14_erase2p3 sd14_erase1000p500 sdxl_debias36x2; do
  timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 > $out/e_${wl}_$rep.json 2> $out/e_${wl}_$rep.log; show $out/e_${wl}_$rep.json
done
done
for wl in sd14_erase100 sd14_erase50; do
UCE_CHAIN_DEBUG=1 timeout 300 python tools/dbg_chain.py $wl > $out/chain_$wl.txt 2>&1
grep "gram/proj" $out/chain_$wl.txt | awk '$6!="None," && $7!="None,"' | head -4; grep -E "^(12|25) " $out/chain_$wl.txt
done
