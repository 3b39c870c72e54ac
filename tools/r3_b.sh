#!/bin/bash
# round-3 GPU pass B: edit tests, N=100 / N=50 benches (product + the UCE_PJ_W1 experiment library), chain timeline
out=gpurun_out/r3b; mkdir -p $out
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
for wl in sd14_erase50 sd14_erase100 sd14_erase2p3; do
  timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 > $out/e_${wl}_$rep.json 2> $out/e_${wl}_$rep.log; show $out/e_${wl}_$rep.json
  UCE_HIP_LIB=$PWD/unified-concept-editing_amd/lib/libuce_hip.15f9257335.so timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 > $out/w1_${wl}_$rep.json 2> $out/w1_${wl}_$rep.log; show $out/w1_${wl}_$rep.json
done
done
UCE_CHAIN_DEBUG=1 timeout 300 python tools/dbg_chain.py sd14_erase100 > $out/chain_sd14_erase100.txt 2>&1
grep -E "^(7|0|5|11|12|25) " $out/chain_sd14_erase100.txt | head; grep "gram/proj" $out/chain_sd14_erase100.txt | awk '$6!="None," && $7!="None,"' | head -12
