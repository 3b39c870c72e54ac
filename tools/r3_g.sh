#!/bin/bash
# round-3 GPU pass G: pipelined sattn variant (UCE_SATTN_QT=3), update-kernel occupancy variants
out=gpurun_out/r3g; mkdir -p $out
UCE_SATTN_QT=3 timeout 600 python -m pytest tests/test_sattn_gpu.py -m gpu -q --timeout 300 > $out/pytest_qt3.log 2>&1; echo "pytest rc=$?" >> $out/pytest_qt3.log
tail -4 $out/pytest_qt3.log
for v in 1 0 3 1 0 3; do
  UCE_SATTN_QT=$v timeout 300 python bench.py --only sattn > $out/sattn_qt$v.json 2> $out/sattn_qt$v.log
  python - <<PY
import json
d=json.load(open("$out/sattn_qt$v.json"))
print("QT variant $v", [(s["L"], s["dh"], s["avg_us"], s["achieved_TFLOPs"]) for s in d["shapes"]])
PY
done
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); r=d["roofline"]
    print("$1", d["ms_per_step_events"], [(r["kernel"], r["avg_ms"])]+[(k["kernel"],k["avg_ms"]) for k in r["kernels"]])
except Exception as e: print("$1 failed", e)
PY
}
for lib in libuce_hip.so libuce_hip.f5cd424d82.so libuce_hip.e97607a0c9.so; do
 for wl in sd14_erase50 sd14_erase2p3 sd14_erase100 sdxl_debias36x2; do
  UCE_HIP_LIB=$PWD/unified-concept-editing_amd/lib/$lib timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 > $out/${lib}_$wl.json 2> $out/${lib}_$wl.log; show $out/${lib}_$wl.json
 done
done
