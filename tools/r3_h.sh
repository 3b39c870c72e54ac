#!/bin/bash
# round-3 GPU pass H: persistent Cholesky with side prefetch; new sattn tests
out=gpurun_out/r3h; mkdir -p $out
timeout 1200 python -m pytest tests/test_edit_gpu.py tests/test_sattn_gpu.py tests/test_stress_gpu.py tests/test_sdxl_gpu.py -m gpu -q --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); r=d["roofline"]
    print("$1", d["ms_per_step_events"], [(r["kernel"], r["avg_ms"])]+[(k["kernel"],k["avg_ms"]) for k in r["kernels"]])
except Exception as e: print("$1 failed", e)
PY
}
for rep in 1 2 3; do
  timeout 300 python bench.py --only edit --workload sd14_erase1000p500 --steps 100 --warmup 10 > $out/la_$rep.json 2> $out/la_$rep.log; show $out/la_$rep.json
done
