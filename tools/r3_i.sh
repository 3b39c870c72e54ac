#!/bin/bash
# round-3 GPU pass I: whole suite, the default bench line, sattn PMC after the XCD remap
out=gpurun_out/r3i; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["ms_per_step_events"])
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("step_traffic_ratio"), c.get("error"))
print(d.get("generate"))
print([(s["L"],s["avg_us"],s.get("torch_sdpa_us"),s.get("unet_dispatch")) for s in d["sattn"]["shapes"]])
print([(s["B"],s["Lq"],s["avg_us"],s["frac"]) for s in d["xattn"]["shapes"]])
print(d.get("cpu_baseline",{}).get("value"))
PY
bash tools/prof_round.sh r03b sattn > $out/prof_sattn.log 2>&1
