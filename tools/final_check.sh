out=gpurun_out/r3final; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["ms_per_step_events"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic_stale"), d["roofline"]["step_frac"], d["roofline"]["step_traffic_ratio"])
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("step_traffic_ratio"), c.get("error"))
print(d.get("generate",{}).get("value"))
print([(s["L"],s["avg_us"],s.get("torch_sdpa_us"),s.get("unet_dispatch")) for s in d["sattn"]["shapes"]])
cb=d.get("cpu_baseline",{}); print(cb.get("value"), cb.get("cores"), cb.get("thread_sweep_seconds_per_edit"))
PY
