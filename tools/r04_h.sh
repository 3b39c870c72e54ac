out=gpurun_out/r04h; mkdir -p $out
timeout 900 python -m pytest tests/test_edit_gpu.py -m gpu -q --timeout 300 -x -k "golden or two_stream or full_size or switches" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
for f in 2 0; do
UCE_EDIT_FUSED=$f timeout 600 python bench.py --no-cpu-baseline --gen-images 0 > $out/bench_fused$f.json 2> $out/bench_fused$f.log; echo "bench fused=$f rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_fused$f.json"))
r=d["roofline"]
print("fused=$f", d["value"], d["ms_per_step"], d["ms_per_step_events"], r["kernel"], r["avg_ms"], r["bound"], r["frac"], r["step_frac"])
for c in d.get("configs",[]): print("  ", c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("error"))
PY
done
