out=gpurun_out/r3h2; mkdir -p $out
timeout 900 python -m pytest tests/test_edit_gpu.py tests/test_stress_gpu.py tests/test_sdxl_gpu.py tests/test_pipeline_gpu.py -q -x --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
for v in 2; do
UCE_APPLY_VARIANT=$v timeout 600 python bench.py --workload sd14_erase1000p500 --only edit --no-cpu-baseline --no-configs > $out/b_$v.json 2> $out/b_$v.log; echo "bench v$v rc=$?"
python - <<PY
import json
d=json.load(open("$out/b_$v.json"))
r=d["roofline"]
print($v, d["ms_per_step"], d["ms_per_step_events"], r["kernel"], r["avg_ms"], [(k["kernel"], k["launches_per_step"], k["avg_ms"], k["frac"]) for k in r["kernels"]])
PY
done
