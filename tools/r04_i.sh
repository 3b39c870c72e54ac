out=$PWD/gpurun_out/r04i; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"])
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("error"))
g=d.get("generate",{}); print("generate", g.get("value"), g.get("prompts_per_unet_call"), g.get("rowwise"), g.get("host_cpu_seconds_per_image"), g.get("host_cpu_cores_busy"))
print([(s["L"],s["avg_us"],s.get("frac")) for s in d["sattn"]["shapes"]])
PY
timeout 900 python bench.py --only generate --gen-batch 128 --gen-images 256 --gen-rowwise 0 > $out/gen128.json 2>> $out/gen.log; python -c "
import json; d=json.load(open('$out/gen128.json')); print('images/s b128', d.get('value'), d.get('seconds'), d.get('error'))"
bash tools/prof_generate.sh r04i/gen 20 64 > $out/gen_prof.log 2>&1; tail -42 $out/gen_prof.log | cut -c1-150
