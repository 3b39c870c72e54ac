out=gpurun_out/r04e; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -12 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"])
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("error"))
g=d.get("generate",{}); print("generate", g.get("value"), g.get("rowwise"), g.get("host_cpu_seconds_per_image"), g.get("host_cpu_cores_busy"))
print("wall", json.dumps(d.get("uce_wall_s"))[:1500])
cb=d.get("cpu_baseline",{}); print(cb.get("value"), cb.get("cores"), cb.get("thread_sweep_seconds_per_edit")); print(json.dumps(cb.get("configs"))[:1200])
PY
for b in 32 64; do
timeout 600 python bench.py --only generate --gen-batch $b --gen-images $((2*b)) --gen-rowwise 0 > $out/gen$b.json 2>> $out/gen.log; python -c "
import json; d=json.load(open('$out/gen$b.json')); print('images/s b$b', d.get('value'), d.get('seconds'), d.get('failure'))"
done
