out=gpurun_out/r3g; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
bash tools/prof_round.sh r03g sd14_erase1000p500 > $out/prof.log 2>&1; tail -3 $out/prof.log
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["ms_per_step_events"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"], d["roofline"]["step_traffic_ratio"])
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("step_traffic_ratio"), c.get("error"))
print(d.get("generate",{}).get("value"))
PY
