# final evidence pass of round 6 (third session), ON THE GPU BOX: full GPU tests, smoke, counter passes of every bench workload
# folded into profiles/traffic.json ON the box, the default bench line and the driver-argument line AFTER the fold (traffic_stale
# false), steady-state kernel stats of the generation loop at 128 prompts per call and row by row.
tag=${1:-r06k}
out=$PWD/gpurun_out/$tag; mkdir -p $out
start=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rs > $out/pytest.log 2>&1; echo "pytest rc=$? seconds=$(( $(date +%s) - start ))" >> $out/pytest.log
tail -4 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/prof_round.sh $tag sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2 xattn sattn > $out/prof.log 2>&1
tail -2 $out/prof.log
for wl in sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2; do python tools/pmc_fold.py edit $wl $out >> $out/fold.log 2>&1; done
python tools/pmc_fold.py xattn $out 2,128 >> $out/fold.log 2>&1
python tools/pmc_fold.py sattn $out 128 >> $out/fold.log 2>&1
cat $out/fold.log | tail -40; cp profiles/traffic.json $out/traffic.json
for wl in sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2 xattn sattn; do python tools/pmc_means.py $out $wl $out; done
for wl in xattn sattn; do for p in fetch write sq; do cp $out/${wl}_pmc_${p}.log $out/${wl}_pmc_${p}_pass.log 2>/dev/null; done; done
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$? seconds=$(( $(date +%s) - start ))"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.log; echo "bench (driver arguments) rc=$?"
python - <<PY
import json
for f in ("bench_default", "bench_driver_args"):
    d=json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"], d["roofline"].get("traffic_stale"))
    for c in d.get("configs",[]): print("  ", c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("error"))
    g=d.get("generate",{}); print("  generate", g.get("value"), (g.get("rowwise") or {}).get("value"))
    for s in d.get("sattn",{}).get("shapes",[]): print("  sattn", s.get("L"), s.get("avg_us"), s.get("exp2_domain_us"), s.get("traffic"), s.get("traffic_stale"))
PY
bash tools/prof_generate.sh $tag/gen128 20 128 > $out/gen128_prof.log 2>&1; tail -14 $out/gen128_prof.log | cut -c1-150
bash tools/prof_generate.sh $tag/gen1 20 1 > $out/gen1_prof.log 2>&1; tail -8 $out/gen1_prof.log | cut -c1-150
python tools/kfamilies.py $out/gen128/steady_kernel_stats.csv $out/gen128/families.json "128 prompts per call, 20 steps, steady half"
python tools/kfamilies.py $out/gen1/steady_kernel_stats.csv $out/gen1/families.json "1 prompt per call, 20 steps, steady half"
find $out -name "*counter_collection.csv" -size +3M -delete
echo "total seconds=$(( $(date +%s) - start ))"; ls $out | head -100
