# final evidence pass of round 4 (second session), ON THE GPU BOX: full GPU tests, smoke, counter passes of every bench workload
# folded into profiles/traffic.json ON the box, the default bench line AFTER the fold (traffic_stale false), steady-state kernel
# stats of the generation loop at 64 prompts per call, the N = 2 line as a dry run (two ranks on one device over gloo).
tag=${1:-r04g}
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
tail -3 $out/fold.log; cp profiles/traffic.json $out/traffic.json
for wl in sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2 xattn sattn; do python tools/pmc_means.py $out $wl $out; done
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$? seconds=$(( $(date +%s) - start ))"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"], d["roofline"].get("traffic_stale"))
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("error"))
g=d.get("generate",{}); print("generate", g.get("value"), g.get("prompts_per_unet_call"), (g.get("rowwise") or {}).get("value"), g.get("host_cpu_seconds_per_image"))
PY
bash tools/prof_generate.sh $tag/gen 20 64 > $out/gen_prof.log 2>&1; tail -22 $out/gen_prof.log | cut -c1-150
UCE_BENCH_SAME_DEVICE=1 UCE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --gen-images 32 --gen-batch 16 --no-configs > $out/bench_n2_gloo_dryrun.json 2> $out/bench_n2_gloo_dryrun.log; echo "n2 rc=$?"; head -c 400 $out/bench_n2_gloo_dryrun.json
bash tools/ab_gen.sh $tag/ab "default" "conv_w1_off UCE_CONV_W1=0" "default2"
find $out -name "*counter_collection.csv" -size +3M -delete
echo "total seconds=$(( $(date +%s) - start ))"; ls $out | head -80
