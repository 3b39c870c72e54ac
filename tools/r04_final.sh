# round-4 evidence pass on the GPU box: full GPU tests, smoke, default bench, rocprofv3 stats + PMC of every bench workload and
# attention leg, steady-state kernel stats of the generation loop at the bench's batch
tag=${1:-r04}
out=$PWD/gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_frac"], d["roofline"].get("traffic_stale"))
for c in d.get("configs",[]): print(c.get("workload"), c.get("ms_per_step_events"), c.get("step_frac"), c.get("error"))
g=d.get("generate",{}); print("generate", g.get("value"), g.get("prompts_per_unet_call"), (g.get("rowwise") or {}).get("value"), g.get("host_cpu_seconds_per_image"))
PY
bash tools/prof_round.sh $tag sd14_erase50 sd14_erase2p3 sd14_erase100 sd14_erase1000p500 sdxl_debias36x2 xattn sattn > $out/prof.log 2>&1
tail -3 $out/prof.log
bash tools/prof_generate.sh $tag/gen 20 64 > $out/gen_prof.log 2>&1; tail -25 $out/gen_prof.log | cut -c1-150
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 valu_mfma.hip -o /tmp/valu_mfma 2>/dev/null && timeout 60 /tmp/valu_mfma) > $out/ubench_valu_mfma.txt 2>&1; tail -3 $out/ubench_valu_mfma.txt
bash tools/pmc_sattn.sh $tag/sattn_h_pmc > $out/sattn_h_pmc.txt 2>&1; grep k_sattn $out/sattn_h_pmc.txt | cut -c1-200
ls $out | head -60
