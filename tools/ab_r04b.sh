# round-4 (second session) A/B pass on the GPU box: parity of the new entry points, then same-box timings of the switches
#   UCE_SATTN_FOLD / UCE_SATTN_PRESCALE (k_sattn_h<FOLD>, q scaled in the projection's epilogue) and UCE_CAT_FREE (two-source GroupNorm / GEMM)
tag=${1:-r04b}
out=$PWD/gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_sattn_gpu.py tests/test_linear_gpu.py tests/test_norm_gpu.py -q -x --timeout 300 > $out/pytest_new.log 2>&1
echo "pytest rc=$?" >> $out/pytest_new.log; tail -5 $out/pytest_new.log
for f in 0 1; do
  UCE_SATTN_FOLD=$f timeout 300 python bench.py --only sattn --gen-batch 64 > $out/sattn_fold$f.json 2> $out/sattn_fold$f.log
  python - <<PY
import json
d=json.load(open("$out/sattn_fold$f.json"))
print("sattn fold=$f", [(s["L"], s["dh"], s["avg_us"], s["frac"]) for s in d["shapes"]])
PY
done
run_gen() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --only generate --gen-images 128 --gen-rowwise 0 > $out/gen_$name.json 2> $out/gen_$name.log
  python - <<PY
import json
try:
    d=json.load(open("$out/gen_$name.json")); print("generate $name", d.get("value"), d.get("unit"))
except Exception as e:
    print("generate $name FAILED", e)
PY
}
run_gen base UCE_SATTN_PRESCALE=0 UCE_CAT_FREE=0
run_gen prescale UCE_SATTN_PRESCALE=1 UCE_CAT_FREE=0
run_gen catfree UCE_SATTN_PRESCALE=0 UCE_CAT_FREE=1
run_gen both UCE_SATTN_PRESCALE=1 UCE_CAT_FREE=1
run_gen base2 UCE_SATTN_PRESCALE=0 UCE_CAT_FREE=0
ls $out
