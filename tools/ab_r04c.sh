# second A/B pass: lazy running maximum (UCE_SATTN_LAZY), the folded kernel with it, diagnostics of the folded kernel at large scores
tag=${1:-r04c}
out=$PWD/gpurun_out/$tag; mkdir -p $out
timeout 300 python tools/diag_fold.py > $out/diag_fold.log 2>&1; tail -30 $out/diag_fold.log
timeout 900 python -m pytest tests/test_sattn_gpu.py -q --timeout 300 > $out/pytest_sattn.log 2>&1
echo "pytest rc=$?" >> $out/pytest_sattn.log; tail -8 $out/pytest_sattn.log
for cfg in "0 8" "1 8" "0 0" "1 0"; do
  set -- $cfg
  UCE_SATTN_FOLD=$1 UCE_SATTN_LAZY=$2 timeout 300 python bench.py --only sattn --gen-batch 64 > $out/sattn_fold$1_lazy$2.json 2> $out/sattn_fold$1_lazy$2.log
  python - <<PY
import json
d=json.load(open("$out/sattn_fold$1_lazy$2.json"))
print("sattn fold=$1 lazy=$2", [(s["L"], s["dh"], s["avg_us"], s["frac"]) for s in d["shapes"]])
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
run_gen lazy0 UCE_SATTN_LAZY=0
run_gen lazy8 UCE_SATTN_LAZY=8
run_gen lazy8_catfree UCE_SATTN_LAZY=8 UCE_CAT_FREE=1
run_gen lazy8_prescale_catfree UCE_SATTN_LAZY=8 UCE_CAT_FREE=1 UCE_SATTN_PRESCALE=1
run_gen lazy8b UCE_SATTN_LAZY=8
ls $out
