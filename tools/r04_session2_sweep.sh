# self-attention dispatch and lazy-threshold sweep at 64 prompts per call (B = 128), SQ counters of k_sattn_h, the ASAN GPU test
out=$PWD/gpurun_out/r04e; mkdir -p $out
timeout 300 python -m pytest tests/test_stress_gpu.py -q -m gpu -k sanitizer -rs > $out/pytest_asan.log 2>&1; tail -4 $out/pytest_asan.log
for cfg in "0 8" "1 8" "2 8" "3 8" "0 4" "0 16"; do
  set -- $cfg
  UCE_SATTN_QT=$1 UCE_SATTN_LAZY=$2 timeout 300 python bench.py --only sattn --gen-batch 64 > $out/sattn_qt$1_lazy$2.json 2> $out/sattn_qt$1_lazy$2.log
  python - <<PY
import json
d=json.load(open("$out/sattn_qt$1_lazy$2.json"))
print("sattn qt=$1 lazy=$2", [(s["L"], s["dh"], s["avg_us"], s["frac"]) for s in d["shapes"]])
PY
done
bash tools/pmc_sattn.sh r04e/sattn_h_pmc > $out/sattn_h_pmc.txt 2>&1; grep "k_sattn" $out/sattn_h_pmc.txt | cut -c1-420
