out=$PWD/gpurun_out/r04d; mkdir -p $out
start=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$? seconds=$(( $(date +%s) - start ))" >> $out/pytest.log
tail -6 $out/pytest.log
timeout 200 python tools/probe_xattn64.py > $out/probe_xattn64.log 2>&1; cat $out/probe_xattn64.log | tail -6
for v in 1 2; do UCE_XATTN_VARIANT=$v timeout 200 python bench.py --only xattn --gen-batch 64 > $out/xattn_variant$v.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$out/xattn_variant$v.json"))
print("xattn variant $v", [(s.get("B"), s.get("Lq"), s.get("dh"), s.get("avg_us"), s.get("frac")) for s in d["shapes"]])
PY
done
bash tools/ab_gen.sh r04d "default" "cat0 UCE_CAT_FREE=0"
