out=gpurun_out/r04j; mkdir -p $out
timeout 900 python -m pytest tests/test_latents_gpu.py -m gpu -q --timeout 600 -x -k generation_batch -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "batch|passed|failed|Error" $out/pytest.log | tail -5 | cut -c1-400
for v in "default:" "alllib:UCE_LINEAR_MIN_TILES=100000000" "noconv:UCE_CONV_IGEMM=never"; do
name=${v%%:*}; envs=${v#*:}
env $envs timeout 600 python bench.py --only generate --gen-images 2 --gen-batch 2 --gen-rowwise 6 > $out/row_$name.json 2>> $out/row.log
python -c "
import json; d=json.load(open('$out/row_$name.json')); print('$name', d.get('value'), (d.get('rowwise') or {}).get('value'), d.get('error'))"
done
