out=$PWD/gpurun_out/r04p; mkdir -p $out
for v in new old new old; do
  if [ $v = old ]; then export UCE_SATTN_QT=2; else unset UCE_SATTN_QT; fi
  timeout 600 python bench.py --only generate > $out/gen_$v.json 2> $out/gen_$v.log; python -c "
import json; d=json.load(open('$out/gen_$v.json')); g=d.get('generate',d); print('$v generate', g.get('value'), g.get('prompts_per_unet_call'))"
done
