out=$PWD/gpurun_out/r04o; mkdir -p $out
timeout 600 python -m pytest tests/test_sattn_gpu.py tests/test_xattn_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -4
OUT=$out timeout 300 python tools/probe_sattn_h.py 2>&1 | grep packed
timeout 900 python bench.py --only generate > $out/gen.json 2> $out/gen.log; python -c "
import json; d=json.load(open('$out/gen.json')); g=d.get('generate',d); print('generate', g.get('value'), g.get('prompts_per_unet_call'))"
