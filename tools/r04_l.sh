out=gpurun_out/r04l; mkdir -p $out
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_norm_gpu.py tests/test_stress_gpu.py tests/test_latents_gpu.py -m gpu -q --timeout 600 -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 600 python tools/probe_r04.py conv > $out/probe.log 2>&1; echo "probe rc=$?"
grep -v amdgpu.ids $out/probe.log | python tools/probe_r04_fmt.py
timeout 600 python bench.py --only generate --gen-rowwise 0 > $out/gen.json 2> $out/gen.log; python -c "
import json; d=json.load(open('$out/gen.json')); print('images/s', d.get('value'), d.get('seconds'), d.get('error'))"
