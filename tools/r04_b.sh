out=gpurun_out/r04b; mkdir -p $out
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_norm_gpu.py tests/test_stress_gpu.py tests/test_latents_gpu.py -m gpu -q --timeout 300 -x > $out/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> $out/pytest_new.log
tail -6 $out/pytest_new.log
timeout 600 python tools/probe_r04.py gemm conv > $out/probe.log 2>&1; echo "probe rc=$?"
grep -v amdgpu.ids $out/probe.log | python tools/probe_r04_fmt.py
timeout 600 python bench.py --only generate > $out/gen.json 2> $out/gen.log; echo "gen rc=$?"; python -c "
import json; d=json.load(open('$out/gen.json')); print('images/s', d.get('value'), d.get('seconds'), d.get('failure'))"
UCE_WIDE_EPILOGUE=0 timeout 600 python bench.py --only generate > $out/gen_narrow.json 2>> $out/gen.log; python -c "
import json; d=json.load(open('$out/gen_narrow.json')); print('images/s narrow epilogue', d.get('value'), d.get('seconds'), d.get('failure'))"
