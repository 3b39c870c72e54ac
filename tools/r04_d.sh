out=gpurun_out/r04d; mkdir -p $out
timeout 900 python -m pytest tests/test_linear_gpu.py -m gpu -q --timeout 300 -x > $out/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> $out/pytest_new.log
tail -4 $out/pytest_new.log
timeout 600 python tools/probe_r04.py gemm > $out/probe.log 2>&1; echo "probe rc=$?"
grep -v amdgpu.ids $out/probe.log | python tools/probe_r04_fmt.py
timeout 600 python bench.py --only generate > $out/gen.json 2> $out/gen.log; echo "gen rc=$?"; python -c "
import json; d=json.load(open('$out/gen.json')); print('images/s b16', d.get('value'), d.get('seconds'), d.get('failure'))"
timeout 600 python bench.py --only generate --gen-batch 32 --gen-images 64 > $out/gen32.json 2>> $out/gen.log; python -c "
import json; d=json.load(open('$out/gen32.json')); print('images/s b32', d.get('value'), d.get('seconds'), d.get('failure'))"
