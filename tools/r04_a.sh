out=gpurun_out/r04a; mkdir -p $out
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_norm_gpu.py tests/test_sattn_gpu.py -m gpu -q --timeout 300 -x > $out/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> $out/pytest_new.log
tail -15 $out/pytest_new.log
timeout 600 python tools/probe_r04.py gemm conv sattn > $out/probe.log 2>&1; echo "probe rc=$?"
cat $out/probe.log | cut -c1-600
