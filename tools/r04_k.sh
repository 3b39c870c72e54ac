out=gpurun_out/r04k; mkdir -p $out
timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_latents_gpu.py -m gpu -q --timeout 600 -x -k "tile_form or generation_batch" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 600 python tools/probe_r04.py gemm > $out/probe.log 2>&1; echo "probe rc=$?"
grep -v amdgpu.ids $out/probe.log | python tools/probe_r04_fmt.py
