#!/bin/bash
# round-3 GPU pass D: new stress tests + attention / pipeline tests, then the PMC passes of the N = 100 and config-3 edits
out=gpurun_out/r3d; mkdir -p $out
timeout 900 python -m pytest tests/test_stress_gpu.py tests/test_xattn_gpu.py tests/test_pipeline_gpu.py tests/test_generate_gpu.py -m gpu -q --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 900 bash tools/prof_round.sh r3d_prof sd14_erase100 sd14_erase1000p500 > $out/prof.log 2>&1
tail -30 $out/prof.log
