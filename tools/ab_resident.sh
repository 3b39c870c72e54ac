cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_edit_gpu.py tests/test_sdxl_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -4
for d in "" "-DUCE_PROJECT_H2=0"; do
  for wl in sdxl_debias36x2; do
  UCE_DEFINES="$d" timeout 300 python bench.py --only edit --workload $wl --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print('H2[$d]', '$wl', p['ms_per_step'], p['ms_per_step_events'], [(k['kernel'],k['avg_ms']) for k in [p['roofline']]+p['roofline']['kernels']])"
  done
  for wl in sd14_erase50 sd14_erase100; do
  UCE_EDIT_RESIDENT=0 UCE_DEFINES="$d" timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print('H2[$d] resident=0', '$wl', p['ms_per_step'], p['ms_per_step_events'], [(k['kernel'],k['avg_ms']) for k in [p['roofline']]+p['roofline']['kernels']])"
  done
done
