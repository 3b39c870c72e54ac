cd $GRAFT_REPO_ROOT
python tools/cmp_resident.py 2>&1 | grep relF
timeout 900 python -m pytest tests/test_edit_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in 1 0; do
  for wl in sd14_erase50 sd14_erase2p3 sd14_erase100; do
  UCE_EDIT_RESIDENT=$v timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print('RES', $v, '$wl', p['ms_per_step'], p['ms_per_step_events'], [(k['kernel'],k['avg_ms']) for k in [p['roofline']]+p['roofline']['kernels']])"
  done
done
