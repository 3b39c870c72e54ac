cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_edit_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in 1 0; do
  for wl in sd14_erase50 sd14_erase2p3 sd14_erase100; do
  UCE_EDIT_RESIDENT=$v timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print('RES', $v, '$wl', p['ms_per_step'], p['ms_per_step_events'])"
  done
done
timeout 300 python bench.py --only edit --workload sdxl_debias36x2 --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print('SDXL', p['ms_per_step'], p['ms_per_step_events'])"
UCE_CHAIN_DEBUG=1 python tools/dbg_resident.py 2>&1 | grep -v None.*None.*None.*None.*None | sed -n 2,9p
