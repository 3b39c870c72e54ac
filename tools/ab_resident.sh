cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_edit_gpu.py -x -q -m gpu -k "resident or erase_golden or every_segment" 2>&1 | tail -2
for wl in sd14_erase50 sd14_erase2p3 sd14_erase100; do
  timeout 300 python bench.py --only edit --workload $wl --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print('RES', '$wl', p['ms_per_step'], p['ms_per_step_events'])"
done
