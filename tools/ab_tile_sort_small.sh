#!/bin/bash
# tile-sort route on / off at small model sizes (SH degree 0, 980x545), three alternating repetitions of 200 steps
cd $GRAFT_REPO_ROOT
run() {
  GSR_OPTS=$1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$*', 'ms %.4f' % d['ms_per_step'], 'median %.4f' % d['step_host_ms']['median'], {k: round(1e3*v,1) for k,v in s.items() if v and k in ('sort_depth','sort_tile','emit','scan')})"
}
for n in ${SIZES:-20000 50000 130000 300000}; do for rep in 1 2 3; do
  run tile_sort=0 --gaussians $n --sh-degree 0
  run tile_sort=1 --gaussians $n --sh-degree 0
done; done
