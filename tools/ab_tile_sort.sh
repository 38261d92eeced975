#!/bin/bash
# Same-box A/B of the tile-sort route (no global depth sort; per-tile depth sort behind an index-order direct binning) against the
# depth sort + direct binning, at the headline and at stage A's sizes.      gpurun -- 'bash tools/ab_tile_sort.sh'
cd $GRAFT_REPO_ROOT
run() {
  GSR_OPTS=$1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 60 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$*', 'img/s %.0f' % d['value'], 'ms %.4f' % d['ms_per_step'], {k: round(1e3*v,1) for k,v in s.items() if v})"
}
for rep in 1 2 3; do
  run tile_sort=0
  run tile_sort=2
done
for rep in 1 2; do
  run tile_sort=0 --gaussians 130000 --sh-degree 0
  run tile_sort=1 --gaussians 130000 --sh-degree 0
  run tile_sort=0 --gaussians 20000 --sh-degree 0
  run tile_sort=1 --gaussians 20000 --sh-degree 0
done
run tile_sort=0 --gaussians 4000000 --steps 10
run tile_sort=2 --gaussians 4000000 --steps 10
run tile_sort=0 --clustered
run tile_sort=2 --clustered
