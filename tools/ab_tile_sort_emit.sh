#!/bin/bash
# tile-sort route behind the EMIT path (frames above 4 096 tiles, batched renders): on / off, same box
cd $GRAFT_REPO_ROOT
run() {
  GSR_OPTS=$1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 60 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$*', 'ms %.4f' % d['ms_per_step'], {k: round(1e3*v,1) for k,v in s.items() if v})"
}
for rep in 1 2; do
  run tile_sort=0 --width 1920 --height 1080
  run tile_sort=1 --width 1920 --height 1080
  run tile_sort=0 --width 1920 --height 1080 --gaussians 300000
  run tile_sort=1 --width 1920 --height 1080 --gaussians 300000
done
for rep in 1 2; do
  GSR_OPTS=tile_sort=0 python tools/prof_batched_step.py 2>&1 | grep "batched step\|k_tile_sort\|k_onesweep\|k_emit\|k_tile_counts\|k_radix" | cut -c1-200
  GSR_OPTS=tile_sort=1 python tools/prof_batched_step.py 2>&1 | grep "batched step\|k_tile_sort\|k_onesweep\|k_emit\|k_tile_counts\|k_radix" | cut -c1-200
done
