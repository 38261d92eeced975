#!/bin/bash
# Same-box A/B: the slabbed direct-binning scatter on a frame that fits one wave's tables (980x545: 2 170 tiles) -- more, lighter waves.
#   gpurun -- 'bash tools/ab_slab_small.sh'
cd $GRAFT_REPO_ROOT
run() {
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$*', 'img/s %.0f' % d['value'], 'ms %.4f' % d['ms_per_step'], {k: round(1e3*v,1) for k,v in s.items() if v})"
}
for rep in 1 2; do
  run X=1
  run GSR_DB_FORCE_SLABS=1 GSR_OPTS=direct_slab_tiles=1116
  run GSR_DB_FORCE_SLABS=1 GSR_OPTS=direct_slab_tiles=744
  run GSR_DB_FORCE_SLABS=1 GSR_OPTS=direct_slab_tiles=558
done
