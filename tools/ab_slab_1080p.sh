#!/bin/bash
# same-box A/B at 1 M Gaussians @1920x1080 of the binning routes: sort route against the slabbed direct route at several slab sizes
#   gpurun -- 'bash tools/ab_slab_1080p.sh [rounds]'
for r in $(seq 1 ${1:-2}); do for o in "direct_slab_tiles=0" "direct_slab_tiles=4096" "direct_slab_tiles=2760" "direct_slab_tiles=2176"; do
GSR_OPTS="$o" python bench.py --width 1920 --height 1080 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', round(d['ms_per_step'],4), {k:(round(v*1000,1) if v is not None else None) for k,v in d['stage_ms'].items() if k in ('scan','emit','sort_tile','ranges')})"
done; done
