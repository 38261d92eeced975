#!/bin/bash
# same-box A/B at stage A's size (130 k Gaussians, SH degree 0) of option sets, on the route WITHOUT the hand-over (a forward that runs its own preprocess):
#   gpurun -- 'bash tools/ab_130k.sh "small_sort9=0" "small_sort9=1" [rounds] [N]'
A=$1; B=$2; R=${3:-2}; N=${4:-130000}
for r in $(seq 1 $R); do for o in "$A" "$B"; do
GSR_OPTS="$o" python bench.py --gaussians $N --sh-degree 0 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --views 1 --no-densify-stats --no-prepare-next 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', round(d['ms_per_step'],4), {k:(round(v*1000,1) if v is not None else None) for k,v in d['stage_ms'].items()})"
done; done
