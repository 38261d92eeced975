#!/bin/bash
# same-box A/B of the list cut (round 6) on the fused training step: gpurun -- 'bash tools/ab_list_cut.sh [rounds] [N] [steps]'
R=${1:-3}; N=${2:-1000000}; S=${3:-60}
for r in $(seq 1 $R); do for o in "list_cut=0" "list_cut=1"; do
GSR_OPTS="$o" python bench.py --gaussians $N --steps $S --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', 'N=$N', round(d['ms_per_step'],4), {k:(round(v*1000,1) if v is not None else None) for k,v in d['stage_ms'].items()})"
done; done
