#!/bin/bash
# Same-box A/B of one library under two option sets:  gpurun -- 'bash tools/ab_opts.sh "direct_binning=0" "direct_binning=1" [N] [rounds] [extra bench args]'
A=$1; B=$2; N=${3:-1000000}; R=${4:-2}; shift 4
for r in $(seq 1 $R); do for o in "$A" "$B"; do
GSR_OPTS="$o" python bench.py --gaussians $N --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v*1000,1) if v is not None else None) for k,v in d['stage_ms'].items()})"; done; done
