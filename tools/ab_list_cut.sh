#!/bin/bash
# same-box A/B of the list cut (round 6) on the fused training step: gpurun -- 'bash tools/ab_list_cut.sh [rounds] [N] [steps] ["opts A" "opts B"]   (list_cut=2: whatever the size)'
R=${1:-3}; N=${2:-1000000}; S=${3:-60}; A=${4:-list_cut=0}; B=${5:-list_cut=2}
for r in $(seq 1 $R); do for o in "$A" "$B"; do
GSR_OPTS="$o" python bench.py --gaussians $N --steps $S --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | TAG="$o N=$N" python -c '
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
lc=d.get("list_cut") or {}
print(os.environ["TAG"], round(d["ms_per_step"],4), {k:(round(v,1) if isinstance(v,float) else v) for k,v in lc.items()}, {k:(round(v*1000,1) if v is not None else None) for k,v in d["stage_ms"].items()})'
done; done
