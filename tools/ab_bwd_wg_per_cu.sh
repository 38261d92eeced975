#!/bin/bash
# persistent backward blend: workgroups per CU (GSR_BWD_WG_PER_CU; 0 = what the occupancy query returns) at stage A's size, same box:
#   gpurun -- 'bash tools/ab_bwd_wg_per_cu.sh'
for r in 1 2; do for w in 0 6 8 10 12 14; do
GSR_BWD_WG_PER_CU=$w python bench.py --gaussians 130000 --sh-degree 0 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --views 1 --no-densify-stats 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wg_per_cu=$w', round(d['ms_per_step'],4), {k:(round(v*1000,1) if v is not None else None) for k,v in d['stage_ms'].items()})"
done; done
