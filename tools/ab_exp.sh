#!/bin/bash
# Same-box A/B of library options: tools/ab_exp.sh "exp_bwd=0" "exp_bwd=1" ...   (each run twice, alternating)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for o in "$@"; do
  GSR_OPTS=$o timeout 180 python bench.py --no-extras --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$o', 'img/s %.0f' % d['value'], 'ms %.4f' % d['ms_per_step'], {k: round(1e3*v,1) for k,v in s.items() if k in ('blend_fwd','blend_bwd','preprocess_bwd')})"
done; done
