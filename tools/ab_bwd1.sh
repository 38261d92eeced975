#!/bin/bash
# Backward blend with one pixel per lane and four waves per tile (blend_bwd_ppt=1, k_blend_bwd1) against the packed two-pixel kernel, same box
cd $GRAFT_REPO_ROOT
run() {
  GSR_OPTS=$1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-100} --warmup 10 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$*', 'ms %.4f' % d['ms_per_step'], {k: round(1e3*v,1) for k,v in s.items() if v and k in ('blend_fwd','blend_bwd','preprocess_bwd')})"
}
for rep in 1 2; do
  for n in 20000 130000 300000; do
    run blend_bwd_ppt=2 --gaussians $n --sh-degree 0
    run blend_bwd_ppt=1 --gaussians $n --sh-degree 0
  done
  STEPS=40 run blend_bwd_ppt=2
  STEPS=40 run blend_bwd_ppt=1
done
for o in blend_bwd_ppt=2 blend_bwd_ppt=1 blend_bwd_ppt=2 blend_bwd_ppt=1; do
  GSR_OPTS=$o python tools/prof_batched_step.py 2>&1 | grep "batched step\|k_blend_bwd" | cut -c1-60,150-215
done
