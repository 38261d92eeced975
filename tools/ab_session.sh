#!/bin/bash
# Same-box A/B of two library builds (gpurun_libs/lib_<a>.so, lib_<b>.so) at the headline and at small-model sizes:
#   gpurun -- 'bash tools/ab_session.sh old final'
cd $GRAFT_REPO_ROOT
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/cur.so
run() {
  cp gpurun_libs/lib_$1.so $D/libgsr_hip.so
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-100} --warmup 10 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', 'images/s %.0f' % d['value'], 'ms %.4f' % d['ms_per_step'], 'median %.4f' % d['step_host_ms']['median'])"
}
for rep in 1 2 3; do
  for lib in "$@"; do STEPS=60 run $lib; done
  for n in 20000 50000 130000 300000; do for lib in "$@"; do run $lib --gaussians $n --sh-degree 0; done; done
done
cp /tmp/cur.so $D/libgsr_hip.so
