#!/bin/bash
# Same-box A/B of several builds of libgsr_hip.so kept under gpurun_libs/ (untracked; they travel with gpurun):
#   gpurun -- 'bash tools/ab_libs.sh "old A B" [rounds] [extra bench args]'
# prints images/s, ms per step and the per-stage device times (us) of every library, `rounds` times round-robin.
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/cur.so
for r in $(seq 1 ${2:-2}); do
  for w in $1; do
    cp gpurun_libs/lib_$w.so $D/libgsr_hip.so
    timeout 600 python bench.py --steps ${STEPS:-30} --warmup ${WARM:-5} --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', round(d['value'], 1), round(d['ms_per_step'], 4), {k: (round(v * 1000) if v is not None else None) for k, v in d.get('stage_ms', {}).items()})"
  done
done
cp /tmp/cur.so $D/libgsr_hip.so
