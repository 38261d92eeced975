#!/bin/bash
# Same-box A/B of two builds of libgsr_hip.so (box-to-box variance on the pool is ~4 %, larger than most single changes).
# Build the old revision first and keep its library as csrc/libgsr_hip_prev.so (untracked, travels with gpurun), build the
# new one in place, then:  gpurun -- 'bash tools/ab_bench.sh [rounds]'
set -e
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/new.so; cp $D/libgsr_hip_prev.so /tmp/prev.so
for r in $(seq 1 ${1:-3}); do
  for w in prev new; do
    cp /tmp/$w.so $D/libgsr_hip.so
    python bench.py --steps ${STEPS:-30} --warmup ${WARM:-5} --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', round(d['value'], 1), round(d['ms_per_step'], 4), round(d.get('fwd_bwd_ms', 0), 3), {k: round(v * 1000) for k, v in d.get('stage_ms', {}).items()})"
  done
done
cp /tmp/new.so $D/libgsr_hip.so
