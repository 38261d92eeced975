#!/bin/bash
# Same-box A/B (csrc/libgsr_hip_prev.so vs csrc/libgsr_hip.so) of the train step at several model sizes:
#   gpurun -- 'bash tools/ab_sizes.sh "20000 100000 1000000" [rounds]'
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/new.so; cp $D/libgsr_hip_prev.so /tmp/prev.so
for r in $(seq 1 ${2:-2}); do for n in ${1:-"20000 1000000"}; do for w in prev new; do cp /tmp/$w.so $D/libgsr_hip.so
python bench.py --gaussians $n --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', $n, round(d['value'],1), round(d['ms_per_step'],4))"; done; done; done
cp /tmp/new.so $D/libgsr_hip.so
