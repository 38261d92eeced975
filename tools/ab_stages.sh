#!/bin/bash
# Same-box A/B of all stage times (csrc/libgsr_hip_prev.so vs csrc/libgsr_hip.so):  gpurun -- 'bash tools/ab_stages.sh [N] [rounds]'
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/new.so; cp $D/libgsr_hip_prev.so /tmp/prev.so
for r in $(seq 1 ${2:-2}); do for w in prev new; do cp /tmp/$w.so $D/libgsr_hip.so
python bench.py --gaussians ${1:-1000000} --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],1), {k:round(v*1000,1) for k,v in d['stage_ms'].items()})"; done; done
cp /tmp/new.so $D/libgsr_hip.so
