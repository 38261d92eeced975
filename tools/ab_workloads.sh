#!/bin/bash
# Same-box A/B (csrc/libgsr_hip_prev.so vs csrc/libgsr_hip.so) of bench.py's other_workloads + headline:  gpurun -- 'bash tools/ab_workloads.sh [rounds]'
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/new.so; cp $D/libgsr_hip_prev.so /tmp/prev.so
for r in $(seq 1 ${1:-2}); do for w in prev new; do cp /tmp/$w.so $D/libgsr_hip.so
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'headline', round(d['value'],1), ' '.join('%s %.4f' % (k.split(' @')[0].split(',')[0], v["ms_per_step"]) for k,v in d["other_workloads"].items() if "ms_per_step" in v), 'dropin', round(d['dropin']['value'],1))"; done; done
cp /tmp/new.so $D/libgsr_hip.so
