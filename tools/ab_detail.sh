D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/cur.so
for r in 1 2 3 4 5 6; do for w in old new; do
cp gpurun_libs/lib_$w.so $D/libgsr_hip.so
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', round(d['value'], 1), round(d['ms_per_step'], 4), d['step_host_ms'], 'ovf', d['spec_overflows'], 'exact', d['exact_forwards'], 'spec', d['speculative_forwards'], 'k7', d['roofline'].get('kernel_us'), {k: round(v*1000) for k, v in d['stage_ms'].items() if v})"
done; done
cp /tmp/cur.so $D/libgsr_hip.so
