# same-box A/B of csrc/libgsr_hip_prev.so (old) vs csrc/libgsr_hip.so (new): headline + clustered, blend stages
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/new.so; cp $D/libgsr_hip_prev.so /tmp/prev.so
for r in 1 2; do for w in prev new; do cp /tmp/$w.so $D/libgsr_hip.so
for extra in "" "--clustered"; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w','$extra', round(d['value'],1), {k:round(v*1000,1) for k,v in d['stage_ms'].items() if 'blend' in k or 'pre' in k})"; done; done; done
cp /tmp/new.so $D/libgsr_hip.so
