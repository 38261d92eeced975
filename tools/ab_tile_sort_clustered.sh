cd $GRAFT_REPO_ROOT
run() {
  GSR_OPTS=$1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$*', 'ms %.4f' % d['ms_per_step'], 'R', d['config'].get('num_rendered_R'), {k: round(1e3*v,1) for k,v in s.items() if v and k in ('sort_depth','sort_tile','emit','scan','blend_fwd')})"
}
for rep in 1 2; do
for n in 130000 300000; do
  run tile_sort=0 --gaussians $n --sh-degree 0 --clustered
  run tile_sort=1 --gaussians $n --sh-degree 0 --clustered
done; done
