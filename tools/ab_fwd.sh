for v in 6 7 6 7; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --fwd-ppt $v 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppt',$v, round(d['value'],1), {k:round(v*1000,1) for k,v in d['stage_ms'].items()})"; done
for v in 6 7; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --clustered --fwd-ppt $v 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('clustered ppt',$v, round(d['value'],1), {k:round(v*1000,1) for k,v in d['stage_ms'].items()})"; done
