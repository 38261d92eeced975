for o in "early_r=1" "early_r=0"; do for extra in "" "--no-prepare-next"; do
GSR_OPTS="$o" python bench.py --gaussians 130000 --sh-degree 0 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --views 1 --no-densify-stats $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o','$extra', round(d['ms_per_step'],4), {k:(round(v*1000,1) if v is not None else None) for k,v in d['stage_ms'].items()}, d['step_host_ms'])"
done; done
