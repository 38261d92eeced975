run() { env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*',round(d['value'],1), {k:round(v*1000) for k,v in d.get('stage_ms',{}).items() if k in ('preprocess_fwd','blend_fwd','blend_bwd')})"; }
for r in 1 2 3; do
run A=0
run GSR_DBG_FWD=1024
done
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
