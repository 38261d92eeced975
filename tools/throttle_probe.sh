st() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for mode in 256 cap 256 cap; do
  a=$(st)
  if [ $mode = 256 ]; then export OMP_NUM_THREADS=256; else unset OMP_NUM_THREADS; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', 'headline', round(d['value'],1), 'max step', round(d['step_host_ms']['max'],2), ' '.join('%s %.3f' % (k.split(' @')[0].split(',')[0][:12], v['ms_per_step']) for k,v in d['other_workloads'].items()))"
  b=$(st)
  echo "   cpu.stat before: $a"; echo "   cpu.stat after:  $b"
done
