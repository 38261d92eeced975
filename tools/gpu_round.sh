#!/bin/bash
# One measured state: bench line, kernel trace, (optionally) PMC passes.  Usage on the GPU box:
#   tools/gpu_round.sh <tag> [pmc]     -> gpurun_out/<tag>/{bench.json, bench_under_rocprof.json, kernel_stats.{md,csv}, pmc.json}
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed" >> $OUT/failed.txt
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err ) || echo "rocprof failed" >> $OUT/failed.txt
DB=$(find $OUT/prof -name "*results.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats "state $TAG: python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras under rocprofv3 --kernel-trace --stats"
if [ "$2" = "pmc" ]; then
  tools/pmc_profile.sh $OUT/pmc "k_blend|k_preprocess|k_emit|k_onesweep|k_chunk" > $OUT/pmc.log 2>&1
  python tools/pmc_summary.py $OUT/pmc $OUT/pmc.json "state $TAG: rocprofv3 --pmc passes over python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras" >> $OUT/pmc.log 2>&1
fi
rm -rf $OUT/prof/*/*.db $OUT/pmc/pass*/*/*.db 2>/dev/null
ls $OUT
