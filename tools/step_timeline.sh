#!/bin/bash
# Kernel timeline (start, duration, gap to the previous kernel) of ONE steady-state step of the bench's timed loop.
#   gpurun -- 'bash tools/step_timeline.sh [step index from the end, default -6]'
OUT=$GRAFT_REPO_ROOT/gpurun_out/step_timeline; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras $TL_ARGS > $OUT/bench.json 2> $OUT/err.txt )
DB=$(find $OUT/prof -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $DB ${1:--6}
rm -rf $OUT/prof
