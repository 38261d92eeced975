#!/bin/bash
# kernel timeline (start, duration, gap to the previous kernel) of ONE steady-state step of the autopatched trainer sequence at stage A's
# size:  gpurun -- 'bash tools/autopatch_timeline.sh [args of tools/autopatch_host_profile.py]'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/aptl; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && cd $ROOT && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o r -- python tools/autopatch_host_profile.py --steps 40 --top 1 "$@" > $OUT/run.txt 2> $OUT/err.txt )
grep "wall per step" $OUT/run.txt
DB=$(find $OUT/prof -name "*results.db" | head -1)
python $ROOT/tools/step_timeline.py $DB -60 | tee $OUT/timeline.txt
rm -rf $OUT/prof
