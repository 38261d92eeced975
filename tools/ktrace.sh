#!/bin/bash
# per-kernel average durations of a short bench run:  gpurun -- 'bash tools/ktrace.sh <regex> [bench args]'
RE=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/err.txt )
DB=$(find $OUT/prof -name "*results.db" | head -1)
python - <<PY
import sqlite3, re, collections
cur=sqlite3.connect("$DB").cursor()
rows=cur.execute("select name,start,end from kernels").fetchall()
agg=collections.defaultdict(list)
for n,s,e in rows:
    if re.search(r"$RE", n): agg[n[:90]].append((e-s)/1e3)
for n,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print(f"{len(v):5d} avg {sum(v)/len(v):8.1f} min {min(v):8.1f} max {max(v):8.1f}  {n}")
PY
rm -rf $OUT/prof
