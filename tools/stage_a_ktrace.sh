#!/bin/bash
# Per-kernel totals of batched stage A (B models per launch chain, 130 k Gaussians each @980x545, SH 0): image iterations and pose
# iterations traced apart.      gpurun -- 'bash tools/stage_a_ktrace.sh [B]'
B=${1:-8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/stage_a_ktrace; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/sa_run.py <<PY
import importlib, sys, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
sequence = importlib.import_module("3dgs_hierarchical_training_amd.sequence")
stage_a = importlib.import_module("3dgs_hierarchical_training_amd.stage_a")
host = importlib.import_module("3dgs_hierarchical_training_amd.host"); host.cap_host_threads()
dev = torch.device("cuda:0")
seq = sequence.FrameSequence(12, 400000, 980, 545, dev, seed=0)
for f in range(12):
    seq.target(f); seq.depth(f)
B = $B
img, pose = (int(sys.argv[1]), int(sys.argv[2]))
stage_a.fit_pairs_batched(seq, list(range(B)), dev, n_points=130000, single_image_iters=img, pose_iters=pose)
torch.cuda.synchronize()
PY
for leg in "100 0" "100 100"; do
  set -- $leg
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$1_$2 -o r -- python /tmp/sa_run.py $1 $2 > $OUT/run_$1_$2.txt 2>&1 )
  DB=$(find $OUT/prof_$1_$2 -name "*results.db" | head -1)
  python - <<PY
import sqlite3, collections
cur = sqlite3.connect("$DB").cursor()
rows = cur.execute("select name,start,end from kernels").fetchall()
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[n[:100]].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("== B=$B, image iterations $1, pose iterations $2: total kernel time %.1f ms over %d launches" % (tot / 1e3, len(rows)))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{len(v):6d} x avg {sum(v)/len(v):8.1f} us = {sum(v)/1e3:8.2f} ms ({100*sum(v)/tot:4.1f} %)  {n}")
PY
  rm -rf $OUT/prof_$1_$2
done
