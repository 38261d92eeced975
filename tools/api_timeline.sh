#!/bin/bash
# HIP API calls and kernels of one training step on one time axis (what sits between two kernels when the GPU idles):
#   gpurun -- 'bash tools/api_timeline.sh [bench args]'
OUT=$GRAFT_REPO_ROOT/gpurun_out/api_tl; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/err.txt )
tail -3 $OUT/err.txt
find $OUT/prof -name "*.csv" | head
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
kf = glob.glob("gpurun_out/api_tl/prof/**/*kernel_trace.csv", recursive=True)[0]
hf = glob.glob("gpurun_out/api_tl/prof/**/*hip_api_trace.csv", recursive=True)[0]
ks = list(csv.DictReader(open(kf))); hs = list(csv.DictReader(open(hf)))
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ks) if "k_preprocess_bwd" in r["Kernel_Name"]]
import os
K = int(os.environ.get("STEP", "8"))
a, b = idx[K] + 1, idx[K + 1] + 1
t0 = int(ks[a]["Start_Timestamp"])
ev = []
for r in ks[a:b]:
    ev.append((int(r["Start_Timestamp"]), "K   " + r["Kernel_Name"][:60] + " dur %.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)))
    ev.append((int(r["End_Timestamp"]), "  end " + r["Kernel_Name"][:40]))
h0, h1 = t0 - 600000, int(ks[b - 1]["End_Timestamp"])
for r in hs:
    s = int(r["Start_Timestamp"])
    if h0 <= s <= h1 and not r["Function"].startswith("hipGetLastError") and "hipGetDevice" not in r["Function"]:
        ev.append((s, "        api " + r["Function"] + " %.1fus" % ((int(r["End_Timestamp"]) - s) / 1e3)))
ev.sort()
for t, s in ev:
    print("%9.1f %s" % ((t - t0) / 1e3, s))
PY
rm -rf $OUT/prof
