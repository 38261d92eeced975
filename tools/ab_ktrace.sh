#!/bin/bash
# per-kernel durations of two library builds, same box:  gpurun -- 'bash tools/ab_ktrace.sh "<regex>" libA libB'   (gpurun_libs/lib_<name>.so)
RE=$1; shift
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/cur.so
for w in "$@" "$@"; do cp gpurun_libs/lib_$w.so $D/libgsr_hip.so; echo "== $w"; bash tools/ktrace.sh "$RE" 2>&1 | tail -3; done
cp /tmp/cur.so $D/libgsr_hip.so
