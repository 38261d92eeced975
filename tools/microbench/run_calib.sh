#!/bin/bash
# Build (hipcc, gfx950) and run the VALU issue-rate / copy-ceiling calibration; summary -> $1 (default gpurun_out/valu_calib.txt)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${1:-gpurun_out/valu_calib.txt}
mkdir -p "$(dirname "$OUT")"
[ -x "$HERE/valu_calib" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$HERE/valu_calib.hip" -o "$HERE/valu_calib"
"$HERE/valu_calib" ${2:-5000} | tee "$OUT"
