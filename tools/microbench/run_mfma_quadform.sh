#!/bin/bash
# build + run the matrix-pipe quadratic-form experiment: gpurun -- 'bash tools/microbench/run_mfma_quadform.sh'  (the binary is built HERE: hipcc is on the box)
set -e
D=$(cd "$(dirname "$0")" && pwd); O=${GRAFT_REPO_ROOT:-$D/../..}/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $D/mfma_quadform.hip -o /tmp/mfma_quadform
for b in 6 12; do /tmp/mfma_quadform 8960 $b; done | tee $O/mfma_quadform.txt
