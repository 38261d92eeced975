#!/bin/bash
# Build a variant of libgsr_hip.so into gpurun_libs/lib_<name>.so (untracked; travels with gpurun) for tools/ab_libs.sh:
#   tools/build_variant.sh <name> [-DFLAG=..] ...        (from the working tree; `git stash` / a worktree for an older source)
NAME=$1; shift
CS=${SRC:-3dgs_hierarchical_training_amd/csrc}
mkdir -p gpurun_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -Wno-unused-result -fno-slp-vectorize \
  -Wl,-soname,libgsr_hip.so $CS/gsr_kernels.hip $CS/loss_kernels.hip $CS/optim_kernels.hip $CS/knn_kernels.hip -o gpurun_libs/lib_$NAME.so 2>&1 | grep -i "error"
ls -la gpurun_libs/lib_$NAME.so
