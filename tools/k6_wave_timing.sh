#!/bin/bash
# Run ON THE GPU BOX (a scratch copy of the tree): rebuilds libgsr_hip.so there with -DGSR_K6_TIMING and runs
# tools/k6_wave_timing.py.  Do not run in the working tree -- it replaces the library (python -m ...build restores it).
cd $GRAFT_REPO_ROOT
CS=3dgs_hierarchical_training_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSR_K6_TIMING -Wno-unused-result -fno-slp-vectorize \
  -Wl,-soname,libgsr_hip.so $CS/gsr_kernels.hip $CS/loss_kernels.hip $CS/optim_kernels.hip $CS/knn_kernels.hip -o $CS/libgsr_hip.so 2>&1 | grep -v warning
python tools/k6_wave_timing.py
