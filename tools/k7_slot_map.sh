#!/bin/bash
# Run ON THE GPU BOX: rebuilds libgsr_hip.so there with -DGSR_K6_TIMING and runs tools/k7_slot_map.py (see k6_wave_timing.sh).
cd $GRAFT_REPO_ROOT
CS=3dgs_hierarchical_training_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSR_K6_TIMING -Wno-unused-result -fno-slp-vectorize \
  -Wl,-soname,libgsr_hip.so $CS/gsr_kernels.hip $CS/loss_kernels.hip $CS/optim_kernels.hip $CS/knn_kernels.hip -o $CS/libgsr_hip.so 2>&1 | grep -v warning
mkdir -p gpurun_out
python tools/k7_slot_map.py
python tools/k6_wave_timing.py > gpurun_out/r04_wave_timeline.txt 2>&1; tail -25 gpurun_out/r04_wave_timeline.txt
