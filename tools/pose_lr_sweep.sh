for lr in 1e-4 5e-5 2e-5 5e-6; do echo "pose_lr $lr"; python 3dgs_hierarchical_training_amd/run_segments.py --local --stage-a 130000 1000 300 --fit-pose --pose-lr $lr 2>&1 | grep -E "done" | cut -c1-200; done
python 3dgs_hierarchical_training_amd/run_segments.py --local --stage-a 130000 1000 300 2>&1 | grep -E "done" | cut -c1-200
