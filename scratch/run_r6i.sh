cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/collect.log 2>&1
( python scratch/pose_flip_stats.py 400 1 2>/dev/null | tail -1
  DCS_POSE_EXACT_EDGE=1 python scratch/pose_flip_stats.py 400 1 2>/dev/null | tail -1
  DCS_POSE_FAST=0 python scratch/pose_flip_stats.py 400 1 2>/dev/null | tail -1 ) > gpurun_out/r06/pose_flip_stats.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2 > gpurun_out/r06/pytest_gpu.txt
tail -30 gpurun_out/collect.log | cut -c1-400; cat gpurun_out/r06/pose_flip_stats.txt gpurun_out/r06/pytest_gpu.txt
