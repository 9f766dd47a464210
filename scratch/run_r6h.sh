cd $GRAFT_REPO_ROOT
python scratch/time_ba_batch.py 8 30 2>&1 | grep -v amdgpu | tail -4
python scratch/time_ba_batch.py 8 30 2>&1 | grep -v amdgpu | tail -4
