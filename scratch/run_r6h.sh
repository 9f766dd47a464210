cd $GRAFT_REPO_ROOT
DCS_BA_TRACE=1 python scratch/time_ba_batch.py 1 12 2>&1 | grep -v amdgpu | tail -5
