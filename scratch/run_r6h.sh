cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_variants.py -m gpu -q 2>&1 | tail -3
DCS_BA_TRACE=1 python scratch/time_ba_batch.py 1 12 2>&1 | grep -v amdgpu | tail -5
python scratch/time_ba_batch.py 8 30 2>&1 | grep -v amdgpu | tail -4
