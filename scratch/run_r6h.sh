cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_variants.py -m gpu -q 2>&1 | tail -3
python scratch/time_ba_batch.py 8 40 2>&1 | grep -v amdgpu | tail -4
