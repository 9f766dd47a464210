cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_track.py -m gpu -q -k "three_stages" 2>&1 | tail -4
