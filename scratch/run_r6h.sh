cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1; done
for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_track.py tests/test_gpu_match.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1; done
