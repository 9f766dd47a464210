cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_match.py -m gpu -q -k "search_by_projection" 2>&1 | tail -4
