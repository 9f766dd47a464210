cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | tail -2
