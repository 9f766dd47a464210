cd $GRAFT_REPO_ROOT
for la in 2 3 4 2 3 4; do echo "LOOKAHEAD=$la"; DCS_BA_LOOKAHEAD=$la python scratch/time_ba_batch.py 8 30 2>/dev/null | grep "B=1\|B=8"; done
