cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/tests.log; tail -c 6000 $O/bench.json
