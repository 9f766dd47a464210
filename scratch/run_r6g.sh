cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/tests.log
timeout 600 python bench.py --no-ba --no-bow --no-c3 --no-c5 --no-host-api --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" > $O/headline.txt
cat $O/tests.log $O/headline.txt
