cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/lat -o lat -- python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 2 --warmup 1 > gpurun_out/lat.log 2>&1
python scratch/pp_trace.py
