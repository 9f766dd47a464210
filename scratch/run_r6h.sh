cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
HEAD="--cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api --no-two-lanes --steps 6 --warmup 2"
for lib in lib scratch/ab/one_round; do
  L=$R/orb-slam2-dualcam_amd/lib/libdcs_hip.so; [ $lib != lib ] && L=$R/$lib/libdcs_hip.so
  rm -rf $R/gpurun_out/tr_x; echo "== alone, $lib"
  DCS_LIB_PATH=$L DCS_ORB_NO_OVERLAP=1 DCS_ORB_EMIT=15 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_x -o t -- python $R/bench.py $HEAD --serial > /dev/null 2>&1
  python $R/scratch/trace_by_grid.py $R/gpurun_out/tr_x k_fast_cells | head -9
  rm -rf $R/gpurun_out/tr_x
done
