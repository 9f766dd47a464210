cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_extract.py -q -m gpu -k "emitting" 2>&1 | tail -4 > $O/emit_test.log
( timeout 700 python scratch/stress_track.py 420 601 > $O/track.log 2>&1 ) &
( DCS_POSE_EXACT_EDGE=1 timeout 700 python scratch/stress_track.py 420 611 > $O/track_exact.log 2>&1 ) &
( timeout 700 python scratch/stress_track_dev.py 420 602 > $O/track_dev.log 2>&1 ) &
( timeout 700 python scratch/stress_parity2.py 420 603 > $O/families.log 2>&1 ) &
( DCS_POSE_EXACT_EDGE=1 timeout 700 python scratch/stress_parity2.py 420 613 > $O/families_exact.log 2>&1 ) &
( timeout 700 python scratch/stress_parity.py 420 > $O/extract.log 2>&1 ) &
( timeout 700 python scratch/stress_parity3.py 420 604 > $O/windows.log 2>&1 ) &
wait
cat $O/emit_test.log; for f in track track_exact track_dev families families_exact extract windows; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -4; done
