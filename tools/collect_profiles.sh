#!/bin/bash
# Round-N evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r03
# bench line, rocprofv3 kernel stats of the same command and of the timed region alone, every kernel ALONE (--serial + DCS_ORB_NO_OVERLAP=1),
# BA-only traces, and the FETCH_SIZE / WRITE_SIZE / SQ PMC passes (separate runs, --kernel-trace only) -> gpurun_out/<tag>/;
# the files DESIGN.md cites are copied from there into profiles/ (tracked).
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
HEAD="--cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api"
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --cpu-seconds 0 > $O/bench_prof.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python $R/bench.py $HEAD > $O/bench_headline.json 2>/dev/null
DCS_ORB_FUSED_BLUR=0 timeout 400 python $R/bench.py $HEAD > $O/bench_headline_separate_blur.json 2>/dev/null
DCS_ORB_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -- python $R/bench.py $HEAD --serial > /dev/null 2>&1
DCS_ORB_FUSED_BLUR=0 DCS_ORB_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo_separate_blur -- python $R/bench.py $HEAD --serial > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba1 -- python $R/scratch/time_ba_batch.py 1 20 > $O/ba1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba8 -- python $R/scratch/time_ba_batch.py 8 20 > $O/ba8.log 2>&1
# round 5: the per-frame tracking chain alone (dcs_track_local_map at batch 1 and 16: k_pose_opt2 + the five chain kernels), the C3 shape (configs[2] at one GPU)
# as a headline-only run, the emitting FAST launch by launch (in the pipeline and alone, emitting FAST and resize chain), k_pose_opt2's in-kernel timeline
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/track -- python $R/scratch/time_track.py > $O/track.log 2>&1
DCS_POSE_FAST=0 timeout 300 python $R/scratch/time_track.py >> $O/track.log 2>&1
DCS_POSE_EXACT_EDGE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/track_exact -- python $R/scratch/time_track.py >> $O/track.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python $R/bench.py $HEAD --no-two-lanes --width 1280 --height 720 --nfeatures 2000 --pairs 64 > $O/bench_c3_shape.json 2>/dev/null
( cd $R && bash scratch/emit_trace.sh ) > $O/fast_emit_by_level.txt 2>&1
# round 6: k_pose_opt2's in-kernel timeline (side build -DDCS_POSE_PROF=20, tools/pose_timeline.py), the windows beyond 42 free poses (scratch/time_ba_large.py) and
# a PMC pass of the tracking chain alone (k_pose_opt2's instruction counts)
if [ -f $R/scratch/ab/pose_prof/libdcs_hip.so ]; then
  ( export DCS_LIB_PATH=$R/scratch/ab/pose_prof/libdcs_hip.so; timeout 120 python $R/tools/pose_timeline.py 2000 2000; timeout 120 python $R/tools/pose_timeline.py 4000 4000
    echo "---- DCS_POSE_EXACT_EDGE=1"; DCS_POSE_EXACT_EDGE=1 timeout 120 python $R/tools/pose_timeline.py 2000 2000; DCS_POSE_EXACT_EDGE=1 timeout 120 python $R/tools/pose_timeline.py 4000 4000 ) 2>/dev/null | grep -v amdgpu > $O/pose_timeline.txt
fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba_large -- python $R/scratch/time_ba_large.py 3 > $O/ba_large.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_track -- python $R/scratch/time_track.py 2000 2000 10 > /dev/null 2>&1
python $R/scratch/pmc_kernel_sum.py $O/pmc_track k_pose_opt2 > $O/pmc_pose_opt2.txt 2>&1
PM="--steps 5 --warmup 1 $HEAD"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc_mfma -- python $R/bench.py $PM > /dev/null 2>&1
cd $R
python scratch/pmc_to_json.py $O/pmc_counters.json 256 640 480 1000 1 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma | tail -40
python scratch/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt
for d in stats headline solo solo_separate_blur ba1 ba8 track track_exact c3 ba_large; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; done
rm -rf $O/stats $O/headline $O/solo $O/solo_separate_blur $O/ba1 $O/ba8 $O/track $O/track_exact $O/c3 $O/ba_large $O/pmc_track $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma
echo "== headline"; python scratch/kstats.py $O/headline_kernel_stats.csv 16
echo "== solo"; python scratch/kstats.py $O/solo_kernel_stats.csv 16
echo "== ba"; python scratch/kstats.py $O/ba1_kernel_stats.csv 12
echo "== track"; python scratch/kstats.py $O/track_kernel_stats.csv 10; cat $O/track.log | grep "ms per frame"; cat $O/pose_timeline.txt 2>/dev/null
tail -c 1500 $O/bench.json; tail -4 $O/ba8.log
