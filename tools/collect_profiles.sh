#!/bin/bash
# Round-N evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r03
# bench line, rocprofv3 kernel stats of the same command and of the timed region alone, every kernel ALONE (--serial + DCS_ORB_NO_OVERLAP=1),
# BA-only traces, and the FETCH_SIZE / WRITE_SIZE / SQ PMC passes (separate runs, --kernel-trace only) -> gpurun_out/<tag>/;
# the files DESIGN.md cites are copied from there into profiles/ (tracked).
TAG=${1:-r04}
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
PM="--steps 5 --warmup 1 $HEAD"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc_mfma -- python $R/bench.py $PM > /dev/null 2>&1
cd $R
python scratch/pmc_to_json.py $O/pmc_counters.json 256 640 480 1000 1 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma | tail -40
python scratch/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt
for d in stats headline solo solo_separate_blur ba1 ba8; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; done
rm -rf $O/stats $O/headline $O/solo $O/solo_separate_blur $O/ba1 $O/ba8 $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma
echo "== headline"; python scratch/kstats.py $O/headline_kernel_stats.csv 16
echo "== solo"; python scratch/kstats.py $O/solo_kernel_stats.csv 16
echo "== ba"; python scratch/kstats.py $O/ba1_kernel_stats.csv 12
tail -c 1500 $O/bench.json; tail -4 $O/ba8.log
