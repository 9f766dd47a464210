#!/bin/bash
# Round-N evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r02
# bench line, rocprofv3 kernel stats of the same command, every kernel ALONE (--serial + DCS_ORB_NO_OVERLAP=1),
# a BA-only trace, and the FETCH_SIZE / WRITE_SIZE / SQ PMC passes (separate runs, --kernel-trace only).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LIGHT="--cpu-seconds 0 --no-c3 --no-c5 --no-bow --no-host-api"
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $LIGHT > $O/bench_prof.json 2>/dev/null
DCS_ORB_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -- python $R/bench.py $LIGHT --no-ba --serial > $O/bench_solo.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba -- python $R/scratch/time_ba_batch.py 1 30 > $O/ba_only.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba8 -- python $R/scratch/time_ba_batch.py 8 20 > $O/ba8_only.log 2>&1
PM="--steps 5 --warmup 1 $LIGHT --no-ba"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py $PM > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python $R/bench.py $PM > /dev/null 2>&1
cd $R
python scratch/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json 256 640 480 1000 1 | tail -3
python scratch/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt
for d in stats solo ba ba8; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; done
rm -rf $O/stats $O/solo $O/ba $O/ba8 $O/pmc_fetch $O/pmc_write $O/pmc_sq
echo "== overlapped"; python scratch/kstats.py $O/stats_kernel_stats.csv 16
echo "== solo"; python scratch/kstats.py $O/solo_kernel_stats.csv 16
echo "== ba"; python scratch/kstats.py $O/ba_kernel_stats.csv 12
tail -c 1500 $O/bench.json
