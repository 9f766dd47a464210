#!/bin/bash
# run on the GPU box from the repo root: bench line, rocprofv3 kernel stats of the same command, solo stats, BA-only stats, PMC passes.
# Results land in gpurun_out/<round>/; copy what is to be judged into profiles/ (tracked).
set -x
RND=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RND
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
# the same command under the kernel tracer (CPU legs off: they only add wall time)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --cpu-seconds 0 > $O/bench_prof.json 2>/dev/null
# the timed region alone (what roofline.achieved is checked against)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python $R/bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api > $O/bench_headline.json 2>/dev/null
# every kernel alone (no stream overlap inside the extractor, matcher not underneath): the durations DESIGN.md quotes as "solo"
DCS_ORB_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -- python $R/bench.py --serial --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api > /dev/null 2>&1
# BA only: one C4 problem, then the batch of 8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba1 -- python $R/scratch/time_ba_batch.py 1 20 > $O/ba1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba8 -- python $R/scratch/time_ba_batch.py 8 20 > $O/ba8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api > /dev/null 2>&1
cd $R
python scratch/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json 256 640 480 1000 1 | tail -5
python scratch/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt
for d in stats headline solo ba1 ba8; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; done
rm -rf $O/stats $O/headline $O/solo $O/ba1 $O/ba8 $O/pmc_fetch $O/pmc_write $O/pmc_sq
tail -c 1500 $O/bench.json; cat $O/ba8.log | tail -4
