#!/bin/bash
# run on the GPU box from the repo root: bench line, rocprofv3 kernel stats of the same command, PMC passes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --cpu-seconds 0 > $O/bench_prof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow > /dev/null 2>&1
cd $R
python scratch/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json 256 640 480 1000 1 | tail -5
python scratch/pmc_sum.py $O/pmc_sq > $O/pmc_sq_summary.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
tail -c 3000 $O/bench.json
