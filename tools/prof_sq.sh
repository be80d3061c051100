#!/bin/bash
# One rocprofv3 --pmc pass of eight SQ counters over one step of bench.py; summary (per kernel:
# instructions per kind, LDS bank-conflict cycles, share of vector issue slots filled) to
# gpurun_out/sq_prof/ (copy it to profiles/).  Needs the kernel stats of tools/prof_bench.sh.
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/sq_prof
rm -rf $O && mkdir -p $O
cd /tmp
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-profile --no-shuffled > $O/sq.log 2>&1
cd $R
C=$(ls -t $O/sq/*/*counter_collection.csv | head -1)
python3 profiles/summarize_sq.py $C profiles/${TAG}_bench1b_kernel_stats.csv $O/${TAG}_bench1b_pmc_sq.json > $O/summary.txt
head -40 $O/summary.txt
