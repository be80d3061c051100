#!/bin/bash
# rocprofv3 passes over tools/bench_ops.py: kernel trace, then FETCH_SIZE and WRITE_SIZE on their own.
set -e
ROWS=${1:-200000000}
TAG=${2:-r03}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/ops_prof
rm -rf $O && mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_ops.py --rows $ROWS --reps 2 --no-profile > $O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_ops.py --rows $ROWS --reps 1 --no-profile > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_ops.py --rows $ROWS --reps 1 --no-profile > $O/write.log 2>&1
cd $R
S=$(ls -t $O/trace/*/*kernel_stats.csv | head -1)
F=$(ls -t $O/fetch/*/*counter_collection.csv | head -1)
W=$(ls -t $O/write/*/*counter_collection.csv | head -1)
T=$(ls -t $O/trace/*/*kernel_trace.csv | head -1)
python3 profiles/summarize_ops.py $T $F $W $ROWS $O/${TAG}_secondary_ops_pmc.json
cp $S $O/${TAG}_secondary_kernel_stats.csv
timeout -k 10 300 python3 tools/bench_ops.py --rows 500000000 > $O/${TAG}_ops.jsonl 2> $O/ops.err
tail -3 $O/${TAG}_ops.jsonl
