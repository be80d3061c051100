#!/bin/bash
# rocprofv3 passes over tools/bench_more.py (kernel trace, then FETCH_SIZE and WRITE_SIZE on their own);
# summary to gpurun_out/more_prof/ (copy it to profiles/).
set -e
TAG=${1:-r03}
ROWS=${2:-100000000}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/more_prof
rm -rf $O && mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_more.py --rows $ROWS > $O/trace.json 2> $O/trace.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_more.py --rows $ROWS > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_more.py --rows $ROWS > $O/write.log 2>&1
cd $R
T=$(ls -t $O/trace/*/*kernel_trace.csv | head -1)
F=$(ls -t $O/fetch/*/*counter_collection.csv | head -1)
W=$(ls -t $O/write/*/*counter_collection.csv | head -1)
python3 profiles/summarize_more.py $T $F $W $O/trace.json $O/${TAG}_ops_pmc.json
cp $(ls -t $O/trace/*/*kernel_stats.csv | head -1) $O/${TAG}_more_kernel_stats.csv
