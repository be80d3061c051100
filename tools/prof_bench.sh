#!/bin/bash
# rocprofv3 passes over bench.py at the headline workload: kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in passes of their own; the summaries go to gpurun_out/bench_prof/ (copy them to profiles/).
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/bench_prof
rm -rf $O && mkdir -p $O
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-shuffled > $O/trace.log 2>&1
# the same with the join's two sides on one stream: durations of kernels that do not overlap add up
export TEMPO_JOIN_STREAMS=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-shuffled > $O/trace1.log 2>&1
unset TEMPO_JOIN_STREAMS
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-profile --no-shuffled > $O/fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-profile --no-shuffled > $O/write.log 2>&1
cd $R
S=$(ls -t $O/trace/*/*kernel_stats.csv | head -1)
F=$(ls -t $O/fetch/*/*counter_collection.csv | head -1)
W=$(ls -t $O/write/*/*counter_collection.csv | head -1)
python3 profiles/summarize_pmc.py $S $F $W $O/${TAG}_bench1b_pmc_traffic.json > $O/summary.txt
cp $S $O/${TAG}_bench1b_kernel_stats.csv
cp $(ls -t $O/trace1/*/*kernel_stats.csv | head -1) $O/${TAG}_bench1b_kernel_stats_one_stream.csv
grep "^{\"metric\"" $O/trace1.log | tail -1 > $O/${TAG}_bench1b_traced_one_stream.json
grep "^{\"metric\"" $O/trace.log | tail -1 > $O/${TAG}_bench1b_traced.json
head -30 $O/summary.txt
