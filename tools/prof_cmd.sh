#!/bin/bash
# rocprofv3 counter passes over any command of this repo's tools, one pass per counter group (SQ
# counters, FETCH_SIZE, WRITE_SIZE), and a per-kernel table of the sums:
#   tools/prof_cmd.sh <out-name> python3 tools/bench_join.py --order sorted --reps 1 --no-profile
# Results: gpurun_out/<out-name>/{sq,fetch,write}/... and gpurun_out/<out-name>/summary.txt
set -e
NAME=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$NAME
rm -rf $O && mkdir -p $O
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sq -- "$@" > $O/sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- "$@" > $O/sq2.log 2>&1 || true
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- "$@" > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- "$@" > $O/write.log 2>&1
cd $R
python3 - $O <<'PY' > $O/summary.txt
import collections, csv, glob, re, sys
o = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in ("sq", "sq2", "fetch", "write"):
    for path in glob.glob(f"{o}/{d}/*/*counter_collection.csv"):
        seen = set()
        with open(path) as f:
            for r in csv.DictReader(f):
                m = re.search(r"tmp::(\w+)", r["Kernel_Name"])
                if not m:
                    continue
                k = m.group(1)
                tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if d == "sq" and r["Counter_Name"] == "SQ_WAVES":
                    calls[k] += 1
for k, c in sorted(tab.items()):
    print(k, "launches", calls[k])
    for name, v in sorted(c.items()):
        print("   %-24s %.4g" % (name, v))
    if "FETCH_SIZE" in c:
        print("   hbm_bytes (2*FETCH+WRITE)*1024 = %.4g" % ((2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024))
PY
cat $O/summary.txt
