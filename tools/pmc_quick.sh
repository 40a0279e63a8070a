#!/bin/bash
# PMC passes over tools/prof_transform.py; prints mean counter per kernel.  Usage: tools/pmc_quick.sh <tag> <args of prof_transform.py>
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/prof_transform.py "$@" > $OUT/f.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/w -- python3 tools/prof_transform.py "$@" > $OUT/w.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/s -- python3 tools/prof_transform.py "$@" > $OUT/s.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/prof_transform.py "$@" > $OUT/t.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "nttx::" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][11:80], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-70s %-18s %16.0f (n=%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
for f in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "nttx::" in row["Name"]:
            print("%-70s calls %s avg %.1f us" % (row["Name"][11:80], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
