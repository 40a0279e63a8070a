#!/bin/bash
# Experiment: do a chunk's second-pass reads hit the XCD L2s that its first pass wrote, across a
# kernel boundary?  Counts FETCH_SIZE per dispatch with and without small chunks whose tiles are
# dealt to the XCDs by transform.  Usage: tools/exp_l2_chunk.sh <chunk MiB>
set -e
OUT=gpurun_out/l2chunk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 120 python3 bench.py --batch 512 --steps 20 --kernel-reps 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
}
run base NTTX_NONE=1
run noxb NTTX_NO_CROSS_BATCH=1
run chunk$1 NTTX_DEBUG=1 NTTX_CHUNK_MIB=$1 NTTX_XCD_GROUP=16 NTTX_NO_CROSS_BATCH=1
run chunk$1_nogroup NTTX_DEBUG=1 NTTX_CHUNK_MIB=$1 NTTX_NO_CROSS_BATCH=1
export NTTX_DEBUG=1 NTTX_CHUNK_MIB=$1 NTTX_XCD_GROUP=16 NTTX_NO_CROSS_BATCH=1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$1 -- python3 bench.py --batch 512 --steps 2 --warmup 0 --kernel-reps 1 --no-cpu-baseline > $OUT/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_fetch_$1/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "pass_kernel" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:70], row["Grid_Size"])].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n=%d" % len(v), "mean FETCH_SIZE KB %.0f  (x2 = %.1f MB)" % (sum(v)/len(v), 2*sum(v)/len(v)/1024))
PY
