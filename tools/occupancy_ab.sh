#!/bin/bash
# PMC view of the headline passes under several builds of the library (occupancy A/B):
#   tools/occupancy_ab.sh <lib.so|default> [...]     (run on the GPU box; one PMC group per rocprofv3 run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  path=$PWD/$lib; [ "$lib" = default ] && path=""
  tag=$(basename "$lib" .so)
  OUT=gpurun_out/occ_$tag
  rm -rf $OUT; mkdir -p $OUT
  export NTTX_LIB_PATH=$path
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/s -- python3 tools/prof_transform.py mont 16 > $OUT/s.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES --output-format csv -d $OUT/l -- python3 tools/prof_transform.py mont 16 > $OUT/l.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/g -- python3 tools/prof_transform.py mont 16 > $OUT/g.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/prof_transform.py mont 16 > $OUT/t.log 2>&1 || exit 1
  echo "== $lib"
  python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
def short(n):
    m = re.search(r"pass_kernel<nttx::([\w<>]+), (\w+), (\w+), (\d), (\d)(?:, \w+)*>", n)
    return None if not m else "%s %s" % ("fwd" if m.group(2) == "true" else "bwd", "contig" if m.group(3) == "true" else "strided")
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k: acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
kern = sorted({k for k, _ in acc})
ctrs = sorted({c for _, c in acc})
print("%-22s" % "counter (mean/launch)" + "".join("%16s" % k for k in kern))
for c in ctrs:
    print("%-22s" % c + "".join("%16.0f" % (sum(acc[(k, c)]) / max(1, len(acc[(k, c)]))) for k in kern))
for f in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        if k: print("%-22s %s calls %s avg %.1f us" % ("kernel-trace", k, row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
