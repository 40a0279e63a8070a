#!/bin/bash
# Dynamic instruction counts per wave of every pass kernel (rocprofv3 --pmc over tools/prof_passes.py).
# Usage: tools/valu_per_wave.sh <kind> <logn>   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_vpw
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_vpw -- python3 tools/prof_passes.py $1 $2 > gpurun_out/pmc_vpw.log 2>&1
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_vpw/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m=re.search(r"pass_kernel<nttx::([\w<>]+), (\w+), (\w+), (\d), (\d)((?:, \w+)*)>", row["Kernel_Name"])
        if m: acc[(m.groups(), row["Counter_Name"])].append(float(row["Counter_Value"]))
waves={k[0]:sum(v)/len(v) for k,v in acc.items() if k[1]=="SQ_WAVES"}
for k,v in sorted(acc.items()):
    if k[1]!="SQ_WAVES": print(k[0], k[1], round(sum(v)/len(v)/waves[k[0]],1))
PY
