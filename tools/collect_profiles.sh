#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box:
#   kernel-trace stats of the bench command itself, then PMC passes (each its own run) over
#   tools/prof_passes.py, which launches every pass kernel of the same plan and batch.
# Usage: tools/collect_profiles.sh <tag> <workload> <kind> <logn>   (outputs under gpurun_out/<tag>)
set -e
TAG=$1; WL=$2; KIND=$3; LOGN=$4
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_passes.py $KIND $LOGN > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_passes.py $KIND $LOGN > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 tools/prof_passes.py $KIND $LOGN > $OUT/pmc_sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 tools/prof_passes.py $KIND $LOGN > $OUT/pmc_sq2.log 2>&1
echo collected $OUT
