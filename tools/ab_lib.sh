#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <other libnttx.so> <workload> [<workload> ...]
# Alternates default build / other build twice per workload; prints pairs/s and ms per step.
other=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for lib in "" "$other"; do
      line=$(NTTX_LIB_PATH=$lib timeout -k 10 200 python bench.py --workload "$w" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) || exit 1
      echo "$w ${lib:-default} $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
