#!/bin/bash
# fused16_kernel (NTTX_FUSED16=1) against the two-launch schedule at small batches of n = 2^16: pairs/s, ms per step
for b in 64 128 256 1024; do
  for f in 0 1; do
    line=$(NTTX_FUSED16=$f timeout -k 10 120 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1) || exit 1
    echo "batch $b fused16=$f $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
