#!/bin/bash
# A/B/C of several builds of the library on one box, with per-pass event times:
#   tools/ab_libs.sh "<workload> [<workload> ...]" <lib.so|default> [<lib.so|default> ...]
# Two rounds per workload, the builds taken in turn inside each round.
wls=$1; shift
for w in $wls; do
  for rep in 1 2; do
    for lib in "$@"; do
      path=$lib; [ "$lib" = default ] && path=""
      NTTX_LIB_PATH=$path timeout -k 10 200 python3 bench.py --workload "$w" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-14s %-34s %10.1f pairs/s  %.4f ms/step  ' % ('$w', '$lib', d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (k['name'].split(' stages')[0], k['avg_ms']) for k in d['roofline']['kernels']) + '  clk %s' % ((d['roofline'].get('valu') or {}).get('clock_mhz')))
" || exit 1
    done
  done
done
