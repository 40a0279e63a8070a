#!/bin/bash
# Per-pass event times of a bench workload under a list of NTTX_* environment settings.
# Usage: tools/pass_times.sh <workload> "<env settings>" ["<env settings>" ...]   ("-" = none)
WL=$1; shift
PROF=$PWD/ntt_amd/libnttx_prof.so
for e in "$@"; do
  [ "$e" = "-" ] && e="NTTX_NONE=1"
  case "$e" in *NTTX_DEBUG_*) e="$e NTTX_LIB_PATH=$PROF";; esac  # the switches that skip work exist in the profiling build only
  env $e timeout -k 10 120 python3 bench.py --workload $WL --no-cpu-baseline --no-verify --steps 30 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s %10.1f pairs/s  %.4f ms/step  ' % ('$e', d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (k['name'].split(' stages')[0], k['avg_ms']) for k in d['roofline']['kernels']))
" || exit 1
done
