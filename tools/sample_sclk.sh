#!/bin/bash
# Samples the shader clock (rocm-smi) while a bench.py workload loops.  Usage: tools/sample_sclk.sh <workload> <steps> <out>
WL=$1; STEPS=$2; OUT=$3
(timeout -k 10 200 python bench.py --no-cpu-baseline --workload $WL --steps $STEPS > /tmp/b_$WL.json 2>/dev/null) &
BP=$!
sleep 5
echo "# sclk while bench.py --workload $WL --steps $STEPS runs (first samples may precede the timed loop)" >> $OUT
for i in $(seq 1 14); do
  rocm-smi --showclocks 2>/dev/null | grep -E "GPU\[0\].*sclk" >> $OUT
  sleep 0.7
done
wait $BP
python -c "import json; b=json.load(open('/tmp/b_$WL.json')); print('# result', b['value'], b['unit'], b['ms_per_step'], 'ms per step')" >> $OUT
