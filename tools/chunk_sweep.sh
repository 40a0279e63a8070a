for c in 0 64 128 192 256 384; do
  NTTX_DEBUG=1 NTTX_CHUNK_MIB=$c timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 --kernel-reps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk', $c, d['value'], d['ms_per_step'])" || exit 1
done
