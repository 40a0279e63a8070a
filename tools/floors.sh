#!/bin/bash
# Measured floors of a bench workload's step (run on the GPU box; needs ntt_amd/libnttx_prof.so from
# `make -C ntt_amd/csrc prof`): the shipped library, then the profiling build with the butterflies
# switched off (memory only), with no global loads or stores of data (compute only), with every workgroup
# on 64 L2-resident tiles (the older compute-only proxy: its stores contend for the same lines), and with
# neither butterflies nor data traffic.
# Writes gpurun_out/floors_<workload>.json; tools/stamp_floors.py turns it into profiles/floors_<workload>.json.
# Usage: tools/floors.sh <workload> [<workload> ...]
PROF=$PWD/ntt_amd/libnttx_prof.so
[ -f "$PROF" ] || { echo "no $PROF: run make -C ntt_amd/csrc prof"; exit 1; }
for wl in "$@"; do
  python3 - "$wl" <<'PY' > gpurun_out/floors_$wl.json || exit 1
import json, os, subprocess, sys
wl = sys.argv[1]
prof = os.path.join(os.getcwd(), "ntt_amd", "libnttx_prof.so")
legs = [("shipped", {}),
        ("memory_only", {"NTTX_LIB_PATH": prof, "NTTX_DEBUG": "1", "NTTX_DEBUG_NO_MATH": "1"}),
        ("compute_only", {"NTTX_LIB_PATH": prof, "NTTX_DEBUG": "1", "NTTX_DEBUG_NO_MEM": "1"}),
        ("hot_tiles", {"NTTX_LIB_PATH": prof, "NTTX_DEBUG": "1", "NTTX_DEBUG_HOT_TILES": "1"}),
        ("neither", {"NTTX_LIB_PATH": prof, "NTTX_DEBUG": "1", "NTTX_DEBUG_NO_MEM": "1", "NTTX_DEBUG_NO_MATH": "1"}),
        ("shipped_again", {})]
out = {"workload": wl, "legs": {}}
for name, env in legs:
    e = dict(os.environ, **env)
    args = ["python3", "bench.py", "--workload", wl, "--no-cpu-baseline", "--steps", "30"] + (["--no-verify"] if env else [])
    r = subprocess.run(args, env=e, capture_output=True, text=True, timeout=200)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        sys.exit(1)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out["legs"][name] = {"ms_per_step": d["ms_per_step"], "kernels": {k["name"]: k["avg_ms"] for k in d["roofline"]["kernels"]}}
    sys.stderr.write("%-14s %-14s %.4f ms/step  %s\n" % (wl, name, d["ms_per_step"], "  ".join("%.3f" % k["avg_ms"] for k in d["roofline"]["kernels"])))
print(json.dumps(out, indent=1))
PY
done
