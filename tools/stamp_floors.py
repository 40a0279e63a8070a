"""gpurun_out/floors_<workload>.json (tools/floors.sh) -> profiles/floors_<workload>.json, stamped with the
kernel sources it was measured on (bench.py quotes it only while the stamp matches)."""
import json, os, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_stamp

for wl in sys.argv[1:]:
    d = json.load(open("gpurun_out/floors_%s.json" % wl))
    legs = d["legs"]
    rec = {
        "shipped_ms": round((legs["shipped"]["ms_per_step"] + legs["shipped_again"]["ms_per_step"]) / 2, 4),
        "memory_only_ms": legs["memory_only"]["ms_per_step"],
        "compute_only_ms": legs["compute_only"]["ms_per_step"],
        "neither_ms": legs["neither"]["ms_per_step"],
        "per_kernel_ms": {leg: v["kernels"] for leg, v in legs.items()},
        "_stamp": kernel_source_stamp(),
        "hot_tiles_ms": legs["hot_tiles"]["ms_per_step"],
        "_source": "tools/floors.sh %s (profiling build: butterflies off = memory only; no global loads or stores of data = compute only; hot_tiles = every workgroup on 64 L2-resident tiles)" % wl,
    }
    try:
        rec["_git"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        pass
    json.dump(rec, open("profiles/floors_%s.json" % wl, "w"), indent=1)
    print("profiles/floors_%s.json" % wl, {k: rec[k] for k in ("shipped_ms", "memory_only_ms", "compute_only_ms")})
