"""Board power and shader clock while a bench workload runs (run on the GPU box).

For each leg (the shipped library, then the profiling build with the butterflies or the data traffic
switched off) bench.py runs as a child for a few seconds of steps while this process reads the
amdgpu hwmon files of every card it can see (power1_average / power1_input in microwatts,
freq1_input in Hz, power1_cap) every 20 ms.  The card whose power rises most is the one under test;
its loaded plateau is reported as the median of the samples above 80 % of the leg's maximum.

Usage: python3 tools/power_probe.py <workload> [steps]      -> gpurun_out/power_<workload>.json
"""
import glob, json, os, statistics, subprocess, sys, threading, time

wl = sys.argv[1] if len(sys.argv) > 1 else "mont16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
prof = os.path.join(os.getcwd(), "ntt_amd", "libnttx_prof.so")


def sensors():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        card = hw.split("/")[4]
        power = next((os.path.join(hw, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, f))), None)
        freq = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
        cap = os.path.join(hw, "power1_cap") if os.path.exists(os.path.join(hw, "power1_cap")) else None
        if power:
            out.append({"card": card, "power": power, "freq": freq, "cap": cap})
    return out


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def run_leg(name, env, cards):
    samples = {c["card"]: [] for c in cards}
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            for c in cards:
                p = read_int(c["power"])
                f = read_int(c["freq"]) if c["freq"] else None
                if p is not None:
                    samples[c["card"]].append((p / 1e6, (f or 0) / 1e6))
            time.sleep(0.02)

    th = threading.Thread(target=sampler)
    th.start()
    args = ["python3", "bench.py", "--workload", wl, "--no-cpu-baseline", "--steps", str(steps), "--kernel-reps", "2"]
    if env:
        args.append("--no-verify")
    r = subprocess.run(args, env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    stop.set()
    th.join()
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        sys.exit(1)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    best = None
    for card, s in samples.items():
        if len(s) < 10:
            continue
        watts = [w for w, _ in s]
        swing = max(watts) - min(watts)
        if best is None or swing > best[0]:
            best = (swing, card, s)
    rec = {"ms_per_step": d["ms_per_step"], "clock_probe_mhz": (d["roofline"].get("valu") or {}).get("clock_mhz")}
    if best:
        _, card, s = best
        top = max(w for w, _ in s)
        loaded = [(w, f) for w, f in s if w >= 0.8 * top]
        rec.update({"card": card, "samples": len(s), "idle_w": round(min(w for w, _ in s), 1), "max_w": round(top, 1),
                    "loaded_median_w": round(statistics.median(w for w, _ in loaded), 1),
                    "loaded_median_sclk_mhz": round(statistics.median(f for _, f in loaded), 1),
                    "loaded_samples": len(loaded)})
    sys.stderr.write("%-14s %-14s %s\n" % (wl, name, json.dumps(rec)))
    return rec


cards = sensors()
sys.stderr.write("sensors: %s\n" % json.dumps(cards))
out = {"workload": wl, "steps": steps, "caps_w": {c["card"]: (read_int(c["cap"]) or 0) / 1e6 for c in cards if c["cap"]}, "legs": {}}
legs = [("shipped", {}),
        ("memory_only", {"NTTX_LIB_PATH": prof, "NTTX_DEBUG": "1", "NTTX_DEBUG_NO_MATH": "1"}),
        ("compute_only", {"NTTX_LIB_PATH": prof, "NTTX_DEBUG": "1", "NTTX_DEBUG_NO_MEM": "1"})]
for name, env in legs:
    out["legs"][name] = run_leg(name, env, cards)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/power_%s.json" % wl, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
