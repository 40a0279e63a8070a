"""Turns gpurun_out/<tag> (tools/collect_profiles.sh) into the committed summaries under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc.txt (mean counter per kernel),
traffic_<workload>.json (HBM bytes per launch: FETCH_SIZE KB x 1024 x 2 -- gfx950 tallies a
128-byte read request as 64 bytes -- plus WRITE_SIZE KB x 1024)."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_stamp  # the stamp bench.py checks before quoting these files

tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
# gpurun_out/ keeps the files of earlier collections (one per process id): take the newest
stats = sorted(glob.glob(src + "/trace/*/*kernel_stats.csv"), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], "profiles/%s_kernel_stats.csv" % tag)
shutil.copy(src + "/bench.json", "profiles/%s_bench.json" % tag)


def short(name):
    m = re.search(r"fused_small_kernel<nttx::(.+?), (\w+), (\d)>\(", name)
    if m:  # bench.py calls it "fwd fused 4+8 stages ..."
        return (m.group(1), "fwd" if m.group(2) == "true" else "bwd", "fused", int(m.group(3)), 8)
    m = re.search(r"pass_kernel<nttx::(.+?), (\w+), (\w+), (\d), (\d)(?:, \w+)*>\(", name)
    if not m:
        return None
    return (m.group(1), "fwd" if m.group(2) == "true" else "bwd", "contig" if m.group(3) == "true" else "strided",
            int(m.group(4)), int(m.group(5)))


acc = collections.defaultdict(list)
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    for f in glob.glob(src + "/" + d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
with open("profiles/%s_pmc.txt" % tag, "w") as out:
    out.write("# mean counter value per launch (rocprofv3 --pmc, separate runs per counter group), workload %s\n" % workload)
    for (k, c), v in sorted(acc.items()):
        out.write("%-10s %s %-7s %d+%d  %-22s %16.0f  (n=%d)\n" % (k[0], k[1], k[2], k[3], k[4], c, sum(v) / len(v), len(v)))

# traffic per launch, keyed by the pass name bench.py reports ("fwd strided 4+4 stages 1-8")
bench = json.load(open(src + "/bench.json"))
traffic = {}
valu = {}  # wave-level VALU instructions per launch (SQ_INSTS_VALU)
for kern in bench["roofline"]["kernels"]:
    name = kern["name"]
    d, layout, ks = name.split(" ")[0], name.split(" ")[1], name.split(" ")[2]
    k1, k2 = (int(x) for x in ks.split("+"))
    cand = [k for (k, c) in acc if c == "FETCH_SIZE" and k[1] == d and k[2] == layout and k[3] == k1 and k[4] == k2]
    if not cand:
        continue
    k = cand[0]
    fetch = sum(acc[(k, "FETCH_SIZE")]) / len(acc[(k, "FETCH_SIZE")])
    write = sum(acc[(k, "WRITE_SIZE")]) / len(acc[(k, "WRITE_SIZE")])
    traffic[name] = int(fetch * 1024 * 2 + write * 1024)
    if (k, "SQ_INSTS_VALU") in acc:
        valu[name] = int(sum(acc[(k, "SQ_INSTS_VALU")]) / len(acc[(k, "SQ_INSTS_VALU")]))
try:
    sha = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    sha = "unknown"
for rec in (traffic, valu):
    rec["_stamp"] = kernel_source_stamp()  # hash of ntt_amd/csrc/*.hpp, *.hip at measurement time
    rec["_git"] = sha                      # commit the working tree was based on (informational)
    rec["_source"] = "gpurun_out/%s -> profiles/%s_pmc.txt" % (tag, tag)
json.dump(traffic, open("profiles/traffic_%s.json" % workload, "w"), indent=1)
json.dump(valu, open("profiles/valu_%s.json" % workload, "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(valu, indent=1))
