"""Prints one line per kernel from a `make asm` resource-usage report: VGPRs, scratch, LDS, occupancy."""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "build/shoup61_near-resource-usage.txt"
cur = {}
rows = []
for line in open(path):
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    text = m.group(1).strip()
    if text.startswith("Function Name:"):
        cur = {"name": text.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in text:
        k, v = text.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"nttx::", "", n).replace("(PassParams, Shoup61Ctx)", "").replace("void ", "")
    print("%-72s vgpr %4s scratch %4s lds %6s occ %s" % (n[:72], r.get("VGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
