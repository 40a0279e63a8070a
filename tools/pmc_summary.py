"""Summarises rocprofv3 counter_collection.csv files: mean counter value per kernel."""
import csv, glob, sys, collections, re
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "pass_kernel" not in name:
                continue
            m = re.search(r"pass_kernel<nttx::(.+?), (\w+), (\w+), (\d), (\d)(?:, \w+)*>\(", name)
            short = "%s %s %s %s+%s" % (m.group(1), "fwd" if m.group(2) == "true" else "bwd",
                                        "contig" if m.group(3) == "true" else "strided", m.group(4), m.group(5))
            acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("%-34s %-24s %16.0f  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
