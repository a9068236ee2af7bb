"""Summarise a tools/prof.sh output directory: per-kernel stats + PMC counters per launch, one block per render kernel
(the timed kernel and the probe-counting kernel `<true, ...>` are different instantiations and are kept apart)."""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("kt/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"),
                                                                row.get("AverageNs"), row.get("Percentage")))
print("== PMC: mean per dispatch, per render kernel (FETCH_SIZE / WRITE_SIZE in KiB as reported) ==")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "k_render" not in k:
                continue
            per_dispatch[(short(k), row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (k, did, cn), v in per_dispatch.items():
            acc[k][cn].append(v)
for k in sorted(acc):
    print("-- %s" % k)
    for cn, vals in sorted(acc[k].items()):
        print("   %-26s n=%d mean=%.6g min=%.6g max=%.6g" % (cn, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
