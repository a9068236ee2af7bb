"""Summarise a tools/prof.sh output directory: per-kernel stats + PMC counters per launch of k_render."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("kt/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
print("== PMC (mean per dispatch of kernels matching k_render) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        per_dispatch = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if "k_render" not in row.get("Kernel_Name", ""):
                continue
            per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (did, cn), v in per_dispatch.items():
            acc[cn].append(v)
        for cn, vals in sorted(acc.items()):
            print("%-28s n=%d mean=%.6g min=%.6g max=%.6g" % (cn, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
