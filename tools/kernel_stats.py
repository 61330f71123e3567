#!/usr/bin/env python3
"""Per-kernel calls / total / average / share from a rocprofv3 --kernel-trace run (csv output).
usage: python tools/kernel_stats.py <dir with *kernel_trace.csv> [header line]"""
import collections, csv, glob, os, sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(v[1] for v in agg.values()) or 1.0
print("# " + (sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --kernel-trace summary"))
print("# %-100s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "share"))
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-102s %8d %12.1f %10.2f %6.2f%%" % (k[:102], c, us, us / c, 100 * us / tot))
