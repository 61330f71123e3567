#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a directory of rocprofv3 --pmc runs (csv output).
usage: python tools/pmc_kernels.py <dir> [kernel-name substring ...]"""
import collections, csv, glob, os, sys

root, want = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if want and not any(w in k for w in want):
                continue
            k = k[:96]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
order = sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("SQ_INST_LEVEL_VMEM", 0)))
for k in order:
    n = max(cnt[k].values())
    print("== %s  (per launch, %d launches)" % (k, n))
    for name, v in sorted(agg[k].items()):
        print("   %-30s %16.0f" % (name, v / max(1, cnt[k][name])))
