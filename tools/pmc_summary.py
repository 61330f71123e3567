"""Per-kernel sums of the rocprofv3 --pmc CSV passes written by tools/pmc_pass.sh -> one table on stdout."""
import csv, glob, os, sys, collections

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?").split("(")[0][:120]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (path, row.get("Dispatch_Id"))
            if row["Counter_Name"] in ("FETCH_SIZE",) and key not in seen:
                seen.add(key)
                calls[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel\tcalls(fetch pass)\t" + "\t".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    print(k + "\t%d\t" % calls[k] + "\t".join("%.4g" % v.get(c, 0) for c in names))
