import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2])]:
    print("%-90s %7d %10.1f %8.2f" % (k[:90], c, us, us / c))
