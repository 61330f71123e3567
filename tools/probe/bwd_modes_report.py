import csv, glob, os, sys, statistics
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "wps_layer_bwd" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r["Kernel_Name"][:64]))
rows.sort()
K, per = int(sys.argv[2]), len(rows) // int(sys.argv[2])
for k in range(K):
    chunk = rows[k * per:(k + 1) * per]
    by = {}
    for _, d, n in chunk: by.setdefault(n, []).append(d)
    print("agent %d: " % k + "  ".join("%s... median %.1f us (n=%d)" % (n[38:62], statistics.median(v), len(v)) for n, v in sorted(by.items())))
