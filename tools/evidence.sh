#!/bin/bash
# Round evidence on the GPU box: full bench line (with cpu_baseline), rocprofv3 kernel trace + stats of the same command,
# PMC passes for roofline.traffic. usage (repo root): tools/evidence.sh <outdir>
set -u
OUT=$(realpath "$1"); mkdir -p "$OUT"
REPO=$(pwd)
export PYTHONPATH=$REPO
python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python $REPO/bench.py --no-cpu-baseline --no-parity > "$OUT/bench_traced.json" 2> "$OUT/trace.log" )
echo "trace rc=$?"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(v[1] for v in agg.values())
with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-parity` (1 warm-up + 3 timed epochs + 1 profiled rollout + update pass)\n")
    f.write("# %-100s %8s %12s %10s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "share"))
    for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%-102s %8d %12.1f %10.2f %6.2f%%\n" % (k[:102], c, us, us / c, 100 * us / tot))
print(open(os.path.join(out, "kernel_stats.txt")).read()[:3000])
PY
tools/pmc_pass.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python tools/pmc_traffic.py "$OUT/pmc/summary.txt" "$OUT/pmc_traffic.json" > /dev/null
find "$OUT" -name "*.csv" -size +1M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
