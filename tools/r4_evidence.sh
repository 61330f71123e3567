#!/bin/bash
# round-4 evidence on the GPU box: tools/evidence.sh (full bench line, rocprofv3 kernel trace + stats, PMC passes), the update
# timeline of that trace, and bench lines of the other workloads / the fp32 parity mode
set -u
O=gpurun_out/r4_ev; mkdir -p $O
# keep the trace CSV long enough for the timeline: evidence.sh deletes large CSVs at its end, so run the trace part here
REPO=$(pwd)
export PYTHONPATH=$REPO
python bench.py --steps 20 --warmup 5 --breakdown $O/breakdown.txt > $O/bench_full.json 2> $O/bench_full.err
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$O/trace" -- \
    python $REPO/bench.py --no-cpu-baseline --no-parity > "$REPO/$O/bench_traced.json" 2> "$REPO/$O/trace.log" )
python tools/update_timeline.py $O/trace $O/update_timeline.txt > /dev/null
python - "$O" <<'PY'
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
PY
tools/pmc_pass.sh "$O/pmc" > "$O/pmc.log" 2>&1
python tools/pmc_traffic.py "$O/pmc/summary.txt" "$O/pmc_traffic.json" > /dev/null
for w in cnn mlp loco64 loco_vis cnn_vis; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-reference-protocol > $O/bench_$w.json 2> $O/bench_$w.err
done
python bench.py --compute f32 --steps 5 --warmup 2 --no-cpu-baseline --no-reference-protocol > $O/bench_f32.json 2> $O/bench_f32.err
find "$O" -name "*.csv" -size +1M -delete; find "$O" -name "*.db" -delete
du -sh $O
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_ev/bench_*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d.get("value_incl_transfers_ratio"))
    except Exception as e: print(f, "ERR", e)
PY
