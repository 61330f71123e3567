#!/bin/bash
# SQ counters of the layer kernels (one pass), summarised per kernel. usage: tools/probe/pmc_wps.sh <outdir>
OUT=$(realpath "$1"); REPO=$(pwd); mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$REPO
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -- python $REPO/tools/probe/wps_run.py > "$OUT/$tag.log" 2>&1
  echo "$tag rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for path in glob.glob(os.path.join(out,"**","*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_INSTS_VALU","SQ_LDS_BANK_CONFLICT"): cnt[(k,r["Counter_Name"])]+=1
with open(os.path.join(out,"summary.txt"),"w") as f:
    for k,c in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0)):
        n=max(1,cnt[(k,"SQ_WAVE_CYCLES")])
        f.write("== %s  (per launch, %d launches)\n" % (k,n))
        for name,v in sorted(c.items()): f.write("   %-28s %14.0f\n" % (name, v/n))
print(open(os.path.join(out,"summary.txt")).read()[:6000])
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
