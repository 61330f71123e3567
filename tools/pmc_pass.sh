#!/bin/bash
# PMC passes over one short bench run (each counter group in its own rocprofv3 run, kernel-trace only).
# usage (on the GPU box, from the repo root): tools/pmc_pass.sh <outdir> [bench args...]
set -u
OUT=$(realpath "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export V4L_GRAPH=0 PYTHONPATH=$REPO
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
run() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run tcc TCC_HIT_sum TCC_MISS_sum
python $REPO/tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete   # raw per-dispatch tables stay on the box
