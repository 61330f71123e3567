#!/bin/bash
# roctx ranges of the library (V4L_ROCTX=1) under rocprofv3 --marker-trace: one smoke() run, prints how many ranges were recorded
# and a sample. usage (repo root, GPU box): tools/probe/roctx_check.sh <outdir>
OUT=$(realpath "$1"); mkdir -p "$OUT"; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
V4L_ROCTX=1 timeout 240 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/mk -- \
  python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" > "$OUT/roctx.log" 2>&1
f=$(find /tmp/mk -name "*marker_api_trace.csv" | head -1)
echo "marker file: $f"
[ -n "$f" ] && wc -l "$f" && head -3 "$f" && sed -n '2,4000p' "$f" | awk -F, '{print $2}' | sort | uniq -c | sort -rn | head -40 > "$OUT/roctx_ranges.txt"
cat "$OUT/roctx_ranges.txt"
