#!/bin/bash
# round-4 GPU session C: suite, XCD A/B, full bench, kernel trace + update timeline
set -u
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -60) > $O/r4c_pytest.log
cp $O/parity.json $O/r4c_parity.json 2>/dev/null
for i in 1 2; do
  V4L_ROLLOUT_XCD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4c_ab_xcd0_$i.json 2>/dev/null
  V4L_ROLLOUT_XCD=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4c_ab_xcd1_$i.json 2>/dev/null
done
python bench.py --steps 10 --warmup 3 --breakdown $O/r4c_breakdown.txt > $O/r4c_bench.json 2> $O/r4c_bench.err
REPO=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4c_trace -- \
    python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4c_traced.json 2> $REPO/$O/r4c_trace.log )
python tools/update_timeline.py $O/r4c_trace $O/r4c_timeline.txt > /dev/null
find $O/r4c_trace -name "*.csv" -size +1M -delete; find $O/r4c_trace -name "*.db" -delete
tail -3 $O/r4c_pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c_ab_*.json'))+['gpurun_out/r4c_bench.json']:
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d.get('value_incl_transfers_ratio'))
    except Exception as e: print(f, 'ERR', e)
PY
