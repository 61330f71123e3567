#!/bin/bash
# round-4 GPU session D: new cross-check first, then the suite, A/B of the external chains, trace + timeline
set -u
O=gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -x -k "external_row_chains" 2>&1 | tail -30) > $O/r4d_xcheck.log
tail -3 $O/r4d_xcheck.log
(timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -80) > $O/r4d_pytest.log
tail -3 $O/r4d_pytest.log
for i in 1 2; do
  V4L_WPS_HEAD_IN=1 V4L_WPS_TOK0_IN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4d_ab_in_$i.json 2>/dev/null
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4d_ab_ext_$i.json 2>/dev/null
done
V4L_WPS_TOK0_IN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4d_ab_headsext_1.json 2>/dev/null
V4L_WPS_HEAD_IN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4d_ab_tok0ext_1.json 2>/dev/null
REPO=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4d_trace -- \
    python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4d_traced.json 2> $REPO/$O/r4d_trace.log )
python tools/update_timeline.py $O/r4d_trace $O/r4d_timeline.txt > /dev/null
find $O/r4d_trace -name "*.csv" -size +1M -delete; find $O/r4d_trace -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d_ab_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
