#!/bin/bash
# round-4 GPU session F: timeline of the update with the external chains (default) and with the proprio chain in-kernel
set -u
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -k "external_row_chains or wave_per_sample_layers_match or forked_weight_grad" 2>&1 | tail -10) > $O/r4f_xcheck.log
tail -2 $O/r4f_xcheck.log
REPO=$(pwd)
for v in ext tok0in; do
  if [ $v = tok0in ]; then export V4L_WPS_TOK0_IN=1; else unset V4L_WPS_TOK0_IN; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4f_trace_$v -- \
      python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4f_traced_$v.json 2> $REPO/$O/r4f_trace_$v.log )
  python tools/update_timeline.py $O/r4f_trace_$v $O/r4f_timeline_$v.txt > /dev/null
  find $O/r4f_trace_$v -name "*.csv" -size +1M -delete; find $O/r4f_trace_$v -name "*.db" -delete
done
unset V4L_WPS_TOK0_IN
for i in 1 2 3; do
  python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4f_ab_ext_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4f_ab_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
