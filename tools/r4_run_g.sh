#!/bin/bash
# round-4 GPU session G: the weight-grad reduction split into the forked section (default) vs one launch behind the join
set -u
O=gpurun_out; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -k "forked_weight_grad or graph" 2>&1 | tail -15) > $O/r4g_xcheck.log
tail -3 $O/r4g_xcheck.log
REPO=$(pwd)
for v in split one; do
  if [ $v = one ]; then export V4L_SPLIT_REDUCE=0; else unset V4L_SPLIT_REDUCE; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4g_trace_$v -- \
      python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4g_traced_$v.json 2> $REPO/$O/r4g_trace_$v.log )
  python tools/update_timeline.py $O/r4g_trace_$v $O/r4g_timeline_$v.txt > /dev/null
  find $O/r4g_trace_$v -name "*.csv" -size +1M -delete; find $O/r4g_trace_$v -name "*.db" -delete
done
for i in 1 2 3; do
  for v in split one; do
    if [ $v = one ]; then export V4L_SPLIT_REDUCE=0; else unset V4L_SPLIT_REDUCE; fi
    python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4g_ab_${v}_$i.json 2>/dev/null
  done
done
unset V4L_SPLIT_REDUCE
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g_ab_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
