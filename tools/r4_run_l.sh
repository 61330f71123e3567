#!/bin/bash
# round-4 GPU session L: the NatureCNN nets' dense stack as one launch per direction (csrc/dense_stack.h)
set -u
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -x -k "fused_dense_stack or ((test_forward or test_backward or test_ppo_update or golden) and cnn)" 2>&1 | tail -15) > $O/r4l_tests.log
tail -4 $O/r4l_tests.log
bash tools/r4_run_i.sh cnn
for i in 1 2; do
for v in stack layers; do
  if [ $v = layers ]; then export V4L_NO_DENSE_STACK=1; else unset V4L_NO_DENSE_STACK; fi
  for w in cnn cnn_vis; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4l_bench_${w}_${v}_$i.json 2>/dev/null; done
done
done
unset V4L_NO_DENSE_STACK
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4l_bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
