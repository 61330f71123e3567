#!/bin/bash
# round-4 GPU session N: 16-row chain blocks + padded operand-block stride: tests, then the update timeline of three processes
set -u
O=gpurun_out; mkdir -p $O
REPO=$(pwd)
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_contractions.py tests/test_gpu_guard.py -q -m gpu --tb=short -x 2>&1 | tail -4) > $O/r4n_tests.log; tail -2 $O/r4n_tests.log
for i in 1 2 3; do
  v=final_$i
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4n_trace_$v -- \
      python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4n_traced_$v.json 2> $REPO/$O/r4n_trace_$v.log )
  python tools/update_timeline.py $O/r4n_trace_$v $O/r4n_timeline_$v.txt > /dev/null
  rm -rf $O/r4n_trace_$v
  echo "== $v: $(head -1 $O/r4n_timeline_$v.txt | cut -c60-110) bwd: $(grep wps_layer_bwd $O/r4n_timeline_$v.txt | awk '{printf "%s ", $3}') loss: $(grep loss $O/r4n_timeline_$v.txt | awk '{printf "%s ", $3}') wgrad: $(grep wps_wgrad $O/r4n_timeline_$v.txt | awk '{printf "%s ", $3}')"
done
for i in 1 2 3; do python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4n_bench_$i.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4n_bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
PY
