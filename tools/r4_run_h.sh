#!/bin/bash
# round-4 GPU session H: MFMA accumulators in arch VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1) vs the default AGPR form
set -u
O=gpurun_out; mkdir -p $O
REPO=$(pwd)
export V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_vform.so
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -x 2>&1 | tail -15) > $O/r4h_vform_tests.log
tail -3 $O/r4h_vform_tests.log
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4h_trace_vform -- \
      python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4h_traced_vform.json 2> $REPO/$O/r4h_trace_vform.log )
python tools/update_timeline.py $O/r4h_trace_vform $O/r4h_timeline_vform.txt > /dev/null
find $O/r4h_trace_vform -name "*.csv" -size +1M -delete; find $O/r4h_trace_vform -name "*.db" -delete
for i in 1 2 3; do
  for v in vform agpr; do
    if [ $v = vform ]; then export V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_vform.so; else unset V4L_LIB; fi
    python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4h_ab_${v}_$i.json 2>/dev/null
  done
done
unset V4L_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h_ab_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
