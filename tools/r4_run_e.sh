#!/bin/bash
# round-4 GPU session E: bit-equality checks after the contraction pragma, max_pool cases, A/B x3 of the external chains
set -u
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -k "external_row_chains or wave_per_sample_layers_match or forked_weight_grad or loco_max or loco_vis_max" 2>&1 | tail -40) > $O/r4e_xcheck.log
tail -4 $O/r4e_xcheck.log
for i in 1 2 3; do
  V4L_WPS_HEAD_IN=1 V4L_WPS_TOK0_IN=1 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4e_ab_in_$i.json 2>/dev/null
  python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4e_ab_ext_$i.json 2>/dev/null
  V4L_WPS_HEAD_IN=1 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4e_ab_tok0ext_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e_ab_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
(timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -40) > $O/r4e_pytest.log
tail -4 $O/r4e_pytest.log
