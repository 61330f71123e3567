#!/bin/bash
# round-4 GPU session J: NatureCNN update after the forked weight-grad section / templated tanh / padded-row epilogue
set -u
O=gpurun_out; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_bench_path.py -q -m gpu --tb=short -x 2>&1 | tail -15) > $O/r4j_tests.log
tail -3 $O/r4j_tests.log
bash tools/r4_run_i.sh cnn
for w in mlp cnn_vis loco_vis loco; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/r4j_bench_$w.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4j_bench_*.json'))+['gpurun_out/r4i_bench_cnn.json']:
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['rollout_inference_ms_per_step'], d['update_only_env_steps_per_s'])
    except Exception as e: print(f, 'ERR', e)
PY
