#!/bin/bash
# interleaved env-only A/B of bench.py: tools/probe/ab_env.sh <out> <rounds> "name=ENV1=a ENV2=b" ...
OUT=$1; R=$2; shift 2
for i in $(seq 1 $R); do for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  ( for kv in $envs; do export "$kv"; done
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name', 'value %.0f update_only %.0f update_us %.1f fb_us %.1f rest_us %.1f' % (d['value'], d['update_only_env_steps_per_s'], r['update_us'], r['avg_launch_us'], r['update_us']-2*r['avg_launch_us']))" )
done; done > $OUT 2>&1
