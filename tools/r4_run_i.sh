#!/bin/bash
# round-4 GPU session I: update timeline of the NatureCNN workload (BASELINE configs[1])
set -u
O=gpurun_out; mkdir -p $O
REPO=$(pwd)
W=${1:-cnn}
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4i_trace_$W -- \
      python $REPO/bench.py --workload $W --no-cpu-baseline --no-parity > $REPO/$O/r4i_traced_$W.json 2> $REPO/$O/r4i_trace_$W.log )
python tools/update_timeline.py $O/r4i_trace_$W $O/r4i_timeline_$W.txt > /dev/null
find $O/r4i_trace_$W -name "*.csv" -size +1M -delete; find $O/r4i_trace_$W -name "*.db" -delete
python bench.py --workload $W --steps 15 --warmup 3 --no-cpu-baseline --no-parity > $O/r4i_bench_$W.json 2>/dev/null
