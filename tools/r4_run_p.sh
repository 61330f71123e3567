#!/bin/bash
set -u
O=gpurun_out; mkdir -p $O; REPO=$(pwd)
for i in 1 2; do
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4p_trace_$i -- python $REPO/tools/probe/bwd_modes.py 4 > $REPO/$O/r4p_$i.log 2>&1 )
echo "== process $i"; python tools/probe/bwd_modes_report.py $O/r4p_trace_$i 4
rm -rf $O/r4p_trace_$i
done
