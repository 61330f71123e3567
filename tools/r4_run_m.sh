#!/bin/bash
# round-4 GPU session M: rows per block of the external row chains (loss-heads launches / the proprio chain in wps_wgrad)
set -u
O=gpurun_out; mkdir -p $O
REPO=$(pwd)
for lib in base rc22 rc12 rc11; do
  if [ $lib = base ]; then unset V4L_LIB; else export V4L_LIB=$REPO/vision4leg_amd/libv4l_hip_$lib.so; fi
  for crit in 0 1; do
    if [ $crit = 1 ]; then export V4L_WPS_HEAD_EXT_CRITIC=1; else unset V4L_WPS_HEAD_EXT_CRITIC; fi
    v=${lib}_c$crit
    ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$O/r4m_trace_$v -- \
        python $REPO/bench.py --no-cpu-baseline --no-parity > $REPO/$O/r4m_traced_$v.json 2> $REPO/$O/r4m_trace_$v.log )
    python tools/update_timeline.py $O/r4m_trace_$v $O/r4m_timeline_$v.txt > /dev/null
    rm -rf $O/r4m_trace_$v
    echo "== $v: $(head -1 $O/r4m_timeline_$v.txt | cut -c1-110)"
    grep -E "loss|wps_layer_bwd|wps_wgrad" $O/r4m_timeline_$v.txt | head -8
  done
done
