#!/bin/bash
# Diagnostic variant builds of the library (never the shipped one): tools/probe/build_variant.sh <name> [extra hipcc flags...]
#   roll   -DV4L_WPS_ROLL_LAYERS   layer loops of the wave-per-sample stack kernels as run-time loops (code size / I-cache probe)
#   timing -DV4L_INFER_TIMING      clock64 phase stamps
#   onetile -DV4L_WPS_PROBE_ONE_TILE  TIMING ONLY, wrong results: the layer functions walk one token tile (what the padding tile costs)
#   eu1    -DV4L_WPS_EU1           amdgpu_waves_per_eu(1,1) on the wave-per-sample stack kernels
#   x3     -DV4L_PROBE_F32_SPLIT3  the fp32 mode's MFMAs as three bf16 MFMAs on high / low operand halves (accuracy / speed probe)
#   only1 / only2 / only0  -DV4L_DEV_ONLY=<mode>  development build holding ONE compute mode (bf16 / f16 / f32): a third of the compile time
#   ilp    -mllvm -amdgpu-sched-strategy=max-ilp ; bias0  -mllvm -amdgpu-schedule-metric-bias=0   (whole library)
# -> vision4leg_amd/libv4l_hip_<name>.so, selected at run time with V4L_LIB=<path>
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
case $name in
  roll) flags="-DV4L_WPS_ROLL_LAYERS" ;;
  timing) flags="-DV4L_INFER_TIMING" ;;
  eu1) flags="-DV4L_WPS_EU1" ;;
  onetile) flags="-DV4L_WPS_PROBE_ONE_TILE" ;;
  x3) flags="-DV4L_PROBE_F32_SPLIT3" ;;
  only0) flags="-DV4L_DEV_ONLY=0" ;;
  only1) flags="-DV4L_DEV_ONLY=1" ;;
  only2) flags="-DV4L_DEV_ONLY=2" ;;
  ilp) flags="-mllvm -amdgpu-sched-strategy=max-ilp" ;;
  bias0) flags="-mllvm -amdgpu-schedule-metric-bias=0" ;;
  *) flags="" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags "$@" vision4leg_amd/csrc/v4l_hip.hip \
  -o vision4leg_amd/libv4l_hip_$name.so
ls -la vision4leg_amd/libv4l_hip_$name.so
