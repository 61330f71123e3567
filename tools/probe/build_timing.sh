#!/bin/bash
# diagnostic build with clock64 phase stamps (not the shipped library)
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DV4L_INFER_TIMING \
  vision4leg_amd/csrc/v4l_hip.hip -o tools/probe/libv4l_timing.so
