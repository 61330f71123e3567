#!/bin/bash
# AddressSanitizer build of the C-ABI library's HOST side (plans, layouts, pack tables, descriptor bookkeeping, error paths) and a
# run of the CPU tests that drive it without a GPU (net creation for every net kind, parameter tables, unsupported-config
# errors, exported symbols). Device code is compiled as usual; nothing is launched. usage (repo root): tools/asan_check.sh [out.txt]
set -u
OUT=${1:-/dev/stdout}
R=$(pwd)
LIB=/tmp/libv4l_asan.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=address -shared-libasan -fno-omit-frame-pointer \
  vision4leg_amd/csrc/v4l_hip.hip -o $LIB 2> /tmp/asan_build.log || { tail -5 /tmp/asan_build.log; exit 1; }
{
  echo "# tools/asan_check.sh: hipcc -fsanitize=address build of csrc/v4l_hip.hip (host side), CPU tests of the C ABI under it"
  echo "# runtime: $RT"
  V4L_LIB=$LIB LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 \
    timeout 900 python -m pytest tests/test_cpu.py -q -m "not gpu" -k "library_exports or plan_matches or unsupported_configs" -p no:cacheprovider 2>&1 | tail -6
  echo "exit code: ${PIPESTATUS[0]}"
} > "$OUT"
