#!/bin/bash
# GPU-side AddressSanitizer run (SURVEY.md section 5 "sanitizers" row): libv4l_hip.so built with
#   hipcc --offload-arch=gfx950:xnack+ -O1 -g -fsanitize=address -shared-libsan      (device code instrumented: asanrtl.bc)
# The build takes ~75 minutes (the fused kernels are 1-2.5 K lines of unrolled MFMA code each), so it is done ahead of time in
# the build container (`tools/asan_gpu_check.sh build`) and the library travels to the GPU box as tools/asan/libv4l_gpu_asan.so
# (git-ignored like every .so). On the box: `tools/asan_gpu_check.sh run [out.txt]` runs smoke() and the offset-stressing tests
# under it with HSA_XNACK=1.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
LIB=$R/tools/asan/libv4l_gpu_asan.so
case "${1:-run}" in
build)
  mkdir -p "$R/tools/asan"
  /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address -shared-libsan \
    -fno-omit-frame-pointer "$R/vision4leg_amd/csrc/v4l_hip.hip" -o "$LIB"
  ;;
run)
  OUT=${2:-/dev/stdout}
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  cd "$R"
  {
    echo "# tools/asan_gpu_check.sh run: device-ASAN build of csrc/v4l_hip.hip ($(stat -c %s "$LIB") bytes), HSA_XNACK=1, runtime $RT"
    export HSA_XNACK=1 V4L_LIB=$LIB LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:abort_on_error=0
    echo "## smoke()"
    timeout 420 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15
    echo "smoke exit code: ${PIPESTATUS[0]}"
    echo "## ragged batches (n = 1, 30), dense rollout step E = 33"
    timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
      -k "fused_kernels_on_ragged_batches or (dense_rollout_step_env_counts and 33)" 2>&1 | tail -15
    echo "pytest exit code: ${PIPESTATUS[0]}"
  } > "$OUT" 2>&1
  ;;
esac
