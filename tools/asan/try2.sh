#!/bin/bash
R=$(pwd); LIB=$R/tools/asan/libv4l_gpu_asan.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
O=gpurun_out/r4_gpu_asan_try2.txt
{
echo "## rocminfo xnack"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -i -m3 "xnack\|gfx950"
for v in A B; do
  export HSA_XNACK=1 V4L_LIB=$LIB LD_PRELOAD=$RT
  if [ $v = A ]; then export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:allocator_may_return_null=1; echo "## variant A: allocator_may_return_null=1";
  else export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}; echo "## variant B: LD_LIBRARY_PATH=/opt/rocm/lib (ROCm's own libamdhip64 / libhsa-runtime64 ahead of torch's bundled ones)"; fi
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -12 | cut -c1-400
  echo "exit: ${PIPESTATUS[0]}"
done
} > $O 2>&1
tail -30 $O
