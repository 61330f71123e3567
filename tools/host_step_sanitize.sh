#!/bin/bash
# ThreadSanitizer and AddressSanitizer runs of the collector's host-side cast pool (csrc/host_step.h: persistent worker threads,
# spin / sleep / wake, static partition with claim flags, AVX-512 and scalar casts) through tools/host/cast_pool_check.cpp.
# Host code only — GPU sanitizers are not available on this pool. usage (repo root): tools/host_step_sanitize.sh [out.txt]
set -u
OUT=${1:-/dev/stdout}
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
{
  echo "# tools/host_step_sanitize.sh: $CXX -fsanitize={thread,address} of tools/host/cast_pool_check.cpp (includes csrc/host_step.h)"
  for san in thread address; do
    $CXX -std=c++17 -O1 -g -fsanitize=$san -fno-omit-frame-pointer -pthread tools/host/cast_pool_check.cpp -o /tmp/cast_check_$san || exit 1
    echo "== -fsanitize=$san"
    timeout 1200 /tmp/cast_check_$san 240 2>&1 | tail -20
    echo "exit code: ${PIPESTATUS[0]}"
  done
} > "$OUT"
