// Device-side hand-over latency between two blocks of one launch (the mechanism DESIGN.md section 7 proposes for splitting a
// rollout sample's stack over several CUs): block A writes a 4 KB payload, releases a flag at agent scope; block B spins on
// it (bounded), acquires, reads the payload, answers the same way. 1000 ping-pongs; prints ns per one-way hop for a pair of
// blocks on the same XCD (block ids 0, 8) and on different XCDs (0, 1).  hipcc --offload-arch=gfx950 -O3 flag_hop.hip -o flag_hop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool wait_flag(const unsigned* f, unsigned want) {
  for (long long spin = 0; spin < (1ll << 24); ++spin) {  // bounded: a lost hand-over must not hang the GPU
    if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

__global__ __launch_bounds__(256) void hop_kernel(float* buf, unsigned* flags, int partner, int iters, long long* out, int* bad) {
  const int b = blockIdx.x;
  if (b != 0 && b != partner) return;
  float* mine = buf + (b == 0 ? 0 : 1024);
  const float* theirs = buf + (b == 0 ? 1024 : 0);
  unsigned* fm = flags + (b == 0 ? 0 : 32);
  const unsigned* ft = flags + (b == 0 ? 32 : 0);
  float acc = 0.f;
  __shared__ int ok;
  long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (b != 0) {  // B: wait for A's i-th message first
      if (threadIdx.x == 0) ok = wait_flag(ft, (unsigned)i);
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
      for (int k = threadIdx.x; k < 1024; k += 256) acc += theirs[k];
    }
    for (int k = threadIdx.x; k < 1024; k += 256) mine[k] = (float)(i + k) + acc * 1e-30f;
    __syncthreads();  // all of the block's payload stores are issued
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_store(fm, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b == 0) {  // A: wait for B's answer
      if (threadIdx.x == 0) ok = wait_flag(ft, (unsigned)i);
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
      for (int k = threadIdx.x; k < 1024; k += 256) acc += theirs[k];
      // payload check: B wrote (i + k) (+ negligible)
      if (threadIdx.x == 0 && fabsf(theirs[5] - (float)(i + 5)) > 0.5f) atomicAdd(bad, 1);
    }
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0 && b == 0) { out[0] = t1 - t0; out[1] = (long long)acc; }
}

int main() {
  float* buf; unsigned* flags; long long* out; int* bad;
  CHECK(hipMalloc(&buf, 2048 * 4)); CHECK(hipMalloc(&flags, 64 * 4)); CHECK(hipMalloc(&out, 16)); CHECK(hipMalloc(&bad, 4));
  int rate_khz = 0;
  CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const int iters = 1000;
  const int partners[3] = {8, 1, 4};
  const char* what[3] = {"same XCD (blocks 0 and 8)", "neighbour XCD (blocks 0 and 1)", "XCD 4 (blocks 0 and 4)"};
  for (int p = 0; p < 3; ++p) {
    CHECK(hipMemset(flags, 0, 64 * 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(out, 0, 16));
    hipLaunchKernelGGL(hop_kernel, dim3(16), dim3(256), 0, 0, buf, flags, partners[p], iters, out, bad);
    CHECK(hipDeviceSynchronize());
    long long h[2]; int hb;
    CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    const double ns = (double)h[0] / (double)rate_khz * 1e6 / (2.0 * iters);
    printf("%-34s %8.0f ns per one-way hand-over of 4 KB (wall clock %d kHz), errors %d\n", what[p], ns, rate_khz, hb);
  }
  return 0;
}
