// How fast can ONE CU pull L2-resident data? Every block (one per CU, all 256 at once — the situation of the fused kernels)
// streams the same 128 KB region `reps` times with W waves and K loads in flight per wave, through global_load_dwordx4 /
// dwordx2 / dword (lane-contiguous) or the global->LDS DMA; prints bytes per shader clock per CU.
//   hipcc --offload-arch=gfx950 -O3 cu_fetch.hip -o cu_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REGION = 128 * 1024;  // bytes, shared by all blocks (L2 hits after the first touch)

template <typename V, int K>
__global__ __launch_bounds__(1024) void stream_kernel(const V* __restrict__ src, int reps, long long* cyc, float* sink) {
  const int tid = threadIdx.x, nth = blockDim.x;
  const int nvec = REGION / sizeof(V);
  float acc = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    for (int base = 0; base < nvec; base += nth * K) {
      V v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int i = base + k * nth + tid;
        v[k] = src[i < nvec ? i : 0];
      }
#pragma unroll
      for (int k = 0; k < K; ++k) acc += ((const float*)&v[k])[0];
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void dma_kernel(const float4* __restrict__ src, int reps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 64 KB staging
  const int tid = threadIdx.x, nth = blockDim.x;
  const int nvec = REGION / 16;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    for (int base = 0; base < nvec; base += nth) {
      const int i = base + tid;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                       (__attribute__((address_space(3))) void*)(smem + (((base % 4096) + (tid & ~63)) * 16)), 16, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename V, int K>
static void run(const char* what, int threads, const void* src, long long* cyc, float* sink, int blocks) {
  const int reps = 20;
  hipLaunchKernelGGL((stream_kernel<V, K>), dim3(blocks), dim3(threads), 0, 0, (const V*)src, 2, cyc, sink);  // warm
  hipLaunchKernelGGL((stream_kernel<V, K>), dim3(blocks), dim3(threads), 0, 0, (const V*)src, reps, cyc, sink);
  CHECK(hipDeviceSynchronize());
  long long h[256];
  CHECK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("%-28s %2d waves x %d in flight, %3d blocks: %6.1f B/clk per CU\n", what, threads / 64, K, blocks, (double)REGION * reps / (s / blocks));
}

// One burst: W waves each request K fragments (1 KB per wave load) at once and wait for them — the shape of a GEMM phase
// of the fused kernels (e.g. conv3': 8 waves x 9). Cycles from the first request to the last arrival, per block.
template <int K>
__global__ __launch_bounds__(1024) void burst_kernel(const float4* __restrict__ src, int off_vec, long long* cyc, float* sink) {
  const int tid = threadIdx.x, nth = blockDim.x;
  float4 v[K];
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  const long long t0 = clock64();
  __builtin_amdgcn_sched_barrier(0);  // nothing is requested before the first clock read
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = src[off_vec + k * nth + tid];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) acc += v[k].x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}
template <int K>
static void burst(int threads, const void* src, long long* cyc, float* sink, int blocks, bool warm) {
  // warm: the same bytes were just read by every block (L2 / L1 state of a second sample); cold: a region nobody touched in
  // this launch sequence (another kernel's output, the first touch of a phase)
  const int bytes = threads * K * 16;
  if (warm) hipLaunchKernelGGL((burst_kernel<K>), dim3(blocks), dim3(threads), 0, 0, (const float4*)src, 0, cyc, sink);
  hipLaunchKernelGGL((burst_kernel<K>), dim3(blocks), dim3(threads), 0, 0, (const float4*)src, warm ? 0 : 65536 / 16, cyc, sink);
  CHECK(hipDeviceSynchronize());
  long long h[256];
  CHECK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("burst %2d waves x %d fragments = %3d KB, %3d blocks, %s: %6.0f cycles = %5.1f B/clk per CU\n", threads / 64, K, bytes / 1024, blocks,
         warm ? "L2-warm" : "first touch", s / blocks, bytes / (s / blocks));
}

int main() {
  void* src; long long* cyc; float* sink;
  CHECK(hipMalloc(&src, REGION)); CHECK(hipMemset(src, 0, REGION)); CHECK(hipMalloc(&cyc, 256 * 8)); CHECK(hipMalloc(&sink, 4));
  for (int blocks : {256, 32}) {
    run<float4, 1>("global_load_dwordx4", 256, src, cyc, sink, blocks);
    run<float4, 4>("global_load_dwordx4", 256, src, cyc, sink, blocks);
    run<float4, 8>("global_load_dwordx4", 256, src, cyc, sink, blocks);
    run<float4, 4>("global_load_dwordx4", 512, src, cyc, sink, blocks);
    run<float4, 8>("global_load_dwordx4", 512, src, cyc, sink, blocks);
    run<float4, 4>("global_load_dwordx4", 1024, src, cyc, sink, blocks);
    run<float2, 8>("global_load_dwordx2", 512, src, cyc, sink, blocks);
    run<float, 8>("global_load_dword", 512, src, cyc, sink, blocks);
    run<float, 16>("global_load_dword", 1024, src, cyc, sink, blocks);
    {
      const int reps = 20;
      CHECK(hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      hipLaunchKernelGGL(dma_kernel, dim3(blocks), dim3(512), 65536, 0, (const float4*)src, 2, cyc);
      hipLaunchKernelGGL(dma_kernel, dim3(blocks), dim3(512), 65536, 0, (const float4*)src, reps, cyc);
      CHECK(hipDeviceSynchronize());
      long long h[256];
      CHECK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
      double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
      printf("%-28s  8 waves, DMA to LDS,     %3d blocks: %6.1f B/clk per CU\n", "global_load_lds dwordx4", blocks, (double)REGION * reps / (s / blocks));
    }
  }
  for (int blocks : {256, 64}) {
    for (int w = 0; w < 2; ++w) {
      burst<1>(256, src, cyc, sink, blocks, w);
      burst<2>(256, src, cyc, sink, blocks, w);
      burst<4>(512, src, cyc, sink, blocks, w);
      burst<9>(512, src, cyc, sink, blocks, w);
    }
  }
  return 0;
}
