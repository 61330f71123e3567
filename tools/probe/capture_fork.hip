// Probe: does ROCm's stream capture survive repeated fork/join through the SAME event pair (variant 0) vs a pool of
// distinct events (variant 1), with nested forks and memset nodes?  usage: capture_fork <variant> <nested> <memset>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(err_)); exit(2); } } while (0)
__global__ void add(float* p, float v, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += v; }
int main(int argc, char** argv) {
  int variant = argc > 1 ? atoi(argv[1]) : 0, nested = argc > 2 ? atoi(argv[2]) : 0, use_memset = argc > 3 ? atoi(argv[3]) : 0;
  const int iters = argc > 4 ? atoi(argv[4]) : 20;
  const int n = 1 << 16;
  float *a, *b, *c;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
  hipStream_t s, aux, aux2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&aux2, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(4096);
  for (size_t i = 0; i < ev.size(); ++i) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev[i] = e; }
  int ei = 0;
  auto nextev = [&]() { return variant == 0 ? ev[(ei++) % 2] : ev[ei++]; };
  auto nextev2 = [&]() { return variant == 0 ? ev[2 + (ei++) % 2] : ev[ei++]; };
  hipGraph_t g; hipGraphExec_t ge;
  hipStream_t aux3; CK(hipStreamCreateWithFlags(&aux3, hipStreamNonBlocking));
  const int warm = argc > 5 ? atoi(argv[5]) : 0, sib = argc > 6 ? atoi(argv[6]) : 0;
  auto body = [&]() {
    ei = 0;
    hipEvent_t fs = ev[4000], js = ev[4001];
    if (sib) { CK(hipEventRecord(fs, s)); CK(hipStreamWaitEvent(aux3, fs, 0)); hipLaunchKernelGGL(add, dim3(n / 256), dim3(256), 0, aux3, c, 0.f, n); }
    for (int it = 0; it < iters; ++it) {
      hipLaunchKernelGGL(add, dim3(n / 256), dim3(256), 0, s, a, 1.f, n);
      hipEvent_t f = nextev();
      CK(hipEventRecord(f, s)); CK(hipStreamWaitEvent(aux, f, 0));
      if (use_memset) CK(hipMemsetAsync(b, 0, 16, aux));
      hipLaunchKernelGGL(add, dim3(n / 256), dim3(256), 0, aux, b, 1.f, n);
      if (nested) {
        hipEvent_t f2 = nextev2();
        CK(hipEventRecord(f2, aux)); CK(hipStreamWaitEvent(aux2, f2, 0));
        hipLaunchKernelGGL(add, dim3(n / 256), dim3(256), 0, aux2, c, 1.f, n);
        hipEvent_t j2 = nextev2();
        CK(hipEventRecord(j2, aux2)); CK(hipStreamWaitEvent(aux, j2, 0));
      }
      hipLaunchKernelGGL(add, dim3(n / 256), dim3(256), 0, s, a, 1.f, n);
      hipEvent_t j = nextev();
      CK(hipEventRecord(j, aux)); CK(hipStreamWaitEvent(s, j, 0));
    }

    if (sib) { CK(hipEventRecord(js, aux3)); CK(hipStreamWaitEvent(s, js, 0)); }
  };
  if (warm) { body(); CK(hipStreamSynchronize(s)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4)); }
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  float ha, hb, hc;
  CK(hipMemcpy(&ha, a + 100, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b + 100, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, c + 100, 4, hipMemcpyDeviceToHost));
  printf("variant %d nested %d memset %d: a=%g (want 120) b=%g (want 60) c=%g (want %d) OK\n", variant, nested, use_memset, ha, hb, hc, nested ? 60 : 0);
  return 0;
}
