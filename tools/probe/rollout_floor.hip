// What is the floor of the LocoTransformer rollout step's launch structure?
// The step is two DEPENDENT launches at E = 32 env rows (csrc/infer.h):
//   rollout_encoder2_kernel  E + E/32 = 33 blocks x 1024 threads: a block reads one observation row (66 KB fp32), the conv
//                            stack's weights (164 KB bf16) and writes a token tensor (17 x 64 fp32 = 4.3 KB);
//   rollout_stack_kernel     2 E = 64 blocks x 512 threads: a block reads one sample's tokens and ONE net's layer + head weights
//                            (401 KB bf16) through a chain of ~25 dependent phases and writes 64 bytes.
// This probe launches two kernels with exactly those grids, block sizes and dynamic LDS sizes that do NOTHING but move those bytes
// (every lane keeps D loads of 16 bytes in flight, the whole stream in one burst: the best case of the per-CU fetch path — no
// MFMA, no LDS traffic, no phase chain) and that depend on each other the way the real pair does (kernel 2 reads what kernel 1
// wrote). It reports the time per step of T eager steps on one stream, the same with the pair captured as a hipGraph, the pair
// with empty bodies (launch / drain only), and each kernel alone.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/rollout_floor.hip -o tools/probe/rollout_floor && tools/probe/rollout_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int E = 32;
constexpr int OBS_BYTES = (93 + 16384) * 4, ENC_W_BYTES = 164 * 1024, TOK_BYTES = 17 * 64 * 4, STACK_W_BYTES = 401 * 1024;

// every thread streams `bytes` of `src` (rounded to 16 B per lane) with D loads in flight, then one lane per block writes
template <int D>
__device__ __forceinline__ float stream(const uint4* __restrict__ src, int bytes, int tid, int nth) {
  const int nvec = bytes / 16;
  float acc = 0.f;
  for (int base = 0; base < nvec; base += nth * D) {
    uint4 v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int i = base + k * nth + tid;
      v[k] = src[i < nvec ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) acc += __uint_as_float(v[k].x ^ v[k].w);
  }
  return acc;
}

template <bool BODY>
__global__ __launch_bounds__(1024) void enc_like(const uint4* obs, const uint4* w, float* tokens, const float* prev_out) {
  extern __shared__ unsigned char smem[];
  if (!BODY) return;
  const int tid = threadIdx.x, b = blockIdx.x;
  float acc = prev_out[0];  // the step depends on the previous step's action (the env would have consumed it)
  if (b < E) acc += stream<4>(obs + (size_t)b * (OBS_BYTES / 16), OBS_BYTES / 16 * 16, tid, 1024);
  acc += stream<4>(w, ENC_W_BYTES, tid, 1024);
  if (tid < TOK_BYTES / 4 && b < E) tokens[b * (TOK_BYTES / 4) + tid] = acc;
}

template <bool BODY>
__global__ __launch_bounds__(512) void stack_like(const float* tokens, const uint4* w_pf, const uint4* w_vf, float* out) {
  extern __shared__ unsigned char smem[];
  if (!BODY) return;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int smp = b >> 1, net = b & 1;
  float acc = tokens[smp * (TOK_BYTES / 4) + (tid % (TOK_BYTES / 4))];
  acc += stream<8>(net ? w_vf : w_pf, STACK_W_BYTES, tid, 512);
  if (tid < 16) out[b * 16 + tid] = acc;
}

template <bool BODY>
static void step(hipStream_t s, const uint4* obs, const uint4* wenc, float* tokens, const uint4* wpf, const uint4* wvf, float* out, int t) {
  hipLaunchKernelGGL(enc_like<BODY>, dim3(E + E / 32), dim3(1024), 100 * 1024, s, obs + (size_t)(t % 64) * E * (OBS_BYTES / 16), wenc, tokens, out);
  hipLaunchKernelGGL(stack_like<BODY>, dim3(2 * E), dim3(512), 60 * 1024, s, tokens, wpf, wvf, out);
}

template <class F> static double time_us(hipStream_t s, int T, F f) {
  for (int t = 0; t < 32; ++t) f(t);
  CHECK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < T; ++t) f(t);
  CHECK(hipStreamSynchronize(s));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / T;
}

int main() {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  uint4 *obs, *wenc, *wpf, *wvf;
  float *tokens, *out;
  CHECK(hipMalloc(&obs, (size_t)64 * E * OBS_BYTES));
  CHECK(hipMalloc(&wenc, ENC_W_BYTES)); CHECK(hipMalloc(&wpf, STACK_W_BYTES)); CHECK(hipMalloc(&wvf, STACK_W_BYTES));
  CHECK(hipMalloc(&tokens, E * TOK_BYTES)); CHECK(hipMalloc(&out, 2 * E * 64));
  CHECK(hipMemset(obs, 0, (size_t)64 * E * OBS_BYTES)); CHECK(hipMemset(wenc, 0, ENC_W_BYTES));
  CHECK(hipMemset(wpf, 0, STACK_W_BYTES)); CHECK(hipMemset(wvf, 0, STACK_W_BYTES)); CHECK(hipMemset(out, 0, 2 * E * 64));
  CHECK(hipFuncSetAttribute((const void*)enc_like<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CHECK(hipFuncSetAttribute((const void*)enc_like<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const int T = 2048;
  const double pair = time_us(s, T, [&](int t) { step<true>(s, obs, wenc, tokens, wpf, wvf, out, t); });
  const double empty = time_us(s, T, [&](int t) { step<false>(s, obs, wenc, tokens, wpf, wvf, out, t); });
  const double k1 = time_us(s, T, [&](int t) {
    hipLaunchKernelGGL(enc_like<true>, dim3(E + E / 32), dim3(1024), 100 * 1024, s, obs + (size_t)(t % 64) * E * (OBS_BYTES / 16), wenc, tokens, out); });
  const double k2 = time_us(s, T, [&](int t) { hipLaunchKernelGGL(stack_like<true>, dim3(2 * E), dim3(512), 60 * 1024, s, tokens, wpf, wvf, out); });
  // the pair as a graph, replayed
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  step<true>(s, obs, wenc, tokens, wpf, wvf, out, 0);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  const double graph = time_us(s, T, [&](int) { (void)hipGraphLaunch(ge, s); });
  // host-side issue cost of the two launches (no synchronisation inside the loop: what the CPU pays per step)
  printf("rollout step floor probe (E = %d): bytes per step: obs %d KB x %d + encoder weights %d KB x %d blocks, tokens %d B x %d,\n"
         "  stack weights %d KB x %d blocks\n", E, OBS_BYTES / 1024, E, ENC_W_BYTES / 1024, E + E / 32, TOK_BYTES, E, STACK_W_BYTES / 1024, 2 * E);
  printf("  two dependent launches, bodies only move the step's bytes : %6.2f us per step (eager, %d steps back to back)\n", pair, T);
  printf("  the same pair as one hipGraph replay                      : %6.2f us per step\n", graph);
  printf("  two dependent EMPTY launches (same grids / LDS)           : %6.2f us per step\n", empty);
  printf("  encoder-like launch alone                                 : %6.2f us\n", k1);
  printf("  stack-like launch alone                                   : %6.2f us\n", k2);
  return 0;
}
