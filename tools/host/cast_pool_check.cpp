// Sanitizer harness for csrc/host_step.h (the collector's host-side cast pool): many casts with changing thread counts, pauses
// longer than the workers' spin window (sleep / wake path), pool resizes, ragged sizes; every result compared with the scalar
// two-step rounding. Built with -fsanitize=thread and -fsanitize=address by tools/host_step_sanitize.sh (no GPU, no HIP).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../vision4leg_amd/csrc/host_step.h"

using namespace v4l::host;

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 240;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd(0.0, 1.0);
  long long checked = 0;
  const int shapes[][3] = {{32, 93, 16384}, {5, 0, 4099}, {3, 7, 2048 * 3 + 5}, {64, 93, 1000}};
  for (int it = 0; it < iters; ++it) {
    const int* sh = shapes[it % 4];
    const int E = sh[0], S = sh[1];
    const int64_t img = sh[2], ld = S + img;
    std::vector<double> rows((size_t)E * ld);
    for (double& x : rows) x = nd(rng) * (it % 7 == 0 ? 1e-6 : 1.0);
    rows[S] = 65520.0; rows[S + 1] = 1.0 + std::ldexp(1.0, -11) + std::ldexp(1.0, -30); rows[S + 2] = NAN; rows[S + 3] = 6e-8;
    const int kind = it % 3;  // 0: fp32 image, 1: bf16, 2: f16
    const int threads = 1 + (it * 5) % 13;
    std::vector<float> prop((size_t)E * (S > 0 ? S : 1), -7.f);
    std::vector<uint16_t> out16((size_t)E * img);
    std::vector<float> out32((size_t)E * img);
    void* dst = kind == 0 ? (void*)out32.data() : (void*)out16.data();
    if (cast_rows(rows.data(), ld, E, S, img, S ? prop.data() : nullptr, dst, kind, threads) != 0) return 2;
    for (int e = 0; e < E; ++e) {
      for (int s = 0; s < S; ++s)
        if (prop[(size_t)e * S + s] != (float)rows[(size_t)e * ld + s]) { std::printf("proprio mismatch it %d\n", it); return 1; }
      for (int64_t c = 0; c < img; ++c) {
        const float f = (float)rows[(size_t)e * ld + S + c];
        if (kind == 0) {
          const float g = out32[(size_t)e * img + c];
          if (!(g == f || (f != f && g != g))) { std::printf("fp32 mismatch it %d\n", it); return 1; }
        } else {
          const uint16_t want = kind == CAST_BF16 ? f32_to_bf16_rne(f) : f32_to_f16_rne(f), got = out16[(size_t)e * img + c];
          const bool nan = f != f;
          if (!nan && want != got) { std::printf("16-bit mismatch it %d kind %d e %d c %lld: %04x vs %04x\n", it, kind, e, (long long)c, got, want); return 1; }
        }
        ++checked;
      }
    }
    if (it % 40 == 39) std::this_thread::sleep_for(std::chrono::milliseconds(2));  // longer than SPIN_NS: the workers go to sleep
  }
  // two host threads (two collectors) share the process-wide pool and ask for different thread counts: they must take turns
  {
    std::atomic<int> bad{0};
    auto caller = [&](int seed, int threads) {
      std::mt19937_64 r2(seed);
      std::normal_distribution<double> n2(0.0, 1.0);
      const int E = 8, S = 5; const int64_t img = 4099, ld = S + img;
      std::vector<double> rows((size_t)E * ld);
      std::vector<float> prop((size_t)E * S);
      std::vector<uint16_t> out((size_t)E * img);
      for (int it = 0; it < 25; ++it) {
        for (double& x : rows) x = n2(r2);
        if (cast_rows(rows.data(), ld, E, S, img, prop.data(), out.data(), CAST_F16, threads) != 0) { bad++; return; }
        for (int e = 0; e < E; ++e)
          for (int64_t c = 0; c < img; ++c)
            if (out[(size_t)e * img + c] != f32_to_f16_rne((float)rows[(size_t)e * ld + S + c])) { bad++; return; }
      }
    };
    std::thread a(caller, 11, 4), b(caller, 12, 7);
    a.join(); b.join();
    if (bad.load() != 0) { std::printf("concurrent callers: mismatch\n"); return 1; }
  }
  std::printf("cast pool check: %lld elements over %d jobs (1..13 threads, pool resized, sleep / wake), simd %d: OK\n", checked, iters, (int)have_avx512());
  return 0;
}
