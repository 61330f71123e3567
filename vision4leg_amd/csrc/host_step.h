// Host side of one env step of the collector (torchrl/collector/on_policy.py:90-100: `torch.Tensor(self.current_ob).to(device)`,
// pf.explore, vf, `.cpu().numpy()`), as ONE C call with no interpreter in the loop (round 6):
//
//   float64 rows [E][S + C*H*W] (what the env wrappers hand over)
//     -> cast_rows: fp32 proprio block | 16-bit depth block in pinned memory, on a persistent pool of worker threads (AVX-512:
//        vcvtpd2ps, then vcvtps2ph for half / an integer round-to-nearest-even for bfloat16 — the same two-step rounding
//        torch's float64 -> float16 / bfloat16 copy performs, i.e. torch.Tensor(ob) followed by the kernels' ingest cast)
//     -> the step's two launches, reading the pinned rows in place (v4l_actor_step_split)
//     -> wait until the [E][A] action AND the [E] value have arrived in pinned memory (both armed with NaN before the launch;
//        every block writes its output last, so the observation buffers are free when the call returns)
//
// The pool: workers spin on a generation word for SPIN_NS after their last job, then block on a condition variable — a
// simulator that steps for milliseconds gets its cores back, a fast one never pays a wake-up.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace v4l {
namespace host {

enum { CAST_BF16 = 1, CAST_F16 = 2 };

static inline void cpu_relax() {
#if defined(__x86_64__)
  _mm_pause();
#endif
}

// ---- scalar reference of the two-step rounding (also the tail / non-AVX-512 path)
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // NaN stays NaN (quiet)
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline uint16_t f32_to_f16_rne(float f) {
  const _Float16 h = (_Float16)f;  // IEEE round-to-nearest-even, subnormals kept, overflow -> inf
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static void cast_span_scalar(const double* src, uint16_t* dst, int64_t n, int kind) {
  if (kind == CAST_BF16) for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_bf16_rne((float)src[i]);
  else for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_f16_rne((float)src[i]);
}
static void cast_span_f32_scalar(const double* src, float* dst, int64_t n) {
  for (int64_t i = 0; i < n; ++i) dst[i] = (float)src[i];
}

#if defined(__x86_64__)
__attribute__((target("avx512f,avx512vl,avx512bw,f16c"))) static void cast_span_avx512(const double* src, uint16_t* dst, int64_t n,
                                                                                       int kind) {
  int64_t i = 0;
  if (kind == CAST_F16) {
    for (; i + 16 <= n; i += 16) {
      const __m256 a = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)), b = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8));
      _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm256_cvtps_ph(a, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
      _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i + 8), _mm256_cvtps_ph(b, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
    }
  } else {
    const __m256i bias = _mm256_set1_epi32(0x7fff), one = _mm256_set1_epi32(1), absm = _mm256_set1_epi32(0x7fffffff),
                  inf = _mm256_set1_epi32(0x7f800000), quiet = _mm256_set1_epi32(0x0040);
    for (; i + 8 <= n; i += 8) {
      const __m256i u = _mm256_castps_si256(_mm512_cvtpd_ps(_mm512_loadu_pd(src + i)));
      const __m256i hi = _mm256_srli_epi32(u, 16);
      const __m256i r = _mm256_srli_epi32(_mm256_add_epi32(u, _mm256_add_epi32(bias, _mm256_and_si256(hi, one))), 16);
      const __mmask8 nan = _mm256_cmpgt_epi32_mask(_mm256_and_si256(u, absm), inf);
      const __m256i out = _mm256_mask_blend_epi32(nan, r, _mm256_or_si256(hi, quiet));
      _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm256_cvtepi32_epi16(out));
    }
  }
  if (i < n) cast_span_scalar(src + i, dst + i, n - i, kind);
}
__attribute__((target("avx512f"))) static void cast_span_f32_avx512(const double* src, float* dst, int64_t n) {
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) _mm256_storeu_ps(dst + i, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)));
  if (i < n) cast_span_f32_scalar(src + i, dst + i, n - i);
}
static bool have_avx512() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") &&
                         __builtin_cpu_supports("f16c");
  return ok;
}
#else
static bool have_avx512() { return false; }
#endif

static inline void cast_span(const double* src, uint16_t* dst, int64_t n, int kind, bool simd) {
#if defined(__x86_64__)
  if (simd) return cast_span_avx512(src, dst, n, kind);
#endif
  cast_span_scalar(src, dst, n, kind);
}
static inline void cast_span_f32(const double* src, float* dst, int64_t n, bool simd) {
#if defined(__x86_64__)
  if (simd) return cast_span_f32_avx512(src, dst, n);
#endif
  cast_span_f32_scalar(src, dst, n);
}

// ---- one cast job: rows [E][ld] float64 -> prop [E][S] fp32 (S may be 0) | img [E][img] 16-bit (or fp32 when kind == 0)
// Split into `parts` contiguous runs of (row, 2048-column chunk) units, one per participant (static: no shared counter is touched
// per unit — a first version with a shared "next unit" word scaled to 3.3 x on 8 threads and got SLOWER beyond 16); a participant
// that finishes early walks the other parts and takes any that nobody has claimed yet (a worker that was asleep).
constexpr int64_t CAST_CHUNK = 2048;
constexpr int CAST_MAX_PARTS = 64;
struct alignas(64) CastClaim { std::atomic<int> taken{0}; };
struct CastJob {
  const double* rows; int64_t ld; int E, S; int64_t img;
  float* prop; void* img_out; int kind; bool simd;
  int units, parts;
  CastClaim claim[CAST_MAX_PARTS];
  alignas(64) std::atomic<int> done{0};  // parts finished
};
static inline void cast_unit(const CastJob& j, int u) {
  const int64_t per_row = (j.img + CAST_CHUNK - 1) / CAST_CHUNK;
  const int e = (int)(u / per_row);
  const int64_t c0 = (u - (int64_t)e * per_row) * CAST_CHUNK, c1 = c0 + CAST_CHUNK < j.img ? c0 + CAST_CHUNK : j.img;
  const double* src = j.rows + (int64_t)e * j.ld;
  if (c0 == 0 && j.S > 0) cast_span_f32(src, j.prop + (int64_t)e * j.S, j.S, j.simd);
  if (j.kind == 0) cast_span_f32(src + j.S + c0, reinterpret_cast<float*>(j.img_out) + (int64_t)e * j.img + c0, c1 - c0, j.simd);
  else cast_span(src + j.S + c0, reinterpret_cast<uint16_t*>(j.img_out) + (int64_t)e * j.img + c0, c1 - c0, j.kind, j.simd);
}
// participant `id` (0 = the caller, 1.. = workers): its own part first, then whatever is still unclaimed
static inline void cast_drain(CastJob& j, int id) {
  for (int k = 0; k < j.parts; ++k) {
    const int p = (id + k) % j.parts;
    if (j.claim[p].taken.load(std::memory_order_relaxed) != 0 || j.claim[p].taken.exchange(1, std::memory_order_acq_rel) != 0) continue;
    const int u0 = (int)((int64_t)j.units * p / j.parts), u1 = (int)((int64_t)j.units * (p + 1) / j.parts);
    for (int u = u0; u < u1; ++u) cast_unit(j, u);
    j.done.fetch_add(1, std::memory_order_release);
  }
}

class CastPool {
 public:
  static constexpr int64_t SPIN_NS = 300000;  // a worker keeps spinning this long after its last job, then sleeps
  explicit CastPool(int workers) {
    for (int i = 0; i < workers; ++i) th_.emplace_back([this, i] { loop(i + 1); });
  }
  ~CastPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_.store(true, std::memory_order_release);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
  }
  int workers() const { return (int)th_.size(); }
  // runs the job on the pool's workers + the calling thread; returns when every unit is done
  void run(CastJob& j) {
    job_.store(&j, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_release);
    if (sleepers_.load(std::memory_order_acquire) > 0) {
      std::lock_guard<std::mutex> g(mu_);
      cv_.notify_all();
    }
    cast_drain(j, 0);
    while (j.done.load(std::memory_order_acquire) < j.parts) cpu_relax();
    job_.store(nullptr, std::memory_order_release);
    // (a worker that read the pointer before it was cleared finds every part claimed and touches nothing else of the job; the
    // job object must outlive that look: busy_ counts the workers inside a job)
    while (busy_.load(std::memory_order_acquire) > 0) cpu_relax();
  }

 private:
  void loop(int id) {
    uint64_t seen = 0;
    auto last = std::chrono::steady_clock::now();
    for (;;) {
      const uint64_t g = gen_.load(std::memory_order_acquire);
      if (g != seen) {
        seen = g;
        if (stop_.load(std::memory_order_acquire)) return;
        busy_.fetch_add(1, std::memory_order_acq_rel);
        CastJob* j = job_.load(std::memory_order_acquire);
        if (j != nullptr) cast_drain(*j, id);
        busy_.fetch_sub(1, std::memory_order_acq_rel);
        last = std::chrono::steady_clock::now();
        continue;
      }
      cpu_relax();
      if ((++spins_ & 1023) == 0 &&
          std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - last).count() > SPIN_NS) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<CastJob*> job_{nullptr};
  std::atomic<int> busy_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  static thread_local uint64_t spins_;
};
thread_local uint64_t CastPool::spins_ = 0;

// process-wide pool, resized on demand (threads - 1 workers: the caller works too). One job at a time: the library's contract is
// one HANDLE per host thread at a time, but the pool is shared by all handles of the process — two collectors in two threads
// take turns here (uncontended: one atomic exchange).
static std::mutex& pool_mutex() { static std::mutex mu; return mu; }
static CastPool* pool_for(int threads) {  // (call with pool_mutex() held)
  static CastPool* pool = nullptr;
  const int want = threads > 1 ? threads - 1 : 0;
  if (pool == nullptr || pool->workers() != want) {
    delete pool;
    pool = new CastPool(want);
  }
  return pool;
}

static int cast_rows(const double* rows, int64_t ld, int E, int S, int64_t img, float* prop, void* img_out, int kind, int threads) {
  CastJob j;
  j.rows = rows; j.ld = ld; j.E = E; j.S = S; j.img = img; j.prop = prop; j.img_out = img_out; j.kind = kind;
  j.simd = have_avx512();
  j.units = E * (int)((img + CAST_CHUNK - 1) / CAST_CHUNK);
  if (threads > CAST_MAX_PARTS) threads = CAST_MAX_PARTS;
  j.parts = threads < 1 ? 1 : (threads > j.units ? j.units : threads);
  if (j.parts <= 1) { cast_drain(j, 0); return 0; }
  std::lock_guard<std::mutex> g(pool_mutex());
  pool_for(j.parts)->run(j);
  return 0;
}

// all of `n` floats are numbers (the buffers were armed with NaN)?
static inline bool arrived(const volatile float* p, int n) {
  for (int i = 0; i < n; ++i) { const float x = p[i]; if (x != x) return false; }
  return true;
}

}  // namespace host
}  // namespace v4l
