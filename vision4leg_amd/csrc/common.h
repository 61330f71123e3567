// vision4leg_amd — shared device/host definitions for the gfx950 (MI355X) PPO hot path.
// Everything here is CDNA4-only: wave = 64 lanes, MFMA 16x16 tiles, LDS staging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace v4l {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// N adjacent 16-bit operand elements (H = __bf16 | _Float16) as one register vector
template <typename H, int N> struct HVec { typedef H type __attribute__((ext_vector_type(N))); };

// ---------------------------------------------------------------- error state
// The C ABI never throws; every entry returns int and leaves a message here.
void set_error(const char* fmt, ...);
const char* last_error();

#define V4L_HIP_CHECK(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      v4l::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                     hipGetErrorString(_e));                                       \
      return -2;                                                                   \
    }                                                                              \
  } while (0)

#define V4L_REQUIRE(cond, ...)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      v4l::set_error(__VA_ARGS__);                                                 \
      return -1;                                                                   \
    }                                                                              \
  } while (0)

#define V4L_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      v4l::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,             \
                     hipGetErrorString(_e));                                       \
      return -2;                                                                   \
    }                                                                              \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---------------------------------------------------------------- operand types
// The contraction operand type T is float (exact-f32 MFMA 16x16x4, the parity mode), __bf16 (MFMA 16x16x32 bf16, fp32
// accumulate) or _Float16 (MFMA 16x16x32 f16, fp32 accumulate: bf16's rate and bytes, three more significand bits; the
// backward runs on loss-gradient rows scaled by a power of two, see v4l_net_grad_scale). Kernels branch on sizeof(T) == 2.
template <typename T> struct Op;
template <> struct Op<float> {
  static __device__ __forceinline__ float from_f32(float x) { return x; }
  static __device__ __forceinline__ float to_f32(float x) { return x; }
};
template <> struct Op<__bf16> {
  static __device__ __forceinline__ __bf16 from_f32(float x) { return (__bf16)x; }  // v_cvt_pk_bf16_f32: RNE
  static __device__ __forceinline__ float to_f32(__bf16 x) { return (float)x; }
};
template <> struct Op<_Float16> {
  static __device__ __forceinline__ _Float16 from_f32(float x) { return (_Float16)x; }  // v_cvt_f16_f32: RNE, subnormals kept, overflow -> inf
  static __device__ __forceinline__ float to_f32(_Float16 x) { return (float)x; }
};
// compute-mode number of an operand type as device code sees it in `int mode` arguments (= V4L_F32 / V4L_BF16 / V4L_F16)
template <typename T> struct ModeOf;
template <> struct ModeOf<float> { static constexpr int value = 0; };
template <> struct ModeOf<__bf16> { static constexpr int value = 1; };
template <> struct ModeOf<_Float16> { static constexpr int value = 2; };

// x as the contraction sees it: rounded to the operand type (bf16: RNE; fp32: unchanged)
template <typename T> __device__ __forceinline__ float rt(float x) { return Op<T>::to_f32(Op<T>::from_f32(x)); }
template <typename T> __device__ __forceinline__ float4 rt4(float4 v) { return float4{rt<T>(v.x), rt<T>(v.y), rt<T>(v.z), rt<T>(v.w)}; }

// One K=32 step of a 16x16 output tile. Every lane holds 8 operand elements whose k index is
// 8*(lane>>4)+j for BOTH operands; row (A) / column (B) is lane&15.
//   bf16 / f16: a single v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x32_f16.
//   f32 : eight v_mfma_f32_16x16x4_f32; in the j-th one lane group g=lane>>4 supplies k=8g+j. The
//         hardware pairs A's and B's k by lane group, so the permuted k order is still a full
//         contraction over the 32 k's (exact f32 fma chain, order differs from a CPU dot only).
// C/D layout (both): col = lane&15, row = 4*(lane>>4)+r.
__device__ __forceinline__ void mma_k32(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_k32(f32x4& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
struct __attribute__((aligned(16))) f32x8 { float v[8]; };
__device__ __forceinline__ void mma_k32(f32x4& acc, const f32x8& a, const f32x8& b) {
#ifdef V4L_PROBE_F32_SPLIT3
  // probe: fp32 operands split into bf16 high + low parts, three bf16 MFMAs (the low x low product is dropped)
  bf16x8 ah, al, bh, bl;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ah[j] = (__bf16)a.v[j]; al[j] = (__bf16)(a.v[j] - (float)ah[j]);
    bh[j] = (__bf16)b.v[j]; bl[j] = (__bf16)(b.v[j] - (float)bh[j]);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
#else
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
#endif
}
template <typename T> struct Frag;
template <> struct Frag<__bf16> { typedef bf16x8 type; };
template <> struct Frag<_Float16> { typedef f16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };

// 4 consecutive elements of one row -> one 8/16-byte store (LDS or global), T = __bf16 | _Float16 | float
template <typename H>
__device__ __forceinline__ void st4(H* p, float a, float b, float c, float d) {
  static_assert(sizeof(H) == 2, "16-bit operand type");
  typename HVec<H, 4>::type v;
  v[0] = (H)a; v[1] = (H)b; v[2] = (H)c; v[3] = (H)d;
  *reinterpret_cast<typename HVec<H, 4>::type*>(p) = v;
}
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = float4{a, b, c, d};
}
// 4 consecutive elements (8/16-byte load) -> float4
template <typename H>
__device__ __forceinline__ float4 ld4(const H* p) {
  static_assert(sizeof(H) == 2, "16-bit operand type");
  const typename HVec<H, 4>::type v = *reinterpret_cast<const typename HVec<H, 4>::type*>(p);
  return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// 64-lane all-reduce. Rotations inside each 16-lane DPP row (v_add_f32_dpp row_ror:8/4/2/1: no LDS round trip, unlike
// the ds_bpermute a __shfl_xor compiles to, ~10x faster) leave every lane with its row's total; the four row totals
// are then combined through the scalar unit (v_readlane). All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0x128>(v);
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x122>(v);
  v += dpp_mov<0x121>(v);
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));
  v = fmaxf(v, dpp_mov<0x124>(v));
  v = fmaxf(v, dpp_mov<0x122>(v));
  v = fmaxf(v, dpp_mov<0x121>(v));
  return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_mov<0x128>(v));
  v = fminf(v, dpp_mov<0x124>(v));
  v = fminf(v, dpp_mov<0x122>(v));
  v = fminf(v, dpp_mov<0x121>(v));
  return fminf(fminf(lane_bcast(v, 0), lane_bcast(v, 16)), fminf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// A pointer read out of a device-side table is a generic ("flat") pointer to the compiler: its loads become flat_load,
// which also tick the LDS counter (lgkmcnt) and so serialise against every ds_read of the MFMA loop. Such pointers
// are converted to address space 1 and dereferenced AS global pointers (-> global_load / global_store).
#define V4L_GLOBAL __attribute__((address_space(1)))
template <typename P> __device__ __forceinline__ V4L_GLOBAL P* as_global(P* p) { return (V4L_GLOBAL P*)p; }

// Which descriptor of a table sorted by first block (`blk0`, d[0].blk0 == 0) owns block b. The whole wave looks at once: lane l
// tests descriptors l, l + 64, ... and a ballot counts the ones at or below b — one L2 round trip, where a binary search is
// log2(n) dependent ones at the head of every table-driven kernel. Call with all 64 lanes active.
template <class D>
__device__ __forceinline__ int find_desc(const D* __restrict__ d, int n, int64_t b) {
  const int lane = threadIdx.x & 63;
  int cnt = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool le = i < n && d[i < n ? i : 0].blk0 <= b;
    cnt += (int)__popcll(__ballot(le));
  }
  return cnt - 1;
}

// Four consecutive floats from a 4-byte aligned address as ONE global_load_dwordx4 (the hardware takes dword-aligned
// multi-dword global loads; the observation rows [S | 4x64x64] start on a 16-byte boundary only when S % 4 == 0)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld4u(const float* p) {
  const f32x4_a4 v = *reinterpret_cast<const f32x4_a4*>(p);
  return float4{v[0], v[1], v[2], v[3]};
}

}  // namespace v4l
