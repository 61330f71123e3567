// MFMA GEMM cores for the gfx950 PPO hot path.
//
//   gemm_nt : C[m][n] = epi( sum_k A(m,k) * Bp[n][k] )        forward linears/convs and data-grads
//   gemm_tn : dW[n][k]  = sum_m Y(m,n) * X(m,k)               weight-grads (split over m: one partial slab per split,
//                                                               summed in a fixed order by wgrad_reduce_kernel — no atomics)
//
// A / X / Y are *loader functors*: a row context (all integer divisions hoisted out of the k loop) plus
// a "give me 8 consecutive k as fp32" call. That is what turns one MFMA core into dense linear,
// implicit-im2col convolution (CHW depth stack and NHWC feature maps) and gather-form conv data-grad
// without ever materialising an im2col matrix in HBM. Operands are rounded to T (bf16 or exact f32)
// when they are staged into LDS; accumulation is always fp32 in the MFMA accumulators.
//
// Tiling: 256 threads = 4 waves. gemm_nt: 128(M) x BN(N) x 64(K) per stage, wave w owns rows
// [32w,32w+32). gemm_tn: BN(n) x 64(k) output per block, 64 rows of the reduction (m) per stage, both
// operands are transposed on their way into LDS so the MFMA fragments are contiguous 16-byte reads.
// Global loads for stage t+1 are issued before the MFMAs of stage t (register prefetch).
#pragma once
#include "common.h"

namespace v4l {

struct RowCtx {
  int64_t base;   // element offset of the row's first element
  int a, b;       // loader specific (e.g. output pixel coordinates)
  int valid;
};

__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 x = *reinterpret_cast<const float4*>(p);
  const float4 y = *reinterpret_cast<const float4*>(p + 4);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
}
// 8 16-bit operands that are only guaranteed 8-byte aligned (conv1 windows start at 4*ox elements)
template <typename H>
__device__ __forceinline__ void ld8(const H* p, float (&v)[8]) {
  static_assert(sizeof(H) == 2, "16-bit operand type");
  typedef typename HVec<H, 4>::type bf16x4;
  const bf16x4 x = *reinterpret_cast<const bf16x4*>(p);
  const bf16x4 y = *reinterpret_cast<const bf16x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = (float)x[j]; v[4 + j] = (float)y[j]; }
}

// W (1, 2 or 4) adjacent elements with one W*sizeof(T)-byte load (address aligned to that size)
template <int W> __device__ __forceinline__ void ldv(const float* p, float (&v)[W]) {
  if constexpr (W == 1) v[0] = p[0];
  else if constexpr (W == 2) { const float2 x = *reinterpret_cast<const float2*>(p); v[0] = x.x; v[1] = x.y; }
  else { const float4 x = *reinterpret_cast<const float4*>(p); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
}
template <int W, typename H> __device__ __forceinline__ void ldv(const H* p, float (&v)[W]) {
  static_assert(sizeof(H) == 2, "16-bit operand type");
  if constexpr (W == 1) v[0] = (float)p[0];
  else {
    typedef typename HVec<H, W>::type bv_t;
    const bv_t x = *reinterpret_cast<const bv_t*>(p);
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = (float)x[i];
  }
}

// hipcc turns "cond ? load : 0" into a branch around the load plus an s_waitcnt vmcnt(0) per element, which
// serialises every load of an unrolled batch behind the previous one. All loaders therefore ALWAYS load (from
// offset 0 of their array when the element is out of range) and select afterwards.
__device__ __forceinline__ void keep8(float (&v)[8], bool ok) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;
}

// ------------------------------------------------------------------------------------ loaders
// Dense fp32 row-major matrix. lda % 4 == 0, K % 8 == 0, base 16-byte aligned.
// rowidx: optional gather (minibatch row -> rollout slot). tokmap: 1 = rows are the 16 depth tokens
// of each sample inside a [B,17,D] token tensor (m -> (m/16)*17 + 1 + m%16).
struct ADense {
  const float* p;
  int lda, M, K;
  const int* rowidx;
  int tokmap;
  const float* mask;  // optional, same indexing as p: value passes only where mask > 0 (ReLU backward)
  __device__ __forceinline__ RowCtx row(int m) const {
    RowCtx rc;
    rc.valid = m < M;
    rc.a = rc.b = 0;
    int r = m;
    if (tokmap == 1) r = (m >> 4) * 17 + 1 + (m & 15);
    if (rowidx != nullptr) r = rowidx[rc.valid ? m : 0];
    rc.base = (int64_t)r * lda;
    return rc;
  }
  __device__ __forceinline__ void load(const RowCtx& rc, int k0, float (&v)[8]) const {
    const bool ok = rc.valid && k0 < K;
    const int64_t off = ok ? rc.base + k0 : 0;
    ld8(p + off, v);
    if (mask != nullptr) {
      float mk[8];
      ld8(mask + off, mk);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mk[j] > 0.f ? v[j] : 0.f;
    }
    keep8(v, ok);
  }
  // separable form used by the weight-grad kernel: address = row_off(m) + col_off(k). RowIt walks consecutive rows
  // without re-deriving (sample, pixel) coordinates: the weight-grad kernel visits 16 consecutive rows per stage and
  // the integer divisions of row() would otherwise dominate its scalar-unit time.
  struct RowIt { int m; };
  __device__ __forceinline__ RowIt iter(int m) const { return RowIt{m}; }
  __device__ __forceinline__ int64_t off(const RowIt& it, int& valid) const { return row_off(it.m, valid); }
  __device__ __forceinline__ void next(RowIt& it) const { ++it.m; }
  __device__ __forceinline__ int64_t row_off(int m, int& valid) const { RowCtx rc = row(m); valid = rc.valid; return rc.base; }
  __device__ __forceinline__ int64_t col_off(int k, int& valid) const { valid = k < K; return k; }
  __device__ __forceinline__ float get(int64_t off) const {
    const float v = p[off];
    if (mask == nullptr) return v;  // uniform branch
    const float mk = mask[off];
    return mk > 0.f ? v : 0.f;
  }
  template <int W> __device__ __forceinline__ void getv(int64_t off, float (&v)[W]) const {
    if (mask == nullptr) ldv<W>(p + off, v);
    else {
#pragma unroll
      for (int i = 0; i < W; ++i) v[i] = get(off + i);
    }
  }
};

// conv1: 8x8 stride-4 windows over the [n][C][IH][IW] depth stack (reference layout of the image
// part of an observation row, torchrl/networks/nets.py:997-1000). k = (c, ky, kx) = PyTorch's weight
// order, one chunk = the 8 kx of one (c,ky). SrcT is float (exact mode) or __bf16 (bf16 mode: the
// image is stored rounded once at ingest, which is the same rounding the contraction would apply).
template <typename SrcT>
struct AIm2colCHW {
  const SrcT* p;
  int C, IH, IW, OH, OW, stride;  // KH = KW = 8
  int M;                          // n * OH * OW
  const int* rowidx;              // per-sample gather
  __device__ __forceinline__ RowCtx row(int m) const {
    RowCtx rc;
    rc.valid = m < M;
    const int opix = OH * OW;
    int b = m / opix, q = m - b * opix;
    int oy = q / OW, ox = q - oy * OW;
    if (rowidx != nullptr) b = rowidx[rc.valid ? b : 0];
    rc.a = oy; rc.b = ox;
    rc.base = (int64_t)b * C * IH * IW + (int64_t)(oy * stride) * IW + ox * stride;
    return rc;
  }
  __device__ __forceinline__ void load(const RowCtx& rc, int k0, float (&v)[8]) const {
    const int c = k0 >> 6, ky = (k0 >> 3) & 7;
    const bool ok = rc.valid && c < C;
    ld8(p + (ok ? rc.base + (int64_t)c * IH * IW + ky * IW : 0), v);
    keep8(v, ok);
  }
  __device__ __forceinline__ int64_t row_off(int m, int& valid) const { RowCtx rc = row(m); valid = rc.valid; return rc.base; }
  struct RowIt { int m, b, oy, ox; };
  __device__ __forceinline__ RowIt iter(int m) const {
    const int opix = OH * OW;
    const int b = m / opix, q = m - b * opix;
    const int oy = q / OW;
    return RowIt{m, b, oy, q - oy * OW};
  }
  __device__ __forceinline__ int64_t off(const RowIt& it, int& valid) const {
    valid = it.m < M;
    const int b = rowidx != nullptr ? rowidx[valid ? it.b : 0] : it.b;
    return (int64_t)b * C * IH * IW + (int64_t)(it.oy * stride) * IW + it.ox * stride;
  }
  __device__ __forceinline__ void next(RowIt& it) const {
    ++it.m;
    if (++it.ox == OW) { it.ox = 0; if (++it.oy == OH) { it.oy = 0; ++it.b; } }
  }
  __device__ __forceinline__ int64_t col_off(int k, int& valid) const {
    const int c = k >> 6, ky = (k >> 3) & 7, kx = k & 7;
    valid = c < C;
    return (int64_t)c * IH * IW + ky * IW + kx;
  }
  __device__ __forceinline__ float get(int64_t off) const { return (float)p[off]; }
  template <int W> __device__ __forceinline__ void getv(int64_t off, float (&v)[W]) const { ldv<W>(p + off, v); }
};

// conv2/conv3 forward (and the X side of their weight-grads): fp32 NHWC feature map [n][IH][IW][Cin],
// k = (ky, kx, c) so that 8 consecutive k are 8 consecutive channels (Cin % 8 == 0).
struct AIm2colNHWC {
  const float* p;
  int IH, IW, Cin, OH, OW, KH, KW, stride;
  int M, K;  // n*OH*OW, KH*KW*Cin
  __device__ __forceinline__ RowCtx row(int m) const {
    RowCtx rc;
    rc.valid = m < M;
    const int opix = OH * OW;
    int b = m / opix, q = m - b * opix;
    int oy = q / OW, ox = q - oy * OW;
    rc.a = oy; rc.b = ox;
    rc.base = (((int64_t)b * IH + oy * stride) * IW + ox * stride) * Cin;
    return rc;
  }
  __device__ __forceinline__ void load(const RowCtx& rc, int k0, float (&v)[8]) const {
    const bool ok = rc.valid && k0 < K;
    const int tap = k0 / Cin, c0 = k0 - tap * Cin;
    const int ky = tap / KW, kx = tap - ky * KW;
    ld8(p + (ok ? rc.base + (int64_t)(ky * IW + kx) * Cin + c0 : 0), v);
    keep8(v, ok);
  }
  __device__ __forceinline__ int64_t row_off(int m, int& valid) const { RowCtx rc = row(m); valid = rc.valid; return rc.base; }
  struct RowIt { int m, b, oy, ox; };
  __device__ __forceinline__ RowIt iter(int m) const {
    const int opix = OH * OW;
    const int b = m / opix, q = m - b * opix;
    const int oy = q / OW;
    return RowIt{m, b, oy, q - oy * OW};
  }
  __device__ __forceinline__ int64_t off(const RowIt& it, int& valid) const {
    valid = it.m < M;
    return (((int64_t)it.b * IH + it.oy * stride) * IW + it.ox * stride) * Cin;
  }
  __device__ __forceinline__ void next(RowIt& it) const {
    ++it.m;
    if (++it.ox == OW) { it.ox = 0; if (++it.oy == OH) { it.oy = 0; ++it.b; } }
  }
  __device__ __forceinline__ int64_t col_off(int k, int& valid) const {
    valid = k < K;
    const int tap = k / Cin, c = k - tap * Cin;
    const int ky = tap / KW, kx = tap - ky * KW;
    return (int64_t)(ky * IW + kx) * Cin + c;
  }
  __device__ __forceinline__ float get(int64_t off) const { return p[off]; }
  template <int W> __device__ __forceinline__ void getv(int64_t off, float (&v)[W]) const { ldv<W>(p + off, v); }
};

// Gather-form convolution data-grad for one stride-parity class (py,px) of input pixels:
//   dX[b,iy,ix,c] = sum_{a,bb,n} dY[b, jy-a, jx-bb, n] * W[n,c,py+s*a,px+s*bb],  iy = py+s*jy, ix = px+s*jx
// rows m enumerate (b, jy, jx); k = (a, bb, n). Needs KH % s == 0 (4/2 and 3/1 in NatureCNN).
struct ADgradNHWC {
  const float* p;  // dY [n][OH][OW][Cout]
  int OH, OW, Cout, TH, TW;  // TH = KH/s taps per axis
  int nIy, nIx;              // pixels of this class per axis
  int M, K;                  // n*nIy*nIx, TH*TW*Cout
  __device__ __forceinline__ RowCtx row(int m) const {
    RowCtx rc;
    rc.valid = m < M;
    const int npix = nIy * nIx;
    int b = m / npix, q = m - b * npix;
    int jy = q / nIx, jx = q - jy * nIx;
    rc.a = jy; rc.b = jx;
    rc.base = (int64_t)b * OH * OW * Cout;
    return rc;
  }
  __device__ __forceinline__ void load(const RowCtx& rc, int k0, float (&v)[8]) const {
    const int tap = k0 / Cout, n0 = k0 - tap * Cout;
    const int a = tap / TW, bb = tap - a * TW;
    const int oy = rc.a - a, ox = rc.b - bb;
    const bool ok = rc.valid && k0 < K && oy >= 0 && oy < OH && ox >= 0 && ox < OW;
    ld8(p + (ok ? rc.base + (int64_t)(oy * OW + ox) * Cout + n0 : 0), v);
    keep8(v, ok);
  }
};

// ------------------------------------------------------------------------------------ epilogue
enum { ROWMAP_IDENT = 0, ROWMAP_TOK_DEPTH = 1, ROWMAP_TOK_STATE = 2, ROWMAP_DGRAD = 3 };

struct Epi {
  float* C;
  int ldc, M, N;
  const float* bias;  // [N] or null
  int relu;           // max(x,0) after bias
  const float* mask;  // null, or multiply by (mask[orow*ldmask+n] > 0): ReLU backward of the layer below
  int ldmask;
  int accumulate;     // C += instead of C =   (shorthand for addend == C)
  const float* addend; // optional: C = result + addend[same row/col layout as C]
  int rowmap;
  int npad;           // > N: columns [N, npad) of every output row are written as zeros (the heads' padded [OUT_LD] rows)
  // ROWMAP_DGRAD: class (py,px), stride s, class pixel counts, input plane
  int py, px, s, nIy, nIx, IH, IW;
  __device__ __forceinline__ int64_t out_row(int m) const {
    switch (rowmap) {
      case ROWMAP_TOK_DEPTH: return (int64_t)(m >> 4) * 17 + 1 + (m & 15);
      case ROWMAP_TOK_STATE: return (int64_t)m * 17;
      case ROWMAP_DGRAD: {
        const int npix = nIy * nIx;
        int b = m / npix, q = m - b * npix;
        int jy = q / nIx, jx = q - jy * nIx;
        return ((int64_t)b * IH + (py + s * jy)) * IW + (px + s * jx);
      }
      default: return m;
    }
  }
};

template <typename T> struct Tile {
  static constexpr int BK = 64;
  static constexpr int LD = BK + (sizeof(T) == 2 ? 8 : 4);  // +16 bytes of row padding
};

template <typename T>
__device__ __forceinline__ void st8(T* dst, const float (&v)[8]) {
  typename Frag<T>::type f;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (T)v[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = v[j];
  }
  *reinterpret_cast<typename Frag<T>::type*>(dst) = f;
}

// ------------------------------------------------------------------------------------ gemm_nt
template <typename T, int BN, class AL>
__device__ __forceinline__ void nt_body(const AL& al, const T* __restrict__ Bp, int Kp, const Epi& ep, int bx, int by) {
  constexpr int BM = 128, BK = Tile<T>::BK, LD = Tile<T>::LD;
  constexpr int NT = BN / 16;
  constexpr int BCH = (BN * 8 + 255) / 256;  // B chunks per thread
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) T sA[BM * LD];
  __shared__ __attribute__((aligned(16))) T sB[BN * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = bx * BM, n0 = by * BN;

  // staging ownership: A chunk q = tid + 256*i -> row q>>3, k-chunk q&7
  RowCtx rc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rc[i] = al.row(m0 + ((tid + 256 * i) >> 3));
  const int akc = (tid & 7) * 8;

  float ra[4][8];
  frag_t rb[BCH];

  auto gload = [&](int kt) {
    const int kb = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) al.load(rc[i], kb + akc, ra[i]);
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int q = tid + 256 * i;
      if (q < BN * 8) {
        const int r = q >> 3, c = (q & 7) * 8;
        rb[i] = *reinterpret_cast<const frag_t*>(Bp + (int64_t)(n0 + r) * Kp + kb + c);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) st8<T>(&sA[((tid + 256 * i) >> 3) * LD + akc], ra[i]);
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int q = tid + 256 * i;
      if (q < BN * 8) *reinterpret_cast<frag_t*>(&sB[(q >> 3) * LD + (q & 7) * 8]) = rb[i];
    }
  };

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = Kp / BK;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      frag_t fa[2], fb[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = *reinterpret_cast<const frag_t*>(&sA[(wave * 32 + i * 16 + fr) * LD + ks * 32 + fg]);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j] = *reinterpret_cast<const frag_t*>(&sB[(j * 16 + fr) * LD + ks * 32 + fg]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_k32(acc[i][j], fb[j], fa[i]);  // transposed tile: see the epilogue
    }
    __syncthreads();
  }

  // epilogue. The MFMAs above were issued with the weight fragment as the "row" operand, so the accumulator tile is
  // transposed: acc[i][j][r] = C[m][n4 + r] with m = m0 + 32*wave + 16*i + (lane&15), n4 = n0 + 16*j + 4*(lane>>4):
  // every lane owns 4 consecutive columns of ONE output row -> one 16-byte load/store per (i,j) for bias, ReLU mask,
  // accumulate target and result, and one row-map evaluation per i. Loads are issued from always-valid addresses and
  // only the stores are predicated (see keep8 for why). vec: all row strides / N are multiples of 4.
  const bool vec = ((ep.N | ep.ldc) & 3) == 0 && (ep.mask == nullptr || (ep.ldmask & 3) == 0);
  const int ng = (lane >> 4) * 4;
  const float* add = ep.accumulate ? ep.C : ep.addend;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wave * 32 + i * 16 + (lane & 15);
    const bool rok = m < ep.M;
    const int64_t orow = ep.out_row(rok ? m : 0);
    if (vec) {
      float4 bv[NT], mk[NT], old[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = n0 + j * 16 + ng;
        const bool ok = rok && n4 < ep.N;
        bv[j] = ep.bias != nullptr ? *reinterpret_cast<const float4*>(ep.bias + (n4 < ep.N ? n4 : 0)) : float4{0.f, 0.f, 0.f, 0.f};
        if (ep.mask != nullptr) mk[j] = *reinterpret_cast<const float4*>(ep.mask + (ok ? orow * ep.ldmask + n4 : 0));
        if (add != nullptr) old[j] = *reinterpret_cast<const float4*>(add + (ok ? orow * ep.ldc + n4 : 0));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n4 = n0 + j * 16 + ng;
        float4 v = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
        if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (ep.mask != nullptr) {
          v.x = mk[j].x > 0.f ? v.x : 0.f; v.y = mk[j].y > 0.f ? v.y : 0.f;
          v.z = mk[j].z > 0.f ? v.z : 0.f; v.w = mk[j].w > 0.f ? v.w : 0.f;
        }
        if (add != nullptr) { v.x += old[j].x; v.y += old[j].y; v.z += old[j].z; v.w += old[j].w; }
        if (rok && n4 < ep.N) *reinterpret_cast<float4*>(ep.C + orow * ep.ldc + n4) = v;
        else if (rok && n4 < ep.npad) *reinterpret_cast<float4*>(ep.C + orow * ep.ldc + n4) = float4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + j * 16 + ng + r;
          const bool ok = rok && n < ep.N;
          float v = acc[i][j][r];
          if (ep.bias != nullptr) v += ep.bias[n < ep.N ? n : 0];
          if (ep.relu) v = fmaxf(v, 0.f);
          if (ep.mask != nullptr) v = ep.mask[ok ? orow * ep.ldmask + n : 0] > 0.f ? v : 0.f;
          if (add != nullptr) v += add[ok ? orow * ep.ldc + n : 0];
          if (ok) ep.C[orow * ep.ldc + n] = v;
          else if (rok && n < ep.npad) ep.C[orow * ep.ldc + n] = 0.f;
        }
    }
  }
}

template <typename T, int BN, class AL>
__global__ __launch_bounds__(256) void gemm_nt_kernel(AL al, const T* __restrict__ Bp, int Kp, Epi ep) {
  nt_body<T, BN, AL>(al, Bp, Kp, ep, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------ gemm_nt, skinny M
// C = epi(A * W^T) for the dense layers of a minibatch (M ~ 1 K rows, N <= 1 K, K <= 1 K): gemm_nt_kernel's 128 x 64 tiles
// make 32 blocks of such a problem and walk K one 64-wide stage at a time with ONE stage in flight — 16 exposed round trips
// for K = 1024 (27 us). Here a block is 32 rows x 64 columns (M/32 x N/64 blocks: 128 for the NatureCNN projector), wave w
// owns column tile w for both row tiles, the weight arrives in fragment order (PK_FRAG / PK_FRAGT: one contiguous 1 KB read per
// fragment, straight into registers, no LDS), and EIGHT K-stages of A rows and weight fragments are in flight per thread.
template <typename T, class AL>
__global__ __launch_bounds__(256) void gemm_nt_deep_kernel(AL al, const T* __restrict__ Bf, int Np, int Kp, Epi ep) {
  constexpr int BM = 32, BK = Tile<T>::BK, LD = Tile<T>::LD, D = 8;
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) T sA[2][BM * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  const int m0 = blockIdx.x * BM, tile = blockIdx.y * 4 + wave;
  const bool live = tile * 16 < Np;  // (wave-uniform) a column tile past the padded width: stages A, multiplies nothing
  const int KS = Kp >> 5, nkt = Kp / BK;
  const RowCtx rc = al.row(m0 + (tid >> 3));
  const int akc = (tid & 7) * 8;
  const frag_t* Bw = reinterpret_cast<const frag_t*>(Bf) + (size_t)(live ? tile : 0) * KS * 64 + lane;
  float ra[D][8];
  frag_t rb[D][2];
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nkt) {
      al.load(rc, d * BK + akc, ra[d]);
      rb[d][0] = Bw[(d * 2) * 64];
      rb[d][1] = Bw[(d * 2 + 1) * 64];
    }
  for (int kt0 = 0; kt0 < nkt; kt0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kt = kt0 + d;
      if (kt < nkt) {  // (uniform)
        T* buf = sA[d & 1];  // D is even: stage kt uses buffer kt & 1
        st8<T>(buf + (tid >> 3) * LD + akc, ra[d]);
        __syncthreads();  // one barrier per stage: the other buffer is only rewritten after the NEXT barrier
        if (live) {
#pragma unroll
          for (int ks = 0; ks < BK / 32; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              mma_k32(acc[i], rb[d][ks], *reinterpret_cast<const frag_t*>(buf + (i * 16 + fr) * LD + ks * 32 + fg));
        }
        if (kt + D < nkt) {
          al.load(rc, (kt + D) * BK + akc, ra[d]);
          rb[d][0] = Bw[((kt + D) * 2) * 64];
          rb[d][1] = Bw[((kt + D) * 2 + 1) * 64];
        }
      }
    }
  }
  if (!live) return;
  // epilogue as in nt_body: acc[i][r] = C[m0 + 16 i + (lane & 15)][16 tile + 4 (lane >> 4) + r]
  const bool vec = ((ep.N | ep.ldc) & 3) == 0 && (ep.mask == nullptr || (ep.ldmask & 3) == 0);
  const float* add = ep.accumulate ? ep.C : ep.addend;
  const int n4 = tile * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + i * 16 + fr;
    const bool rok = m < ep.M;
    const int64_t orow = ep.out_row(rok ? m : 0);
    if (vec) {
      const bool ok = rok && n4 < ep.N;
      float4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
      if (ep.bias != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(ep.bias + (n4 < ep.N ? n4 : 0));
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (ep.mask != nullptr) {
        const float4 mk = *reinterpret_cast<const float4*>(ep.mask + (ok ? orow * ep.ldmask + n4 : 0));
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      }
      if (add != nullptr) {
        const float4 old = *reinterpret_cast<const float4*>(add + (ok ? orow * ep.ldc + n4 : 0));
        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
      }
      if (ok) *reinterpret_cast<float4*>(ep.C + orow * ep.ldc + n4) = v;
      else if (rok && n4 < ep.npad) *reinterpret_cast<float4*>(ep.C + orow * ep.ldc + n4) = float4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n4 + r;
        const bool ok = rok && n < ep.N;
        float v = acc[i][r];
        if (ep.bias != nullptr) v += ep.bias[n < ep.N ? n : 0];
        if (ep.relu) v = fmaxf(v, 0.f);
        if (ep.mask != nullptr) v = ep.mask[ok ? orow * ep.ldmask + n : 0] > 0.f ? v : 0.f;
        if (add != nullptr) v += add[ok ? orow * ep.ldc + n : 0];
        if (ok) ep.C[orow * ep.ldc + n] = v;
        else if (rok && n < ep.npad) ep.C[orow * ep.ldc + n] = 0.f;
      }
    }
  }
}

// The stride-parity classes of a gather-form conv data-grad (different row counts, weight slices and output pixel
// maps, same tile shape) as ONE launch: blockIdx.z = class.
template <typename T> struct DgradClasses {
  ADgradNHWC a[4];
  const T* B[4];
  Epi ep[4];
  int Kp;
};
template <typename T, int BN>
__global__ __launch_bounds__(256) void gemm_nt_dgrad_kernel(DgradClasses<T> cls) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x * 128 >= cls.a[z].M) return;  // classes differ in size: surplus blocks of the smaller ones
  nt_body<T, BN, ADgradNHWC>(cls.a[z], cls.B[z], cls.Kp, cls.ep[z], blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------ gemm_tn
// Weight-grad: partial[z][n][k] = sum_{m in slab z} Y(m,n) * X(m,k), then wgrad_reduce_kernel sums the slabs
// (deterministic, no atomics) and scatters into the PyTorch-layout gradient.
//
// "lane = column" staging: for a row m every lane reads ONE element of Y (lane = n) and KT elements of X
// (lane = k), so a wave-level load is one coalesced 256/512-byte row segment, and each lane collects 8 consecutive
// m of its column in registers -> one 16-byte ds_write of an MFMA-ready fragment (the transpose costs no scalar
// LDS traffic). Row addresses are wave-uniform (scalar unit), column offsets are per-lane constants hoisted out of
// the m loop. Block = 4 waves, stage = 64 rows (wave w stages rows 16w..16w+15), output tile BN(n) x 64*KT(k),
// wave w owns k-tiles [w*KT, (w+1)*KT).
// LDS of one tn_body block: sY [BN][LD] | sX [64 KT][LD] | bias partials [4][64]
template <typename T, int BN, int KT> struct TnBodyLds {
  static constexpr int LD = 64 + (sizeof(T) == 2 ? 8 : 4);
  static constexpr size_t bytes = (size_t)(BN + 64 * KT) * LD * sizeof(T) + 4 * 64 * 4;
};
// DYN: the block's LDS comes from the caller (a kernel that hosts several kinds of blocks shares ONE dynamic allocation
// between them instead of adding every kind's static arrays up)
template <typename T, int BN, int KT, class YL, class XL, bool DYN = false>
__device__ __forceinline__ void tn_body(const YL& yl, const XL& xl, int M, int m_per_block, float* __restrict__ slab,
                                        float* __restrict__ bslab, int Npad, int Kpad, int bx, int by, int bz,
                                        unsigned char* dyn_smem = nullptr) {
  constexpr int BMR = 64;
  constexpr int LD = BMR + (sizeof(T) == 2 ? 8 : 4);
  constexpr int NT = BN / 16;
  constexpr int BKO = 64 * KT;
  typedef typename Frag<T>::type frag_t;
  T* sY;                  // [n][m]
  T* sX;                  // [k][m]
  float (*sBias)[64];
  if constexpr (DYN) {
    sY = reinterpret_cast<T*>(dyn_smem);
    sX = sY + BN * LD;
    sBias = reinterpret_cast<float (*)[64]>(sX + BKO * LD);
  } else {
    __shared__ __attribute__((aligned(16))) T sY_[BN * LD];
    __shared__ __attribute__((aligned(16))) T sX_[BKO * LD];
    __shared__ float sBias_[4][64];
    sY = sY_; sX = sX_; sBias = sBias_;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = bx * BKO, n0 = by * BN;
  const int mb = bz * m_per_block;
  const int me = min(M, mb + m_per_block);
  const bool do_bias = (bslab != nullptr) && (bx == 0);

  // per-lane column offsets (constant for the whole block). The KT columns of a lane are adjacent in memory for
  // every loader (dense rows, NHWC channels, the 8 kx of a CHW window row) and share validity: one vector load.
  int yok, xok;
  const int64_t yco = yl.col_off(n0 + lane, yok);
  yok = yok && (lane < BN);
  const int64_t xco = xl.col_off(k0 + lane * KT, xok);

  float yv[16], xv[KT][16];
  float bsum = 0.f;
  auto gload = [&](int ms) {
    typename YL::RowIt yit = yl.iter(ms + wave * 16);
    typename XL::RowIt xit = xl.iter(ms + wave * 16);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int m = ms + wave * 16 + j;
      int vy, vx;
      const int64_t yro = yl.off(yit, vy);
      const int64_t xro = xl.off(xit, vx);
      yl.next(yit);
      xl.next(xit);
      const bool in = m < me;
      const bool oky = in && vy && yok;
      const float ty = yl.get(oky ? yro + yco : 0);  // unconditional load, select afterwards (see keep8)
      yv[j] = oky ? ty : 0.f;
      const bool okx = in && vx && xok;
      float tx[KT];
      xl.template getv<KT>(okx ? xro + xco : 0, tx);
#pragma unroll
      for (int i = 0; i < KT; ++i) xv[i][j] = okx ? tx[i] : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float t8[8];
      if (lane < BN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { t8[j] = yv[g * 8 + j]; bsum += t8[j]; }
        st8<T>(&sY[lane * LD + wave * 16 + g * 8], t8);
      }
#pragma unroll
      for (int i = 0; i < KT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = xv[i][g * 8 + j];
        st8<T>(&sX[(lane * KT + i) * LD + wave * 16 + g * 8], t8);
      }
    }
  };

  f32x4 acc[NT][KT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = (lane >> 4) * 8;
  if (mb < me) gload(mb);
  for (int ms = mb; ms < me; ms += BMR) {
    lstore();
    __syncthreads();
    if (ms + BMR < me) gload(ms + BMR);
#pragma unroll
    for (int ks = 0; ks < BMR / 32; ++ks) {
      frag_t fx[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j)
        fx[j] = *reinterpret_cast<const frag_t*>(&sX[((wave * KT + j) * 16 + fr) * LD + ks * 32 + fg]);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const frag_t fy = *reinterpret_cast<const frag_t*>(&sY[(i * 16 + fr) * LD + ks * 32 + fg]);
#pragma unroll
        for (int j = 0; j < KT; ++j) mma_k32(acc[i][j], fx[j], fy);  // transposed tile: 4 consecutive k per lane
      }
    }
    __syncthreads();
  }

  // acc[i][j][r] = partial dW[n0 + 16*i + (lane&15)][k0 + 16*(wave*KT+j) + 4*(lane>>4) + r]: one 16-byte store each
  float* out = slab + (int64_t)bz * Npad * Kpad;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = n0 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int k4 = k0 + (wave * KT + j) * 16 + (lane >> 4) * 4;
      if (n < Npad && k4 < Kpad)
        *reinterpret_cast<float4*>(out + (int64_t)n * Kpad + k4) = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
    }
  }
  if (do_bias) {
    sBias[wave][lane] = bsum;
    __syncthreads();
    if (tid < BN && n0 + tid < Npad)
      bslab[(int64_t)bz * Npad + n0 + tid] = sBias[0][tid] + sBias[1][tid] + sBias[2][tid] + sBias[3][tid];
  }
}

template <typename T, int BN, int KT, class YL, class XL>
__global__ __launch_bounds__(256) void gemm_tn_kernel(YL yl, XL xl, int M, int m_per_block, float* __restrict__ slab,
                                                      float* __restrict__ bslab, int Npad, int Kpad) {
  tn_body<T, BN, KT, YL, XL>(yl, xl, M, m_per_block, slab, bslab, Npad, Kpad, blockIdx.x, blockIdx.y, blockIdx.z);
}

// All dense (Linear) weight-grads of a backward pass in ONE launch: they are mutually independent, individually too
// small to fill the chip (12..64 blocks each) and would otherwise serialise as ~30 dependent-in-stream kernels.
struct TnProb {
  ADense y, x;
  int M, mpb, Npad, Kpad, gx, gy;
  float* slab;
  float* bslab;
  int64_t blk0;  // first block of this problem
};
// ADense as the weight-grad staging code uses it, for descriptors that were read from a device-side table: the same
// addressing, with the pointers held as global (address space 1) pointers so that the loads are global_load
struct ADenseG {
  const V4L_GLOBAL float* p;
  const V4L_GLOBAL int* rowidx;
  const V4L_GLOBAL float* mask;
  int lda, M, K, tokmap;
  __device__ __forceinline__ explicit ADenseG(const ADense& a)
      : p(as_global(a.p)), rowidx(as_global(a.rowidx)), mask(as_global(a.mask)), lda(a.lda), M(a.M), K(a.K), tokmap(a.tokmap) {}
  struct RowIt { int m; };
  __device__ __forceinline__ RowIt iter(int m) const { return RowIt{m}; }
  __device__ __forceinline__ void next(RowIt& it) const { ++it.m; }
  __device__ __forceinline__ int64_t off(const RowIt& it, int& valid) const {
    const int m = it.m;
    valid = m < M;
    int r = m;
    if (tokmap == 1) r = (m >> 4) * 17 + 1 + (m & 15);
    if (a_rowidx()) r = rowidx[valid ? m : 0];
    return (int64_t)r * lda;
  }
  __device__ __forceinline__ bool a_rowidx() const { return rowidx != nullptr; }
  __device__ __forceinline__ int64_t col_off(int k, int& valid) const { valid = k < K; return k; }
  __device__ __forceinline__ float get(int64_t o) const {
    const float v = p[o];
    if (mask == nullptr) return v;  // uniform branch
    const float mk = mask[o];
    return mk > 0.f ? v : 0.f;
  }
  template <int W> __device__ __forceinline__ void getv(int64_t o, float (&v)[W]) const {
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = get(o + i);
  }
};
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_group_kernel(const TnProb* __restrict__ probs, int np) {
  const TnProb p = probs[find_desc(probs, np, (int64_t)blockIdx.x)];
  const int lb = (int)((int64_t)blockIdx.x - p.blk0);
  const int bx = lb % p.gx, t = lb / p.gx;
  tn_body<T, 64, 1, ADenseG, ADenseG>(ADenseG(p.y), ADenseG(p.x), p.M, p.mpb, p.slab, p.bslab, p.Npad, p.Kpad, bx, t % p.gy,
                                      t / p.gy);
}

// ------------------------------------------------------------------------------------ gemm_tn_wide
// Weight-grads of the transformer-layer linears (M = 17 tokens x batch rows, N, K in {64, 192, 256}): both operands
// are plain row-major [M][C] arrays ALREADY in the contraction type T (the fused layer kernels write them that way),
// and one block owns the WHOLE N x K output for its slab of rows -> every operand byte is fetched exactly once
// (gemm_tn_group's 64x64 tiles re-fetch X per n-tile and Y per k-tile, from another XCD's L2 more often than not).
// Staging: a lane loads 2 adjacent columns of 8 consecutive rows (4/8-byte loads, 128/256-byte row segments per
// half-wave), regroups them into two 8-row column vectors in registers and writes each with one 16/32-byte ds_write
// into the [column][row] image the MFMA fragments are read from.
struct TnWide {
  const void *y, *x;   // T [M][N], T [M][K]
  int M, mpb, N, K;
  float *slab, *bslab;  // [nsplit][N][K], [nsplit][N]
  int blk0, pad;
};
template <typename T> struct Pair;
template <> struct Pair<__bf16> { typedef __attribute__((ext_vector_type(2))) __bf16 type; };
template <> struct Pair<_Float16> { typedef __attribute__((ext_vector_type(2))) _Float16 type; };
template <> struct Pair<float> { typedef __attribute__((ext_vector_type(2))) float type; };
template <typename H>
__device__ __forceinline__ void st8raw(H* dst, const H (&v)[8]) {
  static_assert(sizeof(H) == 2, "16-bit operand type");
  typename Frag<H>::type t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = v[j];
  *reinterpret_cast<typename Frag<H>::type*>(dst) = t;
}
__device__ __forceinline__ void st8raw(float* dst, const float (&v)[8]) {
  *reinterpret_cast<float4*>(dst) = float4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<float4*>(dst + 4) = float4{v[4], v[5], v[6], v[7]};
}
template <typename T> struct TnWideLds {
  static constexpr int LD = 64 + (sizeof(T) == 2 ? 8 : 4);
  static constexpr size_t bytes(int N, int K) { return (size_t)(N + K) * LD * sizeof(T) + (size_t)8 * N * 4; }
  static constexpr size_t max_bytes = (size_t)(256 + 64) * LD * sizeof(T) + (size_t)8 * 256 * 4;
};

template <typename T, int N, int K>
__device__ __forceinline__ void tn_wide_body(const TnWide& p, int bz, unsigned char* smem) {
  constexpr int LD = TnWideLds<T>::LD;
  constexpr int NC = N / 64, KC = K / 64;
  constexpr bool SPLIT_N = N > 64;           // waves split the n-tiles (K == 64) or the k-tiles (N == 64)
  static_assert(!SPLIT_N || K == 64, "tn_wide: unsupported shape");
  constexpr int NT_W = SPLIT_N ? N / 64 : 4;
  constexpr int KT_W = SPLIT_N ? 4 : K / 64;
  typedef typename Frag<T>::type frag_t;
  typedef typename Pair<T>::type pair_t;
  T* sY = reinterpret_cast<T*>(smem);
  T* sX = sY + N * LD;
  float* sB = reinterpret_cast<float*>(sX + K * LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, cl = (lane & 31) * 2;
  const int mb = bz * p.mpb, me = min(p.M, mb + p.mpb);
  const V4L_GLOBAL T* Y = as_global(reinterpret_cast<const T*>(p.y));
  const V4L_GLOBAL T* X = as_global(reinterpret_cast<const T*>(p.x));
  typedef const V4L_GLOBAL pair_t* gpair_t;
  pair_t yv[NC][8], xv[KC][8];
  float bs[NC][2];
#pragma unroll
  for (int q = 0; q < NC; ++q) bs[q][0] = bs[q][1] = 0.f;
  auto gload = [&](int ms) {
    const int r0 = ms + wave * 16 + half * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = r0 + j < me;
      const int64_t r = ok ? r0 + j : mb;  // unconditional loads from a row of this slab, selected below
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const pair_t v = *(gpair_t)(Y + r * N + q * 64 + cl);
        yv[q][j] = ok ? v : pair_t{(T)0.f, (T)0.f};
      }
#pragma unroll
      for (int q = 0; q < KC; ++q) {
        const pair_t v = *(gpair_t)(X + r * K + q * 64 + cl);
        xv[q][j] = ok ? v : pair_t{(T)0.f, (T)0.f};
      }
    }
  };
  auto lstore = [&]() {
    const int m0 = wave * 16 + half * 8;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      T c0[8], c1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c0[j] = yv[q][j].x; c1[j] = yv[q][j].y;
        bs[q][0] += (float)c0[j]; bs[q][1] += (float)c1[j];
      }
      st8raw(&sY[(q * 64 + cl) * LD + m0], c0);
      st8raw(&sY[(q * 64 + cl + 1) * LD + m0], c1);
    }
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      T c0[8], c1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { c0[j] = xv[q][j].x; c1[j] = xv[q][j].y; }
      st8raw(&sX[(q * 64 + cl) * LD + m0], c0);
      st8raw(&sX[(q * 64 + cl + 1) * LD + m0], c1);
    }
  };
  f32x4 acc[NT_W][KT_W];
#pragma unroll
  for (int i = 0; i < NT_W; ++i)
#pragma unroll
    for (int j = 0; j < KT_W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  const int nbase = SPLIT_N ? wave * NT_W : 0, kbase = SPLIT_N ? 0 : wave * KT_W;
  if (mb < me) gload(mb);
  for (int ms = mb; ms < me; ms += 64) {
    lstore();
    __syncthreads();
    if (ms + 64 < me) gload(ms + 64);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t fx[KT_W];
#pragma unroll
      for (int j = 0; j < KT_W; ++j) fx[j] = *reinterpret_cast<const frag_t*>(&sX[((kbase + j) * 16 + fr) * LD + ks * 32 + fg]);
#pragma unroll
      for (int i = 0; i < NT_W; ++i) {
        const frag_t fy = *reinterpret_cast<const frag_t*>(&sY[((nbase + i) * 16 + fr) * LD + ks * 32 + fg]);
#pragma unroll
        for (int j = 0; j < KT_W; ++j) mma_k32(acc[i][j], fx[j], fy);  // transposed tile: 4 consecutive k per lane
      }
    }
    __syncthreads();
  }
  V4L_GLOBAL float* out = as_global(p.slab) + (int64_t)bz * N * K;
#pragma unroll
  for (int i = 0; i < NT_W; ++i) {
    const int n = (nbase + i) * 16 + fr;
#pragma unroll
    for (int j = 0; j < KT_W; ++j) {
      const int k4 = (kbase + j) * 16 + (lane >> 4) * 4;
      *(V4L_GLOBAL f32x4*)(out + (int64_t)n * K + k4) = acc[i][j];
    }
  }
  // bias grads: column sums of Y. Eight (wave, half) row groups hold partials of every column -> fixed-order sum
  const int g = wave * 2 + half;
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    sB[g * N + q * 64 + cl] = bs[q][0];
    sB[g * N + q * 64 + cl + 1] = bs[q][1];
  }
  __syncthreads();
  if (tid < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sB[k * N + tid];
    as_global(p.bslab)[(int64_t)bz * N + tid] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_wide_kernel(const TnWide* __restrict__ probs, int np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_wide_smem[];
  int pi = 0;
  for (int i = 1; i < np; ++i) pi = probs[i].blk0 <= (int)blockIdx.x ? i : pi;
  const TnWide p = probs[pi];
  const int bz = (int)blockIdx.x - p.blk0;
  if (p.N == 256 && p.K == 64) tn_wide_body<T, 256, 64>(p, bz, tn_wide_smem);
  else if (p.N == 192 && p.K == 64) tn_wide_body<T, 192, 64>(p, bz, tn_wide_smem);
  else if (p.N == 64 && p.K == 256) tn_wide_body<T, 64, 256>(p, bz, tn_wide_smem);
  else tn_wide_body<T, 64, 64>(p, bz, tn_wide_smem);
}
// The grouped dense weight-grads and the layers' whole-output weight-grads in ONE launch: blocks [0, wblocks) are
// gemm_tn_wide_kernel's, the rest gemm_tn_group_kernel's. Both kinds are short latency chains of a few hundred blocks; side by
// side they cost the longer one's time instead of the sum. One dynamic LDS allocation (the larger need) serves either kind.
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_dense_kernel(const TnWide* __restrict__ wprobs, int nw, int wblocks,
                                                            const TnProb* __restrict__ gprobs, int ng) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_wide_smem[];
  if ((int)blockIdx.x < wblocks) {
    int pi = 0;
    for (int i = 1; i < nw; ++i) pi = wprobs[i].blk0 <= (int)blockIdx.x ? i : pi;
    const TnWide p = wprobs[pi];
    const int bz = (int)blockIdx.x - p.blk0;
    if (p.N == 256 && p.K == 64) tn_wide_body<T, 256, 64>(p, bz, tn_wide_smem);
    else if (p.N == 192 && p.K == 64) tn_wide_body<T, 192, 64>(p, bz, tn_wide_smem);
    else if (p.N == 64 && p.K == 256) tn_wide_body<T, 64, 256>(p, bz, tn_wide_smem);
    else tn_wide_body<T, 64, 64>(p, bz, tn_wide_smem);
  } else {
    const int64_t b = (int64_t)blockIdx.x - wblocks;
    const TnProb p = gprobs[find_desc(gprobs, ng, b)];
    const int lb = (int)(b - p.blk0);
    const int bx = lb % p.gx, t = lb / p.gx;
    tn_body<T, 64, 1, ADenseG, ADenseG, true>(ADenseG(p.y), ADenseG(p.x), p.M, p.mpb, p.slab, p.bslab, p.Npad, p.Kpad, bx,
                                              t % p.gy, t / p.gy, tn_wide_smem);
  }
}
static inline bool tn_wide_shape(int N, int K) {
  return (N == 256 && K == 64) || (N == 192 && K == 64) || (N == 64 && K == 256) || (N == 64 && K == 64);
}

constexpr int CONV_BWD_MAX_BLOCKS = 256;  // persistent blocks of the fused conv backward (one per CU)

// Sums the per-slab partials of one or more weight tensors and writes the PyTorch-layout gradients.
// The packed k order of NHWC convs / NHWC flatten is (tap, c); PyTorch's is (c, tap): kt = (k % Cin) * taps + k / Cin.
struct RedDesc {
  const float* slab;   // [nsplit][Npad][Kpad]
  const float* bslab;  // [nsplit][Npad] or null
  float* dW;           // [N][Ktorch]
  float* db;           // [N] or null
  int nsplit, N, K, Npad, Kpad, Ktorch, Cin, taps;
  int64_t blk0;        // first block of this descriptor
  int group, pad_;     // 1: partials of the fused conv-stack backward (bwd_conv_kernel / bwd_conv3_wgrad_kernel), 0: everything else
};
// blk_base: first block of the table this launch covers — the reduction can be issued as two launches (round 4: the conv
// stack's partials on the main stream right behind dW3, the dense ones on the auxiliary stream behind the grouped weight-grads,
// both INSIDE the forked section, so that only clip_adam waits for the join); sq_part is indexed by the table-wide block number.
// unscale: 1 / v4l_net_grad_scale of the pass (1, or a negative power of two in the f16 mode: exact) — the gradients and their
// norm partials leave unscaled.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedDesc* __restrict__ descs, int nd, float* __restrict__ sq_part,
                                                           int blk_base, float unscale) {
  // 64 outputs per block x 4 slab groups: thread (o, g) adds slabs g, g+4, g+8, ... (4 independent accumulators),
  // the 4 group sums are combined through LDS in a fixed order -> deterministic and latency-tolerant
  __shared__ float part[4][64];
  const int64_t bid = (int64_t)blockIdx.x + blk_base;
  const RedDesc d = descs[find_desc(descs, nd, bid)];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t e = (bid - d.blk0) * 64 + o;
  const int64_t nk = (int64_t)d.N * d.K;
  const V4L_GLOBAL float* p = nullptr;
  int64_t stride = 0;
  int n = 0, k = 0;
  if (e < nk) {
    n = (int)(e / d.K); k = (int)(e - (int64_t)n * d.K);
    p = as_global(d.slab) + (int64_t)n * d.Kpad + k;
    stride = (int64_t)d.Npad * d.Kpad;
  } else if (d.db != nullptr && e < nk + d.N) {
    n = (int)(e - nk);
    p = as_global(d.bslab) + n;
    stride = d.Npad;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (p != nullptr) {
    int z = g;
    for (; z + 12 < d.nsplit; z += 16) {
      s0 += p[z * stride]; s1 += p[(z + 4) * stride]; s2 += p[(z + 8) * stride]; s3 += p[(z + 12) * stride];
    }
    for (; z < d.nsplit; z += 4) s0 += p[z * stride];
  }
  part[g][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0) {  // wave 0: the 64 outputs of this block; their squares are the block's share of the gradient norm
    const float sum = p != nullptr ? ((part[0][o] + part[1][o]) + (part[2][o] + part[3][o])) * unscale : 0.f;
    if (sq_part != nullptr) {
      const float sq = wave_sum(sum * sum);
      if (o == 0) sq_part[bid] = sq;
    }
    if (p == nullptr) return;
    if (e < nk) {
      int kt = k;
      if (d.Cin != 0) { const int t = k / d.Cin, c = k - t * d.Cin; kt = c * d.taps + t; }
      as_global(d.dW)[(int64_t)n * d.Ktorch + kt] = sum;
    } else {
      as_global(d.db)[n] = sum;
    }
  }
}

}  // namespace v4l
