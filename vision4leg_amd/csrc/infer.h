// Fused small-batch inference of the LocoTransformer for the rollout step (E = 16..64 envs): the whole per-step
// network evaluation in FOUR launches instead of ~50 (a dependent kernel boundary costs 3.5-5 us on MI355X, more
// than any of these layers computes at batch E):
//
//   infer_encoder_kernel : blocks 0..E-1: one sample each — split/ingest of the observation row, conv1 -> conv2 ->
//                          conv3 -> 1x1 up-conv with every activation in LDS (implicit GEMM straight from the LDS
//                          image), depth tokens out.  block E: the proprio MLP + state_projector for all E rows.
//   infer_layer_kernel   : one TransformerEncoderLayer (in_proj, 17-token attention, out_proj, +res, LN, FFN, +res,
//                          LN) for 4 samples (68 token rows = 5 MFMA row tiles) per block; blockIdx.y = net (pf, vf).
//   infer_head_kernel    : pooling + the three head linears of both nets + Gaussian sampling / value read-out,
//                          filing of action/value into the rollout arrays, advance of the device-side step cursor.
//
// Weights are read as MFMA B fragments straight from the packed operand copies in L2 (each block streams a layer's
// 25-130 KB once); activations never leave LDS inside a kernel. Rounding points are those of the unfused kernels
// (operands rounded to T when they enter an MFMA, fp32 accumulate, fp32 residual/LN/softmax), so both paths agree
// to fp32 summation noise.
#pragma once
#include "elem.h"
#include "gemm.h"

namespace v4l {

constexpr int INF_SPW = 4;                      // samples per block in the layer kernel
constexpr int INF_ROWS = 80;                    // 4*17 = 68 token rows padded to 5 MFMA row tiles
constexpr int INF_MT = INF_ROWS / 16;

template <typename T> struct InfLd {            // LDS row strides (elements) with 16-byte padding
  static constexpr int PAD = sizeof(T) == 2 ? 8 : 4;
};

// 8 consecutive k of an A row held in LDS as fp32 or as T -> MFMA fragment of T
template <typename T>
__device__ __forceinline__ typename Frag<T>::type afrag(const float* p) {
  typename Frag<T>::type f;
  const float4 x = *reinterpret_cast<const float4*>(p);
  const float4 y = *reinterpret_cast<const float4*>(p + 4);
  if constexpr (sizeof(T) == 2) {
    f[0] = (T)x.x; f[1] = (T)x.y; f[2] = (T)x.z; f[3] = (T)x.w;
    f[4] = (T)y.x; f[5] = (T)y.y; f[6] = (T)y.z; f[7] = (T)y.w;
  } else {
    f.v[0] = x.x; f.v[1] = x.y; f.v[2] = x.z; f.v[3] = x.w; f.v[4] = y.x; f.v[5] = y.y; f.v[6] = y.z; f.v[7] = y.w;
  }
  return f;
}
template <typename H>
__device__ __forceinline__ typename Frag<H>::type afrag_t(const H* p) {  // 8-byte aligned source (conv1 windows)
  static_assert(sizeof(H) == 2, "16-bit operand type");
  typedef typename HVec<H, 4>::type bf16x4;
  const bf16x4 x = *reinterpret_cast<const bf16x4*>(p);
  const bf16x4 y = *reinterpret_cast<const bf16x4*>(p + 4);
  typename Frag<H>::type f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[j] = x[j]; f[4 + j] = y[j]; }
  return f;
}
__device__ __forceinline__ f32x8 afrag_t(const float* p) { return *reinterpret_cast<const f32x8*>(p); }

// acc[mt][j] += (A[mt-th row tile] * W[ntile[j]]^T)^T over KS K=32 steps. A: LDS, row stride lda, element AT.
// The weight fragment is the MFMA "row" operand, so a lane ends up with 4 CONSECUTIVE output columns of one row:
// acc[mt][j][r] = out[16*mt + (lane&15)][16*ntile[j] + 4*(lane>>4) + r]  -> vector epilogue stores.
// The weight fragments come straight from L2/HBM: a ring of PD k-steps of them is kept in flight (a load issued per
// step consumed) so the MFMAs never wait for a just-issued global load.
// The first PD k-steps of a GEMM's weight fragments, requested ahead of time (gemm_prefetch) so that the L2 round trip —
// and, on CDNA where loads and stores retire through the same in-order vmcnt queue, the acknowledgement of every global
// store issued before it — overlaps the phase in front of the GEMM instead of opening it.
template <typename T, int NTW, int KS> struct GemmRing {
  // ring depth: every step in flight when that costs <= 16 fragment registers sets, else 4 steps
  static constexpr int PD = KS * NTW <= 16 ? KS : (KS < 4 ? KS : 4);
  typename Frag<T>::type fb[PD][NTW];
};
template <typename T, int NTW, int KS>
__device__ __forceinline__ GemmRing<T, NTW, KS> gemm_prefetch(const T* __restrict__ Wp, int Kp, const int (&ntile)[NTW], int lane) {
  typedef typename Frag<T>::type frag_t;
  GemmRing<T, NTW, KS> ring;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
#pragma unroll
  for (int d = 0; d < GemmRing<T, NTW, KS>::PD; ++d)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      ring.fb[d][j] = *reinterpret_cast<const frag_t*>(Wp + (int64_t)(ntile[j] * 16 + fr) * Kp + fg + d * 32);
  return ring;
}
template <typename T, int MT, int NTW, int KS, typename AT>
__device__ __forceinline__ void block_gemm(f32x4 (&acc)[MT][NTW], const AT* sA, int lda, const T* __restrict__ Wp, int Kp,
                                           const int (&ntile)[NTW], int lane, GemmRing<T, NTW, KS>& ring) {
  typedef typename Frag<T>::type frag_t;
  constexpr int PD = GemmRing<T, NTW, KS>::PD;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  const T* wrow[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) wrow[j] = Wp + (int64_t)(ntile[j] * 16 + fr) * Kp + fg;
  frag_t (&fb)[PD][NTW] = ring.fb;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    frag_t cur[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) cur[j] = fb[ks % PD][j];
    if (ks + PD < KS) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) fb[ks % PD][j] = *reinterpret_cast<const frag_t*>(wrow[j] + (ks + PD) * 32);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      frag_t fa;
      if constexpr (sizeof(AT) == sizeof(T)) fa = *reinterpret_cast<const frag_t*>(sA + (mt * 16 + fr) * lda + ks * 32 + fg);
      else fa = afrag<T>(reinterpret_cast<const float*>(sA) + (mt * 16 + fr) * lda + ks * 32 + fg);
#pragma unroll
      for (int j = 0; j < NTW; ++j) mma_k32(acc[mt][j], cur[j], fa);  // transposed tile
    }
  }
}
template <typename T, int MT, int NTW, int KS, typename AT>
__device__ __forceinline__ void block_gemm(f32x4 (&acc)[MT][NTW], const AT* sA, int lda, const T* __restrict__ Wp, int Kp,
                                           const int (&ntile)[NTW], int lane) {
  GemmRing<T, NTW, KS> ring = gemm_prefetch<T, NTW, KS>(Wp, Kp, ntile, lane);
  block_gemm<T, MT, NTW, KS, AT>(acc, sA, lda, Wp, Kp, ntile, lane, ring);
}
template <int MT, int NTW> __device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NTW]) {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------ attention tile (MFMA)
// softmax(Q K^T / 8) V of ONE 16-query row tile of one sample (17 tokens, one head of 64) by ONE wave, on the matrix
// cores (torchrl/networks/nets.py:948-955 builds nn.TransformerEncoderLayer(64, 1, ...); north star: "QK^T/softmax/V ... use
// MFMA bf16 tiles"). Operands in the compute type T in LDS, written once by in_proj's epilogue:
//   q0   : this tile's 16 query rows  [16][ldq]   (rows past the sample's 17th token are other rows of the block: finite,
//          their results are discarded)
//   k0   : the sample's key rows      [32][ldq]   (rows >= 17: ditto; their scores are masked to -inf)
//   vt   : the sample's values, transposed [64][LDV], key columns 17..31 ZERO (0 x NaN would poison the row)
//   pb   : wave-private scratch [16][LDV]: P in T, re-read as the A operand of P V
// Scores and softmax live in registers (a lane owns query fr, keys 4g..4g+3 and 16+4g..16+4g+3; row max / sum cross the four
// lane groups with two ds_bpermute each). nq: valid queries of the tile (16, or 1 for the tile that holds token 16).
// Results: ctx rows -> cx (T, stride ldc; only valid queries are written), optionally P (fp32 [17][17]) and ctx (T [17][64])
// of the sample to global memory for the backward pass.
constexpr int ATT_LDV = 32 + 8;
template <typename T, int NT = NTOK>
__device__ __forceinline__ void attn_tile(const T* q0, const T* k0, int ldq, const T* vt, T* pb, T* cx, int ldc, int lane,
                                          int q_first, int nq, float* __restrict__ gP, T* __restrict__ gctx) {
  typedef typename Frag<T>::type frag_t;
  const int fr = lane & 15, g = lane >> 4, fg = g * 8, qr = g * 4;
  f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const frag_t fa = *reinterpret_cast<const frag_t*>(q0 + fr * ldq + ks * 32 + fg);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const frag_t fb = *reinterpret_cast<const frag_t*>(k0 + (nt * 16 + fr) * ldq + ks * 32 + fg);
      mma_k32(s[nt], fb, fa);  // s[nt][r] = q[fr] . k[16 nt + 4g + r]
    }
  }
  float pv[2][4];
  float mx = -INFINITY;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = nt * 16 + qr + r;
      pv[nt][r] = key < NT ? s[nt][r] * 0.125f : -INFINITY;
      mx = fmaxf(mx, pv[nt][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[nt][r] = nt * 16 + qr + r < NT ? expf(pv[nt][r] - mx) : 0.f;
      sum += pv[nt][r];
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.f / sum;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[nt][r] *= inv;
      const int key = nt * 16 + qr + r;
      if (gP != nullptr && fr < nq && key < NT) gP[(q_first + fr) * NT + key] = pv[nt][r];
    }
    st4(pb + fr * ATT_LDV + nt * 16 + qr, pv[nt][0], pv[nt][1], pv[nt][2], pv[nt][3]);
  }
  __builtin_amdgcn_wave_barrier();  // pb is wave-private: LDS operations of one wave execute in order
  const frag_t fp = *reinterpret_cast<const frag_t*>(pb + fr * ATT_LDV + fg);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const frag_t fv = *reinterpret_cast<const frag_t*>(vt + (nt * 16 + fr) * ATT_LDV + fg);
    mma_k32(c, fv, fp);  // c[r] = sum_key P[fr][key] v[key][16 nt + 4g + r]
    if (fr < nq) {
      st4(cx + fr * ldc + nt * 16 + qr, c[0], c[1], c[2], c[3]);
      if (gctx != nullptr) st4(gctx + (q_first + fr) * TD + nt * 16 + qr, c[0], c[1], c[2], c[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------ encoder
struct InfEnc {
  const void *w1, *w2, *w3, *wup;          // packed conv weights (T): [32][256] [64][512] [64][576] [64][64]
  const float *b1, *b2, *b3, *bup;
  const void *wf1, *wf2, *wpr;             // packed proprio MLP: [256][Kp1] [256][256] [64][256]
  const float *bf1, *bf2, *bpr;
  int S, Sp, Kp1;                          // proprio length, rollout row stride, padded K of fc1
};

// Training forward: read the ingested rollout rows (gathered by rowidx) instead of raw observation rows, and save the
// activations the backward pass needs (fp32, the layouts of the layer-by-layer kernels). All null for inference.
struct InfEncTrain {
  const void* image;     // [slots][4*64*64] T
  const float* state;    // [slots][Sp]
  const int* rowidx;     // [n] or null
  float *s_c1, *s_c2, *s_c3;  // [n][225][32], [n][36][64], [n][16][64]
  float *s_h1, *s_h2;         // [n][256] encoder-MLP activations
};

template <typename T> struct InfEncLds {
  static constexpr int PAD = InfLd<T>::PAD;
  static constexpr int IMG = 4 * 64 * 64;
  static constexpr int LD1 = 32 + PAD, LD2 = 64 + PAD;
  static constexpr int C1 = 240 * LD1, C2 = 48 * LD2, C3 = 16 * LD2;
  static constexpr size_t conv_bytes = (size_t)(IMG + C1 + C2 + C3) * sizeof(T);
  static constexpr int LDS_IN = 128 + 4;   // fp32 proprio rows
  static constexpr int LDH = 256 + PAD;
  static constexpr int MLP_ROWS = 32;      // proprio rows per MLP block (blocks E .. E + ceil(E/32) - 1)
  static constexpr size_t mlp_bytes = (size_t)MLP_ROWS * LDS_IN * 4 + (size_t)2 * MLP_ROWS * LDH * sizeof(T);
  static constexpr size_t bytes = conv_bytes > mlp_bytes ? conv_bytes : mlp_bytes;
};

template <typename T>
__global__ __launch_bounds__(256) void infer_encoder_kernel(const ActCtl* __restrict__ ctl, const float* __restrict__ obs, int E,
                                                            InfEnc w, float* __restrict__ state_roll, T* __restrict__ image_roll,
                                                            float* __restrict__ x0 /* [E][17][64] */, InfEncTrain tr) {
  typedef typename Frag<T>::type frag_t;
  typedef InfEncLds<T> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8, qr = (lane >> 4) * 4;
  const bool train = tr.image != nullptr;
  const int64_t slot0 = train ? 0 : (int64_t)ctl->t * E;
  const int D = w.S + LY::IMG;

  if ((int)blockIdx.x >= E) {
    // ---------------- proprio branch, 32 rows per block: Linear+ReLU, Linear+ReLU, state_projector+ReLU -> token 0
    constexpr int MR = LY::MLP_ROWS;
    const int r0 = ((int)blockIdx.x - E) * MR;
    float* sin = reinterpret_cast<float*>(smem);
    T* h1 = reinterpret_cast<T*>(smem + (size_t)MR * LY::LDS_IN * 4);
    T* h2 = h1 + MR * LY::LDH;
    for (int idx = tid; idx < MR * 128; idx += 256) {
      const int r = idx >> 7, c = idx & 127;
      const bool ok = r0 + r < E;
      float v = 0.f;
      if (train) {
        if (ok && c < w.Sp) v = tr.state[(int64_t)(tr.rowidx ? tr.rowidx[r0 + r] : r0 + r) * w.Sp + c];
      } else {
        if (ok && c < w.S) v = obs[(int64_t)(r0 + r) * D + c];
        if (ok && c < w.Sp) state_roll[(slot0 + r0 + r) * w.Sp + c] = v;
      }
      sin[r * LY::LDS_IN + c] = v;
    }
    __syncthreads();
    const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
    f32x4 acc[2][4];
    auto store_h = [&](T* h, const float* bias, float* save) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(bias + n4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const float v0 = fmaxf(acc[mt][j][0] + bb.x, 0.f), v1 = fmaxf(acc[mt][j][1] + bb.y, 0.f);
          const float v2 = fmaxf(acc[mt][j][2] + bb.z, 0.f), v3 = fmaxf(acc[mt][j][3] + bb.w, 0.f);
          st4(h + (mt * 16 + fr) * LY::LDH + n4, v0, v1, v2, v3);
          if (save != nullptr && r0 + mt * 16 + fr < E) st4(save + (int64_t)(r0 + mt * 16 + fr) * 256 + n4, v0, v1, v2, v3);
        }
      }
    };
    zero_acc(acc);
    if (w.Kp1 == 128) block_gemm<T, 2, 4, 4>(acc, sin, LY::LDS_IN, (const T*)w.wf1, 128, nt4, lane);
    else block_gemm<T, 2, 4, 2>(acc, sin, LY::LDS_IN, (const T*)w.wf1, 64, nt4, lane);
    store_h(h1, w.bf1, tr.s_h1);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 2, 4, 8>(acc, h1, LY::LDH, (const T*)w.wf2, 256, nt4, lane);
    store_h(h2, w.bf2, tr.s_h2);
    __syncthreads();
    const int nt1[1] = {wave};
    f32x4 ap[2][1];
    zero_acc(ap);
    block_gemm<T, 2, 1, 8>(ap, h2, LY::LDH, (const T*)w.wpr, 256, nt1, lane);
    {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(w.bpr + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = r0 + mt * 16 + fr;
        if (row < E)
          st4(x0 + ((int64_t)row * NTOK) * TD + n4, fmaxf(ap[mt][0][0] + bb.x, 0.f), fmaxf(ap[mt][0][1] + bb.y, 0.f),
              fmaxf(ap[mt][0][2] + bb.z, 0.f), fmaxf(ap[mt][0][3] + bb.w, 0.f));
      }
    }
    return;
  }

  // ---------------- depth branch, one sample per block
  const int b = blockIdx.x;
  T* img = reinterpret_cast<T*>(smem);
  T* c1 = img + LY::IMG;
  T* c2 = c1 + LY::C1;
  T* c3 = c2 + LY::C2;
  if (train) {
    // ingested depth stack (already in the operand type, contiguous): global -> LDS DMA, 16 bytes per lane and transfer,
    // every transfer in flight at once (a load -> ds_write loop waits for each load before issuing the next)
    const T* src = reinterpret_cast<const T*>(tr.image) + (int64_t)(tr.rowidx ? tr.rowidx[b] : b) * LY::IMG;
    constexpr int V = 16 / sizeof(T);
#pragma unroll
    for (int k = 0; k < LY::IMG / V / 256; ++k)
      __builtin_amdgcn_global_load_lds((const V4L_GLOBAL void*)(src + (int64_t)(tid + k * 256) * V),
                                       (__attribute__((address_space(3))) void*)(img + ((tid & ~63) + k * 256) * V), 16, 0, 0);
  } else {
    const float* src = obs + (int64_t)b * D + w.S;  // 16-byte aligned only when S % 4 == 0: dword-aligned vector loads (ld4u)
    T* roll = image_roll + (slot0 + b) * (int64_t)LY::IMG;
    for (int i = tid; i < LY::IMG / 4; i += 256) {
      const float4 v = ld4u(src + i * 4);
      T t4[4] = {Op<T>::from_f32(v.x), Op<T>::from_f32(v.y), Op<T>::from_f32(v.z), Op<T>::from_f32(v.w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) { img[i * 4 + j] = t4[j]; roll[i * 4 + j] = t4[j]; }
    }
  }
  __syncthreads();
  {  // conv1: 225 pixels (15 row tiles; wave w owns tiles w, w+4, w+8, w+12), K = (c,ky,kx) = 256, N = 32
    f32x4 acc[4][2];
    zero_acc(acc);
    int pbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = min((wave + 4 * i) * 16 + fr, 224);
      pbase[i] = (p / 15) * 4 * 64 + (p % 15) * 4;
    }
    frag_t fbn[8][2];  // all 8 k-steps of conv1's two column tiles (16 KB of weights per block) up front
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fbn[ks][j] = *reinterpret_cast<const frag_t*>((const T*)w.w1 + (j * 16 + fr) * 256 + ks * 32 + fg);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = ks * 32 + fg, c = k0 >> 6, ky = (k0 >> 3) & 7;
      frag_t fb[2] = {fbn[ks][0], fbn[ks][1]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave + 4 * i < 15) {
          const frag_t fa = afrag_t(img + c * 4096 + ky * 64 + pbase[i]);
#pragma unroll
          for (int j = 0; j < 2; ++j) mma_k32(acc[i][j], fb[j], fa);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n4 = j * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(w.b1 + n4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = (wave + 4 * i) * 16 + fr;
        if (p < 225) {
          const float v0 = fmaxf(acc[i][j][0] + bb.x, 0.f), v1 = fmaxf(acc[i][j][1] + bb.y, 0.f);
          const float v2 = fmaxf(acc[i][j][2] + bb.z, 0.f), v3 = fmaxf(acc[i][j][3] + bb.w, 0.f);
          st4(c1 + p * LY::LD1 + n4, v0, v1, v2, v3);
          if (tr.s_c1 != nullptr) st4(tr.s_c1 + ((int64_t)b * 225 + p) * 32 + n4, v0, v1, v2, v3);
        }
      }
    }
  }
  __syncthreads();
  {  // conv2: 36 pixels (3 row tiles), K = (ky,kx,c) = 512, N = 64: wave w owns column tile w
    f32x4 acc[3][1];
    zero_acc(acc);
    int pb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = min(i * 16 + fr, 35);
      pb[i] = ((p / 6) * 2 * 15 + (p % 6) * 2) * LY::LD1;
    }
    const T* w2row = (const T*)w.w2 + (wave * 16 + fr) * 512 + fg;
    frag_t ring[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) ring[d] = *reinterpret_cast<const frag_t*>(w2row + d * 32);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int tap = ks, ky = tap >> 2, kx = tap & 3;  // 32 channels per tap == one K=32 step
      const frag_t fb = ring[ks & 3];
      if (ks + 4 < 16) ring[ks & 3] = *reinterpret_cast<const frag_t*>(w2row + (ks + 4) * 32);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const frag_t fa = afrag_t(c1 + pb[i] + (ky * 15 + kx) * LY::LD1 + fg);
        mma_k32(acc[i][0], fb, fa);
      }
    }
    {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(w.b2 + n4);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int p = i * 16 + fr;
        if (p < 36) {
          const float v0 = fmaxf(acc[i][0][0] + bb.x, 0.f), v1 = fmaxf(acc[i][0][1] + bb.y, 0.f);
          const float v2 = fmaxf(acc[i][0][2] + bb.z, 0.f), v3 = fmaxf(acc[i][0][3] + bb.w, 0.f);
          st4(c2 + p * LY::LD2 + n4, v0, v1, v2, v3);
          if (tr.s_c2 != nullptr) st4(tr.s_c2 + ((int64_t)b * 36 + p) * 64 + n4, v0, v1, v2, v3);
        }
      }
    }
  }
  __syncthreads();
  {  // conv3: 16 pixels, K = (ky,kx,c) = 576 (two K=32 steps per tap), N = 64
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int pb = ((fr >> 2) * 6 + (fr & 3)) * LY::LD2;
    const T* w3row = (const T*)w.w3 + (wave * 16 + fr) * 576 + fg;
    frag_t ring[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) ring[d] = *reinterpret_cast<const frag_t*>(w3row + d * 32);
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3, c0 = (ks & 1) * 32 + fg;
      const frag_t fb = ring[ks % 6];
      if (ks + 6 < 18) ring[ks % 6] = *reinterpret_cast<const frag_t*>(w3row + (ks + 6) * 32);
      const frag_t fa = afrag_t(c2 + pb + (ky * 6 + kx) * LY::LD2 + c0);
      mma_k32(acc, fb, fa);
    }
    const int n4 = wave * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(w.b3 + n4);
    const float v0 = fmaxf(acc[0] + bb.x, 0.f), v1 = fmaxf(acc[1] + bb.y, 0.f);
    const float v2 = fmaxf(acc[2] + bb.z, 0.f), v3 = fmaxf(acc[3] + bb.w, 0.f);
    st4(c3 + fr * LY::LD2 + n4, v0, v1, v2, v3);
    if (tr.s_c3 != nullptr) st4(tr.s_c3 + ((int64_t)b * 16 + fr) * 64 + n4, v0, v1, v2, v3);
  }
  __syncthreads();
  {  // depth_up_conv (1x1, no activation) -> tokens 1..16
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < 2; ++ks) {
      const frag_t fb = *reinterpret_cast<const frag_t*>((const T*)w.wup + (wave * 16 + fr) * 64 + ks * 32 + fg);
      const frag_t fa = afrag_t(c3 + fr * LY::LD2 + ks * 32 + fg);
      mma_k32(acc, fb, fa);
    }
    const int n4 = wave * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(w.bup + n4);
    st4(x0 + ((int64_t)b * NTOK + 1 + fr) * TD + n4, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
  }
}

// ------------------------------------------------------------------------------------------ transformer layer
struct InfLayer {
  const void *win, *wo, *w1, *w2;        // packed (T): [192][64] [64][64] [ff][64] [64][ffp]
  const float *bin, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
  const float* xin;                      // [E*17][64] fp32
  float* xout;
  // training forward: what the backward pass needs (all null for inference). fp32: qkv, P, xhat / rstd of both norms;
  // in the contraction type T (they are only ever weight-grad operands / a ReLU mask): the layer input, ctx, x1, f
  float *s_qkv, *s_P, *s_xh1, *s_rs1, *s_xh2, *s_rs2;
  void *s_xin, *s_ctx, *s_x1, *s_f;
};
struct InfLayerPair { InfLayer n[2]; };
constexpr int ROLLOUT_MAX_LAYERS = 4;
struct InfLayerStack { InfLayerPair l[ROLLOUT_MAX_LAYERS]; int nl; };  // several layers of both nets in ONE launch

// Heads (pool + append fcs + last linear), run by the LAST layer's blocks on their own samples
struct InfHead {
  const void *w0, *w1, *w2;            // packed (T): [256][128] [256][256] [16][256]
  const float *b0, *b1, *b2;
  float* out;                          // [E][OUT_LD], columns >= nout zeroed
  int nout;
  int max_pool;                        // rollout_stack_kernel only: the depth tokens pooled by max instead of mean (nets.py:1022-1030, 884-889)
  float *s_pooled, *s_h0, *s_h1;       // training forward: [E][128], [E][256], [E][256] (post-ReLU); null for inference
  // rollout_stack_kernel<..., OPT>: this net's token_ln (OPT & 1: nets.py:1007-1008, in front of the layers) and final encoder
  // norm (OPT & 2: nets.py:955-963, behind them) weight / bias [64], or null
  const float *tn_g, *tn_b, *fn_g, *fn_b;
};
struct InfHeadPair { InfHead n[2]; };

// Rollout step epilogue (blockIdx.y 0 = policy: sample + file the action; 1 = value net: file the value); ctl == null:
// no sampling (training forward). The last block to finish advances the device-side step cursor.
struct InfFinish {
  ActCtl* ctl;
  const float *logstd, *eps;
  int A;
  int tanh_action;  // TanhNormal head (distribution.py:5-80): action = tanh(mean + std eps), log-prob through the stored action
  float *acts_roll, *values_roll, *logp_roll, *action, *mean, *stdv, *ent, *value;
  // eager launches: the env step index + 1 as the host counts it (v4l_actor_seek / one per step); 0: read the device cursor
  // ctl->t and advance it through the last-block counter (graph replays, whose arguments are frozen). With the host's index no
  // block has to find out whether it is the last one: that atomic round trip was 2.6 K cycles at the end of every step.
  long long t_plus1;
  // eager launches of rollout_dense_kernel: the launch sequence number + 1 as the host counts it (v4l_actor::dense_seq mirrors
  // ctl->seq), so that no block of the launch depends on WHEN it reads ctl->seq relative to the finishing block's bump of it;
  // 0: read ctl->seq at entry (graph replays: frozen arguments, seq then moves only after BOTH finishing blocks are done)
  unsigned seq_plus1;
};

template <int SPW> struct InfRows {
  static constexpr int ROWS = (SPW * NTOK + 15) / 16 * 16;  // 1 -> 32, 2 -> 48, 4 -> 80
  static constexpr int MT = ROWS / 16;
  static constexpr int U = ROWS == 80 ? 5 : 4;             // LayerNorm rows in flight per wave (ROWS/4 % U == 0)
};

template <typename T, int SPW> struct InfLayLds {
  static constexpr int PAD = InfLd<T>::PAD;
  static constexpr int ROWS = InfRows<SPW>::ROWS;
  static constexpr int LDX = 64 + 4, LDQ = 192 + 4, LDF = 256 + PAD, LDP = 128 + 4;
  static constexpr int LDT = 64 + PAD;                   // q / k / ctx rows in the compute type
  static constexpr int QROWS = ROWS + 16;                // a sample's 32-row key window may run past the last token row
  static constexpr size_t xs_b = (size_t)ROWS * LDX * 4;
  static constexpr size_t f_b = (size_t)ROWS * LDF * sizeof(T);
  static constexpr size_t head_b = (size_t)16 * LDP * 4 + (size_t)2 * 16 * LDF * sizeof(T) + 16 * 16 * 4;
  // attention operands (attn_tile): q | k rows, transposed values per sample, one P scratch tile per wave
  static constexpr size_t att_b = ((size_t)2 * QROWS * LDT + (size_t)SPW * 64 * ATT_LDV + (size_t)4 * 16 * ATT_LDV) * sizeof(T);
  static constexpr size_t big0_b = att_b > f_b ? att_b : f_b;
  static constexpr size_t big_b = (big0_b > head_b ? big0_b : head_b) / 16 * 16 + 16;
  static constexpr size_t p_b = (size_t)ROWS * LDT * sizeof(T);  // attention context rows (T)
  static constexpr size_t bytes = xs_b + big_b + xs_b + p_b;  // xs | attention operands / z / f / head | z2 | ctx
};

__device__ __forceinline__ float dot64(const float* a, const float* b) {  // 16-byte aligned LDS rows of 64 floats
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int d = 0; d < TD; d += 8) {
    const float4 x0 = *reinterpret_cast<const float4*>(a + d), y0 = *reinterpret_cast<const float4*>(b + d);
    const float4 x1 = *reinterpret_cast<const float4*>(a + d + 4), y1 = *reinterpret_cast<const float4*>(b + d + 4);
    s0 = fmaf(x0.x, y0.x, s0); s0 = fmaf(x0.y, y0.y, s0); s0 = fmaf(x0.z, y0.z, s0); s0 = fmaf(x0.w, y0.w, s0);
    s1 = fmaf(x1.x, y1.x, s1); s1 = fmaf(x1.y, y1.y, s1); s1 = fmaf(x1.z, y1.z, s1); s1 = fmaf(x1.w, y1.w, s1);
  }
  return s0 + s1;
}

// LayerNorm of the ROWS LDS rows; optionally (training) saves xhat / rstd / the output rows < nrows to global memory.
// A row is owned by a quarter wave (16 lanes x 4 consecutive columns): both reductions are 4 DPP row_ror adds inside the
// 16-lane DPP row, every load / store is 16 bytes, and a wave normalises 4 rows per step — ROWS/16 independent steps per
// wave, all in flight together. (One row per wave, lane = column, cost 2 x (4 DPP + 4 v_readlane) and 4-byte accesses
// per row: 7.5 K cycles of the 60 K a training-forward block takes; this form: see DESIGN.md.)
template <int ROWS, int U, typename SO>
__device__ __forceinline__ void ln_rows(const float* z, int ldz, float* out, int ldo, const float* __restrict__ g,
                                        const float* __restrict__ be, int wave, int lane, int nrows, float* s_xh,
                                        float* s_rs, SO* s_out) {
  (void)U;
  const int l16 = lane & 15, c4 = l16 * 4, q = lane >> 4;
  const float4 gg = *reinterpret_cast<const float4*>(g + c4), bb = *reinterpret_cast<const float4*>(be + c4);
  constexpr int IT = ROWS / 16;
  float4 v[IT];
  float mean[IT], var[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) v[it] = *reinterpret_cast<const float4*>(z + (it * 16 + wave * 4 + q) * ldz + c4);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    float s = (v[it].x + v[it].y) + (v[it].z + v[it].w);
    s += dpp_mov<0x128>(s); s += dpp_mov<0x124>(s); s += dpp_mov<0x122>(s); s += dpp_mov<0x121>(s);
    mean[it] = s * (1.f / TD);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    v[it].x -= mean[it]; v[it].y -= mean[it]; v[it].z -= mean[it]; v[it].w -= mean[it];
    float s = (v[it].x * v[it].x + v[it].y * v[it].y) + (v[it].z * v[it].z + v[it].w * v[it].w);
    s += dpp_mov<0x128>(s); s += dpp_mov<0x124>(s); s += dpp_mov<0x122>(s); s += dpp_mov<0x121>(s);
    var[it] = s * (1.f / TD);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int r = it * 16 + wave * 4 + q;
    const float rs = 1.f / sqrtf(var[it] + 1e-5f);
    const float4 xh = {v[it].x * rs, v[it].y * rs, v[it].z * rs, v[it].w * rs};
    const float4 o = {fmaf(xh.x, gg.x, bb.x), fmaf(xh.y, gg.y, bb.y), fmaf(xh.z, gg.z, bb.z), fmaf(xh.w, gg.w, bb.w)};
    if (out != nullptr) *reinterpret_cast<float4*>(out + r * ldo + c4) = o;
    if (r < nrows) {
      if (s_xh != nullptr) {
        *reinterpret_cast<float4*>(s_xh + r * TD + c4) = xh;
        if (l16 == 0) s_rs[r] = rs;
      }
      if (s_out != nullptr) st4(s_out + r * TD + c4, o.x, o.y, o.z, o.w);
    }
  }
}

#ifdef V4L_INFER_TIMING
__device__ long long g_inf_stamps[128];
#define INF_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_inf_stamps[i] = clock64(); } while (0)
// rollout kernels: [64..95] rollout_stack_kernel (layer 0: 64.., head: 80..), [96..111] rollout_encoder2_kernel (depth block 0)
#define ROLL_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_inf_stamps[i] = clock64(); } while (0)
// per-block placement / time log of the fused forward-loss-backward launch (round 6, tools/probe/fb_blocks.py): block b ->
// {wall clock at entry, shader clock at entry, HW_ID | XCC_ID << 32, wall clock at exit, shader clock at exit} — wall clock =
// s_memrealtime (100 MHz, one counter for the chip), shader clock = s_memtime (runs at the XCD's current frequency)
__device__ long long g_blk_log[1024 * 5];
#define BLK_LOG_BEGIN() do { if (threadIdx.x == 0 && blockIdx.x < 1024) { long long* q_ = g_blk_log + 5 * blockIdx.x; \
    q_[0] = wall_clock64(); q_[1] = clock64(); \
    q_[2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } } while (0)
#define BLK_LOG_END() do { if (threadIdx.x == 0 && blockIdx.x < 1024) { long long* q_ = g_blk_log + 5 * blockIdx.x; \
    q_[3] = wall_clock64(); q_[4] = clock64(); } } while (0)
#else
#define INF_STAMP(i)
#define ROLL_STAMP(i)
#define BLK_LOG_BEGIN()
#define BLK_LOG_END()
#endif

// One nn.TransformerEncoderLayer for SPW samples per block (blockIdx.y = net). HEAD: the block continues with the
// pooled heads of its samples (nets.py:1015-1034) and, in a rollout step, with the sampling / filing epilogue.
//   SPW = 4: training forward (68 of 80 MFMA rows used); SPW = 1: rollout steps (E blocks per net, shortest latency)
// NL > 1: the block walks NL consecutive layers (stk.l[0..NL-1]) with the token rows staying in LDS between them.
template <typename T, int SPW, bool HEAD, int NL = 1>
__global__ __launch_bounds__(256) void infer_layer_kernel(InfLayerStack stk, InfHeadPair hd, InfFinish fin, int E, int ff) {
  typedef InfLayLds<T, SPW> LY;
  constexpr int ROWS = InfRows<SPW>::ROWS, MT = InfRows<SPW>::MT, U = InfRows<SPW>::U;
  INF_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, qr = (lane >> 4) * 4;
  float* xs = reinterpret_cast<float*>(smem);
  float* big = reinterpret_cast<float*>(smem + LY::xs_b);            // qkv, later z (fp32), later f (T), later heads
  float* cx = reinterpret_cast<float*>(smem + LY::xs_b + LY::big_b); // ctx, later z2
  T* cb = reinterpret_cast<T*>(smem + LY::xs_b + LY::big_b + LY::xs_b);  // attention context rows (T): out_proj's operand
  constexpr int LDT = LY::LDT;
  T* qb = reinterpret_cast<T*>(big);                                   // q rows | k rows | v^T per sample | P scratch:
  T* kb = qb + LY::QROWS * LDT;                                        // dead once the attention is done, `big` then takes z
  T* vt = kb + LY::QROWS * LDT;
  T* pbuf = vt + SPW * 64 * ATT_LDV;
  const int s0 = blockIdx.x * SPW;
  const int ns = min(SPW, E - s0);
  const int nrows = ns * NTOK;
  const int64_t row0 = (int64_t)s0 * NTOK;  // first token row of this block
  long long t_step = 0;
  if constexpr (HEAD) { if (fin.ctl != nullptr) t_step = fin.ctl->t; }
  const int nt[3] = {wave, wave + 4, wave + 8};
  const int nt1[1] = {wave};
  const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  // every GEMM's first weight fragments are requested one phase early (GemmRing): the L2 round trip and the acknowledgement
  // of the global stores in front of it overlap that phase
  GemmRing<T, 3, 2> ring_in = gemm_prefetch<T, 3, 2>((const T*)stk.l[0].n[blockIdx.y].win, 64, nt, lane);
  // A layer's biases are fetched before the layer stores anything (the next layer's: before this layer's last saves). Loads and
  // stores retire through one in-order queue per wave: a bias fetched inside an epilogue loop made every iteration wait for
  // the acknowledgement of the saves issued just before it (7-8 K cycles per in_proj / linear1 epilogue, clock64 stamps).
  struct LayerBias { float4 in[3], o, f1[4], f2; };
  auto fetch_bias = [&](const InfLayer& wl) {
    LayerBias b;
#pragma unroll
    for (int j = 0; j < 3; ++j) b.in[j] = *reinterpret_cast<const float4*>(wl.bin + nt[j] * 16 + qr);
    b.o = *reinterpret_cast<const float4*>(wl.bo + wave * 16 + qr);
#pragma unroll
    for (int j = 0; j < 4; ++j) b.f1[j] = *reinterpret_cast<const float4*>(wl.b1 + nt4[j] * 16 + qr);
    b.f2 = *reinterpret_cast<const float4*>(wl.b2 + wave * 16 + qr);
    return b;
  };
  LayerBias bias_next = fetch_bias(stk.l[0].n[blockIdx.y]);
#pragma unroll
  for (int l = 0; l < NL; ++l) {
  const InfLayer& w = stk.l[l].n[blockIdx.y];
  const LayerBias lb = bias_next;
  if (l == 0) {
    const float* xg = w.xin + (int64_t)s0 * NTOK * TD;
    for (int i4 = tid; i4 < ROWS * (TD / 4); i4 += 256) {
      const int r = i4 >> 4, c4 = (i4 & 15) * 4;
      const bool ok = r < nrows;
      const float4 v = *reinterpret_cast<const float4*>(xg + (ok ? r : 0) * TD + c4);
      *reinterpret_cast<float4*>(xs + r * LY::LDX + c4) = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
      if (w.s_xin != nullptr && ok) st4(reinterpret_cast<T*>(w.s_xin) + (row0 + r) * TD + c4, v.x, v.y, v.z, v.w);
      st4(cb + r * LDT + c4, 0.f, 0.f, 0.f, 0.f);  // rows the attention does not write (padding / absent samples)
    }
  } else {
    INF_STAMP(0);     // (stacked launch: the last layer's pass overwrites the earlier one's stamps)
    __syncthreads();  // the previous layer's norm2 left this layer's input rows in xs
    if (w.s_xin != nullptr) {
      for (int i4 = tid; i4 < ROWS * (TD / 4); i4 += 256) {
        const int r = i4 >> 4, c4 = (i4 & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(xs + r * LY::LDX + c4);
        if (r < nrows) st4(reinterpret_cast<T*>(w.s_xin) + (row0 + r) * TD + c4, v.x, v.y, v.z, v.w);
      }
    }
  }
  for (int i = tid; i < SPW * 64 * (32 - NTOK); i += 256) {  // key columns 17..31 of v^T: zeros (P there is 0, 0 x NaN is not)
    const int rr = i / (32 - NTOK), cc = NTOK + i - rr * (32 - NTOK);
    vt[rr * ATT_LDV + cc] = (T)0.f;
  }
  if (l == 0) __syncthreads();
  INF_STAMP(1);
  GemmRing<T, 1, 2> ring_o;
  {  // in_proj: [ROWS][64] x [192][64]^T -> qkv (fp32)
    f32x4 acc[MT][3];
    zero_acc(acc);
    block_gemm<T, MT, 3, 2>(acc, xs, LY::LDX, (const T*)w.win, 64, nt, lane, ring_in);
    INF_STAMP(9);
    ring_o = gemm_prefetch<T, 1, 2>((const T*)w.wo, 64, nt1, lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // column tile wave + 4j: j = 0 -> q, 1 -> k, 2 -> v
      const int n4 = nt[j] * 16 + qr;
      const float4 bb = lb.in[j];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + fr;
        const float v0 = acc[mt][j][0] + bb.x, v1 = acc[mt][j][1] + bb.y, v2 = acc[mt][j][2] + bb.z, v3 = acc[mt][j][3] + bb.w;
        if (w.s_qkv != nullptr && row < nrows) st4(w.s_qkv + (row0 + row) * 192 + n4, v0, v1, v2, v3);  // fp32, for the backward
        if (j == 0) st4(qb + row * LDT + n4, v0, v1, v2, v3);
        else if (j == 1) st4(kb + row * LDT + (n4 - TD), v0, v1, v2, v3);
        else {
          const int sm = row / NTOK, key = row - sm * NTOK;
          if (sm < SPW) {  // values transposed per sample for the P V product
            T* vc = vt + (sm * 64 + (n4 - 2 * TD)) * ATT_LDV + key;
            vc[0] = (T)v0; vc[ATT_LDV] = (T)v1; vc[2 * ATT_LDV] = (T)v2; vc[3 * ATT_LDV] = (T)v3;
          }
        }
      }
    }
  }
  INF_STAMP(10);
  __syncthreads();
  INF_STAMP(2);
  // attention on the matrix cores: one (sample, query tile) job per wave pass — tokens 0..15 | token 16 of each sample
  for (int j = wave; j < 2 * ns; j += 4) {
    const int sm = j >> 1, mt = j & 1;
    attn_tile<T>(qb + (sm * NTOK + mt * 16) * LDT, kb + sm * NTOK * LDT, LDT, vt + sm * 64 * ATT_LDV, pbuf + wave * 16 * ATT_LDV,
                 cb + (sm * NTOK + mt * 16) * LDT, LDT, lane, mt * 16, mt == 0 ? 16 : 1,
                 w.s_P != nullptr ? w.s_P + (int64_t)(s0 + sm) * NTOK * NTOK : nullptr,
                 w.s_ctx != nullptr ? reinterpret_cast<T*>(w.s_ctx) + (row0 + sm * NTOK) * TD : nullptr);
  }
  __syncthreads();
  INF_STAMP(3);
  GemmRing<T, 4, 2> ring_1;
  {  // out_proj + residual -> z (in `big`, fp32 [ROWS][LDX])
    f32x4 acc[MT][1];
    zero_acc(acc);
    block_gemm<T, MT, 1, 2>(acc, cb, LDT, (const T*)w.wo, 64, nt1, lane, ring_o);
    ring_1 = gemm_prefetch<T, 4, 2>((const T*)w.w1, 64, nt4, lane);  // in front of norm1 and its saves
    const int n4 = wave * 16 + qr;
    const float4 bb = lb.o;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fr;
      const float4 xr = *reinterpret_cast<const float4*>(xs + row * LY::LDX + n4);
      st4(big + row * LY::LDX + n4, xr.x + acc[mt][0][0] + bb.x, xr.y + acc[mt][0][1] + bb.y, xr.z + acc[mt][0][2] + bb.z,
          xr.w + acc[mt][0][3] + bb.w);
    }
  }
  __syncthreads();
  INF_STAMP(4);
  ln_rows<ROWS, U, T>(big, LY::LDX, xs, LY::LDX, w.g1, w.be1, wave, lane, nrows, w.s_xh1 ? w.s_xh1 + row0 * TD : nullptr,
                      w.s_rs1 ? w.s_rs1 + row0 : nullptr,
                      w.s_x1 ? reinterpret_cast<T*>(w.s_x1) + row0 * TD : nullptr);  // x1 -> xs
  __syncthreads();
  INF_STAMP(5);
  T* f = reinterpret_cast<T*>(big);
  GemmRing<T, 1, 8> ring_2;
  {  // linear1 + ReLU -> f (T) ; ff <= 256: wave w owns column tiles 4w..4w+3
    f32x4 acc[MT][4];
    zero_acc(acc);
    block_gemm<T, MT, 4, 2>(acc, xs, LY::LDX, (const T*)w.w1, 64, nt4, lane, ring_1);
    INF_STAMP(11);
    ring_2 = gemm_prefetch<T, 1, 8>((const T*)w.w2, 256, nt1, lane);  // in front of the f saves
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n4 = nt4[j] * 16 + qr;
      const float4 bb = lb.f1[j];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float f0 = fmaxf(acc[mt][j][0] + bb.x, 0.f), f1 = fmaxf(acc[mt][j][1] + bb.y, 0.f);
        const float f2 = fmaxf(acc[mt][j][2] + bb.z, 0.f), f3 = fmaxf(acc[mt][j][3] + bb.w, 0.f);
        st4(f + (mt * 16 + fr) * LY::LDF + n4, f0, f1, f2, f3);
        if (w.s_f != nullptr && mt * 16 + fr < nrows)
          st4(reinterpret_cast<T*>(w.s_f) + (row0 + mt * 16 + fr) * 256 + n4, f0, f1, f2, f3);
      }
    }
  }
  INF_STAMP(12);
  __syncthreads();
  INF_STAMP(6);
  {  // linear2 + residual -> z2 (in `cx`)
    f32x4 acc[MT][1];
    zero_acc(acc);
    block_gemm<T, MT, 1, 8>(acc, f, LY::LDF, (const T*)w.w2, 256, nt1, lane, ring_2);
    const int n4 = wave * 16 + qr;
    const float4 bb = lb.f2;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fr;
      const float4 xr = *reinterpret_cast<const float4*>(xs + row * LY::LDX + n4);
      st4(cx + row * LY::LDX + n4, xr.x + acc[mt][0][0] + bb.x, xr.y + acc[mt][0][1] + bb.y, xr.z + acc[mt][0][2] + bb.z,
          xr.w + acc[mt][0][3] + bb.w);
    }
  }
  __syncthreads();
  INF_STAMP(7);
  if (l + 1 < NL) {
    ring_in = gemm_prefetch<T, 3, 2>((const T*)stk.l[l + 1 < NL ? l + 1 : l].n[blockIdx.y].win, 64, nt, lane);
    bias_next = fetch_bias(stk.l[l + 1 < NL ? l + 1 : l].n[blockIdx.y]);
  }
  ln_rows<ROWS, U, float>(cx, LY::LDX, (HEAD || l + 1 < NL) ? xs : nullptr, LY::LDX, w.g2, w.be2, wave, lane, nrows,
                   w.s_xh2 ? w.s_xh2 + row0 * TD : nullptr, w.s_rs2 ? w.s_rs2 + row0 : nullptr, w.xout + row0 * TD);
  INF_STAMP(8);
  }  // layers
  if constexpr (HEAD) {
    // ---- heads on this block's samples: [state token | mean of the 16 depth tokens] -> 256 -> 256 -> nout
    const InfHead& h = hd.n[blockIdx.y];
    float* pooled = big;                                              // [16][LDP] fp32 (rows >= ns: zeros)
    T* h1 = reinterpret_cast<T*>(big + 16 * LY::LDP);                 // [16][LDF]
    T* h2 = h1 + 16 * LY::LDF;
    float* so = reinterpret_cast<float*>(h2 + 16 * LY::LDF);          // [16][16] last-layer outputs
    float4 hb0[4], hb1[4];  // head biases: fetched before the head stores anything (see LayerBias)
    float hb2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hb0[j] = *reinterpret_cast<const float4*>(h.b0 + nt4[j] * 16 + qr);
      hb1[j] = *reinterpret_cast<const float4*>(h.b1 + nt4[j] * 16 + qr);
      hb2[j] = h.b2[min(qr + j, h.nout - 1)];
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * 128; idx += 256) {
      const int r = idx >> 7, c = idx & 127;
      float v = 0.f;
      if (r < ns) {
        const float* xb = xs + (r * NTOK) * LY::LDX;
        if (c < TD) v = xb[c];
        else {
          float s = 0.f;
#pragma unroll
          for (int i = 1; i < NTOK; ++i) s += xb[i * LY::LDX + (c - TD)];
          v = s * (1.f / 16.f);
        }
        if (h.s_pooled != nullptr) h.s_pooled[(int64_t)(s0 + r) * 128 + c] = v;
      }
      pooled[r * LY::LDP + c] = v;
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto store_h = [&](T* dst, const float4 (&bias)[4], float* save) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float4 bb = bias[j];
        const float v0 = fmaxf(acc[0][j][0] + bb.x, 0.f), v1 = fmaxf(acc[0][j][1] + bb.y, 0.f);
        const float v2 = fmaxf(acc[0][j][2] + bb.z, 0.f), v3 = fmaxf(acc[0][j][3] + bb.w, 0.f);
        st4(dst + fr * LY::LDF + n4, v0, v1, v2, v3);
        if (save != nullptr && fr < ns) st4(save + (int64_t)(s0 + fr) * 256 + n4, v0, v1, v2, v3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 4>(acc, pooled, LY::LDP, (const T*)h.w0, 128, nt4, lane);
    store_h(h1, hb0, h.s_h0);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, h1, LY::LDF, (const T*)h.w1, 256, nt4, lane);
    store_h(h2, hb1, h.s_h1);
    __syncthreads();
    if (wave == 0) {  // last linear: one 16-column tile
      const int nt0[1] = {0};
      f32x4 a1[1][1];
      zero_acc(a1);
      block_gemm<T, 1, 1, 8>(a1, h2, LY::LDF, (const T*)h.w2, 256, nt0, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = qr + r;
        const float v = c < h.nout ? a1[0][0][r] + hb2[r] : 0.f;
        so[fr * 16 + c] = v;
        if (fr < ns) h.out[(int64_t)(s0 + fr) * OUT_LD + c] = v;
      }
    }
    if (fin.ctl != nullptr) {
      // ---- rollout step epilogue (GaussianContPolicyBase.explore, continuous_policy.py:85-125, and the collector's
      // value read-out, collector/on_policy.py:95-100): action = mean + std * eps, entropy, log pi(a|s); filed at
      // rollout slot t*E + i. Same expressions as act_finish_kernel / actor_loss_kernel.
      __syncthreads();
      if (tid < ns) {
        const int i = s0 + tid;
        const int A = fin.A;
        if (blockIdx.y == 0) {
          float e = 0.f, lp = 0.f;
          for (int a = 0; a < A; ++a) {
            const float mu = so[tid * 16 + a];
            const float ls = fminf(fmaxf(fin.logstd[a], LOG_SIG_MIN), LOG_SIG_MAX);
            const float sg = expf(ls);
            e += 0.5f + HALF_LOG_2PI + logf(sg);
            float act = fmaf(sg, fin.eps[(int64_t)i * A + a], mu);
            if (fin.tanh_action) act = tanhf(act);  // (eps = 0: eval_act's tanh(mean))
            fin.action[(int64_t)i * A + a] = act;
            fin.mean[(int64_t)i * A + a] = mu;
            fin.stdv[(int64_t)i * A + a] = sg;
            if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + a] = act;
            const float d = (fin.tanh_action ? tanh_pre(act) : act) - mu;  // (through the STORED action, like act_finish_kernel)
            lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
            if (fin.tanh_action) lp -= tanh_corr(act);
          }
          fin.ent[i] = e;
          if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + i] = lp;
        } else {
          const float v = so[tid * 16];
          fin.value[i] = v;
          if (fin.values_roll != nullptr) fin.values_roll[t_step * E + i] = v;
        }
      }
      // the last block to get here advances the step cursor: every block read it at entry, none reads it again
      __syncthreads();
      if (tid == 0) {
        __threadfence();
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&fin.ctl->done), 1ull);
        if (done == 2ull * (unsigned long long)E - 1ull) {  // (one block per (sample, net))
          fin.ctl->done = 0;
          fin.ctl->t = t_step + 1;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------ rollout layer stack
template <typename T> struct RollStackLds { static constexpr size_t bytes = InfLayLds<T, 1>::bytes + (size_t)(2 * 832 + 528) * 4 + ((size_t)4 * 32 * (64 + InfLd<T>::PAD) + (size_t)(64 + 32) * (32 + 8)) * sizeof(T); };

// ------------------------------------------------------------------------------------------ rollout layer stack, weights ahead
// One nn.TransformerEncoderLayer stack (+ heads + sampling epilogue) for ONE sample per block, 8 waves: a rollout step runs
// only 2E blocks, one per CU, so its time is the latency of one block. What the phase stamps of its first version showed
// (tools/probe/stamps_rollout.py, E = 32: 67.6 K cycles per block): every kernel starts with a cold L2 — on a multi-XCD part
// the L2s are written back and invalidated at kernel boundaries — so the FIRST touch of each weight tile is an
// Infinity-Cache round trip of ~1.5 K cycles, and the kernel paid one per barrier-separated GEMM phase (in_proj 3.6 K,
// FF1 3.9 K, FF2 4.7 K, head fc0 / fc1 / fc2 6.0 / 9.2 / 2.6 K cycles for a few dozen MFMAs each) plus a serial
// thread-0 sampling epilogue of 7 K cycles (six dependent global loads). Here a wave's weight fragments of EVERY phase
// are requested one layer ahead into registers (bf16: 34 fragments = 136 VGPRs of the 256 a 512-thread block may use):
// layer 0's and the head's widest linear at kernel entry, layer l+1's as soon as layer l's MFMAs have consumed the
// register, the other head linears during the last layer. The fp32 parity mode (8 VGPRs per fragment) loads at use.
// Arithmetic, operand rounding and k order are those of infer_layer_kernel.
template <typename T, int MT, int KS, typename AT>
__device__ __forceinline__ void mm_held(f32x4 (&acc)[MT], const AT* sA, int lda, const typename Frag<T>::type (&fb)[KS], int lane) {
  typedef typename Frag<T>::type frag_t;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      frag_t fa;
      if constexpr (sizeof(AT) == sizeof(T)) fa = *reinterpret_cast<const frag_t*>(sA + (mt * 16 + fr) * lda + ks * 32 + fg);
      else fa = afrag<T>(reinterpret_cast<const float*>(sA) + (mt * 16 + fr) * lda + ks * 32 + fg);
      mma_k32(acc[mt], fb[ks], fa);
    }
}

// NT: tokens per sample — 17 (LocoTransformer: proprio token + 16 depth tokens, head input [token 0 | mean of the depth
// tokens]) or 16 (the vision-only Transformer, nets.py:884-889: head input = mean of all 16 tokens, fc0 contracts 64)
template <typename T, int NL, int NT = NTOK, int OPT = 0>
__global__ __launch_bounds__(512) void rollout_stack_kernel(InfLayerStack stk, InfHeadPair hd, InfFinish fin, int E, int warm,
                                                            int xcd) {
  typedef typename Frag<T>::type frag_t;
  typedef InfLayLds<T, 1> LY;
  constexpr int NW = 8, NTH = 512, ROWS = 32;
  constexpr bool PRE = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8, qr = (lane >> 4) * 4;
  float* xs = reinterpret_cast<float*>(smem);
  float* big = reinterpret_cast<float*>(smem + LY::xs_b);            // qkv, later z (fp32), later f (T), later heads
  float* cx = reinterpret_cast<float*>(smem + LY::xs_b + LY::big_b); // ctx, later z2
  // biases / LayerNorm parameters of every layer and of the head, staged once: a global load in the middle of a phase would
  // have to wait for every weight fragment requested before it (loads return in order)
  float* prm = reinterpret_cast<float*>(smem + LY::bytes);
  constexpr int P_BIN = 0, P_BO = 192, P_B1 = 256, P_B2 = 512, P_G1 = 576, P_BE1 = 640, P_G2 = 704, P_BE2 = 768, P_LAYER = 832;
  constexpr int P_H0 = 0, P_H1 = 256, P_H2 = 512, P_HEAD = 528;
  // (sample, net) of this block. xcd: a 1-D grid of 2 * ceil4(E) blocks in which the net follows the XCD the block lands on
  // (work-groups go round-robin over the 8 XCDs: block b runs on XCD b % 8) — XCDs 0..3 run the policy, 4..7 the value net, so
  // each XCD's L2 holds ONE net's 401 KB of weights instead of both (PMC, round 3: 6.1 MB of HBM traffic per launch against
  // 1.08 MB algorithmic — every L2 fetched both nets). Otherwise the (E, 2) grid: blockIdx.y = net.
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;
  const int net = xcd ? (lin >> 2) & 1 : (int)blockIdx.y, s0 = xcd ? (lin >> 3) * 4 + (lin & 3) : (int)blockIdx.x;
  if (s0 >= E) return;  // (xcd: E rounded up to a multiple of 4)
  constexpr int nl = NL;  // the layer loop is unrolled: the compiler's vmcnt bookkeeping stays exact (no loop-carried merges)
  float* prm_h = prm + 2 * P_LAYER;  // [layer l & 1][832] | head [528]
  // operand-type copies of the two GEMM inputs that also live in fp32 (token rows: residual / LayerNorm; attention context):
  // written once by their producer, read as whole MFMA fragments (one ds_read_b128) by every wave that multiplies with them.
  // (Building the fragment from the fp32 rows cost each wave 6 odd-sized LDS reads + 10 VALU per fragment: the in_proj / FF1
  // phases were bound by the LDS pipe all 8 waves share, 1.5 K cycles per column tile.)
  constexpr int LDT = 64 + LY::PAD;
  T* xb = reinterpret_cast<T*>(prm_h + P_HEAD);   // [32][LDT] layer input, then x1
  T* cb = xb + ROWS * LDT;                          // [32][LDT] attention context
  // attention operands (attn_tile): q / k rows and the transposed values in T, written by in_proj's epilogue
  T* qb = cb + ROWS * LDT;                          // [32][LDT]
  T* kb = qb + ROWS * LDT;                          // [32][LDT]
  T* vt = kb + ROWS * LDT;                          // [64][ATT_LDV], key columns >= 17 zero
  T* pbuf = vt + 64 * ATT_LDV;                      // [2][16][ATT_LDV] P of the two query tiles
  const int64_t row0 = (int64_t)s0 * NT;
  const InfHead& h = hd.n[net];
  ROLL_STAMP(64);
  // weights arrive in fragment order (PK_FRAG): one fragment = 64 consecutive frag_t, whole cache lines per wave load
  auto wfrag = [&](const void* W, int Kp, int tile, int ks) -> frag_t {
    return reinterpret_cast<const frag_t*>(W)[(tile * (Kp >> 5) + ks) * 64 + lane];
  };
  // a wave's fragments: in_proj tiles {wave, wave+8 (waves 0..3)}, out_proj / linear2 tile (wave & 3) for row tile
  // (wave >> 2), linear1 tiles {wave, wave+8}; head: fc0 / fc1 tiles {wave, wave+8}, last linear tile 0 (wave 0)
  frag_t r_in[2][2], r_o[2], r_f1[2][2], r_f2[8], r_h1[2][8];
  auto load_in = [&](const InfLayer& w) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) r_in[j][ks] = wfrag(w.win, 64, min(wave + 8 * j, 11), ks);
  };
  auto load_o = [&](const InfLayer& w) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) r_o[ks] = wfrag(w.wo, 64, wave & 3, ks);
  };
  auto load_f1 = [&](const InfLayer& w) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) r_f1[j][ks] = wfrag(w.w1, 64, wave + 8 * j, ks);
  };
  auto load_f2 = [&](const InfLayer& w) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r_f2[ks] = wfrag(w.w2, 256, wave & 3, ks);
  };
  // head fc0 ([256][128], 4 k-steps per tile) rides in r_in (k-steps 0,1) and r_f1 (k-steps 2,3) once the last layer is
  // done with them; the last linear ([16][256]) in r_f2
  auto load_h0a = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) r_in[j][ks] = wfrag(h.w0, NT == 16 ? 64 : 128, wave + 8 * j, ks);
  };
  auto load_h0b = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) r_f1[j][ks] = wfrag(h.w0, NT == 16 ? 64 : 128, wave + 8 * j, NT == 16 ? ks : 2 + ks);
  };
  auto load_h1 = [&](int j) {  // j: compile-time tile slot
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r_h1[j][ks] = wfrag(h.w1, 256, wave + 8 * j, ks);
  };
  auto load_h2 = [&]() {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r_f2[ks] = wfrag(h.w2, 256, 0, ks);
  };
  // thread t < 208 fetches float4 number t of a layer's [bin | bo | b1 | b2 | g1 | be1 | g2 | be2] (unconditional load from a
  // clamped, always-valid address: a load inside a branch ends in a vmcnt(0) wait)
  auto layer_params = [&](const InfLayer& wl) -> float4 {
    const int o = min(tid, P_LAYER / 4 - 1) * 4;
    const float *q0 = wl.bin, *q1 = wl.bo, *q2 = wl.b1, *q3 = wl.b2, *q4 = wl.g1, *q5 = wl.be1, *q6 = wl.g2, *q7 = wl.be2;
    const float* src = q0 + o;
    src = o >= P_BO ? q1 + (o - P_BO) : src;
    src = o >= P_B1 ? q2 + (o - P_B1) : src;
    src = o >= P_B2 ? q3 + (o - P_B2) : src;
    src = o >= P_G1 ? q4 + (o - P_G1) : src;
    src = o >= P_BE1 ? q5 + (o - P_BE1) : src;
    src = o >= P_G2 ? q6 + (o - P_G2) : src;
    src = o >= P_BE2 ? q7 + (o - P_BE2) : src;
    return *reinterpret_cast<const float4*>(src);
  };
  long long t_step = 0;
  if (fin.t_plus1 > 0) t_step = fin.t_plus1 - 1;
  else if (fin.ctl != nullptr) t_step = fin.ctl->t;
  float lsd = 0.f, ep = 0.f;  // sampling operands of lane a < A of wave 0 (policy blocks)
  unsigned warm_word = 0;
  {
    const InfLayer& w0 = stk.l[0].n[net];
    // (1) L2 warm-up: the kernel starts with a cold L2, and the blocks that share one (block b runs on XCD b % 8) want the
    // same weight lines at the same time. Each touches a different quarter of this net's lines first (one dword per
    // 128-byte line, <= 2 independent loads per thread), so that the fragment loads behind them find most lines already on
    // their way into the L2. Placement is a speed assumption only: every block still loads every fragment it uses.
    {
      const int share = (lin >> 3) & 3;
      const int L0 = tid * 4 + share, L1 = (tid + NTH) * 4 + share;  // line numbers in the concatenation of this net's weights
      const char* p0 = reinterpret_cast<const char*>(w0.win);
      const char* p1 = p0;
      int c = 0;
      auto seg = [&](const void* W, int bytes) {
        const int nlines = bytes >> 7;
        const char* base = reinterpret_cast<const char*>(W);
        p0 = (L0 >= c && L0 < c + nlines) ? base + ((size_t)(L0 - c) << 7) : p0;
        p1 = (L1 >= c && L1 < c + nlines) ? base + ((size_t)(L1 - c) << 7) : p1;
        c += nlines;
      };
      for (int l = 0; l < nl; ++l) {
        const InfLayer& wl = stk.l[l].n[net];
        seg(wl.win, 192 * 64 * (int)sizeof(T)); seg(wl.wo, 64 * 64 * (int)sizeof(T));
        seg(wl.w1, 256 * 64 * (int)sizeof(T)); seg(wl.w2, 64 * 256 * (int)sizeof(T));
      }
      seg(h.w0, 256 * 128 * (int)sizeof(T)); seg(h.w1, 256 * 256 * (int)sizeof(T)); seg(h.w2, 16 * 256 * (int)sizeof(T));
      p0 = warm ? p0 : reinterpret_cast<const char*>(w0.win);
      p1 = warm ? p1 : reinterpret_cast<const char*>(w0.win);
      warm_word = *reinterpret_cast<const unsigned*>(p0) ^ *reinterpret_cast<const unsigned*>(p1);
    }
    // (2) what the first phases need: token rows (the encoder kernel just wrote them), parameters, sampling operands.
    // Every load is unconditional from a clamped, always-valid address (a load inside a branch ends in a vmcnt(0) wait)
    const int r = tid >> 4, c4 = (tid & 15) * 4;  // 32 rows x 16 float4 = 512 threads
    const float4 xv = *reinterpret_cast<const float4*>(w0.xin + (row0 + (r < NT ? r : 0)) * TD + c4);
    const float4 pv = layer_params(w0);
    float4 hv;
    {
      const int o = min(tid, 127) * 4;
      const float *q0 = h.b0, *q1 = h.b1;
      hv = *reinterpret_cast<const float4*>(o >= P_H1 ? q1 + (o - P_H1) : q0 + o);
    }
    const float b2v = h.b2[min(tid, h.nout - 1)];
    // (this kernel is the rollout step: fin.ctl, fin.eps and fin.logstd are always set)
    lsd = fin.logstd[min(tid, fin.A - 1)];
    ep = fin.eps[(int64_t)s0 * fin.A + min(tid, fin.A - 1)];
    __builtin_amdgcn_sched_barrier(0);
    // (3) the weight fragments, in the order the phases consume them
    if constexpr (PRE) load_in(w0);
    __builtin_amdgcn_sched_barrier(0);
    {
      const float4 x0v = r < NT ? xv : float4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<float4*>(xs + r * LY::LDX + c4) = x0v;
      st4(xb + r * LDT + c4, x0v.x, x0v.y, x0v.z, x0v.w);
      if (r >= NT) st4(cb + r * LDT + c4, 0.f, 0.f, 0.f, 0.f);  // the attention only writes the 17 real context rows
    }
    if (tid < P_LAYER / 4) *reinterpret_cast<float4*>(prm + tid * 4) = pv;
    if (tid < 128) *reinterpret_cast<float4*>(prm_h + tid * 4) = hv;
    if (tid < 16) prm_h[P_H2 + tid] = tid < h.nout ? b2v : 0.f;
  }
  __syncthreads();
  T* f = reinterpret_cast<T*>(big);
  auto ln2rows = [&](const float* z, float* out, const float* g, const float* be, float* gout) {
    const int l16 = lane & 15, c4 = l16 * 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c4), bb = *reinterpret_cast<const float4*>(be + c4);
    const int r = wave * 4 + (lane >> 4);  // 32 rows = one step of 8 waves, a quarter wave per row
    float4 v = *reinterpret_cast<const float4*>(z + r * LY::LDX + c4);
    float s = (v.x + v.y) + (v.z + v.w);
    s += dpp_mov<0x128>(s); s += dpp_mov<0x124>(s); s += dpp_mov<0x122>(s); s += dpp_mov<0x121>(s);
    const float mean = s * (1.f / TD);
    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    float q2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    q2 += dpp_mov<0x128>(q2); q2 += dpp_mov<0x124>(q2); q2 += dpp_mov<0x122>(q2); q2 += dpp_mov<0x121>(q2);
    const float rs = 1.f / sqrtf(q2 * (1.f / TD) + 1e-5f);
    const float4 o = {fmaf(v.x * rs, gg.x, bb.x), fmaf(v.y * rs, gg.y, bb.y), fmaf(v.z * rs, gg.z, bb.z),
                      fmaf(v.w * rs, gg.w, bb.w)};
    *reinterpret_cast<float4*>(out + r * LY::LDX + c4) = o;
    st4(xb + r * LDT + c4, o.x, o.y, o.z, o.w);
    if (gout != nullptr && r < NT) *reinterpret_cast<float4*>(gout + (row0 + r) * TD + c4) = o;
  };
  if constexpr ((OPT & 1) != 0) {  // token_norm=True: out = token_ln(tokens), each quarter wave its own row, in place
    ln2rows(xs, xs, h.tn_g, h.tn_b, nullptr);
    __syncthreads();
  }
#pragma unroll
  for (int l = 0; l < NL; ++l) {  // the token rows stay in `xs` from one layer to the next
    const InfLayer& w = stk.l[l].n[net];
    const float* pl = prm + (l & 1) * P_LAYER;
    const bool more = l + 1 < nl;
    const InfLayer& wn = stk.l[more ? l + 1 : l].n[net];
    const float4 pnext = layer_params(wn);  // next layer's parameters: requested now, put into LDS when this layer is done
    if (l > 0) __syncthreads();
    if (l == 0) ROLL_STAMP(65);
    {  // in_proj: 12 column tiles, two row tiles each
      if constexpr (!PRE) load_in(w);
      if constexpr (PRE) { if (l == 0) { if (wave < 4) load_o(w); load_f1(w); } }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = wave + 8 * j;
        if (t < 12) {
          f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
          mm_held<T, 2, 2>(acc, xb, LDT, r_in[j], lane);
          const int n4 = t * 16 + qr;
          const float4 bb = *reinterpret_cast<const float4*>(pl + P_BIN + n4);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 16 + fr;
            const float v0 = acc[mt][0] + bb.x, v1 = acc[mt][1] + bb.y, v2 = acc[mt][2] + bb.z, v3 = acc[mt][3] + bb.w;
            if (t < 4) st4(qb + row * LDT + n4, v0, v1, v2, v3);
            else if (t < 8) st4(kb + row * LDT + (n4 - TD), v0, v1, v2, v3);
            else {  // values, transposed for the P V product; keys past the 17 tokens are zeros
              T* vc = vt + (n4 - 2 * TD) * ATT_LDV + row;
              const bool ok = row < NT;
              vc[0] = (T)(ok ? v0 : 0.f); vc[ATT_LDV] = (T)(ok ? v1 : 0.f);
              vc[2 * ATT_LDV] = (T)(ok ? v2 : 0.f); vc[3 * ATT_LDV] = (T)(ok ? v3 : 0.f);
            }
          }
        }
      }
      if constexpr (PRE) { if (more) load_in(wn); else load_h0a(); }
    }
    __syncthreads();
    if (l == 0) ROLL_STAMP(66);
    if constexpr (PRE) { if (l == 0 && wave < 4) load_f2(w); if (!more) load_h1(0); }
    if (wave < (NT > 16 ? 2 : 1))  // attention on the matrix cores: wave = query tile (tokens 0..15 | token 16)
      attn_tile<T, NT>(qb + wave * 16 * LDT, kb, LDT, vt, pbuf + wave * 16 * ATT_LDV, cb + wave * 16 * LDT, LDT, lane, wave * 16,
                       wave == 0 ? 16 : NT - 16, nullptr, nullptr);
    if (l == 0) { ROLL_STAMP(67); ROLL_STAMP(68); }
    __syncthreads();
    if (l == 0) ROLL_STAMP(69);
    {  // out_proj + residual -> z (in `big`, fp32 [32][LDX]): column tile = wave (waves 0..3), both row tiles: the 8 KB of
      // weights enter the CU once
      if (wave < 4) {
        if constexpr (!PRE) load_o(w);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mm_held<T, 2, 2>(acc, cb, LDT, r_o, lane);
        if constexpr (PRE) { if (more) load_o(wn); }
        const int n4 = wave * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(pl + P_BO + n4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 16 + fr;
          const float4 xr = *reinterpret_cast<const float4*>(xs + row * LY::LDX + n4);
          st4(big + row * LY::LDX + n4, xr.x + acc[mt][0] + bb.x, xr.y + acc[mt][1] + bb.y, xr.z + acc[mt][2] + bb.z,
              xr.w + acc[mt][3] + bb.w);
        }
      }
      if constexpr (PRE) { if (!more) load_h1(1); }
    }
    __syncthreads();
    if (l == 0) ROLL_STAMP(70);
    ln2rows(big, xs, pl + P_G1, pl + P_BE1, nullptr);  // x1 -> xs
    __syncthreads();
    if (l == 0) ROLL_STAMP(71);
    {  // linear1 + ReLU -> f (T): 16 column tiles, two per wave
      if constexpr (!PRE) load_f1(w);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = wave + 8 * j;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mm_held<T, 2, 2>(acc, xb, LDT, r_f1[j], lane);
        if (l == 0 && j == 0) ROLL_STAMP(75);
        const int n4 = t * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(pl + P_B1 + n4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          st4(f + (mt * 16 + fr) * LY::LDF + n4, fmaxf(acc[mt][0] + bb.x, 0.f), fmaxf(acc[mt][1] + bb.y, 0.f),
              fmaxf(acc[mt][2] + bb.z, 0.f), fmaxf(acc[mt][3] + bb.w, 0.f));
      }
      if (l == 0) ROLL_STAMP(76);
      if constexpr (PRE) { if (more) load_f1(wn); else load_h0b(); }
      if (l == 0) ROLL_STAMP(77);
    }
    __syncthreads();
    if (l == 0) ROLL_STAMP(72);
    if (wave < 4) {  // linear2 + residual -> z2 (in `cx`): column tile = wave, both row tiles
      if constexpr (!PRE) load_f2(w);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 8>(acc, f, LY::LDF, r_f2, lane);
      if constexpr (PRE) { if (more) load_f2(wn); else if (wave == 0) load_h2(); }
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(pl + P_B2 + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 16 + fr;
        const float4 xr = *reinterpret_cast<const float4*>(xs + row * LY::LDX + n4);
        st4(cx + row * LY::LDX + n4, xr.x + acc[mt][0] + bb.x, xr.y + acc[mt][1] + bb.y, xr.z + acc[mt][2] + bb.z,
            xr.w + acc[mt][3] + bb.w);
      }
    }
    __syncthreads();
    if (l == 0) ROLL_STAMP(73);
    ln2rows(cx, xs, pl + P_G2, pl + P_BE2, w.xout);  // -> xs (next layer / heads) and the net's token tensor
    if (tid < P_LAYER / 4) *reinterpret_cast<float4*>(prm + ((l + 1) & 1) * P_LAYER + tid * 4) = pnext;
    if (l == 0) ROLL_STAMP(74);
  }
  // use_pytorch_encoder=True: nn.TransformerEncoder's final norm on the rows the last norm2 left (each lane re-reads what it wrote)
  if constexpr ((OPT & 2) != 0) ln2rows(xs, xs, h.fn_g, h.fn_b, nullptr);
  // ---- head on this sample: [state token | mean of the 16 depth tokens] -> 256 -> 256 -> nout (fragment row 0 carries data)
  float* pooled = big;                                              // [16][LDP] fp32, row 0 = this sample
  T* h1 = reinterpret_cast<T*>(big + 16 * LY::LDP);                 // [16][LDF]
  T* h2 = h1 + 16 * LY::LDF;
  float* so = reinterpret_cast<float*>(h2 + 16 * LY::LDF);          // [16] last-layer outputs of row 0
  __syncthreads();
  ROLL_STAMP(80);
  constexpr bool VIS = NT == 16;
  if (tid < (VIS ? TD : 2 * TD)) {  // rows 1..15 of the operand tiles are never read back: only fragment row 0 is filled
    float v;
    if (!VIS && tid < TD) v = xs[tid];
    else {
      const int d = VIS ? tid : tid - TD;
      float s = 0.f, m = -INFINITY;
#pragma unroll
      for (int i = VIS ? 0 : 1; i < NT; ++i) {
        const float t = xs[i * LY::LDX + d];
        s += t;
        m = fmaxf(m, t);
      }
      v = h.max_pool ? m : s * (1.f / 16.f);  // max_pool=True: `.max(dim=0)[0]` over the same tokens (nets.py:1022-1023, 886-887)
    }
    pooled[tid] = v;
  }
  __syncthreads();
  ROLL_STAMP(81);
  auto store_h = [&](T* dst, const f32x4& a, const float* bias, int t) {
    const int n4 = t * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(bias + n4);
    if (fr == 0) st4(dst + n4, fmaxf(a[0] + bb.x, 0.f), fmaxf(a[1] + bb.y, 0.f), fmaxf(a[2] + bb.z, 0.f), fmaxf(a[3] + bb.w, 0.f));
  };
  {
    if constexpr (!PRE) { load_h0a(); load_h0b(); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      if constexpr (VIS) {
        const frag_t fb[2] = {r_in[j][0], r_in[j][1]};
        mm_held<T, 1, 2>(acc, pooled, LY::LDP, fb, lane);
      } else {
        const frag_t fb[4] = {r_in[j][0], r_in[j][1], r_f1[j][0], r_f1[j][1]};
        mm_held<T, 1, 4>(acc, pooled, LY::LDP, fb, lane);
      }
      store_h(h1, acc[0], prm_h + P_H0, wave + 8 * j);
    }
  }
  __syncthreads();
  ROLL_STAMP(82);
  {
    if constexpr (!PRE) { load_h1(0); load_h1(1); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 1, 8>(acc, h1, LY::LDF, r_h1[j], lane);
      store_h(h2, acc[0], prm_h + P_H1, wave + 8 * j);
    }
  }
  __syncthreads();
  ROLL_STAMP(83);
  if (wave == 0) {
    if constexpr (!PRE) load_h2();
    // epilogue operands requested before the last GEMM: lane a < A carries action dimension a
    const int A = fin.A, i = s0;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    mm_held<T, 1, 8>(acc, h2, LY::LDF, r_f2, lane);
    if (fr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = qr + r;
        const float v = c < h.nout ? acc[0][r] + prm_h[P_H2 + c] : 0.f;
        so[c] = v;
        h.out[(int64_t)s0 * OUT_LD + c] = v;
      }
    }
    if (fin.ctl != nullptr) {
      // ---- rollout step epilogue (GaussianContPolicyBase.explore, continuous_policy.py:85-125, and the collector's value
      // read-out, collector/on_policy.py:95-100): same expressions and summation order as act_finish_kernel /
      // infer_layer_kernel, the per-dimension terms evaluated by A lanes side by side
      __builtin_amdgcn_wave_barrier();  // `so` was written by this wave: LDS operations of one wave execute in order
      if (net == 0) {
        const float mu = so[lane < A ? lane : 0];
        const float ls = fminf(fmaxf(lsd, LOG_SIG_MIN), LOG_SIG_MAX);
        const float sg = expf(ls);
        const float et = 0.5f + HALF_LOG_2PI + logf(sg);
        float act = fmaf(sg, ep, mu);
        if (fin.tanh_action) act = tanhf(act);  // TanhNormal.rsample (distribution.py:61-80); eps = 0: eval_act's tanh(mean)
        const float d = (fin.tanh_action ? tanh_pre(act) : act) - mu;
        const float lt = -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
        const float ct = fin.tanh_action ? tanh_corr(act) : 0.f;
        if (lane < A) {
          fin.action[(int64_t)i * A + lane] = act;
          fin.mean[(int64_t)i * A + lane] = mu;
          fin.stdv[(int64_t)i * A + lane] = sg;
          if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + lane] = act;
        }
        float e = 0.f, lp = 0.f;
        if (fin.tanh_action) {  // (act_finish_kernel's order: the Gaussian term in, the tanh correction out, dimension by dimension)
          for (int a = 0; a < A; ++a) { e += lane_bcast(et, a); lp += lane_bcast(lt, a); lp -= lane_bcast(ct, a); }
        } else {
          for (int a = 0; a < A; ++a) { e += lane_bcast(et, a); lp += lane_bcast(lt, a); }  // a = 0 .. A-1, in order
        }
        if (lane == 0) {
          fin.ent[i] = e;
          if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + i] = lp;
        }
      } else if (lane == 0) {
        const float v = so[0];
        fin.value[i] = v;
        if (fin.values_roll != nullptr) fin.values_roll[t_step * E + i] = v;
      }
      ROLL_STAMP(84);
      if (fin.t_plus1 > 0) {  // nobody in this launch reads the cursor: one plain store keeps it in step for graph replays
        if (lane == 0 && s0 == 0 && net == 0) fin.ctl->t = t_step + 1;
      } else if (lane == 0) {  // the last block to get here advances the step cursor: every block read it at entry
        __threadfence();
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&fin.ctl->done), 1ull);
        if (done == (unsigned long long)(gridDim.x * gridDim.y) - 1) {
          fin.ctl->done = 0;
          fin.ctl->t = t_step + 1;
        }
      }
      ROLL_STAMP(85);
    }
  }
  if (warm_word == 0x7fc00123u && fin.ctl == nullptr) h.out[(int64_t)s0 * OUT_LD + 15] = 0.f;  // keeps the warm-up loads alive
}


// ------------------------------------------------------------------------------------------ rollout encoder (16 waves)
// infer_encoder_kernel's inference path with every phase spread over 16 waves (see rollout_stack_kernel): conv1 one row
// tile per wave, conv2 one (row tile, column tile) per wave, conv3 column tile x K-quarter per wave + an LDS sum, the
// proprio MLP one column tile per wave. Same arithmetic except conv3's fp32 partial sums (4 K-quarters added in order).
template <typename T>
__global__ __launch_bounds__(1024) void rollout_encoder_kernel(const ActCtl* __restrict__ ctl, const float* __restrict__ obs,
                                                               int E, InfEnc w, float* __restrict__ state_roll,
                                                               T* __restrict__ image_roll, float* __restrict__ x0) {
  typedef typename Frag<T>::type frag_t;
  typedef InfEncLds<T> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8, qr = (lane >> 4) * 4;
  const int64_t slot0 = (int64_t)ctl->t * E;
  const int D = w.S + LY::IMG;

  if ((int)blockIdx.x >= E) {
    // ---------------- proprio branch, 32 rows per block: Linear+ReLU, Linear+ReLU, state_projector+ReLU -> token 0
    constexpr int MR = LY::MLP_ROWS;
    const int r0 = ((int)blockIdx.x - E) * MR;
    float* sin = reinterpret_cast<float*>(smem);
    T* h1 = reinterpret_cast<T*>(smem + (size_t)MR * LY::LDS_IN * 4);
    T* h2 = h1 + MR * LY::LDH;
    for (int idx = tid; idx < MR * 128; idx += 1024) {
      const int r = idx >> 7, c = idx & 127;
      const bool ok = r0 + r < E;
      float v = 0.f;
      if (ok && c < w.S) v = obs[(int64_t)(r0 + r) * D + c];
      if (ok && c < w.Sp) state_roll[(slot0 + r0 + r) * w.Sp + c] = v;
      sin[r * LY::LDS_IN + c] = v;
    }
    __syncthreads();
    const int nt[1] = {wave};
    f32x4 acc[2][1];
    auto store_h = [&](T* h, const float* bias) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bias + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        st4(h + (mt * 16 + fr) * LY::LDH + n4, fmaxf(acc[mt][0][0] + bb.x, 0.f), fmaxf(acc[mt][0][1] + bb.y, 0.f),
            fmaxf(acc[mt][0][2] + bb.z, 0.f), fmaxf(acc[mt][0][3] + bb.w, 0.f));
    };
    zero_acc(acc);
    if (w.Kp1 == 128) block_gemm<T, 2, 1, 4>(acc, sin, LY::LDS_IN, (const T*)w.wf1, 128, nt, lane);
    else block_gemm<T, 2, 1, 2>(acc, sin, LY::LDS_IN, (const T*)w.wf1, 64, nt, lane);
    store_h(h1, w.bf1);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 2, 1, 8>(acc, h1, LY::LDH, (const T*)w.wf2, 256, nt, lane);
    store_h(h2, w.bf2);
    __syncthreads();
    if (wave < 8) {
      const int ntp[1] = {wave & 3}, mt = wave >> 2;
      f32x4 ap[1][1];
      zero_acc(ap);
      block_gemm<T, 1, 1, 8>(ap, h2 + mt * 16 * LY::LDH, LY::LDH, (const T*)w.wpr, 256, ntp, lane);
      const int n4 = ntp[0] * 16 + qr, row = r0 + mt * 16 + fr;
      const float4 bb = *reinterpret_cast<const float4*>(w.bpr + n4);
      if (row < E)
        st4(x0 + ((int64_t)row * NTOK) * TD + n4, fmaxf(ap[0][0][0] + bb.x, 0.f), fmaxf(ap[0][0][1] + bb.y, 0.f),
            fmaxf(ap[0][0][2] + bb.z, 0.f), fmaxf(ap[0][0][3] + bb.w, 0.f));
    }
    return;
  }

  // ---------------- depth branch, one sample per block
  const int b = blockIdx.x;
  T* img = reinterpret_cast<T*>(smem);
  T* c1 = img + LY::IMG;
  T* c2 = c1 + LY::C1;
  T* c3 = c2 + LY::C2;
  ROLL_STAMP(96);
  {
    const float* src = obs + (int64_t)b * D + w.S;  // 16-byte aligned only when S % 4 == 0: dword-aligned vector loads (ld4u)
    T* roll = image_roll + (slot0 + b) * (int64_t)LY::IMG;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld4u(src + (tid + k * 1024) * 4);  // 4096 float4 per image: four per thread, all in flight
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 1024;
      st4(img + i * 4, v[k].x, v[k].y, v[k].z, v[k].w);
      st4(roll + i * 4, v[k].x, v[k].y, v[k].z, v[k].w);
    }
  }
  __syncthreads();
  ROLL_STAMP(97);
  if (wave < 15) {  // conv1: 225 pixels = 15 row tiles, one per wave; K = (c,ky,kx) = 256, N = 32
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int p = min(wave * 16 + fr, 224);
    const int pbase = (p / 15) * 4 * 64 + (p % 15) * 4;
    frag_t ring[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int j = 0; j < 2; ++j) ring[d][j] = *reinterpret_cast<const frag_t*>((const T*)w.w1 + (j * 16 + fr) * 256 + d * 32 + fg);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = ks * 32 + fg, c = k0 >> 6, ky = (k0 >> 3) & 7;
      const frag_t fb0 = ring[ks & 3][0], fb1 = ring[ks & 3][1];
      if (ks + 4 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          ring[ks & 3][j] = *reinterpret_cast<const frag_t*>((const T*)w.w1 + (j * 16 + fr) * 256 + (ks + 4) * 32 + fg);
      }
      const frag_t fa = afrag_t(img + c * 4096 + ky * 64 + pbase);
      mma_k32(acc[0], fb0, fa);
      mma_k32(acc[1], fb1, fa);
    }
    const int pp = wave * 16 + fr;
    if (pp < 225) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n4 = j * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(w.b1 + n4);
        st4(c1 + pp * LY::LD1 + n4, fmaxf(acc[j][0] + bb.x, 0.f), fmaxf(acc[j][1] + bb.y, 0.f), fmaxf(acc[j][2] + bb.z, 0.f),
            fmaxf(acc[j][3] + bb.w, 0.f));
      }
    }
  }
  __syncthreads();
  ROLL_STAMP(98);
  if (wave < 12) {  // conv2: 36 pixels (3 row tiles) x 4 column tiles, one pair per wave; K = (ky,kx,c) = 512
    const int mt = wave >> 2, nt = wave & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int p = min(mt * 16 + fr, 35);
    const int pb = ((p / 6) * 2 * 15 + (p % 6) * 2) * LY::LD1;
    const T* w2row = (const T*)w.w2 + (nt * 16 + fr) * 512 + fg;
    frag_t ring[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) ring[d] = *reinterpret_cast<const frag_t*>(w2row + d * 32);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int ky = ks >> 2, kx = ks & 3;  // 32 channels per tap == one K=32 step
      const frag_t fb = ring[ks & 7];
      if (ks + 8 < 16) ring[ks & 7] = *reinterpret_cast<const frag_t*>(w2row + (ks + 8) * 32);
      const frag_t fa = afrag_t(c1 + pb + (ky * 15 + kx) * LY::LD1 + fg);
      mma_k32(acc, fb, fa);
    }
    const int pp = mt * 16 + fr, n4 = nt * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(w.b2 + n4);
    if (pp < 36)
      st4(c2 + pp * LY::LD2 + n4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f), fmaxf(acc[2] + bb.z, 0.f),
          fmaxf(acc[3] + bb.w, 0.f));
  }
  __syncthreads();
  ROLL_STAMP(99);
  float* part = reinterpret_cast<float*>(img);  // [4 K-quarters][16 pixels][64] fp32 partial sums (the image is dead)
  {  // conv3: 16 pixels, K = (ky,kx,c) = 576 = 18 steps: wave = (column tile, K-quarter of 5/5/5/3 steps)
    const int nt = wave & 3, kq = wave >> 2;
    const int ks0 = kq * 5, ks1 = min(18, ks0 + 5);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int pb = ((fr >> 2) * 6 + (fr & 3)) * LY::LD2;
    const T* w3row = (const T*)w.w3 + (nt * 16 + fr) * 576 + fg;
    frag_t fbv[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) fbv[d] = *reinterpret_cast<const frag_t*>(w3row + min(ks0 + d, 17) * 32);
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int ks = ks0 + d;
      if (ks < ks1) {
        const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3, c0 = (ks & 1) * 32 + fg;
        const frag_t fa = afrag_t(c2 + pb + (ky * 6 + kx) * LY::LD2 + c0);
        mma_k32(acc, fbv[d], fa);
      }
    }
    st4(part + (kq * 16 + fr) * 64 + nt * 16 + qr, acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  ROLL_STAMP(100);
  {  // sum of the four K-quarters + bias + ReLU -> c3: one output per thread
    const int pix = tid >> 6, n = tid & 63;
    const float v = ((part[pix * 64 + n] + part[(16 + pix) * 64 + n]) + part[(32 + pix) * 64 + n]) + part[(48 + pix) * 64 + n];
    c3[pix * LY::LD2 + n] = (T)fmaxf(v + w.b3[n], 0.f);
  }
  __syncthreads();
  ROLL_STAMP(101);
  if (wave < 4) {  // depth_up_conv (1x1, no activation) -> tokens 1..16
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const frag_t fb = *reinterpret_cast<const frag_t*>((const T*)w.wup + (wave * 16 + fr) * 64 + ks * 32 + fg);
      const frag_t fa = afrag_t(c3 + fr * LY::LD2 + ks * 32 + fg);
      mma_k32(acc, fb, fa);
    }
    const int n4 = wave * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(w.bup + n4);
    st4(x0 + ((int64_t)b * NTOK + 1 + fr) * TD + n4, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
  }
  ROLL_STAMP(102);
}


// ------------------------------------------------------------------------------------------ rollout encoder, weights once per CU
// rollout_encoder_kernel with its weight traffic cut to what the block needs ONCE. Phase stamps of that kernel
// (tools/probe/stamps_rollout.py: 40 K cycles per depth block, conv1 16.4 K, conv2 13.0 K, conv3 6.2 K) against its MFMA work
// (a few hundred MFMAs) said the time is the weight stream: all 15 conv1 waves pulled the same 16 KB of w1 (240 KB per block),
// the three row-tile waves of a conv2 column tile the same 16 KB (192 KB), and a CU only takes ~16-25 B/clk of such loads.
// Here every byte enters the CU once, as whole-cache-line fragment-order reads (PK_FRAG packs) requested at kernel entry in
// the order of use: the observation row, w1 and w2 into LDS (16 + 64 KB, shared by all waves), w3 / w_up straight into the
// registers of the wave that multiplies with them; biases are staged in LDS (a late global load would wait for every older
// one). bf16 only (the fp32 parity mode keeps rollout_encoder_kernel: its fragments are twice the size).
struct InfEncFrag {
  const void *w1, *w2, *w3, *wup;          // fragment-order packs: [2][8][64] [4][16][64] [4][18][64] [4][2][64] fragments
  const float *b1, *b2, *b3, *bup;
  const void *wf1, *wf2, *wpr;             // proprio MLP: [16][4][64] [16][8][64] [4][8][64] fragments
  const float *bf1, *bf2, *bpr;
  int S, Sp;
};
struct RollEnc2Lds {
  typedef InfEncLds<__bf16> E;
  static constexpr size_t w1_b = 2 * 8 * 64 * 16, w2_b = 4 * 16 * 64 * 16, bias_b = 256 * 4;
  static constexpr size_t conv_bytes = E::conv_bytes + w1_b + w2_b + bias_b;
  static constexpr int LDB = 128 + 8;      // bf16 proprio rows
  static constexpr size_t mlp_bytes = (size_t)32 * LDB * 2 + (size_t)2 * 32 * E::LDH * 2 + 576 * 4;
  static constexpr size_t bytes = conv_bytes > mlp_bytes ? conv_bytes : mlp_bytes;
};
// MODE: what leaves the block. ENC_TOK17: LocoTransformer tokens (fp32 x0[E][17][64]: token 0 = proprio branch, 1..16 = depth
// up-conv). ENC_TOK16: the vision-only Transformer's 16 depth tokens (x0[E][16][64], no proprio blocks in the grid).
// ENC_FLAT: conv3's NHWC flatten (k = pixel*64 + c) as operand-type rows featv[E][1024] for the dense layer that follows
// (NatureEncoder(flatten), base.py:304-342), no proprio blocks. ENC_FUSE: the NatureFuseEncoder (base.py:345-385): featv as
// ENC_FLAT, and the proprio blocks stop after the second Linear+ReLU and write featp[E][512] columns 256..511 (the right half
// of the concat the head reads; rollout_linear_kernel fills the left half with the visual projector's output).
enum { ENC_TOK17 = 0, ENC_FUSE = 1, ENC_FLAT = 2, ENC_TOK16 = 3 };
// featv / featp rows are handed to the dense launches in MFMA A-fragment order, so that a consumer wave reads a whole
// fragment as one contiguous 1 KB block: element (row, k) of a [rows][32*KS] operand sits at act_frag_off(row, k, KS)
// (a row-major [E][1024] operand would be read 16 bytes per lane at a 2 KB row stride: measured 6 us per row tile)
__device__ __forceinline__ int64_t act_frag_off(int row, int k, int KS) {
  return ((((int64_t)(row >> 4) * KS + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (row & 15)) << 3) + (k & 7);
}
// IMG16: the observation arrives SPLIT (v4l_actor_step_split) — proprio rows [E][ld_obs] fp32 at `obs`, the depth stacks
// [E][ld16] already in bf16 at `img16` (the collector's host threads round fp64 -> fp32 -> bf16, exactly the two roundings the
// fp32 row path applies: torch.Tensor(ob) on the host, the cast to the operand type here) — half the bytes over PCIe when the
// kernel reads pinned host memory in place. Otherwise one fp32 row [ld_obs = S + C*H*W] per sample.
// T: the 16-bit operand type (__bf16 | _Float16; both share InfEncLds<__bf16>'s byte layout).
template <typename T, int MODE, bool IMG16 = false>
__global__ __launch_bounds__(1024) void rollout_encoder2_kernel(const ActCtl* __restrict__ ctl, const float* __restrict__ obs,
                                                                int ld_obs, const T* __restrict__ img16, int64_t ld16,
                                                                int E, InfEncFrag w, float* __restrict__ state_roll,
                                                                T* __restrict__ image_roll, float* __restrict__ x0,
                                                                T* __restrict__ featv, T* __restrict__ featp,
                                                                long long t_plus1) {
  static_assert(sizeof(T) == 2, "16-bit operand types only");
  typedef typename Frag<T>::type frag_t;
  typedef InfEncLds<T> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8, qr = (lane >> 4) * 4;
  const int64_t slot0 = (t_plus1 > 0 ? t_plus1 - 1 : ctl->t) * (int64_t)E;  // (InfFinish::t_plus1)
  const int D = ld_obs;
  auto gfrag = [&](const void* W, int idx) -> frag_t { return reinterpret_cast<const frag_t*>(W)[idx * 64 + lane]; };

  if ((int)blockIdx.x >= E) {
    // ---------------- proprio branch, 32 rows per block: Linear+ReLU, Linear+ReLU, state_projector+ReLU -> token 0
    constexpr int MR = LY::MLP_ROWS, LDB = RollEnc2Lds::LDB;
    const int r0 = ((int)blockIdx.x - E) * MR;
    T* sb = reinterpret_cast<T*>(smem);                         // [32][LDB] proprio rows (T)
    T* h1 = sb + MR * LDB;
    T* h2 = h1 + MR * LY::LDH;
    float* bs = reinterpret_cast<float*>(h2 + MR * LY::LDH);    // bf1[256] | bf2[256] | bpr[64]
    float sv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // 32 x 128 values, 4 per thread; unconditional loads from clamped addresses
      const int idx = tid + k * 1024, r = idx >> 7, c = idx & 127;
      sv[k] = obs[(int64_t)min(r0 + r, E - 1) * D + min(c, w.S - 1)];
    }
    float bv = 0.f;
    {
      const int o = min(tid, 575);
      const float *q0 = w.bf1, *q1 = w.bf2, *q2 = w.bpr;
      bv = *(o >= 512 ? q2 + (o - 512) : o >= 256 ? q1 + (o - 256) : q0 + o);
    }
    __builtin_amdgcn_sched_barrier(0);
    frag_t r1[4], r2[8], r3[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) r1[ks] = gfrag(w.wf1, wave * 4 + ks);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r2[ks] = gfrag(w.wf2, wave * 8 + ks);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 1024, r = idx >> 7, c = idx & 127;
      const bool ok = r0 + r < E;
      const float v = ok && c < w.S ? sv[k] : 0.f;
      if (ok && c < w.Sp) state_roll[(slot0 + r0 + r) * w.Sp + c] = v;
      sb[r * LDB + c] = (T)v;
    }
    if (tid < 576) bs[tid] = bv;
    __syncthreads();
    auto store_h = [&](T* h, const f32x4 (&acc)[2], const float* bias) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bias + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        st4(h + (mt * 16 + fr) * LY::LDH + n4, fmaxf(acc[mt][0] + bb.x, 0.f), fmaxf(acc[mt][1] + bb.y, 0.f),
            fmaxf(acc[mt][2] + bb.z, 0.f), fmaxf(acc[mt][3] + bb.w, 0.f));
    };
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 4>(acc, sb, LDB, r1, lane);
      if (MODE == ENC_TOK17 && wave < 4) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) r3[ks] = gfrag(w.wpr, wave * 8 + ks);
      }
      store_h(h1, acc, bs);
    }
    __syncthreads();
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 8>(acc, h1, LY::LDH, r2, lane);
      if constexpr (MODE == ENC_FUSE) {  // the concat's right half, operand type
        const int n4 = wave * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(bs + 256 + n4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = r0 + mt * 16 + fr;
          if (row < E)
            st4(featp + act_frag_off(row, 256 + n4, 16), fmaxf(acc[mt][0] + bb.x, 0.f), fmaxf(acc[mt][1] + bb.y, 0.f),
                fmaxf(acc[mt][2] + bb.z, 0.f), fmaxf(acc[mt][3] + bb.w, 0.f));
        }
        return;
      }
      store_h(h2, acc, bs + 256);
    }
    __syncthreads();
    if (wave < 4) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 8>(acc, h2, LY::LDH, r3, lane);
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bs + 512 + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = r0 + mt * 16 + fr;
        if (row < E)
          st4(x0 + ((int64_t)row * NTOK) * TD + n4, fmaxf(acc[mt][0] + bb.x, 0.f), fmaxf(acc[mt][1] + bb.y, 0.f),
              fmaxf(acc[mt][2] + bb.z, 0.f), fmaxf(acc[mt][3] + bb.w, 0.f));
      }
    }
    return;
  }

  // ---------------- depth branch, one sample per block
  const int b = blockIdx.x;
  T* img = reinterpret_cast<T*>(smem);
  T* c1 = img + LY::IMG;
  T* c2 = c1 + LY::C1;
  T* c3 = c2 + LY::C2;
  frag_t* w1s = reinterpret_cast<frag_t*>(smem + LY::conv_bytes);   // [2][8][64]
  frag_t* w2s = w1s + 2 * 8 * 64;                                    // [4][16][64]
  float* bs = reinterpret_cast<float*>(w2s + 4 * 16 * 64);           // b1[32] | b2[64] | b3[64] | bup[64]
  ROLL_STAMP(96);
  float4 v[4];
  frag_t u[2];
  if constexpr (IMG16) {
    const frag_t* row = reinterpret_cast<const frag_t*>(img16 + (int64_t)b * ld16);
#pragma unroll
    for (int k = 0; k < 2; ++k) u[k] = row[tid + k * 1024];  // 2048 x 16 bytes per image: two per thread
  } else {
    const float* row = obs + (int64_t)b * D + w.S;  // 16-byte aligned only when S % 4 == 0: dword-aligned vector loads (ld4u)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld4u(row + (tid + k * 1024) * 4);  // 4096 float4 per image: four per thread, all in flight
  }
  const frag_t w1v = reinterpret_cast<const frag_t*>(w.w1)[tid];
  float bv;
  {
    const int o = min(tid, 223);
    const float *q0 = w.b1, *q1 = w.b2, *q2 = w.b3, *q3 = w.bup;
    bv = *(o >= 160 ? q3 + (o - 160) : o >= 96 ? q2 + (o - 96) : o >= 32 ? q1 + (o - 32) : q0 + o);
  }
  __builtin_amdgcn_sched_barrier(0);
  frag_t w2v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w2v[k] = reinterpret_cast<const frag_t*>(w.w2)[tid + k * 1024];
  // conv3: wave = (column tile, K-quarter of 5/5/5/3 k-steps); up-conv: waves 0..3, one column tile each
  const int nt3 = wave & 3, kq = wave >> 2, ks0 = kq * 5, ks1 = min(18, ks0 + 5);
  frag_t w3v[5], wuv[2];
#pragma unroll
  for (int d = 0; d < 5; ++d) w3v[d] = gfrag(w.w3, nt3 * 18 + min(ks0 + d, 17));
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wuv[ks] = gfrag(MODE == ENC_TOK17 || MODE == ENC_TOK16 ? w.wup : w.w3, nt3 * 2 + ks);
  __builtin_amdgcn_sched_barrier(0);
  {
    T* roll = image_roll + (slot0 + b) * (int64_t)LY::IMG;
    if constexpr (IMG16) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = tid + k * 1024;
        *reinterpret_cast<frag_t*>(img + i * 8) = u[k];
        *reinterpret_cast<frag_t*>(roll + i * 8) = u[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = tid + k * 1024;
        st4(img + i * 4, v[k].x, v[k].y, v[k].z, v[k].w);
        st4(roll + i * 4, v[k].x, v[k].y, v[k].z, v[k].w);
      }
    }
    w1s[tid] = w1v;
    if (tid < 224) bs[tid] = bv;
    if constexpr (MODE == ENC_FLAT || MODE == ENC_TOK16) {  // vision-only: the all-zero dummy state row of this slot
      if (tid < w.Sp) state_roll[(slot0 + b) * w.Sp + tid] = 0.f;
    }
  }
  __syncthreads();
  ROLL_STAMP(97);
  if (wave < 15) {  // conv1: 225 pixels = 15 row tiles, one per wave; K = (c,ky,kx) = 256, N = 32
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int p = min(wave * 16 + fr, 224);
    const int pbase = (p / 15) * 4 * 64 + (p % 15) * 4;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = ks * 32 + fg, c = k0 >> 6, ky = (k0 >> 3) & 7;
      const frag_t fa = afrag_t(img + c * 4096 + ky * 64 + pbase);
      mma_k32(acc[0], w1s[(0 * 8 + ks) * 64 + lane], fa);
      mma_k32(acc[1], w1s[(1 * 8 + ks) * 64 + lane], fa);
    }
    const int pp = wave * 16 + fr;
    if (pp < 225) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n4 = j * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(bs + n4);
        st4(c1 + pp * LY::LD1 + n4, fmaxf(acc[j][0] + bb.x, 0.f), fmaxf(acc[j][1] + bb.y, 0.f), fmaxf(acc[j][2] + bb.z, 0.f),
            fmaxf(acc[j][3] + bb.w, 0.f));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) w2s[tid + k * 1024] = w2v[k];
  __syncthreads();
  ROLL_STAMP(98);
  if (wave < 12) {  // conv2: 36 pixels (3 row tiles) x 4 column tiles, one pair per wave; K = (ky,kx,c) = 512
    const int mt = wave >> 2, nt = wave & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int p = min(mt * 16 + fr, 35);
    const int pb = ((p / 6) * 2 * 15 + (p % 6) * 2) * LY::LD1;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int ky = ks >> 2, kx = ks & 3;  // 32 channels per tap == one K=32 step
      const frag_t fa = afrag_t(c1 + pb + (ky * 15 + kx) * LY::LD1 + fg);
      mma_k32(acc, w2s[(nt * 16 + ks) * 64 + lane], fa);
    }
    const int pp = mt * 16 + fr, n4 = nt * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(bs + 32 + n4);
    if (pp < 36)
      st4(c2 + pp * LY::LD2 + n4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f), fmaxf(acc[2] + bb.z, 0.f),
          fmaxf(acc[3] + bb.w, 0.f));
  }
  __syncthreads();
  ROLL_STAMP(99);
  float* part = reinterpret_cast<float*>(img);  // [4 K-quarters][16 pixels][64] fp32 partial sums (the image is dead)
  {  // conv3: 16 pixels, K = (ky,kx,c) = 576 = 18 steps
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int pb = ((fr >> 2) * 6 + (fr & 3)) * LY::LD2;
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int ks = ks0 + d;
      if (ks < ks1) {
        const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3, c0 = (ks & 1) * 32 + fg;
        const frag_t fa = afrag_t(c2 + pb + (ky * 6 + kx) * LY::LD2 + c0);
        mma_k32(acc, w3v[d], fa);
      }
    }
    st4(part + (kq * 16 + fr) * 64 + nt3 * 16 + qr, acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  ROLL_STAMP(100);
  {  // sum of the four K-quarters + bias + ReLU -> c3: one output per thread
    const int pix = tid >> 6, n = tid & 63;
    const float s4 = ((part[pix * 64 + n] + part[(16 + pix) * 64 + n]) + part[(32 + pix) * 64 + n]) + part[(48 + pix) * 64 + n];
    const T o = (T)fmaxf(s4 + bs[96 + n], 0.f);
    if constexpr (MODE == ENC_FUSE || MODE == ENC_FLAT) { featv[act_frag_off(b, tid, 32)] = o; return; }
    c3[pix * LY::LD2 + n] = o;
  }
  __syncthreads();
  ROLL_STAMP(101);
  constexpr int TOKS = MODE == ENC_TOK16 ? 16 : NTOK, TOK0 = MODE == ENC_TOK16 ? 0 : 1;
  if (wave < 4) {  // depth_up_conv (1x1, no activation) -> tokens 1..16 (vision-only: 0..15)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const frag_t fa = afrag_t(c3 + fr * LY::LD2 + ks * 32 + fg);
      mma_k32(acc, wuv[ks], fa);
    }
    const int n4 = wave * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(bs + 160 + n4);
    st4(x0 + ((int64_t)b * TOKS + TOK0 + fr) * TD + n4, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
  }
  ROLL_STAMP(102);
}


// ------------------------------------------------------------------------------------------ training encoder, persistent blocks
// The update's encoder forward (infer_encoder_kernel launched one 4-wave block per sample: 62 us at B = 1024, every block
// streaming the 160 KB of conv weights for its single sample) as rollout_encoder2_kernel's block made persistent: a 16-wave
// block keeps w1 / w2 in LDS and w3 / w_up in registers and walks its samples smp = cb, cb + nconv, ...; the next sample's
// depth stack (bf16 rollout row, gathered through rowidx) is requested while the current one is convolved. Saves c1 / c2 /
// c3 (fp32 NHWC, what the backward kernels read) and the depth tokens. Blocks 0 .. nmlp-1 run the proprio MLP for 32 rows
// each (they finish early; the conv blocks queued behind them start on their CUs). bf16 only.
struct TrainEnc {
  const void* image;     // [slots][4*64*64] in the 16-bit operand type
  const float* state;    // [slots][Sp]
  const int* rowidx;     // [n] or null
  float *s_c1, *s_c2, *s_c3;  // [n][225][32], [n][36][64], [n][16][64]
  float *s_h1, *s_h2;         // [n][256] encoder-MLP activations (s_h2: row stride ld_h2)
  int n, nmlp, nconv, ld_h2;
  int acts16;                 // s_c1 / s_c2 are written in the operand type (bf16) for bwd_conv_kernel<T, true> (the trainer's passes)
};
struct TrainEncLds {
  static constexpr size_t part_b = 4 * 16 * 64 * 4;
  static constexpr size_t bytes = RollEnc2Lds::conv_bytes + part_b;
};
// MODE as in rollout_encoder2_kernel: ENC_TOK17 (LocoTransformer), ENC_TOK16 (vision-only Transformer: no proprio blocks,
// 16 tokens per sample), ENC_FLAT (NatureEncoder(flatten): the saved conv3 rows ARE the output), ENC_FUSE (NatureFuseEncoder:
// the proprio blocks stop after the second Linear+ReLU, whose rows go to tr.s_h2 with row stride tr.ld_h2 — the right half
// of the concat buffer; the visual projector over the saved conv3 rows is the caller's next launch).
template <typename T, int MODE>
__global__ __launch_bounds__(1024) void train_encoder_kernel(InfEncFrag w, TrainEnc tr, float* __restrict__ x0) {
  static_assert(sizeof(T) == 2, "16-bit operand types only");
  typedef typename Frag<T>::type frag_t;
  typedef InfEncLds<T> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8, qr = (lane >> 4) * 4;
  const int n = tr.n;
  auto gfrag = [&](const void* W, int idx) -> frag_t { return reinterpret_cast<const frag_t*>(W)[idx * 64 + lane]; };

  if ((int)blockIdx.x < tr.nmlp) {
    // ---------------- proprio branch, 32 rows per block: Linear+ReLU, Linear+ReLU, state_projector+ReLU -> token 0
    constexpr int MR = LY::MLP_ROWS, LDB = RollEnc2Lds::LDB;
    const int r0 = (int)blockIdx.x * MR;
    T* sb = reinterpret_cast<T*>(smem);
    T* h1 = sb + MR * LDB;
    T* h2 = h1 + MR * LY::LDH;
    float* bs = reinterpret_cast<float*>(h2 + MR * LY::LDH);    // bf1[256] | bf2[256] | bpr[64]
    float sv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // 32 x 128 values, 4 per thread; unconditional loads from clamped addresses (Sp >= 32)
      const int idx = tid + k * 1024, r = idx >> 7, c = idx & 127;
      const int row = min(r0 + r, n - 1);
      sv[k] = tr.state[(int64_t)(tr.rowidx ? tr.rowidx[row] : row) * w.Sp + min(c, w.Sp - 1)];
    }
    float bv;
    {
      const int o = min(tid, 575);
      const float *q0 = w.bf1, *q1 = w.bf2, *q2 = w.bpr;
      bv = *(o >= 512 ? q2 + (o - 512) : o >= 256 ? q1 + (o - 256) : q0 + o);
    }
    __builtin_amdgcn_sched_barrier(0);
    frag_t r1[4], r2[8], r3[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) r1[ks] = gfrag(w.wf1, wave * 4 + ks);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r2[ks] = gfrag(w.wf2, wave * 8 + ks);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 1024, r = idx >> 7, c = idx & 127;
      sb[r * LDB + c] = (T)((r0 + r < n && c < w.Sp) ? sv[k] : 0.f);  // columns S..Sp-1 of a rollout row are zero
    }
    if (tid < 576) bs[tid] = bv;
    __syncthreads();
    auto store_h = [&](T* h, const f32x4 (&acc)[2], const float* bias, float* save, int lds) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bias + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float v0 = fmaxf(acc[mt][0] + bb.x, 0.f), v1 = fmaxf(acc[mt][1] + bb.y, 0.f);
        const float v2 = fmaxf(acc[mt][2] + bb.z, 0.f), v3 = fmaxf(acc[mt][3] + bb.w, 0.f);
        st4(h + (mt * 16 + fr) * LY::LDH + n4, v0, v1, v2, v3);
        if (r0 + mt * 16 + fr < n) st4(save + (int64_t)(r0 + mt * 16 + fr) * lds + n4, v0, v1, v2, v3);
      }
    };
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 4>(acc, sb, LDB, r1, lane);
      if (MODE == ENC_TOK17 && wave < 4) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) r3[ks] = gfrag(w.wpr, wave * 8 + ks);
      }
      store_h(h1, acc, bs, tr.s_h1, 256);
    }
    __syncthreads();
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 8>(acc, h1, LY::LDH, r2, lane);
      store_h(h2, acc, bs + 256, tr.s_h2, tr.ld_h2);
    }
    if constexpr (MODE != ENC_TOK17) return;
    __syncthreads();
    if (wave < 4) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      mm_held<T, 2, 8>(acc, h2, LY::LDH, r3, lane);
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bs + 512 + n4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = r0 + mt * 16 + fr;
        if (row < n)
          st4(x0 + ((int64_t)row * NTOK) * TD + n4, fmaxf(acc[mt][0] + bb.x, 0.f), fmaxf(acc[mt][1] + bb.y, 0.f),
              fmaxf(acc[mt][2] + bb.z, 0.f), fmaxf(acc[mt][3] + bb.w, 0.f));
      }
    }
    return;
  }

  // ---------------- depth branch: a persistent block over samples cb, cb + nconv, ...
  const int cb = (int)blockIdx.x - tr.nmlp;
  if (cb >= n) return;
  T* img = reinterpret_cast<T*>(smem);
  T* c1 = img + LY::IMG;
  T* c2 = c1 + LY::C1;
  T* c3 = c2 + LY::C2;
  frag_t* w1s = reinterpret_cast<frag_t*>(smem + LY::conv_bytes);   // [2][8][64]
  frag_t* w2s = w1s + 2 * 8 * 64;                                    // [4][16][64]
  float* bs = reinterpret_cast<float*>(w2s + 4 * 16 * 64);           // b1[32] | b2[64] | b3[64] | bup[64]
  float* part = bs + 256;                                            // [4 K-quarters][16 pixels][64] fp32
  auto image_of = [&](int smp) { return reinterpret_cast<const T*>(tr.image) + (int64_t)(tr.rowidx ? tr.rowidx[smp] : smp) * LY::IMG; };
  frag_t iv[2];
  {
    const frag_t* src = reinterpret_cast<const frag_t*>(image_of(cb));  // 2048 x 16 bytes, two per thread
    iv[0] = src[tid]; iv[1] = src[tid + 1024];
  }
  const frag_t w1v = reinterpret_cast<const frag_t*>(w.w1)[tid];
  float bv;
  {
    const int o = min(tid, 223);
    const float *q0 = w.b1, *q1 = w.b2, *q2 = w.b3, *q3 = w.bup;
    bv = *(o >= 160 ? q3 + (o - 160) : o >= 96 ? q2 + (o - 96) : o >= 32 ? q1 + (o - 32) : q0 + o);
  }
  __builtin_amdgcn_sched_barrier(0);
  frag_t w2v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w2v[k] = reinterpret_cast<const frag_t*>(w.w2)[tid + k * 1024];
  const int nt3 = wave & 3, kq = wave >> 2, ks0 = kq * 5, ks1 = min(18, ks0 + 5);
  frag_t w3v[5], wuv[2];
#pragma unroll
  for (int d = 0; d < 5; ++d) w3v[d] = gfrag(w.w3, nt3 * 18 + min(ks0 + d, 17));
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wuv[ks] = gfrag(MODE == ENC_TOK17 || MODE == ENC_TOK16 ? w.wup : w.w3, nt3 * 2 + ks);
  __builtin_amdgcn_sched_barrier(0);
  reinterpret_cast<frag_t*>(img)[tid] = iv[0];
  reinterpret_cast<frag_t*>(img)[tid + 1024] = iv[1];
  w1s[tid] = w1v;
  if (tid < 224) bs[tid] = bv;
  bool w2_pending = true;
  __syncthreads();
  for (int smp = cb; smp < n; smp += tr.nconv) {
    const int nxt = smp + tr.nconv;
    if (nxt < n) {  // the next depth stack starts its trip now
      const frag_t* src = reinterpret_cast<const frag_t*>(image_of(nxt));
      iv[0] = src[tid]; iv[1] = src[tid + 1024];
    }
    if (wave < 15) {  // conv1: 225 pixels = 15 row tiles, one per wave; K = (c,ky,kx) = 256, N = 32
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const int p = min(wave * 16 + fr, 224);
      const int pbase = (p / 15) * 4 * 64 + (p % 15) * 4;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int k0 = ks * 32 + fg, c = k0 >> 6, ky = (k0 >> 3) & 7;
        const frag_t fa = afrag_t(img + c * 4096 + ky * 64 + pbase);
        mma_k32(acc[0], w1s[(0 * 8 + ks) * 64 + lane], fa);
        mma_k32(acc[1], w1s[(1 * 8 + ks) * 64 + lane], fa);
      }
      const int pp = wave * 16 + fr;
      if (pp < 225) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n4 = j * 16 + qr;
          const float4 bb = *reinterpret_cast<const float4*>(bs + n4);
          const float v0 = fmaxf(acc[j][0] + bb.x, 0.f), v1 = fmaxf(acc[j][1] + bb.y, 0.f);
          const float v2 = fmaxf(acc[j][2] + bb.z, 0.f), v3 = fmaxf(acc[j][3] + bb.w, 0.f);
          st4(c1 + pp * LY::LD1 + n4, v0, v1, v2, v3);
          if (tr.acts16) st4(reinterpret_cast<T*>(tr.s_c1) + ((int64_t)smp * 225 + pp) * 32 + n4, v0, v1, v2, v3);
          else st4(tr.s_c1 + ((int64_t)smp * 225 + pp) * 32 + n4, v0, v1, v2, v3);
        }
      }
    }
    if (w2_pending) {
#pragma unroll
      for (int k = 0; k < 4; ++k) w2s[tid + k * 1024] = w2v[k];
      w2_pending = false;
    }
    __syncthreads();
    if (nxt < n) {  // conv1 has consumed the image: the next one takes its place
      reinterpret_cast<frag_t*>(img)[tid] = iv[0];
      reinterpret_cast<frag_t*>(img)[tid + 1024] = iv[1];
    }
    if (wave < 12) {  // conv2: 36 pixels (3 row tiles) x 4 column tiles, one pair per wave; K = (ky,kx,c) = 512
      const int mt = wave >> 2, nt = wave & 3;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int p = min(mt * 16 + fr, 35);
      const int pb = ((p / 6) * 2 * 15 + (p % 6) * 2) * LY::LD1;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int ky = ks >> 2, kx = ks & 3;
        const frag_t fa = afrag_t(c1 + pb + (ky * 15 + kx) * LY::LD1 + fg);
        mma_k32(acc, w2s[(nt * 16 + ks) * 64 + lane], fa);
      }
      const int pp = mt * 16 + fr, n4 = nt * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bs + 32 + n4);
      if (pp < 36) {
        const float v0 = fmaxf(acc[0] + bb.x, 0.f), v1 = fmaxf(acc[1] + bb.y, 0.f);
        const float v2 = fmaxf(acc[2] + bb.z, 0.f), v3 = fmaxf(acc[3] + bb.w, 0.f);
        st4(c2 + pp * LY::LD2 + n4, v0, v1, v2, v3);
        if (tr.acts16) st4(reinterpret_cast<T*>(tr.s_c2) + ((int64_t)smp * 36 + pp) * 64 + n4, v0, v1, v2, v3);
        else st4(tr.s_c2 + ((int64_t)smp * 36 + pp) * 64 + n4, v0, v1, v2, v3);
      }
    }
    __syncthreads();
    {  // conv3: 16 pixels, K = (ky,kx,c) = 576 = 18 steps: wave = (column tile, K-quarter of 5/5/5/3 steps)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int pb = ((fr >> 2) * 6 + (fr & 3)) * LY::LD2;
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const int ks = ks0 + d;
        if (ks < ks1) {
          const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3, c0 = (ks & 1) * 32 + fg;
          const frag_t fa = afrag_t(c2 + pb + (ky * 6 + kx) * LY::LD2 + c0);
          mma_k32(acc, w3v[d], fa);
        }
      }
      st4(part + (kq * 16 + fr) * 64 + nt3 * 16 + qr, acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    {  // sum of the four K-quarters + bias + ReLU -> c3: one output per thread
      const int pix = tid >> 6, nn = tid & 63;
      const float s4 = ((part[pix * 64 + nn] + part[(16 + pix) * 64 + nn]) + part[(32 + pix) * 64 + nn]) + part[(48 + pix) * 64 + nn];
      const float v = fmaxf(s4 + bs[96 + nn], 0.f);
      c3[pix * LY::LD2 + nn] = (T)v;
      tr.s_c3[((int64_t)smp * 16 + pix) * 64 + nn] = v;
    }
    __syncthreads();
    constexpr int TOKS = MODE == ENC_TOK16 ? 16 : NTOK, TOK0 = MODE == ENC_TOK16 ? 0 : 1;
    if ((MODE == ENC_TOK17 || MODE == ENC_TOK16) && wave < 4) {  // depth_up_conv (1x1, no activation) -> tokens 1..16 (vision-only: 0..15)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const frag_t fa = afrag_t(c3 + fr * LY::LD2 + ks * 32 + fg);
        mma_k32(acc, wuv[ks], fa);
      }
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bs + 160 + n4);
      st4(x0 + ((int64_t)smp * TOKS + TOK0 + fr) * TD + n4, acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
    }
    // (no barrier: the next round's conv1 only reads img / w1s and writes c1, which conv2 of this round has finished reading)
  }
}


// ------------------------------------------------------------------------------------------ rollout step, NatureCNN fuse net
// The whole env step of the NatureCNN policy / value pair (networks/nets.py:194-262, base.py:345-398) for ONE sample
// per block (blockIdx.y = net: both recompute the encoder they share): ingest -> conv1..3 -> visual projector (1024 ->
// 256) || proprio MLP -> concat -> 3-layer head -> sample / file. 16 waves, one 16-column tile per wave in every
// matrix-vector phase (only fragment row 0 carries data).
struct InfCnn {
  const void *w1, *w2, *w3;                  // packed conv weights (T): [32][256] [64][512] [64][576]
  const float *b1, *b2, *b3;
  const void *wpr, *wf1, *wf2;               // visual projector [256][1024] (NHWC-flatten k order), proprio MLP [256][Kp1] [256][256]
  const float *bpr, *bf1, *bf2;
  int S, Sp, Kp1;
};
struct InfCnnHead { const void *w0, *w1, *w2; const float *b0, *b1, *b2; float* out; int nout; };  // [256][512] [256][256] [16][256]
struct InfCnnHeadPair { InfCnnHead n[2]; };
template <typename T> struct RollCnnLds {
  typedef InfEncLds<T> E;
  static constexpr int LDC = 512 + InfLd<T>::PAD;
  static constexpr size_t conv_b = E::conv_bytes;
  static constexpr size_t sin_b = 128 * 4, cat_b = (size_t)LDC * sizeof(T), h_b = (size_t)256 * sizeof(T), so_b = 16 * 4;
  static constexpr size_t bytes = conv_b + sin_b + cat_b + 3 * h_b + so_b + 64;
};
// y[16 columns of tile nt] = W[nt*16 .. +16][0..32*KS) . x: x is ONE row held in LDS (fp32 or T), so only lane group
// fr == 0 supplies a non-zero A fragment; result acc[r] (lanes with fr == 0) = y[nt*16 + 4*(lane>>4) + r]
template <typename T, int KS, int PD, typename AT>
__device__ __forceinline__ f32x4 gemv_tile(const AT* x, const T* __restrict__ Wp, int Kp, int nt, int lane) {
  typedef typename Frag<T>::type frag_t;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  const T* wrow = Wp + (int64_t)(nt * 16 + fr) * Kp + fg;
  frag_t ring[PD];
#pragma unroll
  for (int d = 0; d < PD; ++d) ring[d] = *reinterpret_cast<const frag_t*>(wrow + d * 32);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  frag_t zero;
  if constexpr (sizeof(T) == 2) { for (int j = 0; j < 8; ++j) zero[j] = (T)0.f; } else { for (int j = 0; j < 8; ++j) zero.v[j] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const frag_t fb = ring[ks % PD];
    if (ks + PD < KS) ring[ks % PD] = *reinterpret_cast<const frag_t*>(wrow + (ks + PD) * 32);
    frag_t fa;
    if constexpr (sizeof(AT) == sizeof(T)) fa = *reinterpret_cast<const frag_t*>(x + ks * 32 + fg);
    else fa = afrag<T>(reinterpret_cast<const float*>(x) + ks * 32 + fg);
    mma_k32(acc, fb, fr == 0 ? fa : zero);
  }
  return acc;
}

template <typename T>
__global__ __launch_bounds__(1024) void rollout_cnn_kernel(const ActCtl* __restrict__ ctlc, const float* __restrict__ obs, int E,
                                                           InfCnn w, InfCnnHeadPair hd, InfFinish fin,
                                                           float* __restrict__ state_roll, T* __restrict__ image_roll) {
  typedef typename Frag<T>::type frag_t;
  typedef InfEncLds<T> LY;
  typedef RollCnnLds<T> RL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = (lane >> 4) * 8, qr = (lane >> 4) * 4;
  const long long t_step = ctlc->t;
  const int64_t slot0 = (int64_t)t_step * E;
  const int D = w.S + LY::IMG;
  const int b = blockIdx.x, net = blockIdx.y;
  T* img = reinterpret_cast<T*>(smem);
  T* c1 = img + LY::IMG;
  T* c2 = c1 + LY::C1;
  T* c3 = c2 + LY::C2;
  float* sin = reinterpret_cast<float*>(smem + RL::conv_b);
  T* cat = reinterpret_cast<T*>(smem + RL::conv_b + RL::sin_b);
  T* h1 = reinterpret_cast<T*>(smem + RL::conv_b + RL::sin_b + RL::cat_b);
  T* h2 = h1 + 256;
  T* hs = h2 + 256;  // proprio fc1 output
  float* so = reinterpret_cast<float*>(smem + RL::conv_b + RL::sin_b + RL::cat_b + 3 * RL::h_b);
  {
    const float* src = obs + (int64_t)b * D + w.S;  // 16-byte aligned only when S % 4 == 0: dword-aligned vector loads (ld4u)
    T* roll = image_roll + (slot0 + b) * (int64_t)LY::IMG;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld4u(src + (tid + k * 1024) * 4);
    if (tid < 128) {
      const float x = tid < w.S ? obs[(int64_t)b * D + tid] : 0.f;
      sin[tid] = x;
      if (net == 0 && tid < w.Sp) state_roll[(slot0 + b) * w.Sp + tid] = x;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 1024;
      st4(img + i * 4, v[k].x, v[k].y, v[k].z, v[k].w);
      if (net == 0) st4(roll + i * 4, v[k].x, v[k].y, v[k].z, v[k].w);
    }
  }
  __syncthreads();
  {  // proprio fc1: column tile `wave` (its output is first read after the next barriers)
    const f32x4 a = w.Kp1 == 128 ? gemv_tile<T, 4, 4>(sin, (const T*)w.wf1, 128, wave, lane)
                                 : gemv_tile<T, 2, 2>(sin, (const T*)w.wf1, 64, wave, lane);
    if (fr == 0) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(w.bf1 + n4);
      st4(hs + n4, fmaxf(a[0] + bb.x, 0.f), fmaxf(a[1] + bb.y, 0.f), fmaxf(a[2] + bb.z, 0.f), fmaxf(a[3] + bb.w, 0.f));
    }
  }
  if (wave < 15) {  // conv1: one 16-pixel row tile per wave
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int p = min(wave * 16 + fr, 224);
    const int pbase = (p / 15) * 4 * 64 + (p % 15) * 4;
    frag_t ring[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int j = 0; j < 2; ++j) ring[d][j] = *reinterpret_cast<const frag_t*>((const T*)w.w1 + (j * 16 + fr) * 256 + d * 32 + fg);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = ks * 32 + fg, c = k0 >> 6, ky = (k0 >> 3) & 7;
      const frag_t fb0 = ring[ks & 3][0], fb1 = ring[ks & 3][1];
      if (ks + 4 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          ring[ks & 3][j] = *reinterpret_cast<const frag_t*>((const T*)w.w1 + (j * 16 + fr) * 256 + (ks + 4) * 32 + fg);
      }
      const frag_t fa = afrag_t(img + c * 4096 + ky * 64 + pbase);
      mma_k32(acc[0], fb0, fa);
      mma_k32(acc[1], fb1, fa);
    }
    const int pp = wave * 16 + fr;
    if (pp < 225) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n4 = j * 16 + qr;
        const float4 bb = *reinterpret_cast<const float4*>(w.b1 + n4);
        st4(c1 + pp * LY::LD1 + n4, fmaxf(acc[j][0] + bb.x, 0.f), fmaxf(acc[j][1] + bb.y, 0.f), fmaxf(acc[j][2] + bb.z, 0.f),
            fmaxf(acc[j][3] + bb.w, 0.f));
      }
    }
  }
  __syncthreads();
  if (wave < 12) {  // conv2: (row tile, column tile) per wave
    const int mt = wave >> 2, nt = wave & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int p = min(mt * 16 + fr, 35);
    const int pb = ((p / 6) * 2 * 15 + (p % 6) * 2) * LY::LD1;
    const T* w2row = (const T*)w.w2 + (nt * 16 + fr) * 512 + fg;
    frag_t ring[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) ring[d] = *reinterpret_cast<const frag_t*>(w2row + d * 32);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int ky = ks >> 2, kx = ks & 3;
      const frag_t fb = ring[ks & 7];
      if (ks + 8 < 16) ring[ks & 7] = *reinterpret_cast<const frag_t*>(w2row + (ks + 8) * 32);
      const frag_t fa = afrag_t(c1 + pb + (ky * 15 + kx) * LY::LD1 + fg);
      mma_k32(acc, fb, fa);
    }
    const int pp = mt * 16 + fr, n4 = nt * 16 + qr;
    const float4 bb = *reinterpret_cast<const float4*>(w.b2 + n4);
    if (pp < 36)
      st4(c2 + pp * LY::LD2 + n4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f), fmaxf(acc[2] + bb.z, 0.f),
          fmaxf(acc[3] + bb.w, 0.f));
  }
  __syncthreads();
  float* part = reinterpret_cast<float*>(img);  // [4 K-quarters][16 pixels][64] fp32 partial sums (the image is dead)
  {  // conv3: wave = (column tile, K-quarter)
    const int nt = wave & 3, kq = wave >> 2;
    const int ks0 = kq * 5, ks1 = min(18, ks0 + 5);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int pb = ((fr >> 2) * 6 + (fr & 3)) * LY::LD2;
    const T* w3row = (const T*)w.w3 + (nt * 16 + fr) * 576 + fg;
    frag_t fbv[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) fbv[d] = *reinterpret_cast<const frag_t*>(w3row + min(ks0 + d, 17) * 32);
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int ks = ks0 + d;
      if (ks < ks1) {
        const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3, c0 = (ks & 1) * 32 + fg;
        const frag_t fa = afrag_t(c2 + pb + (ky * 6 + kx) * LY::LD2 + c0);
        mma_k32(acc, fbv[d], fa);
      }
    }
    st4(part + (kq * 16 + fr) * 64 + nt * 16 + qr, acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  {
    const int pix = tid >> 6, n = tid & 63;
    const float v = ((part[pix * 64 + n] + part[(16 + pix) * 64 + n]) + part[(32 + pix) * 64 + n]) + part[(48 + pix) * 64 + n];
    c3[pix * LY::LD2 + n] = (T)fmaxf(v + w.b3[n], 0.f);
  }
  __syncthreads();
  {  // visual projector over the NHWC flatten of conv3 (k = pixel*64 + c: 32 steps) and proprio fc2: tile `wave` of each
    typedef typename Frag<T>::type fr_t;
    const T* wrow = (const T*)w.wpr + (int64_t)(wave * 16 + fr) * 1024 + fg;
    fr_t ring[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) ring[d] = *reinterpret_cast<const fr_t*>(wrow + d * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    fr_t zero;
    if constexpr (sizeof(T) == 2) { for (int j = 0; j < 8; ++j) zero[j] = (T)0.f; } else { for (int j = 0; j < 8; ++j) zero.v[j] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const fr_t fb = ring[ks & 7];
      if (ks + 8 < 32) ring[ks & 7] = *reinterpret_cast<const fr_t*>(wrow + (ks + 8) * 32);
      const fr_t fa = afrag_t(c3 + (ks >> 1) * LY::LD2 + (ks & 1) * 32 + fg);
      mma_k32(acc, fb, fr == 0 ? fa : zero);
    }
    const f32x4 a2 = gemv_tile<T, 8, 8>(hs, (const T*)w.wf2, 256, wave, lane);
    if (fr == 0) {
      const int n4 = wave * 16 + qr;
      const float4 bp = *reinterpret_cast<const float4*>(w.bpr + n4), b2 = *reinterpret_cast<const float4*>(w.bf2 + n4);
      st4(cat + n4, fmaxf(acc[0] + bp.x, 0.f), fmaxf(acc[1] + bp.y, 0.f), fmaxf(acc[2] + bp.z, 0.f), fmaxf(acc[3] + bp.w, 0.f));
      st4(cat + 256 + n4, fmaxf(a2[0] + b2.x, 0.f), fmaxf(a2[1] + b2.y, 0.f), fmaxf(a2[2] + b2.z, 0.f), fmaxf(a2[3] + b2.w, 0.f));
    }
  }
  __syncthreads();
  const InfCnnHead& h = hd.n[net];
  {
    const f32x4 a = gemv_tile<T, 16, 8>(cat, (const T*)h.w0, 512, wave, lane);
    if (fr == 0) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(h.b0 + n4);
      st4(h1 + n4, fmaxf(a[0] + bb.x, 0.f), fmaxf(a[1] + bb.y, 0.f), fmaxf(a[2] + bb.z, 0.f), fmaxf(a[3] + bb.w, 0.f));
    }
  }
  __syncthreads();
  {
    const f32x4 a = gemv_tile<T, 8, 8>(h1, (const T*)h.w1, 256, wave, lane);
    if (fr == 0) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(h.b1 + n4);
      st4(h2 + n4, fmaxf(a[0] + bb.x, 0.f), fmaxf(a[1] + bb.y, 0.f), fmaxf(a[2] + bb.z, 0.f), fmaxf(a[3] + bb.w, 0.f));
    }
  }
  __syncthreads();
  if (wave == 0) {
    const f32x4 a = gemv_tile<T, 8, 8>(h2, (const T*)h.w2, 256, 0, lane);
    if (fr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = qr + r;
        const float v = c < h.nout ? a[r] + h.b2[c] : 0.f;
        so[c] = v;
        h.out[(int64_t)b * OUT_LD + c] = v;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {  // sampling / filing epilogue: see infer_layer_kernel
    const int i = b, A = fin.A;
    if (net == 0) {
      float e = 0.f, lp = 0.f;
      for (int a = 0; a < A; ++a) {
        const float mu = so[a];
        const float ls = fminf(fmaxf(fin.logstd[a], LOG_SIG_MIN), LOG_SIG_MAX);
        const float sg = expf(ls);
        e += 0.5f + HALF_LOG_2PI + logf(sg);
        float act = fmaf(sg, fin.eps[(int64_t)i * A + a], mu);
            if (fin.tanh_action) act = tanhf(act);  // (eps = 0: eval_act's tanh(mean))
        fin.action[(int64_t)i * A + a] = act;
        fin.mean[(int64_t)i * A + a] = mu;
        fin.stdv[(int64_t)i * A + a] = sg;
        if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + a] = act;
        const float d = (fin.tanh_action ? tanh_pre(act) : act) - mu;  // (through the STORED action, like act_finish_kernel)
        lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
        if (fin.tanh_action) lp -= tanh_corr(act);
      }
      fin.ent[i] = e;
      if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + i] = lp;
    } else {
      const float v = so[0];
      fin.value[i] = v;
      if (fin.values_roll != nullptr) fin.values_roll[t_step * E + i] = v;
    }
    __threadfence();
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&fin.ctl->done), 1ull);
    if (done == (unsigned long long)(gridDim.x * gridDim.y) - 1) {
      fin.ctl->done = 0;
      fin.ctl->t = t_step + 1;
    }
  }
}


// ------------------------------------------------------------------------------------------ rollout step, state MLP
// Net / GaussianContPolicyBasicBias pair on proprioception only (networks/nets.py:16-55, starter/ppo_state.py): shared
// base MLP (S -> 256 -> 256, ReLU) + per-net head (256 -> 256 -> 256 -> out). One sample per block, blockIdx.y = net.
struct InfMlp { const void *wf1, *wf2; const float *bf1, *bf2; int S, Sp, Kp1; };
template <typename T>
__global__ __launch_bounds__(1024) void rollout_mlp_kernel(const ActCtl* __restrict__ ctlc, const float* __restrict__ obs, int E,
                                                           InfMlp w, InfCnnHeadPair hd, InfFinish fin,
                                                           float* __restrict__ state_roll) {
  __shared__ __attribute__((aligned(16))) float sin[128];
  __shared__ __attribute__((aligned(16))) T hbuf[4][256];
  __shared__ float so[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, qr = (lane >> 4) * 4;
  const long long t_step = ctlc->t;
  const int b = blockIdx.x, net = blockIdx.y;
  if (tid < 128) {
    const float x = tid < w.S ? obs[(int64_t)b * w.S + tid] : 0.f;
    sin[tid] = x;
    if (net == 0 && tid < w.Sp) state_roll[((int64_t)t_step * E + b) * w.Sp + tid] = x;
  }
  __syncthreads();
  auto relu_store = [&](T* dst, const f32x4& a, const float* bias) {
    if (fr == 0) {
      const int n4 = wave * 16 + qr;
      const float4 bb = *reinterpret_cast<const float4*>(bias + n4);
      st4(dst + n4, fmaxf(a[0] + bb.x, 0.f), fmaxf(a[1] + bb.y, 0.f), fmaxf(a[2] + bb.z, 0.f), fmaxf(a[3] + bb.w, 0.f));
    }
  };
  relu_store(hbuf[0], w.Kp1 == 128 ? gemv_tile<T, 4, 4>(sin, (const T*)w.wf1, 128, wave, lane)
                                   : gemv_tile<T, 2, 2>(sin, (const T*)w.wf1, 64, wave, lane), w.bf1);
  __syncthreads();
  relu_store(hbuf[1], gemv_tile<T, 8, 8>(hbuf[0], (const T*)w.wf2, 256, wave, lane), w.bf2);
  __syncthreads();
  const InfCnnHead& h = hd.n[net];
  relu_store(hbuf[2], gemv_tile<T, 8, 8>(hbuf[1], (const T*)h.w0, 256, wave, lane), h.b0);
  __syncthreads();
  relu_store(hbuf[3], gemv_tile<T, 8, 8>(hbuf[2], (const T*)h.w1, 256, wave, lane), h.b1);
  __syncthreads();
  if (wave == 0) {
    const f32x4 a = gemv_tile<T, 8, 8>(hbuf[3], (const T*)h.w2, 256, 0, lane);
    if (fr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = qr + r;
        const float v = c < h.nout ? a[r] + h.b2[c] : 0.f;
        so[c] = v;
        h.out[(int64_t)b * OUT_LD + c] = v;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {  // sampling / filing epilogue: see infer_layer_kernel
    const int i = b, A = fin.A;
    if (net == 0) {
      float e = 0.f, lp = 0.f;
      for (int a = 0; a < A; ++a) {
        const float mu = so[a];
        const float ls = fminf(fmaxf(fin.logstd[a], LOG_SIG_MIN), LOG_SIG_MAX);
        const float sg = expf(ls);
        e += 0.5f + HALF_LOG_2PI + logf(sg);
        float act = fmaf(sg, fin.eps[(int64_t)i * A + a], mu);
            if (fin.tanh_action) act = tanhf(act);  // (eps = 0: eval_act's tanh(mean))
        fin.action[(int64_t)i * A + a] = act;
        fin.mean[(int64_t)i * A + a] = mu;
        fin.stdv[(int64_t)i * A + a] = sg;
        if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + a] = act;
        const float d = (fin.tanh_action ? tanh_pre(act) : act) - mu;  // (through the STORED action, like act_finish_kernel)
        lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
        if (fin.tanh_action) lp -= tanh_corr(act);
      }
      fin.ent[i] = e;
      if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + i] = lp;
    } else {
      const float v = so[0];
      fin.value[i] = v;
      if (fin.values_roll != nullptr) fin.values_roll[t_step * E + i] = v;
    }
    __threadfence();
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&fin.ctl->done), 1ull);
    if (done == (unsigned long long)(gridDim.x * gridDim.y) - 1) {
      fin.ctl->done = 0;
      fin.ctl->t = t_step + 1;
    }
  }
}

}  // namespace v4l
