// Fused backward kernels of the PPO update (gfx950). Counterparts of csrc/infer.h's training-forward kernels: one
// launch walks a whole sub-network backward with every intermediate gradient held in LDS; HBM only sees what the
// deferred weight-grad contraction (gemm_tn_group_kernel) reads afterwards and the gradient handed to the next stage.
//
// bwd_layer_kernel: backward of one nn.TransformerEncoderLayer (post-norm, one head, ReLU FFN; the module the
// reference stacks in torchrl/networks/nets.py:948-955) for 4 samples = 68 token rows per block:
//   dy -> LN2' -> dz2 -> (W2^T, ReLU mask) -> df -> (W1^T) + dz2 -> dx1 -> LN1' -> dz1 -> (Wo^T) -> dctx
//      -> softmax-attention' -> dqkv -> (Win^T) + dz1 -> dx_in
#pragma once
#include "infer.h"

namespace v4l {

struct BwdLayer {
  const void *w2t, *w1t, *wot, *wint;  // data-grad packs [K][N] (T): [ff][64] [64][ff] [64][64] [64][192]
  const float *g1, *g2;                // LayerNorm weights
  const float* dy;                     // [R][64] grad w.r.t. the layer output
  const float *s_qkv, *s_P, *s_xh1, *s_rs1, *s_xh2, *s_rs2;  // saved by the forward pass (fp32)
  const void* s_f;                       // [R][256] FFN activation in the contraction type T (ReLU mask)
  void *o_dz2, *o_df, *o_dz1, *o_dqkv;   // dY operands (T) of the weight-grads of linear2 / linear1 / out_proj / in_proj
  float* o_dx;                           // [R][64] grad w.r.t. the layer input
  float *gp2, *bp2, *gp1, *bp1;          // [gridDim.x][64] per-block dgamma / dbeta partials of norm2 / norm1
};

// HEAD (last layer): the block first walks the pooled heads of its samples backward (nets.py:1015-1034 reversed):
//   dout -> (W2^T, mask h1) -> dh1 -> (W1^T, mask h0) -> dh0 -> (W0^T) -> dpool -> un-pool -> dy rows in LDS
struct BwdHead {
  const void *w2t, *w1t, *w0t;   // data-grad packs [K][N]: [256][64] [256][256] [128][256]
  const float* dout;             // [n][OUT_LD] grad w.r.t. the head output (columns >= out_dim are zero)
  const float *s_h1, *s_h0;      // [n][256] post-ReLU activations of the two hidden layers
  float *o_dh1, *o_dh0;          // [n][256] dY operands of the weight-grads of fcs[1] / fcs[0]
};
// TAIL (layer 0): the block continues from dx_in into the encoder (base.py:602-622 reversed): token 0 ->
// state_projector' -> encoder MLP'; tokens 1..16 -> depth_up_conv' (-> dc3, ReLU mask of conv3)
struct BwdTail {
  const void *wpt, *wf2t, *wupt;  // data-grad packs: state_projector [256][64], encoder fc2 [256][256], up-conv [64][64]
  const float* x0;                // [R][64] layer-0 input tokens (ReLU mask of token 0)
  const float *s_e1, *s_e0;       // [n][256] encoder-MLP activations (post-ReLU)
  const float* s_c3;              // [n*16][64] conv3 output (post-ReLU)
  float *o_dhc, *o_de0;           // [n][256] grads w.r.t. the pre-activations of encoder fc2 / fc1
  float* o_dc3;                   // [n*16][64]
};

template <typename T> struct BwdLayLds {
  static constexpr int PAD = InfLd<T>::PAD;
  static constexpr int LDX = 64 + 4, LDQ = 192 + 4, LDF = 256 + PAD;
  static constexpr size_t a_b = (size_t)INF_ROWS * LDX * 4;
  static constexpr size_t qkv_b = (size_t)INF_ROWS * LDQ * 4;
  static constexpr size_t f_b = (size_t)INF_ROWS * LDF * sizeof(T);
  static constexpr size_t big_b = qkv_b > f_b ? qkv_b : f_b;
  static constexpr size_t p_b = (size_t)2 * 4 * NTOK * ATT_PLD * 4;  // P and dS of the 4 samples
  static constexpr size_t red_b = (size_t)2 * 4 * TD * 4;
  static constexpr size_t bytes = 2 * a_b + big_b + p_b + red_b;     // a | b | df / qkv->dqkv | P,dS | LN partials
};

// LayerNorm backward, in place over the 80 LDS rows of `d` (rows >= nrows hold zeros and stay zero); the rows < nrows
// also go to o_dz (global). Leaves the block's dgamma/dbeta partial in gpart/bpart[64]. Contains one __syncthreads.
template <typename T>
__device__ __forceinline__ void ln_bwd_rows(float* d, int ld, const float* __restrict__ xh, const float* __restrict__ rs,
                                            const float* __restrict__ gamma, int wave, int lane, int nrows,
                                            T* __restrict__ o_dz, float* red, float* __restrict__ gpart,
                                            float* __restrict__ bpart) {
  const float g = gamma[lane];
  float ag = 0.f, ab = 0.f;
  constexpr int U = INF_ROWS / 16;
  for (int r0 = wave; r0 < INF_ROWS; r0 += 4 * U) {
    float dd[U], x[U], rr[U], dxh[U], c1[U], c2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // loads are unconditional (row 0 stands in for the padding rows), then selected
      const int r = r0 + 4 * u;
      const bool ok = r < nrows;
      const int o = ok ? r : 0;
      const float xv = xh[o * TD + lane], rv = rs[o];
      dd[u] = d[r * ld + lane];
      x[u] = ok ? xv : 0.f;
      rr[u] = ok ? rv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ag = fmaf(dd[u], x[u], ag);
      ab += dd[u];
      dxh[u] = dd[u] * g;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) c1[u] = wave_sum(dxh[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) c2[u] = wave_sum(dxh[u] * x[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 4 * u;
      const float dz = rr[u] * (dxh[u] - c1[u] * (1.f / TD) - x[u] * (c2[u] * (1.f / TD)));
      d[r * ld + lane] = dz;
      if (r < nrows) o_dz[r * TD + lane] = (T)dz;
    }
  }
  red[wave * TD + lane] = ag;
  red[4 * TD + wave * TD + lane] = ab;
  __syncthreads();
  if (wave == 0) {
    gpart[lane] = (red[lane] + red[TD + lane]) + (red[2 * TD + lane] + red[3 * TD + lane]);
    bpart[lane] = (red[4 * TD + lane] + red[5 * TD + lane]) + (red[6 * TD + lane] + red[7 * TD + lane]);
  }
}

template <typename T, bool HEAD, bool TAIL>
__global__ __launch_bounds__(256) void bwd_layer_kernel(BwdLayer w, BwdHead hd, BwdTail tl, int n) {
  typedef BwdLayLds<T> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, qr = (lane >> 4) * 4;
  float* a = reinterpret_cast<float*>(smem);                         // dy -> dz2 -> dctx
  float* b = reinterpret_cast<float*>(smem + LY::a_b);               // dx1 -> dz1
  float* big = reinterpret_cast<float*>(smem + 2 * LY::a_b);         // df (T) -> qkv -> dqkv (fp32)
  float* sp = reinterpret_cast<float*>(smem + 2 * LY::a_b + LY::big_b);
  float* red = reinterpret_cast<float*>(smem + 2 * LY::a_b + LY::big_b + LY::p_b);
  const int s0 = blockIdx.x * INF_SPW;
  const int ns = min(INF_SPW, n - s0);
  const int nrows = ns * NTOK;
  const int64_t row0 = (int64_t)s0 * NTOK;
  const int nt1[1] = {wave};
  const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  if constexpr (HEAD) {
    constexpr int LDP = 128 + 4;
    float* dt = big;                                            // [16][LDX]: dout rows, zero padded to 64 columns
    T* dh1 = reinterpret_cast<T*>(big + 16 * LY::LDX);          // [16][LDF]
    T* dh0 = dh1 + 16 * LY::LDF;
    float* dpool = reinterpret_cast<float*>(dh0 + 16 * LY::LDF);  // [16][LDP]
    for (int idx = tid; idx < 16 * TD; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      const bool ok = r < ns && c < OUT_LD;
      const float v = hd.dout[ok ? (int64_t)(s0 + r) * OUT_LD + c : 0];
      dt[r * LY::LDX + c] = ok ? v : 0.f;
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto masked = [&](const float* act, T* dst, float* save) {  // ReLU mask from the saved activation, rows < ns
      const bool ok = fr < ns;
      float4 m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = *reinterpret_cast<const float4*>(act + (int64_t)(s0 + (ok ? fr : 0)) * 256 + nt4[j] * 16 + qr);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[0][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[0][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[0][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[0][j][3] : 0.f;
        st4(dst + fr * LY::LDF + n4, d0, d1, d2, d3);
        if (ok) st4(save + (int64_t)(s0 + fr) * 256 + n4, d0, d1, d2, d3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 2>(acc, dt, LY::LDX, (const T*)hd.w2t, 64, nt4, lane);
    masked(hd.s_h1, dh1, hd.o_dh1);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh1, LY::LDF, (const T*)hd.w1t, 256, nt4, lane);
    masked(hd.s_h0, dh0, hd.o_dh0);
    __syncthreads();
    {
      const int nt2[2] = {wave * 2, wave * 2 + 1};
      f32x4 a2[1][2];
      zero_acc(a2);
      block_gemm<T, 1, 2, 8>(a2, dh0, LY::LDF, (const T*)hd.w0t, 256, nt2, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) st4(dpool + fr * LDP + nt2[j] * 16 + qr, a2[0][j][0], a2[0][j][1], a2[0][j][2], a2[0][j][3]);
    }
    __syncthreads();
    // un-pool (pool_bwd_kernel): token 0 <- dpool[:, 0:64], tokens 1..16 <- dpool[:, 64:128] / 16
    for (int idx = tid; idx < INF_ROWS * TD; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      float v = 0.f;
      if (r < nrows) {
        const int sm = r / NTOK, t = r - sm * NTOK;
        v = t == 0 ? dpool[sm * LDP + c] : dpool[sm * LDP + TD + c] * (1.f / 16.f);
      }
      a[r * LY::LDX + c] = v;
    }
  } else {
    const float* dyg = w.dy + row0 * TD;
    for (int i4 = tid; i4 < INF_ROWS * (TD / 4); i4 += 256) {
      const int r = i4 >> 4, c4 = (i4 & 15) * 4;
      const bool ok = r < nrows;
      const float4 v = *reinterpret_cast<const float4*>(dyg + (ok ? r : 0) * TD + c4);
      *reinterpret_cast<float4*>(a + r * LY::LDX + c4) = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  // ---- norm2 backward: a = dz2
  ln_bwd_rows(a, LY::LDX, w.s_xh2 + row0 * TD, w.s_rs2 + row0, w.g2, wave, lane, nrows,
              reinterpret_cast<T*>(w.o_dz2) + row0 * TD, red,
              w.gp2 + (int64_t)blockIdx.x * TD, w.bp2 + (int64_t)blockIdx.x * TD);
  __syncthreads();
  // ---- df = (dz2 W2) o [f > 0]   (T in LDS for the next contraction, fp32 to HBM for linear1's weight-grad)
  T* f = reinterpret_cast<T*>(big);
  {
    f32x4 acc[INF_MT][4];
    zero_acc(acc);
    block_gemm<T, INF_MT, 4, 2>(acc, a, LY::LDX, (const T*)w.w2t, 64, nt4, lane);
#pragma unroll
    for (int mt = 0; mt < INF_MT; ++mt) {
      const int row = mt * 16 + fr;
      const bool ok = row < nrows;
      float4 m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        m[j] = ld4(reinterpret_cast<const T*>(w.s_f) + (row0 + (ok ? row : 0)) * 256 + nt4[j] * 16 + qr);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[mt][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[mt][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[mt][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[mt][j][3] : 0.f;
        st4(f + row * LY::LDF + n4, d0, d1, d2, d3);  // padding rows: acc == 0
        if (ok) st4(reinterpret_cast<T*>(w.o_df) + (row0 + row) * 256 + n4, d0, d1, d2, d3);
      }
    }
  }
  __syncthreads();
  {  // ---- dx1 = dz2 + df W1 -> b
    f32x4 acc[INF_MT][1];
    zero_acc(acc);
    block_gemm<T, INF_MT, 1, 8>(acc, f, LY::LDF, (const T*)w.w1t, 256, nt1, lane);
    const int n4 = wave * 16 + qr;
#pragma unroll
    for (int mt = 0; mt < INF_MT; ++mt) {
      const int row = mt * 16 + fr;
      const float4 r = *reinterpret_cast<const float4*>(a + row * LY::LDX + n4);
      st4(b + row * LY::LDX + n4, r.x + acc[mt][0][0], r.y + acc[mt][0][1], r.z + acc[mt][0][2], r.w + acc[mt][0][3]);
    }
  }
  __syncthreads();
  // qkv and P of the four samples come in while norm1' and the out_proj data-grad run (`big` is free: df was consumed)
  {
    const float* qg = w.s_qkv + row0 * 192;
    for (int i4 = tid; i4 < INF_ROWS * 48; i4 += 256) {
      const int r = i4 / 48, c4 = (i4 - r * 48) * 4;
      const bool ok = r < nrows;
      const float4 v = *reinterpret_cast<const float4*>(qg + (ok ? r : 0) * 192 + c4);
      *reinterpret_cast<float4*>(big + r * LY::LDQ + c4) = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
    const float* pg = w.s_P + (int64_t)s0 * NTOK * NTOK;
    for (int idx = tid; idx < ns * NTOK * NTOK; idx += 256) {
      const int sm = idx / (NTOK * NTOK), pr = idx - sm * NTOK * NTOK;
      const int i = pr / NTOK, j = pr - i * NTOK;
      sp[(sm * NTOK + i) * ATT_PLD + j] = pg[idx];
    }
  }
  // ---- norm1 backward: b = dz1
  ln_bwd_rows(b, LY::LDX, w.s_xh1 + row0 * TD, w.s_rs1 + row0, w.g1, wave, lane, nrows,
              reinterpret_cast<T*>(w.o_dz1) + row0 * TD, red,
              w.gp1 + (int64_t)blockIdx.x * TD, w.bp1 + (int64_t)blockIdx.x * TD);
  __syncthreads();
  {  // ---- dctx = dz1 Wo -> a
    f32x4 acc[INF_MT][1];
    zero_acc(acc);
    block_gemm<T, INF_MT, 1, 2>(acc, b, LY::LDX, (const T*)w.wot, 64, nt1, lane);
    const int n4 = wave * 16 + qr;
#pragma unroll
    for (int mt = 0; mt < INF_MT; ++mt)
      st4(a + (mt * 16 + fr) * LY::LDX + n4, acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]);
  }
  __syncthreads();
  // ---- attention backward of sample `wave` (fp32 VALU like the forward):
  //   dP = dctx V^T ; dS = P o (dP - rowsum(P o dP)) ; dV = P^T dctx ; dQ = dS K / 8 ; dK = dS^T Q / 8
  {
    const bool act = wave < ns;
    float* qs = big + wave * NTOK * LY::LDQ;
    float* p = sp + wave * NTOK * ATT_PLD;
    float* ds = sp + (4 + wave) * NTOK * ATT_PLD;
    const float* dc = a + wave * NTOK * LY::LDX;
    if (act) {
      for (int pr = lane; pr < NTOK * NTOK; pr += 64) {
        const int i = pr / NTOK, j = pr - i * NTOK;
        ds[i * ATT_PLD + j] = dot64(dc + i * LY::LDX, qs + j * LY::LDQ + 2 * TD);
      }
      if (lane < 3 * NTOK) {  // zero the 3 padding columns of every P / dS row (read below as float4)
        const int i = lane / 3, j = NTOK + lane - i * 3;
        p[i * ATT_PLD + j] = 0.f;
        ds[i * ATT_PLD + j] = 0.f;
      }
    }
    __syncthreads();
    if (act && lane < NTOK) {
      float rd = 0.f;
#pragma unroll
      for (int j = 0; j < NTOK; ++j) rd = fmaf(p[lane * ATT_PLD + j], ds[lane * ATT_PLD + j], rd);
#pragma unroll
      for (int j = 0; j < NTOK; ++j) ds[lane * ATT_PLD + j] = p[lane * ATT_PLD + j] * (ds[lane * ATT_PLD + j] - rd);
    }
    __syncthreads();
    if (act) {  // lane = feature column; everything this lane needs of Q, K, dctx sits in registers before the rows
                // of `qs` are overwritten with dQ | dK | dV (a wave only touches its own sample's rows)
      float dcr[NTOK], kr[NTOK], qq[NTOK], dq[NTOK], dk[NTOK], dv[NTOK];
#pragma unroll
      for (int i = 0; i < NTOK; ++i) {
        dcr[i] = dc[i * LY::LDX + lane];
        qq[i] = qs[i * LY::LDQ + lane];
        kr[i] = qs[i * LY::LDQ + TD + lane];
        dq[i] = dk[i] = dv[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < NTOK; ++i) {
        float pi[ATT_PLD], si[ATT_PLD];
#pragma unroll
        for (int j4 = 0; j4 < ATT_PLD; j4 += 4) {
          const float4 pv = *reinterpret_cast<const float4*>(p + i * ATT_PLD + j4);
          const float4 sv = *reinterpret_cast<const float4*>(ds + i * ATT_PLD + j4);
          pi[j4] = pv.x; pi[j4 + 1] = pv.y; pi[j4 + 2] = pv.z; pi[j4 + 3] = pv.w;
          si[j4] = sv.x; si[j4 + 1] = sv.y; si[j4 + 2] = sv.z; si[j4 + 3] = sv.w;
        }
#pragma unroll
        for (int j = 0; j < NTOK; ++j) {
          dv[j] = fmaf(pi[j], dcr[i], dv[j]);
          dq[i] = fmaf(si[j], kr[j], dq[i]);
          dk[j] = fmaf(si[j], qq[i], dk[j]);
        }
      }
      T* og = reinterpret_cast<T*>(w.o_dqkv) + (row0 + wave * NTOK) * 192;
#pragma unroll
      for (int i = 0; i < NTOK; ++i) {
        const float q8 = dq[i] * 0.125f, k8 = dk[i] * 0.125f;
        qs[i * LY::LDQ + lane] = q8;
        qs[i * LY::LDQ + TD + lane] = k8;
        qs[i * LY::LDQ + 2 * TD + lane] = dv[i];
        og[i * 192 + lane] = (T)q8;
        og[i * 192 + TD + lane] = (T)k8;
        og[i * 192 + 2 * TD + lane] = (T)dv[i];
      }
    }
  }
  __syncthreads();
  {  // ---- dx_in = dz1 + dqkv Win -> global
    f32x4 acc[INF_MT][1];
    zero_acc(acc);
    block_gemm<T, INF_MT, 1, 6>(acc, big, LY::LDQ, (const T*)w.wint, 192, nt1, lane);
    const int n4 = wave * 16 + qr;
#pragma unroll
    for (int mt = 0; mt < INF_MT; ++mt) {
      const int row = mt * 16 + fr;
      const float4 r = *reinterpret_cast<const float4*>(b + row * LY::LDX + n4);
      const float v0 = r.x + acc[mt][0][0], v1 = r.y + acc[mt][0][1], v2 = r.z + acc[mt][0][2], v3 = r.w + acc[mt][0][3];
      if (row < nrows) st4(w.o_dx + (row0 + row) * TD + n4, v0, v1, v2, v3);
      if constexpr (TAIL) st4(a + row * LY::LDX + n4, v0, v1, v2, v3);  // dctx is dead: `a` takes dx_in (0 beyond nrows)
    }
  }
  if constexpr (TAIL) {
    __syncthreads();
    {  // ---- tokens 1..16: dc3 = (dx_in Wup) o [c3 > 0]; the token-0 rows of the tile are computed and dropped
      f32x4 acc[INF_MT][1];
      zero_acc(acc);
      block_gemm<T, INF_MT, 1, 2>(acc, a, LY::LDX, (const T*)tl.wupt, 64, nt1, lane);
      const int n4 = wave * 16 + qr;
#pragma unroll
      for (int mt = 0; mt < INF_MT; ++mt) {
        const int row = mt * 16 + fr;
        const int sm = row / NTOK, t = row - sm * NTOK;
        const bool ok = row < nrows && t > 0;
        const int64_t o = ok ? ((int64_t)(s0 + sm) * 16 + (t - 1)) * TD + n4 : 0;
        const float4 m = *reinterpret_cast<const float4*>(tl.s_c3 + o);
        if (ok)
          st4(tl.o_dc3 + o, m.x > 0.f ? acc[mt][0][0] : 0.f, m.y > 0.f ? acc[mt][0][1] : 0.f, m.z > 0.f ? acc[mt][0][2] : 0.f,
              m.w > 0.f ? acc[mt][0][3] : 0.f);
      }
    }
    // ---- token 0: (dx_in o [x0 > 0]) -> state_projector' -> [e1 > 0] -> dhc -> fc2' -> [e0 > 0] -> de0
    float* dt = big;                                   // [16][LDX] (dqkv was consumed by the in_proj data-grad)
    T* dh = reinterpret_cast<T*>(big + 16 * LY::LDX);  // [16][LDF]
    __syncthreads();
    for (int idx = tid; idx < 16 * TD; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      const bool ok = r < ns;
      const float m = tl.x0[(row0 + (ok ? r * NTOK : 0)) * TD + c];
      dt[r * LY::LDX + c] = ok && m > 0.f ? a[(r * NTOK) * LY::LDX + c] : 0.f;
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto masked = [&](const float* act, T* dst, float* save) {
      const bool ok = fr < ns;
      float4 m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = *reinterpret_cast<const float4*>(act + (int64_t)(s0 + (ok ? fr : 0)) * 256 + nt4[j] * 16 + qr);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[0][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[0][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[0][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[0][j][3] : 0.f;
        if (dst != nullptr) st4(dst + fr * LY::LDF + n4, d0, d1, d2, d3);
        if (ok) st4(save + (int64_t)(s0 + fr) * 256 + n4, d0, d1, d2, d3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 2>(acc, dt, LY::LDX, (const T*)tl.wpt, 64, nt4, lane);
    masked(tl.s_e1, dh, tl.o_dhc);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh, LY::LDF, (const T*)tl.wf2t, 256, nt4, lane);
    masked(tl.s_e0, (T*)nullptr, tl.o_de0);
  }
}

}  // namespace v4l
