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

struct BwdLayerStack { BwdLayer l[2]; };  // NL = 2: the block walks layer l[0] (the upper one) and then l[1] backward, dx staying in LDS

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

template <typename T, int SPW> struct BwdLayLds {
  static constexpr int ROWS = InfRows<SPW>::ROWS;
  static constexpr int PAD = InfLd<T>::PAD;
  static constexpr int LDX = 64 + 4, LDQ = 192 + 4, LDF = 256 + PAD;
  static constexpr size_t a_b = (size_t)ROWS * LDX * 4;
  static constexpr size_t qkv_b = (size_t)ROWS * LDQ * 4;
  static constexpr size_t f_b = (size_t)ROWS * LDF * sizeof(T);
  static constexpr size_t big_b = qkv_b > f_b ? qkv_b : f_b;
  static constexpr size_t p_b = (size_t)2 * SPW * NTOK * ATT_PLD * 4;  // P and dS of the block's samples
  static constexpr size_t red_b = (size_t)2 * 16 * TD * 4;
  // bf16: the attention backward runs on MFMA tiles; every sample (= wave) owns 32 KB of operands in T, laid out so that each
  // fragment is one 16-byte LDS read (the contraction index is contiguous): dctx [tok][64] and V [tok][64] for dP = dctx V^T,
  // dctx^T / K^T / Q^T [64][tok] and dS, dS^T, P^T [tok][tok] for dV = P^T dctx, dQ = dS K, dK = dS^T Q. The dq | dk | dv rows
  // (operand of the in_proj data-grad) then take the place of dctx / V. The region is a union with a | big | P,dS.
  static constexpr bool MFMA_ATT = sizeof(T) == 2;
  static constexpr int LDA = 64 + 8, LDK = 32 + 8, LDR = 192 + 8;
  static constexpr size_t o_dc = 0, o_v = o_dc + (size_t)32 * LDA * 2, o_dct = o_v + (size_t)32 * LDA * 2;
  static constexpr size_t o_kt = o_dct + (size_t)64 * LDK * 2, o_qt = o_kt + (size_t)64 * LDK * 2, o_ds = o_qt + (size_t)64 * LDK * 2;
  static constexpr size_t o_dst = o_ds + (size_t)32 * LDK * 2, o_pt = o_dst + (size_t)32 * LDK * 2, att_w = 32768;
  static_assert(o_pt + (size_t)32 * LDK * 2 <= att_w && (size_t)18 * LDR * 2 <= o_dct, "attention operand block");
  static constexpr size_t att_b = MFMA_ATT ? (size_t)SPW * att_w : 0;
  static constexpr size_t uni_b = a_b + big_b + p_b > att_b ? a_b + big_b + p_b : att_b;
  static constexpr size_t bytes = a_b + red_b + uni_b;  // b | LN partials [2][16 quarter waves][64] | { a | df / qkv->dqkv | P,dS } U attention blocks
  static_assert(bytes <= 160 * 1024, "one block per CU");
};

// LayerNorm backward, in place over the ROWS LDS rows of `d` (rows >= nrows hold zeros and stay zero); the rows < nrows
// also go to o_dz (global). Leaves the block's dgamma/dbeta partial in gpart/bpart[64]. Contains one __syncthreads.
// Like ln_rows (infer.h): a row is owned by a quarter wave (16 lanes x 4 consecutive columns), both row reductions are
// DPP row_ror adds, accesses are 16 bytes; the dgamma / dbeta column sums are kept per lane and combined over the 16
// quarter waves of the block through `red` ([2][16][64]) in a fixed order.
// The saved xhat / rstd rows of a LayerNorm backward, fetched from HBM ahead of the phase that consumes them (the kernel
// runs one wave per SIMD with registers to spare: the fetch overlaps the GEMM in front of the norm).
template <int ROWS> struct LnPre { float4 x[ROWS / 16]; float rr[ROWS / 16]; float4 g; };
template <int ROWS>
__device__ __forceinline__ LnPre<ROWS> ln_bwd_fetch(const float* __restrict__ xh, const float* __restrict__ rs,
                                                    const float* __restrict__ gamma, int wave, int lane, int nrows) {
  LnPre<ROWS> pre;
  const int c4 = (lane & 15) * 4, q = lane >> 4;
  pre.g = *reinterpret_cast<const float4*>(gamma + c4);
#pragma unroll
  for (int it = 0; it < ROWS / 16; ++it) {  // loads are unconditional (row 0 stands in for the padding rows), then selected
    const int r = it * 16 + wave * 4 + q;
    const bool ok = r < nrows;
    const int o = ok ? r : 0;
    const float4 xv = *reinterpret_cast<const float4*>(xh + o * TD + c4);
    const float rv = rs[o];
    pre.x[it] = ok ? xv : float4{0.f, 0.f, 0.f, 0.f};
    pre.rr[it] = ok ? rv : 0.f;
  }
  return pre;
}
template <int ROWS, typename T>
__device__ __forceinline__ void ln_bwd_rows(float* d, int ld, const LnPre<ROWS>& pre, int wave, int lane, int nrows,
                                            T* __restrict__ o_dz, float* red, float* __restrict__ gpart,
                                            float* __restrict__ bpart) {
  const int l16 = lane & 15, c4 = l16 * 4, q = lane >> 4;
  const float4 g = pre.g;
  constexpr int IT = ROWS / 16;
  float4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  float4 dd[IT];
  const float4 (&x)[IT] = pre.x;
  const float (&rr)[IT] = pre.rr;
#pragma unroll
  for (int it = 0; it < IT; ++it) dd[it] = *reinterpret_cast<const float4*>(d + (it * 16 + wave * 4 + q) * ld + c4);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int r = it * 16 + wave * 4 + q;
    ag.x = fmaf(dd[it].x, x[it].x, ag.x); ag.y = fmaf(dd[it].y, x[it].y, ag.y);
    ag.z = fmaf(dd[it].z, x[it].z, ag.z); ag.w = fmaf(dd[it].w, x[it].w, ag.w);
    ab.x += dd[it].x; ab.y += dd[it].y; ab.z += dd[it].z; ab.w += dd[it].w;
    const float4 dxh = {dd[it].x * g.x, dd[it].y * g.y, dd[it].z * g.z, dd[it].w * g.w};
    float c1 = (dxh.x + dxh.y) + (dxh.z + dxh.w);
    float c2 = (dxh.x * x[it].x + dxh.y * x[it].y) + (dxh.z * x[it].z + dxh.w * x[it].w);
    c1 += dpp_mov<0x128>(c1); c1 += dpp_mov<0x124>(c1); c1 += dpp_mov<0x122>(c1); c1 += dpp_mov<0x121>(c1);
    c2 += dpp_mov<0x128>(c2); c2 += dpp_mov<0x124>(c2); c2 += dpp_mov<0x122>(c2); c2 += dpp_mov<0x121>(c2);
    c1 *= (1.f / TD);
    c2 *= (1.f / TD);
    const float4 dz = {rr[it] * (dxh.x - c1 - x[it].x * c2), rr[it] * (dxh.y - c1 - x[it].y * c2),
                       rr[it] * (dxh.z - c1 - x[it].z * c2), rr[it] * (dxh.w - c1 - x[it].w * c2)};
    *reinterpret_cast<float4*>(d + r * ld + c4) = dz;
    if (r < nrows) st4(o_dz + r * TD + c4, dz.x, dz.y, dz.z, dz.w);
  }
  const int slot = wave * 4 + q;
  *reinterpret_cast<float4*>(red + slot * TD + c4) = ag;
  *reinterpret_cast<float4*>(red + 16 * TD + slot * TD + c4) = ab;
  __syncthreads();
  if (wave == 0) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { sg += red[k * TD + lane]; sb += red[16 * TD + k * TD + lane]; }
    gpart[lane] = sg;
    bpart[lane] = sb;
  }
}

#ifdef V4L_INFER_TIMING
#define LAY_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_inf_stamps[32 + (i)] = clock64(); } while (0)  // stacked launch: the lowest layer's pass overwrites the upper one's
#else
#define LAY_STAMP(i)
#endif

template <typename T, int SPW, bool HEAD, bool TAIL, int NL = 1>
__global__ __launch_bounds__(256) void bwd_layer_kernel(BwdLayerStack stk, BwdHead hd, BwdTail tl, int n) {
  typedef BwdLayLds<T, SPW> LY;
  constexpr int ROWS = InfRows<SPW>::ROWS, MT = InfRows<SPW>::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, qr = (lane >> 4) * 4;
  float* b = reinterpret_cast<float*>(smem);                         // dx1 -> dz1
  float* red = reinterpret_cast<float*>(smem + LY::a_b);
  unsigned char* att = smem + LY::a_b + LY::red_b;                   // bf16: per-sample attention operand blocks (union with a | big | sp)
  float* a = reinterpret_cast<float*>(att);                          // dy -> dz2 -> dctx (fp32 mode)
  float* big = reinterpret_cast<float*>(att + LY::a_b);              // df (T) -> qkv -> dqkv (fp32 mode)
  float* sp = reinterpret_cast<float*>(att + LY::a_b + LY::big_b);   // P, dS (fp32 mode)
  (void)sp;
  // TAIL: dx_in rows for the encoder-side data-grads. fp32 mode: `a` (dctx is dead by then); bf16: the other waves may still be
  // reading dq | dk | dv out of sample 0's block, so the rows go behind sample 1's (dead: its attention is done)
  float* at = LY::MFMA_ATT ? reinterpret_cast<float*>(att + LY::att_w + LY::o_dct) : a;
  (void)at;
  LAY_STAMP(0);
  const int s0 = blockIdx.x * SPW;
  const int ns = min(SPW, n - s0);
  const int nrows = ns * NTOK;
  const int64_t row0 = (int64_t)s0 * NTOK;
  // norm2's saved rows start their trip from HBM now; they are consumed after dy is staged (or the heads are done)
  LnPre<ROWS> pre2 = ln_bwd_fetch<ROWS>(stk.l[0].s_xh2 + row0 * TD, stk.l[0].s_rs2 + row0, stk.l[0].g2, wave, lane, nrows);
  const int nt1[1] = {wave};
  const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  if constexpr (HEAD) {
    constexpr int LDP = 128 + 4;
    float* dt = big;                                            // [16][LDX]: dout rows, zero padded to 64 columns
    T* dh1 = reinterpret_cast<T*>(big + 16 * LY::LDX);          // [16][LDF]
    T* dh0 = dh1 + 16 * LY::LDF;
    float* dpool = reinterpret_cast<float*>(dh0 + 16 * LY::LDF);  // [16][LDP]
    // all HBM operands of the head chain are requested up front: dout, both ReLU masks, the three GEMMs' first fragments
    float dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 256, r = idx >> 6, c = idx & 63;
      const bool ok = r < ns && c < OUT_LD;
      const float v = hd.dout[ok ? (int64_t)(s0 + r) * OUT_LD + c : 0];
      dv[k] = ok ? v : 0.f;
    }
    float4 m1[4], m0[4];
    {
      const int64_t mrow = (int64_t)(s0 + (fr < ns ? fr : 0)) * 256;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        m1[j] = *reinterpret_cast<const float4*>(hd.s_h1 + mrow + nt4[j] * 16 + qr);
        m0[j] = *reinterpret_cast<const float4*>(hd.s_h0 + mrow + nt4[j] * 16 + qr);
      }
    }
    const int nt2[2] = {wave * 2, wave * 2 + 1};
    GemmRing<T, 4, 2> ring2 = gemm_prefetch<T, 4, 2>((const T*)hd.w2t, 64, nt4, lane);
    GemmRing<T, 4, 8> ring1 = gemm_prefetch<T, 4, 8>((const T*)hd.w1t, 256, nt4, lane);
    GemmRing<T, 2, 8> ring0 = gemm_prefetch<T, 2, 8>((const T*)hd.w0t, 256, nt2, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 256, r = idx >> 6, c = idx & 63;
      dt[r * LY::LDX + c] = dv[k];
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto masked = [&](const float4 (&m)[4], T* dst, float* save) {  // ReLU mask from the saved activation, rows < ns
      const bool ok = fr < ns;
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
    block_gemm<T, 1, 4, 2>(acc, dt, LY::LDX, (const T*)hd.w2t, 64, nt4, lane, ring2);
    masked(m1, dh1, hd.o_dh1);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh1, LY::LDF, (const T*)hd.w1t, 256, nt4, lane, ring1);
    masked(m0, dh0, hd.o_dh0);
    __syncthreads();
    {
      f32x4 a2[1][2];
      zero_acc(a2);
      block_gemm<T, 1, 2, 8>(a2, dh0, LY::LDF, (const T*)hd.w0t, 256, nt2, lane, ring0);
#pragma unroll
      for (int j = 0; j < 2; ++j) st4(dpool + fr * LDP + nt2[j] * 16 + qr, a2[0][j][0], a2[0][j][1], a2[0][j][2], a2[0][j][3]);
    }
    __syncthreads();
    // un-pool (pool_bwd_kernel): token 0 <- dpool[:, 0:64], tokens 1..16 <- dpool[:, 64:128] / 16
    for (int idx = tid; idx < ROWS * TD; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      float v = 0.f;
      if (r < nrows) {
        const int sm = r / NTOK, t = r - sm * NTOK;
        v = t == 0 ? dpool[sm * LDP + c] : dpool[sm * LDP + TD + c] * (1.f / 16.f);
      }
      a[r * LY::LDX + c] = v;
    }
  } else {
    const float* dyg = stk.l[0].dy + row0 * TD;
    for (int i4 = tid; i4 < ROWS * (TD / 4); i4 += 256) {
      const int r = i4 >> 4, c4 = (i4 & 15) * 4;
      const bool ok = r < nrows;
      const float4 v = *reinterpret_cast<const float4*>(dyg + (ok ? r : 0) * TD + c4);
      *reinterpret_cast<float4*>(a + r * LY::LDX + c4) = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  LAY_STAMP(1);
  float4 tm_c3[MT], tm_e1[4], tm_e0[4];  // TAIL: encoder-side masks / first weight fragments, requested by the lowest layer
  float tm_x0[4];
  GemmRing<T, 1, 2> ring_up;
  GemmRing<T, 4, 2> ring_pr;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
  const BwdLayer& w = stk.l[l];
  const bool last = l == NL - 1;  // compile-time after unrolling: the TAIL part belongs to the lowest layer
  if (l > 0) { LAY_STAMP(0); LAY_STAMP(1); }
  __syncthreads();
  // ---- norm2 backward: a = dz2 (the next GEMM's first weight fragments are requested before the norm's global stores)
  GemmRing<T, 4, 2> ring_df = gemm_prefetch<T, 4, 2>((const T*)w.w2t, 64, nt4, lane);
  ln_bwd_rows<ROWS>(a, LY::LDX, pre2, wave, lane, nrows, reinterpret_cast<T*>(w.o_dz2) + row0 * TD, red,
              w.gp2 + (int64_t)blockIdx.x * TD, w.bp2 + (int64_t)blockIdx.x * TD);
  __syncthreads();
  LAY_STAMP(2);
  // ---- df = (dz2 W2) o [f > 0]   (T in LDS for the next contraction, fp32 to HBM for linear1's weight-grad)
  T* f = reinterpret_cast<T*>(big);
  GemmRing<T, 1, 8> ring_dx1;
  {
    f32x4 acc[MT][4];
    zero_acc(acc);
    float4 fm[MT][4];  // the ReLU mask (saved f) is fetched before the GEMM, not after it
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fm[mt][j] = ld4(reinterpret_cast<const T*>(w.s_f) + (row0 + (mt * 16 + fr < nrows ? mt * 16 + fr : 0)) * 256 +
                        nt4[j] * 16 + qr);
    block_gemm<T, MT, 4, 2>(acc, a, LY::LDX, (const T*)w.w2t, 64, nt4, lane, ring_df);
    LAY_STAMP(10);
    ring_dx1 = gemm_prefetch<T, 1, 8>((const T*)w.w1t, 256, nt1, lane);  // ahead of this epilogue's global stores
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fr;
      const bool ok = row < nrows;
      const float4 (&m)[4] = fm[mt];
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
  LAY_STAMP(11);
  __syncthreads();
  LAY_STAMP(3);
  // qkv and P of the block's samples and norm1's saved rows are requested before the dx1 GEMM and parked in registers;
  // they go to LDS once `big` (df) has been consumed
  constexpr int QV = (ROWS * 48 + 255) / 256, PV = (SPW * NTOK * NTOK + 255) / 256;
  float4 qpre[QV];
  float ppre[PV];
  float pp[2][2][4];
  (void)ppre; (void)pp;
  {
    const float* qg = w.s_qkv + row0 * 192;
#pragma unroll
    for (int k = 0; k < QV; ++k) {
      const int i4 = tid + k * 256;
      const int r = i4 / 48, c4 = (i4 - r * 48) * 4;
      const bool ok = i4 < ROWS * 48 && r < nrows;
      const float4 v = *reinterpret_cast<const float4*>(qg + (ok ? r * 192 + c4 : 0));
      qpre[k] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (LY::MFMA_ATT) {  // P of sample `wave`, as the dP accumulator tiles will hold it: row 16 m + fr, columns 16 nn + qr + r
      const float* pg = w.s_P + (int64_t)(s0 + (wave < ns ? wave : 0)) * NTOK * NTOK;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = m * 16 + fr, j = nn * 16 + qr + r;
            const bool ok = i < NTOK && j < NTOK && wave < ns;
            const float v = pg[ok ? i * NTOK + j : 0];
            pp[m][nn][r] = ok ? v : 0.f;
          }
    } else {
      const float* pg = w.s_P + (int64_t)s0 * NTOK * NTOK;
#pragma unroll
      for (int k = 0; k < PV; ++k) {
        const int idx = tid + k * 256;
        ppre[k] = pg[idx < ns * NTOK * NTOK ? idx : 0];
      }
    }
  }
  const LnPre<ROWS> pre1 = ln_bwd_fetch<ROWS>(w.s_xh1 + row0 * TD, w.s_rs1 + row0, w.g1, wave, lane, nrows);
  {  // ---- dx1 = dz2 + df W1 -> b
    f32x4 acc[MT][1];
    zero_acc(acc);
    LAY_STAMP(12);
    block_gemm<T, MT, 1, 8>(acc, f, LY::LDF, (const T*)w.w1t, 256, nt1, lane, ring_dx1);
    LAY_STAMP(13);
    const int n4 = wave * 16 + qr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fr;
      const float4 r = *reinterpret_cast<const float4*>(a + row * LY::LDX + n4);
      st4(b + row * LY::LDX + n4, r.x + acc[mt][0][0], r.y + acc[mt][0][1], r.z + acc[mt][0][2], r.w + acc[mt][0][3]);
    }
  }
  __syncthreads();
  LAY_STAMP(4);
  // park -> LDS (`big` is free: df was consumed by the dx1 GEMM)
  if constexpr (LY::MFMA_ATT) {
    // contraction columns 17..31 of dctx^T / K^T / Q^T meet exact zeros of dS / dS^T / P^T: they only have to be finite.
    // Zeroed without touching column 16 (2 + 4 + 8 + 16 bytes), so no ordering against the data writes is needed.
    for (int rr = tid; rr < ns * 3 * 64; rr += 256) {
      const int sm = rr / 192, q = rr - sm * 192;  // q = matrix * 64 + row: dctx^T, K^T, Q^T are consecutive
      T* row = reinterpret_cast<T*>(att + (size_t)sm * LY::att_w + LY::o_dct) + q * LY::LDK;
      row[17] = (T)0.f;
      *reinterpret_cast<uint32_t*>(row + 18) = 0u;
      *reinterpret_cast<uint2*>(row + 20) = uint2{0u, 0u};
      *reinterpret_cast<uint4*>(row + 24) = uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < QV; ++k) {
      const int i4 = tid + k * 256;
      const int r = i4 / 48, c4 = (i4 - r * 48) * 4;
      const int sm = r / NTOK, t = r - sm * NTOK;
      if (i4 < ROWS * 48 && r < nrows) {  // q, k: transposed ([feature][token]); v: as stored. Rounded to T here, once.
        unsigned char* aw = att + (size_t)sm * LY::att_w;
        const float4 v = qpre[k];
        if (c4 < 128) {
          T* col = reinterpret_cast<T*>(aw + (c4 < 64 ? LY::o_qt : LY::o_kt)) + (c4 & 63) * LY::LDK + t;
          col[0] = (T)v.x; col[LY::LDK] = (T)v.y; col[2 * LY::LDK] = (T)v.z; col[3 * LY::LDK] = (T)v.w;
        } else {
          st4(reinterpret_cast<T*>(aw + LY::o_v) + t * LY::LDA + (c4 - 128), v.x, v.y, v.z, v.w);
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < QV; ++k) {
      const int i4 = tid + k * 256;
      const int r = i4 / 48, c4 = (i4 - r * 48) * 4;
      // q | k | v as the attention products see them: rounded to T (the forward's attention runs on MFMA tiles of T, infer.h
      // attn_tile); they are operands of those products only
      if (i4 < ROWS * 48) *reinterpret_cast<float4*>(big + r * LY::LDQ + c4) = rt4<T>(qpre[k]);
    }
#pragma unroll
    for (int k = 0; k < PV; ++k) {
      const int idx = tid + k * 256;
      if (idx < ns * NTOK * NTOK) {
        const int sm = idx / (NTOK * NTOK), pr = idx - sm * NTOK * NTOK;
        const int i = pr / NTOK, j = pr - i * NTOK;
        sp[(sm * NTOK + i) * ATT_PLD + j] = ppre[k];
      }
    }
  }
  LAY_STAMP(14);
  // ---- norm1 backward: b = dz1
  GemmRing<T, 1, 2> ring_dctx = gemm_prefetch<T, 1, 2>((const T*)w.wot, 64, nt1, lane);
  ln_bwd_rows<ROWS>(b, LY::LDX, pre1, wave, lane, nrows, reinterpret_cast<T*>(w.o_dz1) + row0 * TD, red,
              w.gp1 + (int64_t)blockIdx.x * TD, w.bp1 + (int64_t)blockIdx.x * TD);
  __syncthreads();
  LAY_STAMP(5);
  {  // ---- dctx = dz1 Wo -> a
    f32x4 acc[MT][1];
    zero_acc(acc);
    block_gemm<T, MT, 1, 2>(acc, b, LY::LDX, (const T*)w.wot, 64, nt1, lane, ring_dctx);
    const int n4 = wave * 16 + qr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {  // dctx is an operand of the attention products only: kept rounded to T
      if constexpr (LY::MFMA_ATT) {
        const int row = mt * 16 + fr, sm = row / NTOK, t = row - sm * NTOK;
        if (row < nrows) {  // [token][feature] for dP, [feature][token] for dV
          unsigned char* aw = att + (size_t)sm * LY::att_w;
          st4(reinterpret_cast<T*>(aw + LY::o_dc) + t * LY::LDA + n4, acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]);
          T* col = reinterpret_cast<T*>(aw + LY::o_dct) + n4 * LY::LDK + t;
          col[0] = (T)acc[mt][0][0]; col[LY::LDK] = (T)acc[mt][0][1]; col[2 * LY::LDK] = (T)acc[mt][0][2]; col[3 * LY::LDK] = (T)acc[mt][0][3];
        }
      } else {
        st4(a + (mt * 16 + fr) * LY::LDX + n4, rt<T>(acc[mt][0][0]), rt<T>(acc[mt][0][1]), rt<T>(acc[mt][0][2]), rt<T>(acc[mt][0][3]));
      }
    }
  }
  __syncthreads();
  LAY_STAMP(6);
  // ---- attention backward of sample `wave` (fp32 VALU on operands rounded to T: the products the forward's MFMA tiles make):
  //   dP = dctx V^T ; dS = P o (dP - rowsum(P o dP)) ; dV = P^T dctx ; dQ = dS K / 8 ; dK = dS^T Q / 8
  GemmRing<T, 1, 6> ring_dxin = gemm_prefetch<T, 1, 6>((const T*)w.wint, 192, nt1, lane);
  // TAIL: everything the encoder-side data-grads read from HBM (conv3 / x0 / MLP ReLU masks, the first weight fragments of
  // their three GEMMs) is requested here, in front of the long LDS-only attention phase
  LnPre<ROWS> pre2_next = pre2;
  if (!last) {  // the next (lower) layer's norm2 rows start their trip here, in front of the long LDS-only attention phase
    const BwdLayer& wn = stk.l[l + 1 < NL ? l + 1 : l];
    pre2_next = ln_bwd_fetch<ROWS>(wn.s_xh2 + row0 * TD, wn.s_rs2 + row0, wn.g2, wave, lane, nrows);
  }
  if (TAIL && last) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fr;
      const int sm = row / NTOK, t = row - sm * NTOK;
      const bool ok = row < nrows && t > 0;
      tm_c3[mt] = *reinterpret_cast<const float4*>(tl.s_c3 + (ok ? ((int64_t)(s0 + sm) * 16 + (t - 1)) * TD + wave * 16 + qr : 0));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 256, r = idx >> 6, c = idx & 63;
      tm_x0[k] = tl.x0[(row0 + (r < ns ? r * NTOK : 0)) * TD + c];
    }
    const int64_t mrow = (int64_t)(s0 + (fr < ns ? fr : 0)) * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tm_e1[j] = *reinterpret_cast<const float4*>(tl.s_e1 + mrow + nt4[j] * 16 + qr);
      tm_e0[j] = *reinterpret_cast<const float4*>(tl.s_e0 + mrow + nt4[j] * 16 + qr);
    }
    ring_up = gemm_prefetch<T, 1, 2>((const T*)tl.wupt, 64, nt1, lane);
    ring_pr = gemm_prefetch<T, 4, 2>((const T*)tl.wpt, 64, nt4, lane);
  }
  if constexpr (LY::MFMA_ATT) {
    typedef typename Frag<T>::type frag_t;
    if (wave < ns) {  // one wave = one sample, on MFMA tiles out of its own operand block (no block-level sync inside)
      unsigned char* aw = att + (size_t)wave * LY::att_w;
      const T* dC = reinterpret_cast<const T*>(aw + LY::o_dc);
      const T* Vv = reinterpret_cast<const T*>(aw + LY::o_v);
      const T* dCt = reinterpret_cast<const T*>(aw + LY::o_dct);
      const T* Kt = reinterpret_cast<const T*>(aw + LY::o_kt);
      const T* Qt = reinterpret_cast<const T*>(aw + LY::o_qt);
      T* dS = reinterpret_cast<T*>(aw + LY::o_ds);
      T* dSt = reinterpret_cast<T*>(aw + LY::o_dst);
      T* Pt = reinterpret_cast<T*>(aw + LY::o_pt);
      const int fg = (lane >> 4) * 8;
      f32x4 ap[2][2];  // dP[i = 16 m + fr][j = 16 nn + qr + r] = sum_d dctx[i][d] V[j][d]
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) ap[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        frag_t fa[2], fb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          fa[m] = *reinterpret_cast<const frag_t*>(dC + (m * 16 + fr) * LY::LDA + ks * 32 + fg);
          fb[m] = *reinterpret_cast<const frag_t*>(Vv + (m * 16 + fr) * LY::LDA + ks * 32 + fg);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn) mma_k32(ap[m][nn], fb[nn], fa[m]);
      }
      // softmax': dS = P o (dP - rowsum(P o dP)); rows / columns >= 17 of the tiles are padding (their dP may be anything:
      // selected away, never multiplied). dS and P leave rounded to T: operands of the three products below only.
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int i = m * 16 + fr;
        float rd = 0.f;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = i < NTOK && nn * 16 + qr + r < NTOK;
            rd = fmaf(pp[m][nn][r], ok ? ap[m][nn][r] : 0.f, rd);
          }
        rd += __shfl_xor(rd, 16, 64);
        rd += __shfl_xor(rd, 32, 64);
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
          float d[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = i < NTOK && nn * 16 + qr + r < NTOK;
            d[r] = ok ? pp[m][nn][r] * (ap[m][nn][r] - rd) : 0.f;
          }
          const int j0 = nn * 16 + qr;
          st4(dS + i * LY::LDK + j0, d[0], d[1], d[2], d[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dSt[(j0 + r) * LY::LDK + i] = (T)d[r];
            Pt[(j0 + r) * LY::LDK + i] = (T)pp[m][nn][r];
          }
        }
      }
      // dq | dk | dv rows of this sample ([17][LDR] in T + one zero row for the padding rows of the GEMM below) over dctx / V,
      // which the dP tiles have consumed
      T* R = reinterpret_cast<T*>(aw);
      if (lane < LY::LDR * 2 / 16) *reinterpret_cast<uint4*>(R + NTOK * LY::LDR + lane * 8) = uint4{0u, 0u, 0u, 0u};
      T* og = reinterpret_cast<T*>(w.o_dqkv) + (row0 + wave * NTOK) * 192;
      auto product = [&](const T* A, const T* B, float scale, int coff) {  // out[x][d] = scale * sum_y A[x][y] B[d][y]
        f32x4 acc[2][4];
        frag_t fa[2], fb[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) fa[m] = *reinterpret_cast<const frag_t*>(A + (m * 16 + fr) * LY::LDK + fg);
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) fb[nn] = *reinterpret_cast<const frag_t*>(B + (nn * 16 + fr) * LY::LDK + fg);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) {
            acc[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_k32(acc[m][nn], fb[nn], fa[m]);
          }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int x = m * 16 + fr;
          if (x < NTOK) {
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
              const float v0 = acc[m][nn][0] * scale, v1 = acc[m][nn][1] * scale, v2 = acc[m][nn][2] * scale, v3 = acc[m][nn][3] * scale;
              st4(R + x * LY::LDR + coff + nn * 16 + qr, v0, v1, v2, v3);
              st4(og + x * 192 + coff + nn * 16 + qr, v0, v1, v2, v3);
            }
          }
        }
      };
      product(dS, Kt, 0.125f, 0);        // dQ = dS K / 8
      product(dSt, Qt, 0.125f, TD);      // dK = dS^T Q / 8
      product(Pt, dCt, 1.f, 2 * TD);     // dV = P^T dctx
    }
  } else {
    const bool act = wave < ns;
    float* qs = big + wave * NTOK * LY::LDQ;
    float* p = sp + wave * NTOK * ATT_PLD;
    float* ds = sp + (SPW + wave) * NTOK * ATT_PLD;
    const float* dc = a + wave * NTOK * LY::LDX;
    if (act) {
      if (lane < 54) {  // dP in 2x3 register tiles (rows 2a,2a+1 x keys 3b..3b+2): 5 LDS rows feed 6 dot products
        const int a2 = lane / 6, b3 = lane - a2 * 6;
        const int i0 = 2 * a2, i1 = min(i0 + 1, NTOK - 1), j0 = 3 * b3, j1 = min(j0 + 1, NTOK - 1), j2 = min(j0 + 2, NTOK - 1);
        const float *da = dc + i0 * LY::LDX, *db = dc + i1 * LY::LDX;
        const float *va = qs + j0 * LY::LDQ + 2 * TD, *vb = qs + j1 * LY::LDQ + 2 * TD, *vc = qs + j2 * LY::LDQ + 2 * TD;
        float s00 = 0.f, s01 = 0.f, s02 = 0.f, s10 = 0.f, s11 = 0.f, s12 = 0.f;
#pragma unroll 4
        for (int d = 0; d < TD; d += 4) {
          // (dctx and q | k | v sit in LDS already rounded to T: they are operands of these products only)
          const float4 x0 = *reinterpret_cast<const float4*>(da + d), x1 = *reinterpret_cast<const float4*>(db + d);
          const float4 y0 = *reinterpret_cast<const float4*>(va + d), y1 = *reinterpret_cast<const float4*>(vb + d);
          const float4 y2 = *reinterpret_cast<const float4*>(vc + d);
          s00 = fmaf(x0.x, y0.x, s00); s00 = fmaf(x0.y, y0.y, s00); s00 = fmaf(x0.z, y0.z, s00); s00 = fmaf(x0.w, y0.w, s00);
          s01 = fmaf(x0.x, y1.x, s01); s01 = fmaf(x0.y, y1.y, s01); s01 = fmaf(x0.z, y1.z, s01); s01 = fmaf(x0.w, y1.w, s01);
          s02 = fmaf(x0.x, y2.x, s02); s02 = fmaf(x0.y, y2.y, s02); s02 = fmaf(x0.z, y2.z, s02); s02 = fmaf(x0.w, y2.w, s02);
          s10 = fmaf(x1.x, y0.x, s10); s10 = fmaf(x1.y, y0.y, s10); s10 = fmaf(x1.z, y0.z, s10); s10 = fmaf(x1.w, y0.w, s10);
          s11 = fmaf(x1.x, y1.x, s11); s11 = fmaf(x1.y, y1.y, s11); s11 = fmaf(x1.z, y1.z, s11); s11 = fmaf(x1.w, y1.w, s11);
          s12 = fmaf(x1.x, y2.x, s12); s12 = fmaf(x1.y, y2.y, s12); s12 = fmaf(x1.z, y2.z, s12); s12 = fmaf(x1.w, y2.w, s12);
        }
        float* r0 = ds + i0 * ATT_PLD;
        float* r1 = ds + i1 * ATT_PLD;
        r0[j0] = s00;
        if (j0 + 1 < NTOK) r0[j0 + 1] = s01;
        if (j0 + 2 < NTOK) r0[j0 + 2] = s02;
        if (i0 + 1 < NTOK) {
          r1[j0] = s10;
          if (j0 + 1 < NTOK) r1[j0 + 1] = s11;
          if (j0 + 2 < NTOK) r1[j0 + 2] = s12;
        }
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
      for (int j = 0; j < NTOK; ++j) {
        // dS and, from here on, P are operands of the remaining products only (dQ / dK; dV): kept rounded to T
        const float pj = p[lane * ATT_PLD + j];
        ds[lane * ATT_PLD + j] = rt<T>(pj * (ds[lane * ATT_PLD + j] - rd));
        p[lane * ATT_PLD + j] = rt<T>(pj);
      }
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
  LAY_STAMP(7);
  {  // ---- dx_in = dz1 + dqkv Win -> global
    f32x4 acc[MT][1];
    zero_acc(acc);
    if constexpr (LY::MFMA_ATT) {  // the dq | dk | dv rows sit in the samples' operand blocks, already in T
      typedef typename Frag<T>::type frag_t;
      const int fg = (lane >> 4) * 8;
      const T* rp[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + fr, sm = row / NTOK, t = row - sm * NTOK;
        rp[mt] = row < nrows ? reinterpret_cast<const T*>(att + (size_t)sm * LY::att_w) + t * LY::LDR
                             : reinterpret_cast<const T*>(att) + NTOK * LY::LDR;  // sample 0's zero row
      }
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          mma_k32(acc[mt][0], ring_dxin.fb[ks][0], *reinterpret_cast<const frag_t*>(rp[mt] + ks * 32 + fg));
    } else {
      block_gemm<T, MT, 1, 6>(acc, big, LY::LDQ, (const T*)w.wint, 192, nt1, lane, ring_dxin);
    }
    const int n4 = wave * 16 + qr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fr;
      const float4 r = *reinterpret_cast<const float4*>(b + row * LY::LDX + n4);
      acc[mt][0][0] += r.x; acc[mt][0][1] += r.y; acc[mt][0][2] += r.z; acc[mt][0][3] += r.w;
      if (row < nrows) st4(w.o_dx + (row0 + row) * TD + n4, acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]);
      if (TAIL && last) st4(at + row * LY::LDX + n4, acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]);  // `at` takes dx_in (0 beyond nrows)
    }
    if (!last) {
      // dx_in is the lower layer's dy: it goes straight into `a` (bf16: once every wave is done reading dq | dk | dv out of
      // the operand blocks `a` overlaps)
      if constexpr (LY::MFMA_ATT) __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        st4(a + (mt * 16 + fr) * LY::LDX + n4, acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]);
    }
  }
  pre2 = pre2_next;
  LAY_STAMP(8);
  }  // layers
  if constexpr (TAIL) {
    __syncthreads();
    {  // ---- tokens 1..16: dc3 = (dx_in Wup) o [c3 > 0]; the token-0 rows of the tile are computed and dropped
      f32x4 acc[MT][1];
      zero_acc(acc);
      block_gemm<T, MT, 1, 2>(acc, at, LY::LDX, (const T*)tl.wupt, 64, nt1, lane, ring_up);
      const int n4 = wave * 16 + qr;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + fr;
        const int sm = row / NTOK, t = row - sm * NTOK;
        const bool ok = row < nrows && t > 0;
        const int64_t o = ok ? ((int64_t)(s0 + sm) * 16 + (t - 1)) * TD + n4 : 0;
        const float4 m = tm_c3[mt];
        if (ok)
          st4(tl.o_dc3 + o, m.x > 0.f ? acc[mt][0][0] : 0.f, m.y > 0.f ? acc[mt][0][1] : 0.f, m.z > 0.f ? acc[mt][0][2] : 0.f,
              m.w > 0.f ? acc[mt][0][3] : 0.f);
      }
    }
    // ---- token 0: (dx_in o [x0 > 0]) -> state_projector' -> [e1 > 0] -> dhc -> fc2' -> [e0 > 0] -> de0
    float* dt = big;                                   // [16][LDX] (dqkv was consumed by the in_proj data-grad)
    T* dh = reinterpret_cast<T*>(big + 16 * LY::LDX);  // [16][LDF]
    GemmRing<T, 4, 8> ring_f2 = gemm_prefetch<T, 4, 8>((const T*)tl.wf2t, 256, nt4, lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 256, r = idx >> 6, c = idx & 63;
      dt[r * LY::LDX + c] = r < ns && tm_x0[k] > 0.f ? at[(r * NTOK) * LY::LDX + c] : 0.f;
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto masked = [&](const float4 (&m)[4], T* dst, float* save) {
      const bool ok = fr < ns;
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
    block_gemm<T, 1, 4, 2>(acc, dt, LY::LDX, (const T*)tl.wpt, 64, nt4, lane, ring_pr);
    masked(tm_e1, dh, tl.o_dhc);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh, LY::LDF, (const T*)tl.wf2t, 256, nt4, lane, ring_f2);
    masked(tm_e0, (T*)nullptr, tl.o_de0);
  }
  LAY_STAMP(9);
}


// ------------------------------------------------------------------------------------------ conv stack backward
// bwd_conv_kernel: the whole NatureCNN backward (networks/base.py:304-342 reversed) for the shipped geometry
// 4x64x64 -(k8 s4)-> 15x15x32 -(k4 s2)-> 6x6x64 -(k3 s1)-> 4x4x64, one sample at a time out of LDS:
//   dc3 -> dc2 = gather-form data-grad of conv3 (rows = the 36 input pixels, K = 9 taps x 64) o [c2 > 0]
//       -> [dW2 += dc2^T col(c1)]
//       -> dc1 = gather-form data-grad of conv2, one stride-parity class per wave pair (K = 4 taps x 64) o [c1 > 0]
//       -> [dW1 += dc1^T col(image)]
// Blocks are persistent (sample = blockIdx.x, += gridDim.x): the weight-grads accumulate in REGISTERS across a
// block's samples and leave the chip once per block as one slab each — every activation is read from HBM exactly once
// and dc2 / dc1 never exist in HBM. bwd_conv_kernel (512 threads, 1 block per CU) carries dW2 and dW1 (20 MFMA tiles per
// wave); dW3's 144 tiles would not fit beside them, so bwd_conv3_wgrad_kernel (256 threads, 36 tiles per wave) does that
// contraction on its own from dc3 and c2 (13 KB per sample).
// What shapes the per-sample loop (every phase opens with an L2 round trip and vmcnt is per wave, in order):
//   * every weight byte enters the CU once per sample: conv3' splits its ci tiles (and K halves) over the waves, conv2''s
//     64 KB of weights live in LDS in fragment order for the block's whole life (bf16);
//   * dc2 and dc1 exist only as MFMA operands: the data-grad epilogues store them in T, both as [pixel][co] (next data-grad)
//     and transposed [co][pixel] (weight-grad fragments = single 16-byte LDS reads), and add the bias gradients from the same
//     registers;
//   * nothing is waited for where it is requested: the image goes global -> LDS by DMA under dW2 / conv2' (LDS-only phases),
//     the NEXT sample's dc3 / c2 / c1 are requested under dW1 and filed after it.
struct BwdConv {
  const void* w3d;                  // conv3 data-grad pack (T) [ci 64][(a,b,co) 576]
  const void* w2d[4];               // conv2 data-grad packs, one per stride-parity class (py,px): [ci 32][(a,b,co) 256]
  const void* image;                // T [slots][4][64][64]
  const int* rowidx;                // minibatch row -> rollout slot, or null
  const float *c1, *c2;             // [n*225][32], [n*36][64] post-ReLU activations — fp32, or (kernels instantiated with A16) the
                                    // same arrays in T: what train_encoder_kernel writes for the trainer's own passes (round 5)
  const float* dc3;                 // [n*16][64] grad w.r.t. conv3's pre-activation (already ReLU-masked)
  float *slab1, *slab2, *slab3;     // [gridDim.x][32][256], [gridDim.x][64][512], [gridDim.x][64][576]
  float *bslab1, *bslab2, *bslab3;  // [gridDim.x][32], [gridDim.x][64], [gridDim.x][64]
  int n;
  void *t_dc2, *t_dc1;              // test taps (null in production): dc2 [n*36][64], dc1 [n*225][32] as the kernel holds them (T)
};
template <typename T> struct BwdConvLds {
  static constexpr bool B16 = sizeof(T) == 2;
  static constexpr int LF = 64 + 4;                 // c2 rows (fp32, the ReLU mask of conv3')
  static constexpr int LC1 = 32 + (B16 ? 8 : 0);    // c1 rows (T)
  static constexpr int IMGP = B16 ? 1 : 2;          // image passes (fp32: two channels at a time)
  static constexpr int LT = 64 + (B16 ? 8 : 4);     // [pixel][channel] rows in the compute type, read as whole MFMA fragments
  static constexpr int LP2 = 64 + (B16 ? 8 : 4);    // dc2^T rows [co][pixel 0..63]  (weight-grad operand: 8 pixels = one fragment)
  static constexpr int LP1 = 256 + (B16 ? 8 : 4);   // dc1^T rows [co][pixel 0..255]
  static constexpr size_t dc3_b = (size_t)17 * LT * sizeof(T);   // + one zero row (taps that fall outside the 4x4 plane)
  static constexpr size_t c2_b = (size_t)36 * LF * 4;
  static constexpr size_t dc2_b = (size_t)48 * LT * sizeof(T);   // conv2' operand; rows 36..47: zeros (MFMA padding / out-of-plane taps)
  static constexpr size_t dc2T_b = (size_t)64 * LP2 * sizeof(T); // dW2 operand; pixels 36..63: zeros
  static constexpr size_t c1_b = ((size_t)225 * LC1 * sizeof(T) + 15) / 16 * 16;
  static constexpr size_t dc1_b = (size_t)32 * LP1 * sizeof(T);  // dW1 operand (dc1^T); pixels 225..255: zeros
  static constexpr size_t img_b = (size_t)(4 / IMGP) * 4096 * sizeof(T);
  static constexpr size_t w2_b = B16 ? (size_t)4 * 32 * 256 * sizeof(T) : 0;  // conv2' weights, resident in fragment order (bf16 only)
  static constexpr size_t bytes = dc3_b + c2_b + dc2_b + dc2T_b + c1_b + dc1_b + img_b + w2_b;
  static_assert(bytes <= 160 * 1024, "one block per CU");
};

// MFMA operand (row = lane&15, 8 consecutive contraction indices 8*(lane>>4)+j) from 8 scalar values
template <typename T> __device__ __forceinline__ typename Frag<T>::type frag_of(const float (&v)[8]) {
  typename Frag<T>::type f;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (T)v[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = v[j];
  }
  return f;
}
template <typename H>
__device__ __forceinline__ typename Frag<H>::type frag_of_t(const H (&v)[8]) {
  static_assert(sizeof(H) == 2, "16-bit operand type");
  typename Frag<H>::type f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = v[j];
  return f;
}
__device__ __forceinline__ f32x8 frag_of_t(const float (&v)[8]) { return frag_of<float>(v); }

// Gather-form data-grad GEMM of one wave: acc[mt][j] += A(mt, ks) * W[row = (nt0 + j)*16 + lane&15][ks*32 ..]^T over the KS
// K=32 steps starting at KS0, where the A fragment of (row tile mt, step ks) comes from the LDS row (compute type T) arow(mt, ks >> 1) (64 channels = two
// steps per tap) and feeds NT column tiles. Weight fragments stream from L2 through a ring of PD steps, like block_gemm.
template <typename T, int MT, int NT, int KS, int KS0 = 0, int PDEPTH = 4, class RowF>
__device__ __forceinline__ void gather_gemm(f32x4 (&acc)[MT][NT], const T* __restrict__ W, int Kp, int nt0, int lane, RowF arow) {
  typedef typename Frag<T>::type frag_t;
  constexpr int PD = NT >= 2 ? 2 : (KS < PDEPTH ? KS : PDEPTH);
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  const T* wrow = W + (int64_t)(nt0 * 16 + fr) * Kp + fg + KS0 * 32;  // this call covers K steps KS0 .. KS0 + KS - 1
  frag_t fb[PD][NT];
#pragma unroll
  for (int d = 0; d < PD; ++d)
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[d][j] = *reinterpret_cast<const frag_t*>(wrow + (int64_t)j * 16 * Kp + d * 32);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    frag_t cur[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) cur[j] = fb[ks % PD][j];
    if (ks + PD < KS) {
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[ks % PD][j] = *reinterpret_cast<const frag_t*>(wrow + (int64_t)j * 16 * Kp + (ks + PD) * 32);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const frag_t fa = *reinterpret_cast<const frag_t*>(arow(mt, (ks + KS0) >> 1) + ((ks + KS0) & 1) * 32 + fg);  // rows already in T
#pragma unroll
      for (int j = 0; j < NT; ++j)
        mma_k32(acc[mt][j], cur[j], fa);  // transposed tile: acc[mt][j][r] = out[16*mt + lane&15][16*(nt0+j) + 4*(lane>>4) + r]
    }
  }
}

#ifdef V4L_INFER_TIMING
#define CONV_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && smp == 3 * (int)gridDim.x) g_inf_stamps[16 + (i)] = clock64(); } while (0)
#else
#define CONV_STAMP(i)
#endif
// A16 (bf16 only): a.c1 / a.c2 hold the activations in T — the type this kernel rounds them to when it files them into LDS
// anyway (c1: MFMA operand; c2: only its sign is used, as the ReLU mask of conv3'), so the results are bit-identical and the
// launch reads 19 KB less per sample (the training encoder writes 19 KB less).
template <typename T, bool A16 = false>
__global__ __launch_bounds__(512) void bwd_conv_kernel(BwdConv a) {
  static_assert(!A16 || sizeof(T) == 2, "A16: activations saved in the bf16 operand type");
  typedef BwdConvLds<T> LY;
  typedef typename Frag<T>::type frag_t;
  constexpr int NTH = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sdc3 = reinterpret_cast<T*>(smem);
  float* sc2 = reinterpret_cast<float*>(smem + LY::dc3_b);
  T* sdc2t = reinterpret_cast<T*>(smem + LY::dc3_b + LY::c2_b);
  T* sdc2T = reinterpret_cast<T*>(smem + LY::dc3_b + LY::c2_b + LY::dc2_b);
  constexpr size_t o_c1 = LY::dc3_b + LY::c2_b + LY::dc2_b + LY::dc2T_b;
  T* sc1 = reinterpret_cast<T*>(smem + o_c1);
  T* sdc1T = reinterpret_cast<T*>(smem + o_c1 + LY::c1_b);
  T* simg = reinterpret_cast<T*>(smem + o_c1 + LY::c1_b + LY::dc1_b);
  T* sw2 = reinterpret_cast<T*>(smem + o_c1 + LY::c1_b + LY::dc1_b + LY::img_b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  // weight-grad tiles of this wave
  const int kt2 = wave * 4;                    // conv2: k-tiles 4w..4w+3 x all 4 co-tiles
  const int ch1 = wave >> 1, th1 = wave & 1;   // conv1: input channel ch1, k-tiles 4*ch1 + 2*th1 + {0,1} x both co-tiles
  f32x4 acc2[4][4], acc1[2][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc2[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[t >> 1][t & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // bias grads: the data-grad epilogues add what they store (fp32, before the rounding to T) into per-lane partial column sums
  float b2r[4] = {0.f, 0.f, 0.f, 0.f};  // waves 0..3: co = 16*(wave&3) + 4*(lane>>4) + i, partial over the lane's pixels
  float b1r[4] = {0.f, 0.f, 0.f, 0.f};  // co = 16*(wave&1) + 4*(lane>>4) + i
  for (int i = tid; i < 12 * LY::LT; i += NTH) sdc2t[36 * LY::LT + i] = (T)0.f;
  for (int i = tid; i < 64 * LY::LP2; i += NTH) sdc2T[i] = (T)0.f;
  for (int i = tid; i < 32 * LY::LP1; i += NTH) sdc1T[i] = (T)0.f;
  for (int i = tid; i < LY::LT; i += NTH) sdc3[16 * LY::LT + i] = (T)0.f;
  constexpr int CH = 4 / LY::IMGP;  // image channels resident at a time

  constexpr int V = 16 / sizeof(T);  // image elements per 16-byte load
  // dc3 [16][64], c2 [36][64] and c1 [225][32] of a sample are requested under the PREVIOUS sample's dW1 (the CU's fetch path is
  // idle there: that phase runs out of LDS) and filed into LDS right after it — their readers (conv3', dW2, conv2''s mask) are
  // done by then. 256 + 576 float4 of dc3 / c2: every thread takes c2 chunk tid, waves 4..7 dc3 chunk tid - 256, wave 0 the
  // last 64 c2 chunks; c1: 1800 float4, chunk tid + 512 k.
  constexpr int N4 = 225 * 8, PT = (N4 + NTH - 1) / NTH;
  // A16: 16-byte chunks of 8 elements — c2 36 x 64 = 288 chunks (thread t < 288: chunk t), c1 225 x 32 = 900 chunks (t, t + 512)
  constexpr int N8 = 225 * 4, PT8 = (N8 + NTH - 1) / NTH;
  typedef typename Frag<T>::type chunk_t;  // 8 elements of T
  float4 hv0 = float4{0.f, 0.f, 0.f, 0.f}, hv1 = hv0, cv[A16 ? 1 : PT];
  chunk_t h16 = {}, c16[A16 ? PT8 : 1] = {};
  auto load_next = [&](int sm) {
    const float* g3 = a.dc3 + (int64_t)sm * 16 * 64;
    if constexpr (A16) {
      const chunk_t* g2 = reinterpret_cast<const chunk_t*>(reinterpret_cast<const T*>(a.c2) + (int64_t)sm * 36 * 64);
      const chunk_t* g1 = reinterpret_cast<const chunk_t*>(reinterpret_cast<const T*>(a.c1) + (int64_t)sm * 225 * 32);
      h16 = g2[tid < 288 ? tid : 0];
      if (tid >= 256) hv1 = *reinterpret_cast<const float4*>(g3 + (tid - 256) * 4);
#pragma unroll
      for (int k = 0; k < PT8; ++k) {
        const int i8 = tid + k * NTH;
        c16[k] = g1[i8 < N8 ? i8 : 0];
      }
    } else {
      const float* g2 = a.c2 + (int64_t)sm * 36 * 64;
      const float* g1 = a.c1 + (int64_t)sm * 225 * 32;
      hv0 = *reinterpret_cast<const float4*>(g2 + tid * 4);
      if (tid >= 256) hv1 = *reinterpret_cast<const float4*>(g3 + (tid - 256) * 4);
      else if (tid < 64) hv1 = *reinterpret_cast<const float4*>(g2 + (512 + tid) * 4);
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int i4 = tid + k * NTH;
        cv[k] = *reinterpret_cast<const float4*>(g1 + (i4 < N4 ? i4 : 0) * 4);
      }
    }
  };
  auto store_next = [&]() {
    if constexpr (A16) {
      if (tid < 288) {  // c2 chunk t = row t >> 3, columns 8 (t & 7) .. +7, widened to the fp32 mask rows
        float* d = sc2 + (tid >> 3) * LY::LF + (tid & 7) * 8;
        *reinterpret_cast<float4*>(d) = float4{(float)h16[0], (float)h16[1], (float)h16[2], (float)h16[3]};
        *reinterpret_cast<float4*>(d + 4) = float4{(float)h16[4], (float)h16[5], (float)h16[6], (float)h16[7]};
      }
      if (tid >= 256) st4(sdc3 + ((tid - 256) >> 4) * LY::LT + (tid & 15) * 4, hv1.x, hv1.y, hv1.z, hv1.w);
#pragma unroll
      for (int k = 0; k < PT8; ++k) {
        const int i8 = tid + k * NTH;
        if (i8 < N8) *reinterpret_cast<chunk_t*>(sc1 + (i8 >> 2) * LY::LC1 + (i8 & 3) * 8) = c16[k];  // already T: one 16-byte store
      }
    } else {
      *reinterpret_cast<float4*>(sc2 + (tid >> 4) * LY::LF + (tid & 15) * 4) = hv0;
      if (tid >= 256) st4(sdc3 + ((tid - 256) >> 4) * LY::LT + (tid & 15) * 4, hv1.x, hv1.y, hv1.z, hv1.w);  // dc3: MFMA operand only, kept in T
      else if (tid < 64) *reinterpret_cast<float4*>(sc2 + (32 + (tid >> 4)) * LY::LF + (tid & 15) * 4) = hv1;
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int i4 = tid + k * NTH;
        if (i4 < N4) st4(sc1 + (i4 >> 3) * LY::LC1 + (i4 & 7) * 4, cv[k].x, cv[k].y, cv[k].z, cv[k].w);
      }
    }
  };
  {
    // prologue: the first sample's dc3 / c2 / c1 and (bf16) the conv2' weights are all requested before anything is filed
    const bool first = (int)blockIdx.x < a.n;
    if (first) load_next(blockIdx.x);
    if constexpr (LY::B16) {
      // conv2' weights (4 parity classes x [ci 32][K 256]) stay in LDS for the block's whole life, stored as the MFMA fragments
      // the waves read: [class][ci tile][K step][lane][8] — a wave's fragment is one contiguous, conflict-free 1 KB
      float4 wv[4096 / NTH];
#pragma unroll
      for (int k = 0; k < 4096 / NTH; ++k) {  // 16-byte chunk c = (class*32 + ci)*32 + k/8
        const int c = tid + k * NTH, k8 = c & 31, n = (c >> 5) & 31, cl = c >> 10;
        wv[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const T*>(a.w2d[cl]) + n * 256 + k8 * 8);
      }
#pragma unroll
      for (int k = 0; k < 4096 / NTH; ++k) {
        const int c = tid + k * NTH, k8 = c & 31, n = (c >> 5) & 31, cl = c >> 10;
        const int dst = ((cl * 2 + (n >> 4)) * 8 + (k8 >> 2)) * 64 + (k8 & 3) * 16 + (n & 15);
        *reinterpret_cast<float4*>(sw2 + dst * 8) = wv[k];
      }
    }
    if (first) store_next();
  }
  for (int smp = blockIdx.x; smp < a.n; smp += gridDim.x) {
    __syncthreads();  // previous sample's readers are done
    CONV_STAMP(0);
    const int64_t slot = a.rowidx != nullptr ? a.rowidx[smp] : smp;
    const T* gimg = reinterpret_cast<const T*>(a.image) + slot * 16384;
    CONV_STAMP(1);
    {  // ---- dc2 = conv3' (gather form): rows = 36 input pixels (3 row tiles), K = 9 taps x 64 co, N = 64 ci
      //      wave pair (w, w + 4) owns ci tile w & 3 for all three row tiles and splits the 18 K steps in halves: the 72 KB of
      //      w3' enter the CU once per sample, nine 1 KB fragments per wave, all in flight together. (Splitting the ROW tiles
      //      between the pair streamed every fragment twice, and this phase is bound by exactly that stream.)
      const int nt = wave & 3;
      const T* W = reinterpret_cast<const T*>(a.w3d);
      int iy[3], ix[3];
      bool okr[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int px = m * 16 + fr;
        okr[m] = px < 36;
        iy[m] = px / 6; ix[m] = px - iy[m] * 6;
      }
      f32x4 acc[3][1];
#pragma unroll
      for (int m = 0; m < 3; ++m) acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto arow = [&](int m, int tap) -> const T* {
        const int ta = tap / 3, tb = tap - ta * 3;
        const int oy = iy[m] - ta, ox = ix[m] - tb;
        const bool ok = okr[m] && oy >= 0 && oy < 4 && ox >= 0 && ox < 4;
        return sdc3 + (ok ? oy * 4 + ox : 16) * LY::LT;
      };
      float4* part = reinterpret_cast<float4*>(simg);  // the image buffer is free until the DMA below
      if (wave >= 4) {
        gather_gemm<T, 3, 1, 9, 9, LY::B16 ? 9 : 3>(acc, W, 576, nt, lane, arow);
#pragma unroll
        for (int m = 0; m < 3; ++m) part[(nt * 3 + m) * 64 + lane] = float4{acc[m][0][0], acc[m][0][1], acc[m][0][2], acc[m][0][3]};
      } else {
        gather_gemm<T, 3, 1, 9, 0, LY::B16 ? 9 : 3>(acc, W, 576, nt, lane, arow);
      }
      __syncthreads();
      if (wave < 4) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int px = m * 16 + fr, c4 = nt * 16 + qr;
          if (px < 36) {
            const float4 hi = part[(nt * 3 + m) * 64 + lane];
            const float4 mk = *reinterpret_cast<const float4*>(sc2 + px * LY::LF + c4);
            const float d0 = mk.x > 0.f ? acc[m][0][0] + hi.x : 0.f, d1 = mk.y > 0.f ? acc[m][0][1] + hi.y : 0.f;
            const float d2 = mk.z > 0.f ? acc[m][0][2] + hi.z : 0.f, d3 = mk.w > 0.f ? acc[m][0][3] + hi.w : 0.f;
            st4(sdc2t + px * LY::LT + c4, d0, d1, d2, d3);  // [pixel][co]: the A operand of conv2' below
            if (a.t_dc2 != nullptr) st4(reinterpret_cast<T*>(a.t_dc2) + ((int64_t)smp * 36 + px) * 64 + c4, d0, d1, d2, d3);
            T* tp = sdc2T + c4 * LY::LP2 + px;              // [co][pixel]: dW2's column fragments
            tp[0] = (T)d0; tp[LY::LP2] = (T)d1; tp[2 * LY::LP2] = (T)d2; tp[3 * LY::LP2] = (T)d3;
            b2r[0] += d0; b2r[1] += d1; b2r[2] += d2; b2r[3] += d3;
          }
        }
      }
    }
    __syncthreads();
    CONV_STAMP(2);
    // the image (contiguous, unpadded in LDS) goes global -> LDS directly: four 16-byte DMA transfers per lane, no staging
    // registers (LDS destination = wave-uniform base + lane * 16). Issued here, it lands while dW2 and conv2' run out of LDS.
#pragma unroll
    for (int k = 0; k < CH * 4096 / V / NTH; ++k)
      __builtin_amdgcn_global_load_lds((const V4L_GLOBAL void*)(gimg + (int64_t)(tid + k * NTH) * V),
                                       (__attribute__((address_space(3))) void*)(simg + ((tid & ~63) + k * NTH) * V), 16, 0, 0);
    // ---- dW2 += dc2^T col(c1): contraction over the 36 output pixels (two K=32 steps; pixels 36..63 of dc2^T are zeros, so
    //      the c1 side only has to stay in bounds there)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      frag_t fy[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) fy[c] = *reinterpret_cast<const frag_t*>(sdc2T + (c * 16 + fr) * LY::LP2 + st * 32 + g * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kt = kt2 + t, tap = kt >> 1, ci = (kt & 1) * 16 + fr;
        const int ky = tap >> 2, kx = tap & 3;
        T w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int pos = st * 32 + g * 8 + j;
          const int pp = pos < 36 ? pos : 0, oy = pp / 6, ox = pp - oy * 6;
          w[j] = sc1[((2 * oy + ky) * 15 + 2 * ox + kx) * LY::LC1 + ci];
        }
        const frag_t fx = frag_of_t(w);
#pragma unroll
        for (int c = 0; c < 4; ++c) mma_k32(acc2[t][c], fx, fy[c]);
      }
    }
    CONV_STAMP(3);
    {  // ---- dc1 = conv2' (gather form): wave pair = stride-parity class (py,px) with <= 8x8 input pixels (4 row tiles); the
       //      pair splits the two 16-wide ci tiles (N = 32), so a class's 16 KB of weights are streamed once; K = 2x2 taps x 64 co
      const int cls = wave >> 1, py = cls >> 1, pxx = cls & 1, nh = wave & 1;
      const int nIy = (15 - py + 1) >> 1, nIx = (15 - pxx + 1) >> 1;
      int ry[4], rx[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { const int r = m * 16 + fr; ry[m] = r >> 3; rx[m] = r & 7; }
      f32x4 acc[4][1];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto arow = [&](int m, int tap) -> const T* {
        const int oy = ry[m] - (tap >> 1), ox = rx[m] - (tap & 1);
        const bool ok = ry[m] < nIy && rx[m] < nIx && oy >= 0 && oy < 6 && ox >= 0 && ox < 6;
        return sdc2t + (ok ? oy * 6 + ox : 36) * LY::LT;
      };
      if constexpr (LY::B16) {  // weights resident in LDS
        const T* wl = sw2 + (cls * 2 + nh) * 8 * 512 + lane * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const frag_t fw = *reinterpret_cast<const frag_t*>(wl + ks * 512);
#pragma unroll
          for (int m = 0; m < 4; ++m)
            mma_k32(acc[m][0], fw, *reinterpret_cast<const frag_t*>(arow(m, ks >> 1) + (ks & 1) * 32 + g * 8));
        }
      } else {
        gather_gemm<T, 4, 1, 8>(acc, reinterpret_cast<const T*>(a.w2d[cls]), 256, nh, lane, arow);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (ry[m] < nIy && rx[m] < nIx) {
          const int p = (py + 2 * ry[m]) * 15 + pxx + 2 * rx[m];
          const int c4 = nh * 16 + qr;
          const float4 mk = ld4(sc1 + p * LY::LC1 + c4);
          const float d0 = mk.x > 0.f ? acc[m][0][0] : 0.f, d1 = mk.y > 0.f ? acc[m][0][1] : 0.f;
          const float d2 = mk.z > 0.f ? acc[m][0][2] : 0.f, d3 = mk.w > 0.f ? acc[m][0][3] : 0.f;
          T* tp = sdc1T + c4 * LY::LP1 + p;  // dc1 exists only as dW1's operand: [co][pixel]
          tp[0] = (T)d0; tp[LY::LP1] = (T)d1; tp[2 * LY::LP1] = (T)d2; tp[3 * LY::LP1] = (T)d3;
          if (a.t_dc1 != nullptr) st4(reinterpret_cast<T*>(a.t_dc1) + ((int64_t)smp * 225 + p) * 32 + c4, d0, d1, d2, d3);
          b1r[0] += d0; b1r[1] += d1; b1r[2] += d2; b1r[3] += d3;
        }
      }
    }
    __syncthreads();
    CONV_STAMP(4);
    CONV_STAMP(5);
    const bool more = smp + (int)gridDim.x < a.n;
    if (more) load_next(smp + gridDim.x);  // next sample's dc3 / c2 / c1: in flight under dW1
    // ---- dW1 += dc1^T col(image): contraction over the 225 output pixels (eight K=32 steps; pixels 225..255 of dc1^T are zeros)
#pragma unroll 1
    for (int h = 0; h < LY::IMGP; ++h) {
      if (h > 0) {  // fp32 parity mode: the second channel pair replaces the first
        __syncthreads();
        for (int i = tid; i < CH * 4096 / V; i += NTH)
          *reinterpret_cast<float4*>(simg + i * V) = *reinterpret_cast<const float4*>(gimg + (int64_t)h * CH * 4096 + i * V);
        __syncthreads();
      }
      if (ch1 >= h * CH && ch1 < (h + 1) * CH) {  // this wave's input channel is resident
        const T* ich = simg + (ch1 - h * CH) * 4096;
#pragma unroll 1
        for (int st = 0; st < 8; ++st) {
          frag_t fy[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) fy[c] = *reinterpret_cast<const frag_t*>(sdc1T + (c * 16 + fr) * LY::LP1 + st * 32 + g * 8);
          int po2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int pos = st * 32 + g * 8 + j;
            const int pp = pos < 225 ? pos : 0, oy = pp / 15, ox = pp - oy * 15;
            po2[j] = (4 * oy) * 64 + 4 * ox;
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int ky = (th1 * 2 + tt) * 2 + (fr >> 3), kx = fr & 7;
            T w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = ich[po2[j] + ky * 64 + kx];
            const frag_t fx = frag_of_t(w);
#pragma unroll
            for (int c = 0; c < 2; ++c) mma_k32(acc1[c][tt], fx, fy[c]);
          }
        }
      }
    }
    if (more) store_next();
    CONV_STAMP(6);
  }
  // ---- the block's partial weight-grads -> its slab (wgrad_reduce_kernel sums the slabs in a fixed order)
  {
    float* o2 = a.slab2 + (int64_t)blockIdx.x * 64 * 512;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<float4*>(o2 + (c * 16 + fr) * 512 + (kt2 + t) * 16 + qr) =
            float4{acc2[t][c][0], acc2[t][c][1], acc2[t][c][2], acc2[t][c][3]};
    float* o1 = a.slab1 + (int64_t)blockIdx.x * 32 * 256;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        *reinterpret_cast<float4*>(o1 + (c * 16 + fr) * 256 + (ch1 * 4 + th1 * 2 + tt) * 16 + qr) =
            float4{acc1[c][tt][0], acc1[c][tt][1], acc1[c][tt][2], acc1[c][tt][3]};
    __syncthreads();
    float* red2 = reinterpret_cast<float*>(sdc1T);  // [co 64][16 lanes], then [co 32][4 classes x 16 lanes]
    float* red1 = red2 + 64 * 16;
    static_assert(LY::dc1_b >= (64 * 16 + 32 * 64) * 4, "bias reduction scratch");
    if (wave < 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red2[((wave & 3) * 16 + qr + i) * 16 + fr] = b2r[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red1[((wave & 1) * 16 + qr + i) * 64 + (wave >> 1) * 16 + fr] = b1r[i];
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red2[tid * 16 + k];
      a.bslab2[(int64_t)blockIdx.x * 64 + tid] = t;
    }
    if (tid < 32) {
      float t = 0.f;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) t += red1[tid * 64 + k];
      a.bslab1[(int64_t)blockIdx.x * 32 + tid] = t;
    }
  }
}

// dW3 += dc3^T col(c2) (see above): persistent blocks, sample = blockIdx.x, += gridDim.x; wave w owns k-tiles 9w..9w+8 x all
// four co-tiles, the contraction runs over the 16 output pixels (one zero-padded K=32 MFMA step per sample)
template <typename T, bool A16 = false>
__global__ __launch_bounds__(256) void bwd_conv3_wgrad_kernel(BwdConv a) {
  typedef typename Frag<T>::type frag_t;
  constexpr int LF = 64 + 4;
  __shared__ __attribute__((aligned(16))) float sdc3[16 * LF];
  __shared__ __attribute__((aligned(16))) float sc2[36 * LF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  const int kt3 = wave * 9;
  f32x4 acc3[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc3[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bias3 = 0.f;
  // LDS offsets of the im2col elements a lane reads, hoisted out of the sample loop (the tap of a k-tile depends on the wave:
  // the division by 3 and the index arithmetic were ~700 of the loop body's 1 050 VALU instructions, for 36 MFMAs):
  // element (tile t, slot j) = c2[(oy + ky) * 6 + ox + kx][ci] with pos = 8 (g & 1) + j = 4 oy + ox
  int x_base[9], x_pos[8];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kt = kt3 + t, tap = kt >> 2, ky = tap / 3, kx = tap - ky * 3;
    x_base[t] = (ky * 6 + kx) * LF + (kt & 3) * 16 + fr;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int pos = (g & 1) * 8 + j;
    x_pos[j] = ((pos >> 2) * 6 + (pos & 3)) * LF;
  }
  // a sample's rows (dc3 16 x 64, c2 36 x 64 fp32 = 13 KB) are requested one sample ahead into registers: the block used to
  // load -> barrier -> compute one sample after the other, a cold HBM round trip per sample in front of ~100 LDS reads per lane
  // (thread t: dc3 row t >> 4, columns 4 (t & 15); c2 float4s t, t + 256, t + 512 of the sample's 576)
  const int i2 = tid + 512 < 36 * 16 ? tid + 512 : 0;
  float4 r3 = {0.f, 0.f, 0.f, 0.f}, r2a = r3, r2b = r3, r2c = r3;
  // A16: c2 in T, 288 chunks of 8 elements: thread t takes chunk t, threads 0..31 also chunk 256 + t
  typedef typename Frag<T>::type chunk_t;
  chunk_t q2a = {}, q2b = {};
  auto fetch = [&](int sm) {
    const float* g3 = a.dc3 + (int64_t)sm * 16 * 64;
    r3 = *reinterpret_cast<const float4*>(g3 + tid * 4);
    if constexpr (A16) {
      const chunk_t* g2 = reinterpret_cast<const chunk_t*>(reinterpret_cast<const T*>(a.c2) + (int64_t)sm * 36 * 64);
      q2a = g2[tid];
      q2b = g2[tid < 32 ? 256 + tid : 0];
    } else {
      const float* g2 = a.c2 + (int64_t)sm * 36 * 64;
      r2a = *reinterpret_cast<const float4*>(g2 + tid * 4);
      r2b = *reinterpret_cast<const float4*>(g2 + (tid + 256) * 4);
      r2c = *reinterpret_cast<const float4*>(g2 + i2 * 4);
    }
  };
  auto file = [&]() {
    *reinterpret_cast<float4*>(sdc3 + (tid >> 4) * LF + (tid & 15) * 4) = r3;
    if constexpr (A16) {
      auto put = [&](int j, const chunk_t& q) {
        float* d = sc2 + (j >> 3) * LF + (j & 7) * 8;
        *reinterpret_cast<float4*>(d) = float4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
        *reinterpret_cast<float4*>(d + 4) = float4{(float)q[4], (float)q[5], (float)q[6], (float)q[7]};
      };
      put(tid, q2a);
      if (tid < 32) put(256 + tid, q2b);
    } else {
      *reinterpret_cast<float4*>(sc2 + (tid >> 4) * LF + (tid & 15) * 4) = r2a;
      *reinterpret_cast<float4*>(sc2 + ((tid + 256) >> 4) * LF + (tid & 15) * 4) = r2b;
      if (tid + 512 < 36 * 16) *reinterpret_cast<float4*>(sc2 + ((tid + 512) >> 4) * LF + (tid & 15) * 4) = r2c;
    }
  };
  if ((int)blockIdx.x < a.n) fetch(blockIdx.x);
  for (int smp = blockIdx.x; smp < a.n; smp += gridDim.x) {
    __syncthreads();  // the previous sample's readers are done
    file();
    const int nxt = smp + (int)gridDim.x;
    if (nxt < a.n) fetch(nxt);  // (block-uniform)
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 16; ++p) t += sdc3[p * LF + tid];
      bias3 += t;
    }
    float v[8];
    frag_t fy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = sdc3[((g & 1) * 8 + j) * LF + c * 16 + fr];
        v[j] = g < 2 ? x : 0.f;
      }
      fy[c] = frag_of<T>(v);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = sc2[x_base[t] + x_pos[j]];
        v[j] = g < 2 ? x : 0.f;
      }
      const frag_t fx = frag_of<T>(v);
#pragma unroll
      for (int c = 0; c < 4; ++c) mma_k32(acc3[t][c], fx, fy[c]);
    }
  }
  float* o3 = a.slab3 + (int64_t)blockIdx.x * 64 * 576;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<float4*>(o3 + (c * 16 + fr) * 576 + (kt3 + t) * 16 + qr) =
          float4{acc3[t][c][0], acc3[t][c][1], acc3[t][c][2], acc3[t][c][3]};
  if (tid < 64) a.bslab3[(int64_t)blockIdx.x * 64 + tid] = bias3;
}

}  // namespace v4l
