// Wave-per-sample ("WPS") transformer-layer kernels of the PPO update (gfx950).
//
// The block-cooperative layer kernels (infer.h infer_layer_kernel, bwd.h bwd_layer_kernel) walk a layer as ~12 phases
// separated by block barriers, every activation making an LDS round trip between two GEMMs, every block streaming the
// layer's weights out of L2 for 2-4 samples, and both save / re-read ~40 KB per sample and layer with 8-byte-per-lane stores
// scattered over 16 rows — they are store-ISSUE bound (MI355X_MICROARCH.md: a dwordx2 store costs 2.7x a dwordx4 per byte).
// Here ONE WAVE owns ONE SAMPLE (17 token rows = two 16-row MFMA tiles) and carries it through a whole
// nn.TransformerEncoderLayer (torchrl/networks/nets.py:948-955: post-norm, one head, ReLU FFN) IN REGISTERS:
//
//   * a 16x16x32 MFMA leaves lane (fr = lane & 15, g = lane >> 4) with D[4g + r][fr]. With the weight as the A operand
//     ("T layout": fr = token, registers = 4 consecutive features 16 nt + 4g + r of tile nt) the accumulators of two adjacent
//     column tiles ARE the next contraction's B fragment once its k order is permuted to
//         slot (g, j) <-> k = 32 ks + 16 (j >> 2) + 4 g + (j & 3)
//     — both operands only have to agree on the order, so the permutation is baked into the weight packs (PK_FRAGP /
//     PK_FRAGPT, elem.h) and no activation ever goes through LDS between GEMMs;
//   * with the activation as the A operand ("F layout": fr = feature, registers = tokens 4g + r) a GEMM output is a
//     fragment over TOKENS: V comes out of in_proj directly as the transposed operand of P V, and S = Q K^T leaves every
//     lane with one query's scores (softmax = 8 registers + two cross-group shuffles);
//   * T <-> F layout changes of a bf16 operand are ONE MFMA against a constant 0/1 selector fragment (exact: every output is
//     one input times 1.0) — that is how the attention backward gets dS^T, P^T, Q^T, K^T, dctx^T and how every weight-grad
//     operand is turned into a fragment over tokens;
//   * the layer's weights (98 KB in bf16) are DMA'd into LDS once per block in fragment order and read by all waves as
//     conflict-free 16-byte LDS reads (weight-stationary); there is no barrier inside a layer;
//   * the forward saves NOTHING but the layer inputs: the backward recomputes the layer in registers (bit-identical: same
//     code, same order) and then walks it backward; what it hands to the weight-grad kernel are fragment-order operand
//     blocks written with whole-wave 1 KB stores (tokens 0..15 of every sample dense; the 17th token's rows on the side),
//     consumed by wps_wgrad_kernel as MFMA fragments with no shuffling: K = 32 = tokens 0..15 of TWO samples.
//
// Rounding points are those of the block-cooperative kernels and of the oracle's bf16 flavour (operands rounded to T when
// they enter a contraction, fp32 accumulate, fp32 bias / residual / softmax / LayerNorm); only fp32 summation orders differ.
// compute = f32 (parity mode) runs the same code with fp32 fragments streamed from L2 (196 KB per layer do not fit LDS).
#pragma once
#include "bwd.h"

// Floating-point contraction inside this file: per source statement only (a * b + c written as one expression is an FMA, a
// product and a sum in different statements are not fused). HIP's default ("fast") lets the optimiser fuse across statements
// wherever it sees fit, and it decides differently in different template instantiations of the same function — round 4 added
// instantiations of wps_layer_bwd_kernel without the heads / proprio chains and their layer-0 input gradients came out a last
// bit apart from the others' (3.7e-9 of the tensor: the tapped-vs-untapped and forked-vs-serial bit-equality tests caught it).
// With "on" every instantiation performs the same fp32 operations.
#pragma clang fp contract(on)

namespace v4l {

constexpr int WPS_WPB = 4;  // waves = samples per block
#ifdef V4L_INFER_TIMING
#define WPS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_inf_stamps[i] = clock64(); } while (0)
#else
#define WPS_STAMP(i)
#endif
// Diagnostic build switch (tools/probe/build_variant.sh roll): the layer loops of the two stack kernels as run-time loops, i.e.
// ONE copy of the layer code per kernel instead of NL — tests whether the straight-line 100 KB of wps_layer_bwd_kernel against
// the 64 KB instruction cache a CU pair shares is what its launch time (and its two process-dependent modes) comes from.
// Diagnostic build switch (build_variant.sh eu1): tell the compiler that ONE wave per SIMD is the target occupancy of the two
// stack kernels (they need 377 / 512 registers anyway), so its scheduler stops trading latency hiding for register pressure.
#ifdef V4L_WPS_EU1
#define WPS_EU_ATTR __attribute__((amdgpu_waves_per_eu(1, 1)))
#else
#define WPS_EU_ATTR
#endif
// Timing-only probe build (build_variant.sh onetile; results are WRONG, never shipped): the layer functions walk ONE token tile
// instead of two — every MFMA, epilogue and softmax / LayerNorm pass of the padding tile (token 16 + 15 empty rows) is compiled
// out. What is left is the upper bound of what ANY scheme that shares the 17th tokens' tile between samples could gain (VERDICT
// r4 item 1a), before its own exchange costs.
#ifdef V4L_WPS_PROBE_ONE_TILE
#define WPS_NMT_DEF 1
#else
#define WPS_NMT_DEF 2
#endif
// Token tiles a layer function walks: 2 (17 rows) — or 1 in the native 16-token instantiation of the vision-only Transformer
// (VIS = 2, round 5: its 16 depth tokens ARE one MFMA row tile; no dummy row, no padding tile). Per-tile arrays keep two slots;
// the unused one is zero-initialised (dead stores where both tiles are walked).
#define WPS_Z = {}
#ifdef V4L_WPS_ROLL_LAYERS
#define WPS_LAYER_LOOP _Pragma("unroll 1")
#else
#define WPS_LAYER_LOOP _Pragma("unroll")
#endif
// per-layer fragment-order weight block [in_proj 192x64 | out_proj 64x64 | linear1 256x64 | linear2 64x256] (elements of T);
// the transposed block (PK_FRAGPT) has the same four sizes at the same offsets
constexpr int WPS_OFF_WO = 192 * 64, WPS_OFF_W1 = WPS_OFF_WO + 64 * 64, WPS_OFF_W2 = WPS_OFF_W1 + 256 * 64;
constexpr int WPS_LAYER_ELEMS = WPS_OFF_W2 + 64 * 256;  // 49152
// per-layer fp32 parameter block staged beside it
constexpr int WPS_P_BIN = 0, WPS_P_BO = 192, WPS_P_B1 = 256, WPS_P_B2 = 512, WPS_P_G1 = 576, WPS_P_BE1 = 640, WPS_P_G2 = 704,
              WPS_P_BE2 = 768, WPS_P_TOTAL = 832;
// weight-grad operand block of one (sample, layer): 32 "pairs" = 64 feature tiles of 16 = 1024 features x tokens 0..15, pair p =
// [64 lanes][tile 2p: tokens 4g..4g+3 | tile 2p+1: tokens 4g..4g+3] at feature fr. Feature tiles (x side, then dY side):
constexpr int WPS_T_XIN = 0, WPS_T_CTX = 4, WPS_T_X1 = 8, WPS_T_F = 12, WPS_T_DZ2 = 28, WPS_T_DF = 32, WPS_T_DZ1 = 48, WPS_T_DQKV = 52;
constexpr int WPS_WG_ELEMS = 32 * 64 * 8;  // per (sample, layer), elements of T
// sample-to-sample stride of the operand blocks: NOT the power of two the block size is — at any moment the 1024 waves of the
// backward write (and the weight-grad kernel's waves read) the same 1 KB piece of 1024 different samples' blocks, and with a
// 32 KB stride (bf16) those addresses agree in every bit the memory channels are selected by below the page level:
// wps_wgrad_kernel 23.7 -> 21.3 us with 512 bytes of padding per block (4 processes each, round 4)
constexpr int WPS_WG_STRIDE = WPS_WG_ELEMS + 256;
constexpr int WPS_TK_ELEMS = 32 * 4 * 8;   // the 17th token's 1024 features, in fragment (k-permuted) order [pair][g][8]
constexpr int WPS_SPLIT = 32;              // samples per weight-grad partial (one K=32 step of 17th tokens)
constexpr int WPS_ROLES = 12;              // 4x4-tile weight-grad jobs per layer

template <typename T> struct WpsFwdLds {
  static constexpr bool LDSW = sizeof(T) == 2;  // weights resident in LDS (bf16); fp32 fragments stream from L2
  static constexpr size_t w_b = LDSW ? (size_t)WPS_LAYER_ELEMS * sizeof(T) : 0;
  static constexpr int LDP = 128 + 4, LDF = 256 + InfLd<T>::PAD;
  static constexpr size_t head_b = (size_t)16 * LDP * 4 + (size_t)2 * 16 * LDF * sizeof(T) + 16 * 16 * 4;
  static constexpr size_t main_b = ((w_b > head_b ? w_b : head_b) + 15) / 16 * 16;  // heads alias the weight region
  static constexpr size_t bytes = main_b + (size_t)WPS_P_TOTAL * 4;
};
template <typename T> struct WpsBwdLds {
  static constexpr bool LDSW = sizeof(T) == 2;
  static constexpr size_t w_b = LDSW ? (size_t)WPS_LAYER_ELEMS * sizeof(T) : 0;
  static constexpr int LDX = 64 + 4, LDP = 128 + 4, LDF = 256 + InfLd<T>::PAD;
  // heads: dout rows | dh1 | dh0 | dpool ; tail: token-0 rows | dh
  static constexpr size_t head_b = (size_t)16 * LDX * 4 + (size_t)2 * 16 * LDF * sizeof(T) + (size_t)16 * LDP * 4;
  static constexpr size_t main_b = ((w_b > head_b ? w_b : head_b) + 15) / 16 * 16;
  static constexpr size_t red_b = (size_t)WPS_WPB * 4 * TD * 4;  // per wave: dgamma2 | dbeta2 | dgamma1 | dbeta1
  static constexpr size_t bytes = main_b + (size_t)WPS_P_TOTAL * 4 + red_b;
};

// ---- fragment helpers ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ typename Frag<T>::type wps_frag(const float4& a, const float4& b) {
  typename Frag<T>::type f;
  if constexpr (sizeof(T) == 2) {
    f[0] = (T)a.x; f[1] = (T)a.y; f[2] = (T)a.z; f[3] = (T)a.w;
    f[4] = (T)b.x; f[5] = (T)b.y; f[6] = (T)b.z; f[7] = (T)b.w;
  } else {
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  }
  return f;
}
__device__ __forceinline__ float4 f4(const f32x4& a) { return float4{a[0], a[1], a[2], a[3]}; }
__device__ __forceinline__ float4 f4add(const f32x4& a, const float4& b) { return float4{a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w}; }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// fragment `idx` (= tile * KS + ks) of a fragment-order weight matrix: 64 lanes x 8 elements, contiguous
template <typename T, bool LDSW>
__device__ __forceinline__ typename Frag<T>::type wps_w(const T* base, int idx, int lane) {
  typedef typename Frag<T>::type frag_t;
  const T* p = base + ((int64_t)idx * 64 + lane) * 8;
  if constexpr (LDSW) {
    typedef __attribute__((address_space(3))) const frag_t lds_frag;
    return *reinterpret_cast<lds_frag*>((__attribute__((address_space(3))) const T*)p);
  } else {
    return *reinterpret_cast<const frag_t*>(p);
  }
}
// T-layout GEMM step: acc[mt] (features 16 tile + 4g + r of token 16 mt + fr) += W[tile] . x
template <typename T, bool LDSW, int KS, int NMT = WPS_NMT_DEF>
__device__ __forceinline__ void wps_gemm_t(f32x4 (&acc)[2], const T* W, int tile, const typename Frag<T>::type (&xa)[2][KS], int lane) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const typename Frag<T>::type fw = wps_w<T, LDSW>(W, tile * KS + ks, lane);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) mma_k32(acc[mt], fw, xa[mt][ks]);
  }
}
// F-layout GEMM step: acc[mt] (tokens 16 mt + 4g + r of feature 16 tile + fr) += x . W[tile]
template <typename T, bool LDSW, int KS, int NMT = WPS_NMT_DEF>
__device__ __forceinline__ void wps_gemm_f(f32x4 (&acc)[2], const T* W, int tile, const typename Frag<T>::type (&xa)[2][KS], int lane) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const typename Frag<T>::type fw = wps_w<T, LDSW>(W, tile * KS + ks, lane);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) mma_k32(acc[mt], xa[mt][ks], fw);
  }
}
// VIS: 0 = 17 tokens; 1 = the vision-only Transformer on the 17-row machinery (key 0 = the dummy row, masked); 2 = its native
// 16-token instantiation (keys 0..15 = the depth tokens)
template <int VIS> __device__ __forceinline__ bool wps_key_ok(int key) {
  return VIS == 2 ? key < 16 : (key < NTOK && (VIS == 0 || key > 0));
}
__device__ __forceinline__ float xsum(float v) {  // over the four lane groups that share a token (T layout row)
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float xmax(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float rowsum16(float v) {  // over the 16 lanes of a DPP row (= the 16 tokens of a T-layout tile)
  v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);
  return v;
}
__device__ __forceinline__ float rowmax16(float v) {  // max over the 16 lanes of a DPP row
  v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
  return v;
}
__device__ __forceinline__ int rowmin16(int v) {  // (the DPP move carries the bits unchanged)
  v = min(v, __float_as_int(dpp_mov<0x128>(__int_as_float(v)))); v = min(v, __float_as_int(dpp_mov<0x124>(__int_as_float(v))));
  v = min(v, __float_as_int(dpp_mov<0x122>(__int_as_float(v)))); v = min(v, __float_as_int(dpp_mov<0x121>(__int_as_float(v))));
  return v;
}
// The 0/1 selector fragments: as the B operand, E_h picks index 16 h + fr out of a k-step's 32 (permuted) contraction
// indices: mma(A = X fragment, B = E_h) = X^T restricted to those 16 indices, exactly.
template <typename T> __device__ __forceinline__ typename Frag<T>::type wps_sel(int h, int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const bool mine = g == (fr >> 2);
  const int j1 = 4 * h + (fr & 3);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (mine && j == j1) ? 1.f : 0.f;
  return wps_frag<T>(float4{v[0], v[1], v[2], v[3]}, float4{v[4], v[5], v[6], v[7]});
}
// transposed fragment of the two halves (index tiles 2ks, 2ks+1 -> h = 0, 1) of fragments a0 (rows 0..15) / a1 (rows 16..31):
// lane (fr = index 16 h + fr of the k-step, g) <- rows 4g..4g+3 of a0, rows 16+4g.. of a1
template <typename T>
__device__ __forceinline__ typename Frag<T>::type wps_tr(const typename Frag<T>::type& a0, const typename Frag<T>::type& a1,
                                                         const typename Frag<T>::type& E) {
  f32x4 t0 = zero4(), t1 = zero4();
  mma_k32(t0, a0, E);
  mma_k32(t1, a1, E);
  return wps_frag<T>(f4(t0), f4(t1));
}
// One weight-grad operand pair: fragment f0 (tokens 0..15 x 32 features of k-step `pair`) -> F layout -> one whole-wave store;
// f1 (tokens 16..31: only token 16, lane fr = 0, is real) -> the 17th-token block as it is.
struct WpsOut { void *wg, *tk; bool live; };
template <typename T> __device__ __forceinline__ WpsOut wps_out(T* wg, T* tk, bool live) { return WpsOut{wg, tk, live}; }
template <typename T, int NMT = WPS_NMT_DEF>
__device__ __forceinline__ void wps_store_opnd(const WpsOut& o, int pair, const typename Frag<T>::type& f0, const typename Frag<T>::type& f1,
                                               const typename Frag<T>::type& E0, const typename Frag<T>::type& E1, int lane) {
  typedef typename Frag<T>::type frag_t;
  f32x4 d0 = zero4(), d1 = zero4();
  mma_k32(d0, f0, E0);  // d0[r] = f0[token 4g + r][feature fr of tile 2 pair]
  mma_k32(d1, f0, E1);
  if (o.live) {
    *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(o.wg) + ((int64_t)pair * 64 + lane) * 8) = wps_frag<T>(f4(d0), f4(d1));
    if (NMT == 2 && (lane & 15) == 0) *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(o.tk) + ((int64_t)pair * 4 + (lane >> 4)) * 8) = f1;
  }
}

// LayerNorm of one sample's rows in T layout (z[mt][nt]: token 16 mt + fr, features 16 nt + 4g + r): mean, then the variance of
// the centred values (the two-pass form of ln_rows / at::native::layer_norm), eps 1e-5. -> xhat in z, rstd per mt.
template <int NMT = WPS_NMT_DEF>
__device__ __forceinline__ void wps_ln(float4 (&z)[2][4], float (&rs)[2]) {
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s += (z[mt][nt].x + z[mt][nt].y) + (z[mt][nt].z + z[mt][nt].w);
    const float mean = xsum(s) * (1.f / TD);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      z[mt][nt].x -= mean; z[mt][nt].y -= mean; z[mt][nt].z -= mean; z[mt][nt].w -= mean;
      q += (z[mt][nt].x * z[mt][nt].x + z[mt][nt].y * z[mt][nt].y) + (z[mt][nt].z * z[mt][nt].z + z[mt][nt].w * z[mt][nt].w);
    }
    rs[mt] = 1.f / sqrtf(xsum(q) * (1.f / TD) + 1e-5f);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { z[mt][nt].x *= rs[mt]; z[mt][nt].y *= rs[mt]; z[mt][nt].z *= rs[mt]; z[mt][nt].w *= rs[mt]; }
  }
}

// What a recomputed forward leaves in registers for the backward pass of the same layer
template <typename T> struct WpsKeep {
  typedef typename Frag<T>::type frag_t;
  frag_t ka_f[4], qa_f[4];          // K^T, Q^T: rows = features (tile dt), slots = tokens   (dQ = dS K, dK = dS^T Q)
  frag_t va[2][2] WPS_Z;  // V: rows = tokens, slots = features                    (dP = dctx V^T)
  float p[2][2][4] WPS_Z;  // softmax probabilities, fp32: [query tile][key tile][r]
  float4 xh1[2][4] WPS_Z, xh2[2][4] WPS_Z;      // normalised rows of the two LayerNorms
  float rs1[2] WPS_Z, rs2[2] WPS_Z;
  unsigned long long fm[2] WPS_Z;  // ReLU mask of the FFN activation: bit (hidden tile * 4 + r) of token row mt
};

// One nn.TransformerEncoderLayer forward of ONE sample by ONE wave. xr: the layer input rows (T layout, rows >= 17 zero).
// W: the layer's fragment-order weight block; prm: its fp32 parameter block (LDS).
//   * `w` non-null pointers: row-major saves / taps (rows row0 + token, only where ok[mt]): the block-cooperative backward's inputs
//     and the test taps. Production passes only xout (the next layer's input).
//   * wg / tk non-null: the x-side weight-grad operands (layer input, ctx, x1, f) go out in fragment order (wps_store_opnd).
//   * KEEP: fill `kp` for wps_layer_bwd.
//   * VIS: the vision-only Transformer (nets.py:784-906: 16 depth tokens, no proprio token) on the 17-row machinery: its tokens
//     sit in rows 1..16, row 0 is a dummy (zero input rows) that no real token attends to — key 0 is masked out of every
//     softmax. Whatever row 0 computes stays in row 0, the head ignores it (kernel epilogue), and its gradient rows are
//     exactly zero in the backward (P[:, 0] = 0, zero output gradient), so the weight-grads never see it.
template <typename T, bool LDSW, bool KEEP, bool TAPS, int VIS = 0>
__device__ __forceinline__ void wps_layer_fwd(const InfLayer& w, const T* W, const float* prm, const float4 (&xr)[2][4], int lane,
                                              const bool (&ok)[2], int64_t row0, int64_t smp, float4 (&xo)[2][4], const WpsOut* wo,
                                              const typename Frag<T>::type& E0, const typename Frag<T>::type& E1, WpsKeep<T>* kp, int sb = 0) {
  typedef typename Frag<T>::type frag_t;
  constexpr int NMT = VIS == 2 ? 1 : WPS_NMT_DEF;
  constexpr int ROFF = VIS == 2 ? 1 : 0;  // native 16: tile row fr = token fr = memory row 1 + fr of the sample's 17-row slot
  static_assert(!(TAPS && VIS == 2), "the test taps run on the 17-row instantiations");
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  const bool live = ok[0];
  (void)sb;
  // ---- layer input as fragments
  frag_t xa[2][2] WPS_Z;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xa[mt][ks] = wps_frag<T>(xr[mt][2 * ks], xr[mt][2 * ks + 1]);
  if (TAPS && w.s_xin != nullptr) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
      if (ok[mt])
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          st4(reinterpret_cast<T*>(w.s_xin) + (row0 + mt * 16 + fr) * TD + nt * 16 + qr, xr[mt][nt].x, xr[mt][nt].y, xr[mt][nt].z, xr[mt][nt].w);
  }
  if constexpr (KEEP) {  // (the recompute inside the backward kernel is the pass that hands the x-side operands over)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wps_store_opnd<T, NMT>(*wo, WPS_T_XIN / 2 + ks, xa[0][ks], xa[1][ks], E0, E1, lane);
  }
  // ---- in_proj: q | k in T layout (operands of S = Q K^T over the feature index), v in F layout (operand of P V over keys)
  frag_t qa[2][2] WPS_Z, ka[2][2] WPS_Z, vt[4];
#pragma unroll
  for (int part = 0; part < 2; ++part) {  // 0: q, 1: k
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float4 two[2][2] WPS_Z;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tile = part * 4 + 2 * ks + h;
        f32x4 acc[2] = {zero4(), zero4()};
        wps_gemm_t<T, LDSW, 2, NMT>(acc, W, tile, xa, lane);
        const float4 bb = *reinterpret_cast<const float4*>(prm + WPS_P_BIN + tile * 16 + qr);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          two[mt][h] = f4add(acc[mt], bb);
          if (TAPS && w.s_qkv != nullptr && ok[mt])
            st4(w.s_qkv + (row0 + mt * 16 + fr) * 192 + tile * 16 + qr, two[mt][h].x, two[mt][h].y, two[mt][h].z, two[mt][h].w);
        }
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if (part == 0) qa[mt][ks] = wps_frag<T>(two[mt][0], two[mt][1]);
        else ka[mt][ks] = wps_frag<T>(two[mt][0], two[mt][1]);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 acc[2] = {zero4(), zero4()};
    wps_gemm_f<T, LDSW, 2, NMT>(acc, W, 8 + dt, xa, lane);
    const float bv = prm[WPS_P_BIN + 128 + dt * 16 + fr];
    const float4 v0 = {acc[0][0] + bv, acc[0][1] + bv, acc[0][2] + bv, acc[0][3] + bv};   // keys 4g + r
    const float4 v1 = {acc[1][0] + bv, acc[1][1] + bv, acc[1][2] + bv, acc[1][3] + bv};   // keys 16 + 4g + r (only key 16 is real)
    vt[dt] = wps_frag<T>(v0, v1);
    if (TAPS && w.s_qkv != nullptr) {  // v rows for the block-cooperative backward (fp32 [token][192]): one feature of 8 tokens per lane
      const float va[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tok = 16 * (j >> 2) + qr + (j & 3);
        if (tok < NTOK && live) w.s_qkv[(row0 + tok) * 192 + 128 + dt * 16 + fr] = va[j];
      }
    }
  }
  if constexpr (KEEP) {
    // the transposed views the attention backward contracts over tokens / features with: exact selector products
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        kp->qa_f[2 * ks + h] = wps_tr<T>(qa[0][ks], qa[1][ks], h ? E1 : E0);
        kp->ka_f[2 * ks + h] = wps_tr<T>(ka[0][ks], ka[1][ks], h ? E1 : E0);
      }
    float4 vv[2][4] WPS_Z;  // V in T layout: token tile mt, feature tile dt
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 t = zero4();
        mma_k32(t, vt[dt], mt ? E1 : E0);  // t[r] = v[token 16 mt + fr][feature 16 dt + 4g + r]
        vv[mt][dt] = f4(t);
      }
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kp->va[mt][ks] = wps_frag<T>(vv[mt][2 * ks], vv[mt][2 * ks + 1]);
  }
  // ---- attention: S^T tiles (lane = query, registers = keys), softmax in registers, P V
  float4 c[2][4] WPS_Z;
#pragma unroll
  for (int qt = 0; qt < NMT; ++qt) {
    f32x4 s[2] = {zero4(), zero4()};
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) mma_k32(s[kt], ka[kt][ks], qa[qt][ks]);  // s[kt][r] = q[16 qt + fr] . k[16 kt + 4g + r]
    float pv[2][4] WPS_Z;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[kt][r] = wps_key_ok<VIS>(kt * 16 + qr + r) ? s[kt][r] * 0.125f : -INFINITY;
        mx = fmaxf(mx, pv[kt][r]);
      }
    mx = xmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[kt][r] = wps_key_ok<VIS>(kt * 16 + qr + r) ? expf(pv[kt][r] - mx) : 0.f;
        sum += pv[kt][r];
      }
    const float inv = 1.f / xsum(sum);
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[kt][r] *= inv;
        if constexpr (KEEP) kp->p[qt][kt][r] = pv[kt][r];
        const int key = kt * 16 + qr + r;
        if (TAPS && w.s_P != nullptr && ok[qt] && key < NTOK) w.s_P[smp * (NTOK * NTOK) + (qt * 16 + fr) * NTOK + key] = pv[kt][r];
      }
    const frag_t pa = wps_frag<T>(float4{pv[0][0], pv[0][1], pv[0][2], pv[0][3]}, float4{pv[1][0], pv[1][1], pv[1][2], pv[1][3]});
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 a = zero4();
      mma_k32(a, vt[dt], pa);  // a[r] = sum_key P[16 qt + fr][key] v[key][16 dt + 4g + r]
      c[qt][dt] = f4(a);
      if (TAPS && w.s_ctx != nullptr && ok[qt])
        st4(reinterpret_cast<T*>(w.s_ctx) + (row0 + qt * 16 + fr) * TD + dt * 16 + qr, a[0], a[1], a[2], a[3]);
    }
  }
  // ---- out_proj + residual, norm1
  float4 z[2][4] WPS_Z;
  {
    frag_t ca[2][2] WPS_Z;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ca[mt][ks] = wps_frag<T>(c[mt][2 * ks], c[mt][2 * ks + 1]);
    if constexpr (KEEP) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wps_store_opnd<T, NMT>(*wo, WPS_T_CTX / 2 + ks, ca[0][ks], ca[1][ks], E0, E1, lane);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc[2] = {zero4(), zero4()};
      wps_gemm_t<T, LDSW, 2, NMT>(acc, W + WPS_OFF_WO, nt, ca, lane);
      const float4 bb = *reinterpret_cast<const float4*>(prm + WPS_P_BO + nt * 16 + qr);
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
        z[mt][nt] = float4{xr[mt][nt].x + acc[mt][0] + bb.x, xr[mt][nt].y + acc[mt][1] + bb.y, xr[mt][nt].z + acc[mt][2] + bb.z,
                           xr[mt][nt].w + acc[mt][3] + bb.w};
    }
  }
  float rs1[2] WPS_Z;
  wps_ln<NMT>(z, rs1);
  float4 x1[2][4] WPS_Z;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 gg = *reinterpret_cast<const float4*>(prm + WPS_P_G1 + nt * 16 + qr);
    const float4 be = *reinterpret_cast<const float4*>(prm + WPS_P_BE1 + nt * 16 + qr);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      const float4 xh = z[mt][nt];
      if constexpr (KEEP) kp->xh1[mt][nt] = xh;
      x1[mt][nt] = float4{fmaf(xh.x, gg.x, be.x), fmaf(xh.y, gg.y, be.y), fmaf(xh.z, gg.z, be.z), fmaf(xh.w, gg.w, be.w)};
      if (ok[mt]) {
        const int64_t o = (row0 + ROFF + mt * 16 + fr) * TD + nt * 16 + qr;
        if (TAPS && w.s_xh1 != nullptr) *reinterpret_cast<float4*>(w.s_xh1 + o) = xh;
        if (TAPS && w.s_x1 != nullptr) st4(reinterpret_cast<T*>(w.s_x1) + o, x1[mt][nt].x, x1[mt][nt].y, x1[mt][nt].z, x1[mt][nt].w);
      }
    }
  }
  if constexpr (KEEP) { kp->rs1[0] = rs1[0]; kp->rs1[1] = rs1[1]; }
  if (TAPS && w.s_rs1 != nullptr && g == 0) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
      if (ok[mt]) w.s_rs1[row0 + mt * 16 + fr] = rs1[mt];
  }
  // ---- FFN, 32 hidden features at a time: h = relu(W1 x1 + b1) is the B fragment of the linear2 step over those features
  {
    frag_t x1a[2][2] WPS_Z;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) x1a[mt][ks] = wps_frag<T>(x1[mt][2 * ks], x1[mt][2 * ks + 1]);
    if constexpr (KEEP) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wps_store_opnd<T, NMT>(*wo, WPS_T_X1 / 2 + ks, x1a[0][ks], x1a[1][ks], E0, E1, lane);
    }
    f32x4 z2[2][4] WPS_Z;
    unsigned long long fm[2] = {0ull, 0ull};
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) z2[mt][nt] = zero4();
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      float4 hh[2][2] WPS_Z;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tile = 2 * ch + h;
        f32x4 acc[2] = {zero4(), zero4()};
        wps_gemm_t<T, LDSW, 2, NMT>(acc, W + WPS_OFF_W1, tile, x1a, lane);
        const float4 bb = *reinterpret_cast<const float4*>(prm + WPS_P_B1 + tile * 16 + qr);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          hh[mt][h] = float4{fmaxf(acc[mt][0] + bb.x, 0.f), fmaxf(acc[mt][1] + bb.y, 0.f), fmaxf(acc[mt][2] + bb.z, 0.f),
                             fmaxf(acc[mt][3] + bb.w, 0.f)};
          if constexpr (KEEP) {
            // the mask the backward applies is "the ROUNDED activation is positive" (what the saved T-typed f of the
            // block-cooperative kernels encodes): a positive fp32 value that rounds to 0 in T does not exist for bf16 / fp32
            const unsigned long long b4 = (hh[mt][h].x > 0.f ? 1ull : 0ull) | (hh[mt][h].y > 0.f ? 2ull : 0ull) |
                                          (hh[mt][h].z > 0.f ? 4ull : 0ull) | (hh[mt][h].w > 0.f ? 8ull : 0ull);
            fm[mt] |= b4 << (tile * 4);
          }
          if (TAPS && w.s_f != nullptr && ok[mt])
            st4(reinterpret_cast<T*>(w.s_f) + (row0 + mt * 16 + fr) * 256 + tile * 16 + qr, hh[mt][h].x, hh[mt][h].y, hh[mt][h].z, hh[mt][h].w);
        }
      }
      frag_t fa[2] WPS_Z;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) fa[mt] = wps_frag<T>(hh[mt][0], hh[mt][1]);
      if constexpr (KEEP) wps_store_opnd<T, NMT>(*wo, WPS_T_F / 2 + ch, fa[0], fa[1], E0, E1, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const frag_t fw = wps_w<T, LDSW>(W + WPS_OFF_W2, nt * 8 + ch, lane);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) mma_k32(z2[mt][nt], fw, fa[mt]);
      }
    }
    if constexpr (KEEP) { kp->fm[0] = fm[0]; kp->fm[1] = fm[1]; }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float4 bb = *reinterpret_cast<const float4*>(prm + WPS_P_B2 + nt * 16 + qr);
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
        z[mt][nt] = float4{x1[mt][nt].x + z2[mt][nt][0] + bb.x, x1[mt][nt].y + z2[mt][nt][1] + bb.y, x1[mt][nt].z + z2[mt][nt][2] + bb.z,
                           x1[mt][nt].w + z2[mt][nt][3] + bb.w};
    }
  }
  float rs2[2] WPS_Z;
  wps_ln<NMT>(z, rs2);
  if constexpr (KEEP) { kp->rs2[0] = rs2[0]; kp->rs2[1] = rs2[1]; }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 gg = *reinterpret_cast<const float4*>(prm + WPS_P_G2 + nt * 16 + qr);
    const float4 be = *reinterpret_cast<const float4*>(prm + WPS_P_BE2 + nt * 16 + qr);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      const float4 xh = z[mt][nt];
      if constexpr (KEEP) kp->xh2[mt][nt] = xh;
      // rows past the sample's 17th token stay exactly zero: they are the next layer's padding rows
      const bool live_row = mt == 0 || fr == 0;
      xo[mt][nt] = live_row ? float4{fmaf(xh.x, gg.x, be.x), fmaf(xh.y, gg.y, be.y), fmaf(xh.z, gg.z, be.z), fmaf(xh.w, gg.w, be.w)}
                            : float4{0.f, 0.f, 0.f, 0.f};
      if (ok[mt]) {
        const int64_t o = (row0 + ROFF + mt * 16 + fr) * TD + nt * 16 + qr;
        if (TAPS && w.s_xh2 != nullptr) *reinterpret_cast<float4*>(w.s_xh2 + o) = xh;
        if (w.xout != nullptr) *reinterpret_cast<float4*>(w.xout + o) = xo[mt][nt];
      }
    }
  }
  if (TAPS && w.s_rs2 != nullptr && g == 0) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
      if (ok[mt]) w.s_rs2[row0 + mt * 16 + fr] = rs2[mt];
  }
}

// Stage one layer's weight block (bf16: global -> LDS by DMA, 16 bytes per lane and transfer) and, optionally, its parameter
// block (one 16-byte piece per thread, no loop). Every thread of the block calls it; followed by s_waitcnt + a block barrier.
struct WpsPrm { const float *bin, *bo, *b1, *b2, *g1, *be1, *g2, *be2; };
template <typename T, bool LDSW>
__device__ __forceinline__ void wps_stage(const void* wsrc, const WpsPrm* pp, T* wl, float* prm, int tid) {
  if constexpr (LDSW) {
    const T* src = reinterpret_cast<const T*>(wsrc);  // the four matrices' packs are adjacent
    constexpr int V = 16 / sizeof(T);
#pragma unroll
    for (int k = 0; k < WPS_LAYER_ELEMS / V / 256; ++k)
      __builtin_amdgcn_global_load_lds((const V4L_GLOBAL void*)(src + (int64_t)(tid + k * 256) * V),
                                       (__attribute__((address_space(3))) void*)(wl + ((tid & ~63) + k * 256) * V), 16, 0, 0);
  }
  if (pp != nullptr && tid < WPS_P_TOTAL / 4) {
    const int i = tid * 4;  // every segment boundary is a multiple of 4
    const float* p = i < WPS_P_BO ? pp->bin + i
                   : i < WPS_P_B1 ? pp->bo + (i - WPS_P_BO)
                   : i < WPS_P_B2 ? pp->b1 + (i - WPS_P_B1)
                   : i < WPS_P_G1 ? pp->b2 + (i - WPS_P_B2)
                   : i < WPS_P_BE1 ? pp->g1 + (i - WPS_P_G1)
                   : i < WPS_P_G2 ? pp->be1 + (i - WPS_P_BE1)
                   : i < WPS_P_BE2 ? pp->g2 + (i - WPS_P_G2)
                                   : pp->be2 + (i - WPS_P_BE2);
    *reinterpret_cast<float4*>(prm + i) = *reinterpret_cast<const float4*>(p);
  }
}
__device__ __forceinline__ WpsPrm wps_prm_of(const InfLayer& w) {
  return WpsPrm{w.bin, w.bo, w.b1, w.b2, w.g1, w.be1, w.g2, w.be2};
}

// The layer input rows of one sample, row-major fp32 [17][64] -> T layout registers (rows >= 17 and dead samples: zeros)
// ZERO0: row 0 reads as zeros whatever the buffer holds (the vision-only Transformer's dummy row, see wps_layer_fwd)
// VIS = 2 (native 16 tokens): tile 0 row fr <- memory row 1 + fr, tile 1 zero
template <int VIS = 0>
__device__ __forceinline__ void wps_load_rows(const float* __restrict__ xg, int lane, const bool (&ok)[2], float4 (&xr)[2][4]) {
  const int fr = lane & 15, qr = (lane >> 4) * 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (VIS == 2 && mt == 1) { xr[mt][nt] = float4{0.f, 0.f, 0.f, 0.f}; continue; }
      const int r = VIS == 2 ? 1 + fr : mt * 16 + fr;
      const float4 v = *reinterpret_cast<const float4*>(xg + (ok[mt] ? r : 0) * TD + nt * 16 + qr);
      const bool keep = ok[mt] && !(VIS == 1 && mt == 0 && fr == 0);
      xr[mt][nt] = keep ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
}

// Training forward of the transformer stack (+ pooled heads): NL layers for WPS_WPB samples per block, one wave each.
// stk.l[l].n[0].win points at the layer's fragment-order weight block (PK_FRAGP packs, adjacent); the head packs are the
// row-major ones of the block-cooperative kernel (the heads run cooperatively: 4 samples = one MFMA row tile).
template <typename T, bool HEAD, int NL, bool TAPS, int VIS = 0>
__global__ __launch_bounds__(256) WPS_EU_ATTR void wps_layer_fwd_kernel(InfLayerStack stk, InfHeadPair hd, int n) {
  typedef WpsFwdLds<T> LY;
  typedef typename Frag<T>::type frag_t;
  constexpr bool LDSW = LY::LDSW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* wl = reinterpret_cast<T*>(smem);
  float* prm = reinterpret_cast<float*>(smem + LY::main_b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  const int s0 = blockIdx.x * WPS_WPB;
  const int smp = s0 + wave;
  const bool live = smp < n;
  const int64_t srow = live ? smp : 0;
  const int64_t row0 = srow * NTOK;
  const bool ok[2] = {live, live && fr == 0 && VIS != 2};
  WPS_STAMP(0);
  {  // layer 0's weights start their trip first
    const WpsPrm pp = wps_prm_of(stk.l[0].n[0]);
    wps_stage<T, LDSW>(stk.l[0].n[0].win, &pp, wl, prm, tid);
  }
  float4 xr[2][4];
  wps_load_rows<VIS>(stk.l[0].n[0].xin + row0 * TD, lane, ok, xr);
  const frag_t E0 = wps_sel<T>(0, lane), E1 = wps_sel<T>(1, lane);
  WPS_LAYER_LOOP
  for (int l = 0; l < NL; ++l) {
    const InfLayer& w = stk.l[l].n[0];
    if (l > 0) {
      __syncthreads();  // every wave is done with the previous layer's weights
      WPS_STAMP(8 * l);
      const WpsPrm pp = wps_prm_of(w);
      wps_stage<T, LDSW>(w.win, &pp, wl, prm, tid);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    WPS_STAMP(8 * l + 7);
    float4 xo[2][4] WPS_Z;
    wps_layer_fwd<T, LDSW, false, TAPS, VIS>(w, LDSW ? wl : reinterpret_cast<const T*>(w.win), prm, xr, lane, ok, row0, srow, xo,
                                             (const WpsOut*)nullptr, E0, E1, (WpsKeep<T>*)nullptr, 8 * l);
    WPS_STAMP(8 * l + 6);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) xr[mt][nt] = xo[mt][nt];
  }
  WPS_STAMP(16);
  if constexpr (HEAD) {
    // ---- pooled heads of the block's samples (nets.py:1015-1036): [state token | mean of the 16 depth tokens] -> 256 -> 256 -> out
    const InfHead& h = hd.n[0];
    const int ns = min(WPS_WPB, n - s0);
    float* pooled = reinterpret_cast<float*>(smem);                    // [16][LDP] fp32 (rows >= ns: zeros)
    T* h1 = reinterpret_cast<T*>(pooled + 16 * LY::LDP);               // [16][LDF]
    T* h2 = h1 + 16 * LY::LDF;
    const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
    float4 hb0[4], hb1[4];
    float hb2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hb0[j] = *reinterpret_cast<const float4*>(h.b0 + nt4[j] * 16 + qr);
      hb1[j] = *reinterpret_cast<const float4*>(h.b1 + nt4[j] * 16 + qr);
      hb2[j] = h.b2[min(qr + j, h.nout - 1)];
    }
    GemmRing<T, 4, 4> ring0 = gemm_prefetch<T, 4, 4>((const T*)h.w0, 128, nt4, lane);
    __syncthreads();  // the weight region becomes the heads' scratch
    for (int i = tid; i < 16 * LY::LDP; i += 256) pooled[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      // token 0 sits in lane fr = 0 of tile 0; tokens 1..15 in the other lanes of tile 0, token 16 in lane fr = 0 of tile 1
      const float4 m = (VIS != 2 && fr == 0) ? xr[1][nt] : xr[0][nt];  // (native 16: token fr in lane fr)
      // max_pool=True (nets.py:1022-1023, 886-887; round 5): the max over the same 16 tokens instead of their mean
      const float4 mv = h.max_pool ? float4{rowmax16(m.x), rowmax16(m.y), rowmax16(m.z), rowmax16(m.w)}
                                   : float4{rowsum16(m.x) * (1.f / 16.f), rowsum16(m.y) * (1.f / 16.f), rowsum16(m.z) * (1.f / 16.f),
                                            rowsum16(m.w) * (1.f / 16.f)};
      if (fr == 0 && live) {
        // VIS: the head reads the mean of the 16 tokens only — the dummy row's half is zeros here and h.w0 is the [256][128]
        // pack whose columns 0..63 are zero
        const float4 t0 = VIS ? float4{0.f, 0.f, 0.f, 0.f} : xr[0][nt];
        *reinterpret_cast<float4*>(pooled + wave * LY::LDP + nt * 16 + qr) = t0;
        *reinterpret_cast<float4*>(pooled + wave * LY::LDP + TD + nt * 16 + qr) = mv;
        if (h.s_pooled != nullptr) {
          *reinterpret_cast<float4*>(h.s_pooled + (int64_t)smp * 128 + nt * 16 + qr) = t0;
          *reinterpret_cast<float4*>(h.s_pooled + (int64_t)smp * 128 + TD + nt * 16 + qr) = mv;
        }
      }
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
    block_gemm<T, 1, 4, 4>(acc, pooled, LY::LDP, (const T*)h.w0, 128, nt4, lane, ring0);
    GemmRing<T, 4, 8> ring1 = gemm_prefetch<T, 4, 8>((const T*)h.w1, 256, nt4, lane);
    store_h(h1, hb0, h.s_h0);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, h1, LY::LDF, (const T*)h.w1, 256, nt4, lane, ring1);
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
        if (fr < ns) h.out[(int64_t)(s0 + fr) * OUT_LD + c] = v;
      }
    }
  }
  WPS_STAMP(17);
}

// ------------------------------------------------------------------------------------------ backward
struct WpsBwdLayer {
  const void* w;       // fragment-order forward weight block (PK_FRAGP)  — the recompute
  const void* wt;      // fragment-order transposed weight block (PK_FRAGPT) — the data-grads
  const float *bin, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
  const float* xin;    // [R][64] fp32 layer input rows (saved by the forward / produced by the encoder)
  void *wg, *tk;       // weight-grad operand blocks of this layer: [n][WPS_WG_ELEMS], [n][WPS_TK_ELEMS] (T)
  float *gp2, *bp2, *gp1, *bp1;  // [gridDim.x][64] per-block dgamma / dbeta partials of norm2 / norm1
  float* o_dx;         // [R][64] grad w.r.t. the layer input (row-major fp32) or null
  void *t_dz2, *t_df, *t_dz1, *t_dqkv;  // test taps: row-major T rows as the block-cooperative kernel leaves them, or null
};
struct WpsBwdStack { WpsBwdLayer l[2]; };  // l[0] = the upper layer

// Backward of one layer of one sample by one wave. dy (T layout, fp32; rows >= 17 and dead samples exactly zero) is replaced by
// the gradient w.r.t. the layer input. Wt: the transposed weight block. lnred: this wave's [4][64] LDS slots for the LayerNorm
// parameter gradients.
template <typename T, bool LDSW, bool TAPS, int NMT = WPS_NMT_DEF>
__device__ __forceinline__ void wps_layer_bwd(const WpsBwdLayer& w, const T* Wt, const float* prm, const WpsKeep<T>& K, float4 (&dy)[2][4],
                                              int lane, const bool (&ok)[2], int64_t row0, const WpsOut& wo, const typename Frag<T>::type& E0,
                                              const typename Frag<T>::type& E1, float* lnred) {
  typedef typename Frag<T>::type frag_t;
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  const bool live = ok[0];
  auto ln_bwd = [&](float4 (&d)[2][4], const float4 (&xh)[2][4], const float (&rs)[2], int goff, float* red_g, float* red_b) {
    // dz = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)); dgamma / dbeta partials of this sample -> LDS
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float4 a = d[0][nt], xa_ = xh[0][nt];
      float4 b = {0.f, 0.f, 0.f, 0.f}, xb_ = {0.f, 0.f, 0.f, 0.f};
      if constexpr (NMT == 2) { b = d[1][nt]; xb_ = xh[1][nt]; }
      const float4 sg = {rowsum16(fmaf(a.x, xa_.x, b.x * xb_.x)), rowsum16(fmaf(a.y, xa_.y, b.y * xb_.y)),
                         rowsum16(fmaf(a.z, xa_.z, b.z * xb_.z)), rowsum16(fmaf(a.w, xa_.w, b.w * xb_.w))};
      const float4 sb = {rowsum16(a.x + b.x), rowsum16(a.y + b.y), rowsum16(a.z + b.z), rowsum16(a.w + b.w)};
      // (after the row sums all 16 lanes of a group hold the same totals: every lane writes them — same address, same value —
      // rather than lane fr = 0 alone under an exec-mask branch)
      *reinterpret_cast<float4*>(red_g + nt * 16 + qr) = sg;
      *reinterpret_cast<float4*>(red_b + nt * 16 + qr) = sb;
    }
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      float4 dxh[4];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 gg = *reinterpret_cast<const float4*>(prm + goff + nt * 16 + qr);
        dxh[nt] = float4{d[mt][nt].x * gg.x, d[mt][nt].y * gg.y, d[mt][nt].z * gg.z, d[mt][nt].w * gg.w};
        c1 += (dxh[nt].x + dxh[nt].y) + (dxh[nt].z + dxh[nt].w);
        c2 += (dxh[nt].x * xh[mt][nt].x + dxh[nt].y * xh[mt][nt].y) + (dxh[nt].z * xh[mt][nt].z + dxh[nt].w * xh[mt][nt].w);
      }
      c1 = xsum(c1) * (1.f / TD);
      c2 = xsum(c2) * (1.f / TD);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 x = xh[mt][nt];
        d[mt][nt] = float4{rs[mt] * (dxh[nt].x - c1 - x.x * c2), rs[mt] * (dxh[nt].y - c1 - x.y * c2),
                           rs[mt] * (dxh[nt].z - c1 - x.z * c2), rs[mt] * (dxh[nt].w - c1 - x.w * c2)};
      }
    }
  };
  auto tap_rows = [&](void* dst, int ld, int col0, const float4 (&v)[2]) {  // test tap: 4 features of both row tiles, row-major T
    if (!TAPS || dst == nullptr) return;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
      if (ok[mt]) st4(reinterpret_cast<T*>(dst) + (row0 + mt * 16 + fr) * ld + col0 + qr, v[mt].x, v[mt].y, v[mt].z, v[mt].w);
  };
  // ---- norm2 backward: dy -> dz2
  ln_bwd(dy, K.xh2, K.rs2, WPS_P_G2, lnred, lnred + TD);
  frag_t dza[2][2] WPS_Z;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dza[mt][ks] = wps_frag<T>(dy[mt][2 * ks], dy[mt][2 * ks + 1]);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wps_store_opnd<T, NMT>(wo, WPS_T_DZ2 / 2 + ks, dza[0][ks], dza[1][ks], E0, E1, lane);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { const float4 v[2] = {dy[0][nt], dy[1][nt]}; tap_rows(w.t_dz2, TD, nt * 16, v); }
  // ---- df = (dz2 W2) o [f > 0], 32 hidden features at a time, each chunk feeding dx1 += df W1
  f32x4 dx1[2][4] WPS_Z;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dx1[mt][nt] = zero4();
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    float4 dd[2][2] WPS_Z;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tile = 2 * ch + h;
      f32x4 acc[2] = {zero4(), zero4()};
      wps_gemm_t<T, LDSW, 2, NMT>(acc, Wt + WPS_OFF_W2, tile, dza, lane);  // W2^T: rows = hidden features, k = the 64 outputs
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        const unsigned b4 = (unsigned)(K.fm[mt] >> (tile * 4)) & 15u;
        dd[mt][h] = float4{(b4 & 1u) ? acc[mt][0] : 0.f, (b4 & 2u) ? acc[mt][1] : 0.f, (b4 & 4u) ? acc[mt][2] : 0.f,
                           (b4 & 8u) ? acc[mt][3] : 0.f};
      }
      const float4 v[2] = {dd[0][h], dd[1][h]};
      tap_rows(w.t_df, 256, tile * 16, v);
    }
    frag_t dfa[2] WPS_Z;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) dfa[mt] = wps_frag<T>(dd[mt][0], dd[mt][1]);
    wps_store_opnd<T, NMT>(wo, WPS_T_DF / 2 + ch, dfa[0], dfa[1], E0, E1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const frag_t fw = wps_w<T, LDSW>(Wt + WPS_OFF_W1, nt * 8 + ch, lane);  // W1^T: rows = the 64 inputs, k = hidden features
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) mma_k32(dx1[mt][nt], fw, dfa[mt]);
    }
  }
  float4 d1[2][4] WPS_Z;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) d1[mt][nt] = f4add(dx1[mt][nt], dy[mt][nt]);
  // ---- norm1 backward: dx1 -> dz1
  ln_bwd(d1, K.xh1, K.rs1, WPS_P_G1, lnred + 2 * TD, lnred + 3 * TD);
  frag_t dz1a[2][2] WPS_Z;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dz1a[mt][ks] = wps_frag<T>(d1[mt][2 * ks], d1[mt][2 * ks + 1]);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wps_store_opnd<T, NMT>(wo, WPS_T_DZ1 / 2 + ks, dz1a[0][ks], dz1a[1][ks], E0, E1, lane);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { const float4 v[2] = {d1[0][nt], d1[1][nt]}; tap_rows(w.t_dz1, TD, nt * 16, v); }
  // ---- dctx = dz1 Wo  (an operand of the attention products only: kept rounded to T)
  frag_t dca[2][2] WPS_Z;
  {
    float4 dc[2][4] WPS_Z;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc[2] = {zero4(), zero4()};
      wps_gemm_t<T, LDSW, 2, NMT>(acc, Wt + WPS_OFF_WO, dt, dz1a, lane);
      dc[0][dt] = f4(acc[0]);
      dc[1][dt] = f4(acc[1]);
    }
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) dca[mt][ks] = wps_frag<T>(dc[mt][2 * ks], dc[mt][2 * ks + 1]);
  }
  // ---- attention backward: dP = dctx V^T ; dS = P o (dP - rowsum(P o dP)) ; dV = P^T dctx ; dQ = dS K / 8 ; dK = dS^T Q / 8
  frag_t dsa[2] WPS_Z, pa[2] WPS_Z;
#pragma unroll
  for (int qt = 0; qt < NMT; ++qt) {
    f32x4 dp[2] = {zero4(), zero4()};
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) mma_k32(dp[kt], K.va[kt][ks], dca[qt][ks]);  // dp[kt][r] = dctx[16 qt + fr] . v[16 kt + 4g + r]
    float rd = 0.f;
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) rd = fmaf(K.p[qt][kt][r], kt * 16 + qr + r < NTOK ? dp[kt][r] : 0.f, rd);
    rd = xsum(rd);
    float ds[2][4] WPS_Z;
#pragma unroll
    for (int kt = 0; kt < NMT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[kt][r] = kt * 16 + qr + r < NTOK ? K.p[qt][kt][r] * (dp[kt][r] - rd) : 0.f;
    dsa[qt] = wps_frag<T>(float4{ds[0][0], ds[0][1], ds[0][2], ds[0][3]}, float4{ds[1][0], ds[1][1], ds[1][2], ds[1][3]});
    pa[qt] = wps_frag<T>(float4{K.p[qt][0][0], K.p[qt][0][1], K.p[qt][0][2], K.p[qt][0][3]},
                         float4{K.p[qt][1][0], K.p[qt][1][1], K.p[qt][1][2], K.p[qt][1][3]});
  }
  frag_t dsT[2] WPS_Z, pT[2] WPS_Z, dcT[4];
#pragma unroll
  for (int kt = 0; kt < NMT; ++kt) {
    dsT[kt] = wps_tr<T>(dsa[0], dsa[1], kt ? E1 : E0);  // rows = keys of tile kt, slots = queries
    pT[kt] = wps_tr<T>(pa[0], pa[1], kt ? E1 : E0);
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) dcT[2 * ks + h] = wps_tr<T>(dca[0][ks], dca[1][ks], h ? E1 : E0);  // rows = features, slots = queries
  frag_t dqa[2][6] WPS_Z;
  {
    float4 dq[2][4] WPS_Z, dk[2][4] WPS_Z, dv[2][4] WPS_Z;
#pragma unroll
    for (int tt = 0; tt < NMT; ++tt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 a = zero4(), b = zero4(), c = zero4();
        mma_k32(a, K.ka_f[dt], dsa[tt]);  // dQ[query 16 tt + fr][16 dt + 4g + r] = sum_key dS[q][key] K[key][d]
        mma_k32(b, K.qa_f[dt], dsT[tt]);  // dK[key 16 tt + fr][...]              = sum_q dS[q][key] Q[q][d]
        mma_k32(c, dcT[dt], pT[tt]);      // dV[key 16 tt + fr][...]              = sum_q P[q][key] dctx[q][d]
        dq[tt][dt] = float4{a[0] * 0.125f, a[1] * 0.125f, a[2] * 0.125f, a[3] * 0.125f};
        dk[tt][dt] = float4{b[0] * 0.125f, b[1] * 0.125f, b[2] * 0.125f, b[3] * 0.125f};
        dv[tt][dt] = f4(c);
        if (TAPS && w.t_dqkv != nullptr && ok[tt]) {  // test tap: the dq | dk | dv rows, row-major T
          T* o = reinterpret_cast<T*>(w.t_dqkv) + (row0 + tt * 16 + fr) * 192 + dt * 16 + qr;
          st4(o, dq[tt][dt].x, dq[tt][dt].y, dq[tt][dt].z, dq[tt][dt].w);
          st4(o + TD, dk[tt][dt].x, dk[tt][dt].y, dk[tt][dt].z, dk[tt][dt].w);
          st4(o + 2 * TD, dv[tt][dt].x, dv[tt][dt].y, dv[tt][dt].z, dv[tt][dt].w);
        }
      }
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dqa[mt][ks] = wps_frag<T>(dq[mt][2 * ks], dq[mt][2 * ks + 1]);
        dqa[mt][2 + ks] = wps_frag<T>(dk[mt][2 * ks], dk[mt][2 * ks + 1]);
        dqa[mt][4 + ks] = wps_frag<T>(dv[mt][2 * ks], dv[mt][2 * ks + 1]);
      }
  }
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) wps_store_opnd<T, NMT>(wo, WPS_T_DQKV / 2 + ks, dqa[0][ks], dqa[1][ks], E0, E1, lane);
  // ---- dx_in = dz1 + dqkv Win
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    f32x4 acc[2] = {zero4(), zero4()};
    wps_gemm_t<T, LDSW, 6, NMT>(acc, Wt, nt, dqa, lane);  // Win^T: rows = the 64 inputs, k = the 192 outputs
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) dy[mt][nt] = f4add(acc[mt], d1[mt][nt]);
  }
}

// ------------------------------------------------------------------------------------------ batched row chains (round 4)
// The pooled heads' data-grads (dout -> W2' -> dh1 -> W1' -> dh0 -> W0' -> dpool) and the proprio branch's (token-0 gradient ->
// state_projector' -> dhc -> fc2' -> de0) are chains of row-independent GEMMs over the n samples. Inside the wave-per-sample
// backward a block owns 4 samples, so each chain streams its 200 / 170 KB of weights for FOUR rows (19 K + 19 K of the kernel's
// 157 K cycles). Here the same chains run over 16 MT rows per block — the same MFMA steps in the same k order
// per output element (bit-identical to the in-kernel chains) — as extra blocks of launches that are on the update's path anyway:
// the heads beside the loss statistics (critic_loss_kernel / actor_loss_kernel: a row's loss gradient needs nothing but that
// row and the advantage statistics), the proprio chain beside the layers' weight-grads (wps_wgrad_kernel), whose successor
// (the grouped dense weight-grads) is its only consumer.
struct RowsChain {
  const void *wa, *wb, *wc;  // data-grad packs [N][K] row-major: [256][64], [256][256], [128][256] (wc null: two stages)
  const float *ma, *mb;      // [n][256] post-ReLU activations (ReLU masks) of the first / second stage's outputs
  float *oa, *ob;            // [n][256] the masked outputs (dY operands of the weight-grads)
  float* oc;                 // [n][128] third stage's output (un-masked)
};
template <typename T, int MT> struct RowsChainLds {
  static constexpr int LDX = 64 + 4, LDF = 256 + InfLd<T>::PAD;
  static constexpr size_t dt_b = (size_t)MT * 16 * LDX * 4, dh_b = (size_t)MT * 16 * LDF * sizeof(T);
  static constexpr size_t bytes2 = dt_b + dh_b, bytes3 = dt_b + 2 * dh_b;  // two- / three-stage chain
};
// dt: LDS [16 MT][LDX] fp32 input rows (zero beyond the chain's input width and for rows >= n), filled AND synchronised by
// fill_dt(), which runs after the first weight fragments and ReLU masks have been requested (they arrive while the rows are
// being computed: a row of the loss gradient is itself two dependent loads deep); dha / dhb: LDS [16 MT][LDF] T. NW waves per
// block; rows r0 .. r0 + 16 MT - 1.
// EARLY_B: also the second stage's first fragments before fill_dt() (not when fill_dt itself needs most of the registers)
template <typename T, int NW, int MT, bool EARLY_B, class Fill>
__device__ __forceinline__ void rows_chain(const RowsChain& c, const float* dt, T* dha, T* dhb, int r0, int n, int tid, Fill&& fill_dt) {
  constexpr int LDX = RowsChainLds<T, MT>::LDX, LDF = RowsChainLds<T, MT>::LDF;
  constexpr int NTW = 16 / NW;  // column tiles of a 256-wide stage per wave
  static_assert(NW == 4 || NW == 8 || NW == 16, "rows_chain: 4, 8 or 16 waves");
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, qr = (lane >> 4) * 4;
  int nt[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) nt[j] = wave * NTW + j;
  GemmRing<T, NTW, 2> ring_a = gemm_prefetch<T, NTW, 2>((const T*)c.wa, 64, nt, lane);
  GemmRing<T, NTW, 8> ring_b;
  if constexpr (EARLY_B) ring_b = gemm_prefetch<T, NTW, 8>((const T*)c.wb, 256, nt, lane);
  float4 mk[MT][NTW];
  auto load_masks = [&](const float* __restrict__ m) {  // unconditional loads from clamped rows
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = min(r0 + mt * 16 + fr, n - 1);
#pragma unroll
      for (int j = 0; j < NTW; ++j) mk[mt][j] = *reinterpret_cast<const float4*>(m + (int64_t)row * 256 + nt[j] * 16 + qr);
    }
  };
  load_masks(c.ma);
  fill_dt();
  if constexpr (!EARLY_B) ring_b = gemm_prefetch<T, NTW, 8>((const T*)c.wb, 256, nt, lane);
  f32x4 acc[MT][NTW];
  auto masked = [&](T* dst, float* __restrict__ save) {  // ReLU mask from the saved activation
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = r0 + mt * 16 + fr;
      const bool ok = row < n;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n4 = nt[j] * 16 + qr;
        const float4 m = mk[mt][j];
        const float d0 = m.x > 0.f ? acc[mt][j][0] : 0.f, d1 = m.y > 0.f ? acc[mt][j][1] : 0.f;
        const float d2 = m.z > 0.f ? acc[mt][j][2] : 0.f, d3 = m.w > 0.f ? acc[mt][j][3] : 0.f;
        if (dst != nullptr) st4(dst + (mt * 16 + fr) * LDF + n4, d0, d1, d2, d3);
        if (ok) st4(save + (int64_t)row * 256 + n4, d0, d1, d2, d3);
      }
    }
  };
  zero_acc(acc);
  block_gemm<T, MT, NTW, 2>(acc, dt, LDX, (const T*)c.wa, 64, nt, lane, ring_a);
  masked(dha, c.oa);
  load_masks(c.mb);
  __syncthreads();
  zero_acc(acc);
  block_gemm<T, MT, NTW, 8>(acc, dha, LDF, (const T*)c.wb, 256, nt, lane, ring_b);
  const bool three = c.wc != nullptr;  // block-uniform
  constexpr int NT3 = NW >= 8 ? 1 : 8 / NW;  // the 128-wide third stage: 8 column tiles
  int nt3[NT3];
#pragma unroll
  for (int j = 0; j < NT3; ++j) nt3[j] = min(wave * NT3 + j, 7);
  GemmRing<T, NT3, 8> ring_c;
  if (three) ring_c = gemm_prefetch<T, NT3, 8>((const T*)c.wc, 256, nt3, lane);  // ahead of this stage's global stores
  masked(three ? dhb : (T*)nullptr, c.ob);
  if (!three) return;
  __syncthreads();
  if (wave * NT3 < 8) {
    f32x4 a3[MT][NT3];
    zero_acc(a3);
    block_gemm<T, MT, NT3, 8>(a3, dhb, LDF, (const T*)c.wc, 256, nt3, lane, ring_c);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = r0 + mt * 16 + fr;
      if (row < n) {
#pragma unroll
        for (int j = 0; j < NT3; ++j)
          st4(c.oc + (int64_t)row * 128 + nt3[j] * 16 + qr, a3[mt][j][0], a3[mt][j][1], a3[mt][j][2], a3[mt][j][3]);
      }
    }
  }
}
// rows per block of the heads' chain in the loss launches: 16 — a chain block's time is its CU pulling the chain's 224 KB of
// weights plus its rows' masks and saves through one L1, so fewer rows per block and more blocks is faster (measured,
// actor_loss_heads_kernel at B = 1024: 64 rows 21.3 us, 32 rows 17.2 - 18.5, 16 rows 15.8 = the statistics block's own time);
// of the proprio chain inside wps_wgrad_kernel: 32 (16 measured the same; its accumulators must fit beside that kernel's two
// waves per SIMD)
template <typename T> struct RowsChainCfg { static constexpr int MT = 1; static constexpr int MT_TOK0 = 2; };

// the proprio chain of BwdTail for rows r0.. : token-0 rows of the layer-0 input gradient, masked by the token's ReLU
template <typename T, int NW>
__device__ __forceinline__ void tok0_chain_block(const BwdTail& tl, const float* __restrict__ dx0, int n, int r0, unsigned char* smem,
                                                 int tid) {
  constexpr int MT = RowsChainCfg<T>::MT_TOK0;
  typedef RowsChainLds<T, MT> LY;
  float* dt = reinterpret_cast<float*>(smem);
  T* dh = reinterpret_cast<T*>(smem + LY::dt_b);
  RowsChain c;
  c.wa = tl.wpt; c.wb = tl.wf2t; c.wc = nullptr;
  c.ma = tl.s_e1; c.mb = tl.s_e0; c.oa = tl.o_dhc; c.ob = tl.o_de0; c.oc = nullptr;
  rows_chain<T, NW, MT, true>(c, dt, dh, (T*)nullptr, r0, n, tid, [&]() {
    for (int idx = tid; idx < MT * 16 * 16; idx += NW * 64) {
      const int r = idx >> 4, c4 = (idx & 15) * 4, row = r0 + r;
      const bool ok = row < n;
      const int64_t o = (int64_t)(ok ? row : 0) * NTOK * TD + c4;
      const float4 d = *reinterpret_cast<const float4*>(dx0 + o), x = *reinterpret_cast<const float4*>(tl.x0 + o);
      *reinterpret_cast<float4*>(dt + r * LY::LDX + c4) =
          ok ? float4{x.x > 0.f ? d.x : 0.f, x.y > 0.f ? d.y : 0.f, x.z > 0.f ? d.z : 0.f, x.w > 0.f ? d.w : 0.f}
             : float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
  });
}

// The loss launches with the heads' data-grad chain beside the statistics: block 0 = critic_loss_kernel / actor_loss_kernel
// (csrc/elem.h: statistics, the loss gradient rows for the last linear's weight-grad, d log sigma); block 1 + b = rows
// 16 MT b .. of the same loss gradient (recomputed from the row: critic_row / actor_row, the same bits) -> W2' -> dh1 -> W1' ->
// dh0 -> W0' -> dpool, which the wave-per-sample backward (HEAD_IN = false) un-pools. Dynamic LDS: RowsChainLds<T, MT>::bytes3.
template <typename T, int NWAVES, bool EARLY_B, class Fill>
__device__ __forceinline__ void loss_heads_run(const RowsChain& hc, float* dt, int r0, int n, unsigned char* smem, int tid, Fill&& fill) {
  constexpr int MT = RowsChainCfg<T>::MT;
  typedef RowsChainLds<T, MT> LY;
  T* dha = reinterpret_cast<T*>(smem + LY::dt_b);
  T* dhb = reinterpret_cast<T*>(smem + LY::dt_b + LY::dh_b);
  rows_chain<T, NWAVES, MT, EARLY_B>(hc, dt, dha, dhb, r0, n, tid, fill);
}
template <typename T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void critic_loss_heads_kernel(const float* __restrict__ values, const float* __restrict__ ret,
                                                                const float* __restrict__ oldv, const int* __restrict__ rowidx,
                                                                int n, float inv_n, int clipped, float clip,
                                                                float* __restrict__ dvalues, float* __restrict__ st, RowsChain hc,
                                                                float gscale) {
  if (blockIdx.x == 0) {
    critic_loss_body(values, ret, oldv, rowidx, n, inv_n, clipped, clip, dvalues, st, gscale);
    return;
  }
  constexpr int MT = RowsChainCfg<T>::MT;
  typedef RowsChainLds<T, MT> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dt = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, r0 = ((int)blockIdx.x - 1) * MT * 16;
  loss_heads_run<T, NWAVES, true>(hc, dt, r0, n, smem, tid, [&]() {
    for (int idx = tid; idx < MT * 16 * 16; idx += NWAVES * 64) {
      const int r = idx >> 4, c4 = (idx & 15) * 4, i = r0 + r;
      float g = 0.f;
      if (c4 == 0 && i < n) {
        const int slot = rowidx ? rowidx[i] : i;
        float l;
        critic_row(values[(int64_t)i * OUT_LD], ret[slot], clipped ? oldv[slot] : 0.f, clipped, clip, inv_n, l, g);
        int sat = 0;
        g = grad_out(g, gscale, sat);
      }
      *reinterpret_cast<float4*>(dt + r * LY::LDX + c4) = float4{g, 0.f, 0.f, 0.f};
    }
    __syncthreads();
  });
}
template <typename T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void actor_loss_heads_kernel(ActorArgs p, RowsChain hc) {
  if (blockIdx.x == 0) {
    actor_loss_body<false>(p);  // (TanhNormal policies keep the in-kernel heads and actor_loss_kernel<true>)
    return;
  }
  constexpr int MT = RowsChainCfg<T>::MT;
  typedef RowsChainLds<T, MT> LY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dt = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, r0 = ((int)blockIdx.x - 1) * MT * 16;
  loss_heads_run<T, NWAVES, false>(hc, dt, r0, p.n, smem, tid, [&]() {
    for (int idx = tid; idx < MT * 16 * 16; idx += NWAVES * 64)  // columns 8.. (and rows >= n) stay zero
      *reinterpret_cast<float4*>(dt + (idx >> 4) * LY::LDX + (idx & 15) * 4) = float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (tid < MT * 16 && r0 + tid < p.n) {  // one thread per row: the row's d(loss)/d(mean), as block 0 files it in dmean
      const ActorDims D = actor_dims(p);
      const int i = r0 + tid, slot = p.rowidx ? p.rowidx[i] : i;
      const ActorRow o = actor_row<false>(p, D, i, slot, p.st[ST_ADV_MEAN], p.st[ST_ADV_STD]);
      float4* drow = reinterpret_cast<float4*>(dt + tid * LY::LDX);
      float dmr[8];
      int sat = 0;
      actor_dmean_row(o, p.gscale, dmr, sat);
      drow[0] = float4{dmr[0], dmr[1], dmr[2], dmr[3]};
      drow[1] = float4{dmr[4], dmr[5], dmr[6], dmr[7]};
    }
    __syncthreads();
  });
}

// Backward of the transformer stack for WPS_WPB samples per block: pooled heads (cooperative, as bwd_layer_kernel's HEAD) ->
// per layer {recompute the forward in registers, walk it backward} -> encoder-side data-grads (TAIL: up-conv per wave in
// registers, the token-0 chain cooperatively).
// HEAD_IN / TOK0_IN = false: that chain ran / will run outside this kernel over 64 rows per block (rows_chain above) — the
// heads beside the loss statistics, leaving dpool [n][128] (tx.dpool) for the un-pool here; the proprio chain beside the
// layers' weight-grads, from the layer-0 input gradient this kernel writes (stk.l[NL-1].o_dx).
// xlast: the layer stack's output rows [n][17][64] when the head pools the depth tokens by max (max_pool=True): the gradient of
// a max goes to the first token that attained it (torch.max(dim) backward; pool_bwd_kernel), found again from these rows
struct WpsTailExtra { const void* wupt_f; const float* dpool; const float* xlast; const float* dyrows; };  // wupt_f: up-conv's transposed weight as a k-permuted fragment pack
// un-pooling weights of a sample's depth tokens (lane fr: token fr of tile 0, lane fr = 0 of tile 1: token 16), per feature column:
// 1/16 for the mean; for the max 1 on the first token that holds the maximum of its column, 0 elsewhere
template <int VIS>
__device__ __forceinline__ void wps_unpool_weights(const float* xlast_rows, int lane, const bool (&ok)[2], float4 (&w0)[4]) {
  const int fr = lane & 15;
  if (xlast_rows == nullptr) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w0[nt] = float4{1.f / 16.f, 1.f / 16.f, 1.f / 16.f, 1.f / 16.f};
    return;
  }
  float4 xl[2][4];
  wps_load_rows<VIS>(xlast_rows, lane, ok, xl);
  const int idx = VIS == 2 ? fr : (fr == 0 ? 16 : fr);  // the depth token this lane stands for in the row-wide reductions below
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 m = (VIS != 2 && fr == 0) ? xl[1][nt] : xl[0][nt];
    const float mx[4] = {rowmax16(m.x), rowmax16(m.y), rowmax16(m.z), rowmax16(m.w)};
    const float mv[4] = {m.x, m.y, m.z, m.w};
    float wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int first = rowmin16(mv[r] == mx[r] ? idx : 99);
      wv[r] = idx == first ? 1.f : 0.f;
    }
    const float4 wq = {wv[0], wv[1], wv[2], wv[3]};
    // lane fr = 0 carries token 16's weight (its tile-1 slot); in tile 0 that lane is the proprio token (not pooled: unused there)
    w0[nt] = wq;
  }
}
// MODE (round 5, the token_norm / use_pytorch_encoder options around the fused layers): bit 0 = the launch ends with the
// layer-0 input gradient rows (o_dx) — the caller runs token_ln's backward and the encoder-side data-grads itself; bit 1 = the
// launch starts from row-major gradient rows w.r.t. the stack's output (tx.dyrows: the final LayerNorm's backward wrote them)
// instead of running / un-pooling the heads.
template <typename T, int NL, bool TAPS, int VIS = 0, bool HEAD_IN = true, bool TOK0_IN = true, int MODE = 0>
__global__ __launch_bounds__(256) WPS_EU_ATTR void wps_layer_bwd_kernel(WpsBwdStack stk, BwdHead hd, BwdTail tl, WpsTailExtra tx, int n) {
  constexpr int NMT = VIS == 2 ? 1 : WPS_NMT_DEF;
  constexpr int ROFF = VIS == 2 ? 1 : 0;  // (see wps_layer_fwd)
  // VIS (template parameter): see wps_layer_fwd — hd.w0t is then the [128][256] pack whose rows 0..63 are zero (the dummy
  // row's un-pooled gradient is exactly zero), and the TAIL ends after the up-conv data-grad (there is no proprio branch)
  typedef WpsBwdLds<T> LY;
  typedef typename Frag<T>::type frag_t;
  constexpr bool LDSW = LY::LDSW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* wl = reinterpret_cast<T*>(smem);
  float* prm = reinterpret_cast<float*>(smem + LY::main_b);
  float* red = prm + WPS_P_TOTAL;  // [WPS_WPB][4][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  const int s0 = blockIdx.x * WPS_WPB;
  const int ns = min(WPS_WPB, n - s0);
  const int smp = s0 + wave;
  const bool live = smp < n;
  const int64_t srow = live ? smp : 0;
  const int64_t row0 = srow * NTOK;
  const bool ok[2] = {live, live && fr == 0 && VIS != 2};
  const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  const frag_t E0 = wps_sel<T>(0, lane), E1 = wps_sel<T>(1, lane);
  float4 dy[2][4];
  WPS_STAMP(32);
  float4 upw[4];  // un-pooling weights of this lane's depth token (mean: 1/16; max_pool: the arg-max mask), both tiles
  if constexpr ((MODE & 2) == 0) wps_unpool_weights<VIS>(tx.xlast != nullptr ? tx.xlast + row0 * TD : nullptr, lane, ok, upw);
  if constexpr ((MODE & 2) != 0) {
    wps_load_rows<VIS>(tx.dyrows + row0 * TD, lane, ok, dy);
  } else if constexpr (!HEAD_IN) {
    // the heads ran beside the loss statistics: un-pool their dpool rows straight into this wave's registers
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bool tok0 = VIS != 2 && mt == 0 && fr == 0;
        const float4 v = *reinterpret_cast<const float4*>(tx.dpool + srow * (2 * TD) + (tok0 ? 0 : TD) + nt * 16 + qr);
        const float4 sc = tok0 ? float4{1.f, 1.f, 1.f, 1.f} : upw[nt];
        dy[mt][nt] = ok[mt] ? float4{v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w} : float4{0.f, 0.f, 0.f, 0.f};
      }
  } else {
    // ---- heads (nets.py:1015-1034 reversed): dout -> (W2^T, mask h1) -> dh1 -> (W1^T, mask h0) -> dh0 -> (W0^T) -> dpool
    float* dt = reinterpret_cast<float*>(smem);                 // [16][LDX]: dout rows, zero padded to 64 columns
    T* dh1 = reinterpret_cast<T*>(dt + 16 * LY::LDX);           // [16][LDF]
    T* dh0 = dh1 + 16 * LY::LDF;
    float* dpool = reinterpret_cast<float*>(dh0 + 16 * LY::LDF);  // [16][LDP]
    float dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = tid + k * 256, r = idx >> 6, c = idx & 63;
      const bool okd = r < ns && c < OUT_LD;
      const float v = hd.dout[okd ? (int64_t)(s0 + r) * OUT_LD + c : 0];
      dv[k] = okd ? v : 0.f;
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
    WPS_STAMP(60);
    f32x4 acc[1][4];
    auto masked = [&](const float4 (&m)[4], T* dst, float* save) {  // ReLU mask from the saved activation, rows < ns
      const bool okr = fr < ns;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[0][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[0][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[0][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[0][j][3] : 0.f;
        st4(dst + fr * LY::LDF + n4, d0, d1, d2, d3);
        if (okr) st4(save + (int64_t)(s0 + fr) * 256 + n4, d0, d1, d2, d3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 2>(acc, dt, LY::LDX, (const T*)hd.w2t, 64, nt4, lane, ring2);
    masked(m1, dh1, hd.o_dh1);
    __syncthreads();
    WPS_STAMP(61);
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh1, LY::LDF, (const T*)hd.w1t, 256, nt4, lane, ring1);
    masked(m0, dh0, hd.o_dh0);
    __syncthreads();
    WPS_STAMP(62);
    {
      f32x4 a2[1][2];
      zero_acc(a2);
      block_gemm<T, 1, 2, 8>(a2, dh0, LY::LDF, (const T*)hd.w0t, 256, nt2, lane, ring0);
#pragma unroll
      for (int j = 0; j < 2; ++j) st4(dpool + fr * LY::LDP + nt2[j] * 16 + qr, a2[0][j][0], a2[0][j][1], a2[0][j][2], a2[0][j][3]);
    }
    __syncthreads();
    WPS_STAMP(63);
    // un-pool (pool_bwd_kernel) straight into this wave's registers: token 0 <- dpool[:, 0:64], tokens 1..16 <- dpool[:, 64:128] / 16
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bool tok0 = VIS != 2 && mt == 0 && fr == 0;
        const float4 v = *reinterpret_cast<const float4*>(dpool + wave * LY::LDP + (tok0 ? 0 : TD) + nt * 16 + qr);
        const float4 sc = tok0 ? float4{1.f, 1.f, 1.f, 1.f} : upw[nt];
        dy[mt][nt] = ok[mt] ? float4{v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w} : float4{0.f, 0.f, 0.f, 0.f};
      }
  }
  float4 xr[2][4];
  WPS_LAYER_LOOP
  for (int l = 0; l < NL; ++l) {  // stk.l[0] = the upper layer
    const WpsBwdLayer& w = stk.l[l];
    const WpsOut wo = wps_out<T>(reinterpret_cast<T*>(w.wg) + srow * WPS_WG_STRIDE, reinterpret_cast<T*>(w.tk) + srow * WPS_TK_ELEMS, live);
    __syncthreads();  // the scratch / the previous layer's transposed weights are dead
    WPS_STAMP(33 + 8 * l);
    {
      const WpsPrm pp = WpsPrm{w.bin, w.bo, w.b1, w.b2, w.g1, w.be1, w.g2, w.be2};
      wps_stage<T, LDSW>(w.w, &pp, wl, prm, tid);
    }
    wps_load_rows<VIS>(w.xin + row0 * TD, lane, ok, xr);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    WPS_STAMP(34 + 8 * l);
    WpsKeep<T> K;
    {
      InfLayer none;
      none.win = none.wo = none.w1 = none.w2 = nullptr;
      none.bin = none.bo = none.b1 = none.b2 = none.g1 = none.be1 = none.g2 = none.be2 = nullptr;
      none.xin = nullptr; none.xout = nullptr;
      none.s_qkv = none.s_P = none.s_xh1 = none.s_rs1 = none.s_xh2 = none.s_rs2 = nullptr;
      none.s_xin = none.s_ctx = none.s_x1 = none.s_f = nullptr;
      float4 xo[2][4] WPS_Z;
      wps_layer_fwd<T, LDSW, true, false, VIS>(none, LDSW ? wl : reinterpret_cast<const T*>(w.w), prm, xr, lane, ok, row0, srow, xo, &wo, E0, E1, &K);
    }
    WPS_STAMP(35 + 8 * l);
    __syncthreads();  // every wave is done with the forward weights
    WPS_STAMP(36 + 8 * l);
    wps_stage<T, LDSW>(w.wt, (const WpsPrm*)nullptr, wl, prm, tid);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    WPS_STAMP(37 + 8 * l);
    wps_layer_bwd<T, LDSW, TAPS, NMT>(w, LDSW ? wl : reinterpret_cast<const T*>(w.wt), prm, K, dy, lane, ok, row0, wo, E0, E1, red + wave * 4 * TD);
    WPS_STAMP(38 + 8 * l);
    __syncthreads();
    {  // the block's LayerNorm parameter-gradient partials, waves summed in a fixed order
      const int k = tid >> 6, cidx = tid & 63;
      float sacc = 0.f;
#pragma unroll
      for (int wv = 0; wv < WPS_WPB; ++wv) sacc += red[(wv * 4 + k) * TD + cidx];
      float* dst = k == 0 ? w.gp2 : k == 1 ? w.bp2 : k == 2 ? w.gp1 : w.bp1;
      dst[(int64_t)blockIdx.x * TD + cidx] = sacc;
    }
    if (w.o_dx != nullptr) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        if (ok[mt])
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<float4*>(w.o_dx + (row0 + ROFF + mt * 16 + fr) * TD + nt * 16 + qr) = dy[mt][nt];
    }
  }
  if constexpr ((MODE & 1) != 0) return;
  // ---- TAIL (base.py:602-622 reversed). dy = grad w.r.t. the layer-0 input tokens; xr = those tokens (the ReLU mask of token 0)
  WPS_STAMP(50);
  {
    // tokens 1..16: dc3 = (dx_in Wup) o [c3 > 0], per wave in registers (up-conv's transposed weight: 8 KB, straight from L2)
    frag_t da[2][2] WPS_Z;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) da[mt][ks] = wps_frag<T>(dy[mt][2 * ks], dy[mt][2 * ks + 1]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc[2] = {zero4(), zero4()};
      wps_gemm_t<T, false, 2, NMT>(acc, reinterpret_cast<const T*>(tx.wupt_f), nt, da, lane);
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        const int patch = VIS == 2 ? fr : mt * 16 + fr - 1;  // the depth patch of this row's token (17 rows: token t = patch + 1)
        const bool okt = ok[mt] && patch >= 0;
        const int64_t o = okt ? ((int64_t)smp * 16 + patch) * TD + nt * 16 + qr : 0;
        const float4 m = *reinterpret_cast<const float4*>(tl.s_c3 + o);
        if (okt)
          *reinterpret_cast<float4*>(tl.o_dc3 + o) = float4{m.x > 0.f ? acc[mt][0] : 0.f, m.y > 0.f ? acc[mt][1] : 0.f,
                                                            m.z > 0.f ? acc[mt][2] : 0.f, m.w > 0.f ? acc[mt][3] : 0.f};
      }
    }
    WPS_STAMP(66);
    if constexpr (VIS || !TOK0_IN) return;
    // token 0: (dx_in o [x0 > 0]) -> state_projector' -> [e1 > 0] -> dhc -> fc2' -> [e0 > 0] -> de0, cooperatively (4 rows)
    float* dt = reinterpret_cast<float*>(smem);                 // [16][LDX]
    T* dh = reinterpret_cast<T*>(dt + 16 * LY::LDX);            // [16][LDF]
    float4 tm_e1[4], tm_e0[4];
    {
      const int64_t mrow = (int64_t)(s0 + (fr < ns ? fr : 0)) * 256;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tm_e1[j] = *reinterpret_cast<const float4*>(tl.s_e1 + mrow + nt4[j] * 16 + qr);
        tm_e0[j] = *reinterpret_cast<const float4*>(tl.s_e0 + mrow + nt4[j] * 16 + qr);
      }
    }
    GemmRing<T, 4, 2> ring_pr = gemm_prefetch<T, 4, 2>((const T*)tl.wpt, 64, nt4, lane);
    __syncthreads();  // the transposed weights are dead: their region takes the token-0 rows
    for (int i = tid; i < 16 * LY::LDX; i += 256) dt[i] = 0.f;
    __syncthreads();
    if (live && fr == 0) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 x = xr[0][nt], d = dy[0][nt];
        *reinterpret_cast<float4*>(dt + wave * LY::LDX + nt * 16 + qr) =
            float4{x.x > 0.f ? d.x : 0.f, x.y > 0.f ? d.y : 0.f, x.z > 0.f ? d.z : 0.f, x.w > 0.f ? d.w : 0.f};
      }
    }
    __syncthreads();
    WPS_STAMP(67);
    f32x4 acc[1][4];
    auto masked = [&](const float4 (&m)[4], T* dst, float* save) {
      const bool okr = fr < ns;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[0][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[0][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[0][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[0][j][3] : 0.f;
        if (dst != nullptr) st4(dst + fr * LY::LDF + n4, d0, d1, d2, d3);
        if (okr) st4(save + (int64_t)(s0 + fr) * 256 + n4, d0, d1, d2, d3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 2>(acc, dt, LY::LDX, (const T*)tl.wpt, 64, nt4, lane, ring_pr);
    GemmRing<T, 4, 8> ring_f2 = gemm_prefetch<T, 4, 8>((const T*)tl.wf2t, 256, nt4, lane);
    masked(tm_e1, dh, tl.o_dhc);
    __syncthreads();
    WPS_STAMP(68);
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh, LY::LDF, (const T*)tl.wf2t, 256, nt4, lane, ring_f2);
    masked(tm_e0, (T*)nullptr, tl.o_de0);
  }
  WPS_STAMP(51);
}

// ------------------------------------------------------------------------------------------ weight-grads
// dW = sum over samples and tokens of dY^T X for the four linears of every layer, from the fragment-order operand blocks the
// backward kernel left. One wave = one job: a 4 x 4-tile patch of one weight matrix (WPS_ROLES patches per layer) over one
// run of WPS_SPLIT samples, accumulated in registers and written as one partial slab (summed by wgrad_reduce_kernel in a
// fixed order). K = 32 per MFMA = tokens 0..15 of TWO samples (operand blocks are dense in those); the 17th tokens of the
// run's 32 samples make one more K = 32 step. Bias gradients = column sums of dY: one MFMA against an all-ones fragment.
struct WpsWgLayer {
  const void *wg, *tk;     // [n][WPS_WG_ELEMS], [n][WPS_TK_ELEMS] (T)
  float* slab[4];          // per matrix (in_proj, out_proj, linear1, linear2): [nsplit][N][K]
  float* bslab[4];         // [nsplit][N]
};
// chain_blocks > 0: the launch carries that many extra blocks (ahead of the weight-grad ones) that run the proprio branch's
// data-grad chain over RowsChainCfg<T>::MT_TOK0 * 16 rows each (tok0_chain_block; RowsChainLds<T, MT_TOK0>::bytes2 of dynamic LDS)
struct WpsWg { WpsWgLayer l[2]; int n, nsplit, nlayers; int wg_blocks, chain_blocks; BwdTail tl; const float* dx0; };
struct WpsRole { int mat, nt0, kt0, nrow0, kcol0, N, K, bias; };
__device__ __forceinline__ WpsRole wps_role(int r) {
  // dY-side tiles (rows n of dW) x x-side tiles (columns k of dW), as tile numbers inside the operand block
  if (r < 3) return WpsRole{0, WPS_T_DQKV + 4 * r, WPS_T_XIN, 64 * r, 0, 192, 64, 1};            // in_proj: dqkv x xin
  if (r == 3) return WpsRole{1, WPS_T_DZ1, WPS_T_CTX, 0, 0, 64, 64, 1};                          // out_proj: dz1 x ctx
  if (r < 8) return WpsRole{2, WPS_T_DF + 4 * (r - 4), WPS_T_X1, 64 * (r - 4), 0, 256, 64, 1};   // linear1: df x x1
  return WpsRole{3, WPS_T_DZ2, WPS_T_F + 4 * (r - 8), 0, 64 * (r - 8), 64, 256, r == 8};         // linear2: dz2 x f
}
template <typename T>
__global__ __launch_bounds__(256) void wps_wgrad_kernel(WpsWg a) {
  typedef typename Frag<T>::type frag_t;
  if ((int)blockIdx.x < a.chain_blocks) {  // (block-uniform; the chain blocks lead the grid: theirs is the longest latency chain)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tok0_chain_block<T, 4>(a.tl, a.dx0, a.n, (int)blockIdx.x * RowsChainCfg<T>::MT_TOK0 * 16, smem, threadIdx.x);
    return;
  }
  const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
  const int job = ((int)blockIdx.x - a.chain_blocks) * 4 + (threadIdx.x >> 6);
  const int per_split = WPS_ROLES * a.nlayers;
  const int split = job / per_split;
  if (split >= a.nsplit) return;
  const int rl = job - split * per_split;
  const int layer = rl / WPS_ROLES;
  const WpsRole ro = wps_role(rl - layer * WPS_ROLES);
  const WpsWgLayer& L = a.l[layer];
  const T* wg = reinterpret_cast<const T*>(L.wg);
  const T* tk = reinterpret_cast<const T*>(L.tk);
  const int sbeg = split * WPS_SPLIT, send = min(a.n, sbeg + WPS_SPLIT);
  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = zero4();
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4();
  }
  const frag_t ones = wps_frag<T>(float4{1.f, 1.f, 1.f, 1.f}, float4{1.f, 1.f, 1.f, 1.f});
  const frag_t zf = wps_frag<T>(float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f});
  // a loaded pair holds [tile 2p: 4 tokens | tile 2p+1: 4 tokens]; two samples' halves make one K = 32 fragment per tile
  auto halves = [&](const frag_t& ua, const frag_t& ub, frag_t& t0, frag_t& t1) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { t0[j] = ua[j]; t0[4 + j] = ub[j]; t1[j] = ua[4 + j]; t1[4 + j] = ub[4 + j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { t0.v[j] = ua.v[j]; t0.v[4 + j] = ub.v[j]; t1.v[j] = ua.v[4 + j]; t1.v[4 + j] = ub.v[4 + j]; }
    }
  };
  auto step = [&](const frag_t (&fx)[4], const frag_t (&fy)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma_k32(acc[i][j], fx[i], fy[j]);  // acc[i][j][r] = dW[n = 16 j + fr][k = 16 i + 4g + r]
    if (ro.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mma_k32(accb[j], ones, fy[j]);     // every row: sum over the step's tokens of dY[.][n = 16 j + fr]
    }
  };
  // (software-pipelined: the operand pairs of the NEXT two samples are requested before this step's MFMAs — a wave has the
  // SIMD to itself, so nothing else hides the L2 / HBM round trip)
  auto fetch = [&](int s, frag_t (&u)[8]) {  // [x pair 0: A, B | x pair 1: A, B | y pair 0: A, B | y pair 1: A, B]
    const int sa = s < send ? s : sbeg, sb_ = s + 1 < send ? s + 1 : sa;
    const T* pa = wg + (int64_t)sa * WPS_WG_STRIDE + lane * 8;
    const T* pb = wg + (int64_t)sb_ * WPS_WG_STRIDE + lane * 8;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u[2 * q] = *reinterpret_cast<const frag_t*>(pa + (int64_t)(ro.kt0 / 2 + q) * 512);
      u[2 * q + 1] = *reinterpret_cast<const frag_t*>(pb + (int64_t)(ro.kt0 / 2 + q) * 512);
      u[4 + 2 * q] = *reinterpret_cast<const frag_t*>(pa + (int64_t)(ro.nt0 / 2 + q) * 512);
      u[5 + 2 * q] = *reinterpret_cast<const frag_t*>(pb + (int64_t)(ro.nt0 / 2 + q) * 512);
    }
  };
  auto consume = [&](int s, const frag_t (&u)[8]) {
    const bool two = s + 1 < send;
    frag_t fx[4], fy[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      halves(u[2 * q], two ? u[2 * q + 1] : zf, fx[2 * q], fx[2 * q + 1]);
      halves(u[4 + 2 * q], two ? u[5 + 2 * q] : zf, fy[2 * q], fy[2 * q + 1]);
    }
    step(fx, fy);
  };
  {
    frag_t ua[8], ub[8];
    fetch(sbeg, ua);
    for (int s = sbeg; s < send; s += 4) {
      fetch(s + 2, ub);
      consume(s, ua);
      if (s + 2 < send) {
        fetch(s + 4, ua);
        consume(s + 2, ub);
      }
    }
  }
  if (tk != nullptr) {  // the 17th tokens of the run's samples: slot (g, j) <-> sample sbeg + 8 g + j, gathered from the k-permuted
    // side blocks (null: the native 16-token vision-only Transformer has no 17th token)
    frag_t fx[4], fy[4];
    auto gather = [&](int tile) {
      const int p = tile >> 1, h = tile & 1;
      const int loc = (p * 4 + (fr >> 2)) * 8 + 4 * h + (fr & 3);
      T v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = sbeg + 8 * g + j;
        const T x = tk[(int64_t)(s < send ? s : sbeg) * WPS_TK_ELEMS + loc];
        v[j] = s < send ? x : (T)0.f;
      }
      frag_t f;
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = v[j];
      }
      return f;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) { fx[i] = gather(ro.kt0 + i); fy[i] = gather(ro.nt0 + i); }
    step(fx, fy);
  }
  float* slab = L.slab[ro.mat] + (int64_t)split * ro.N * ro.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(slab + (int64_t)(ro.nrow0 + 16 * j + fr) * ro.K + ro.kcol0 + 16 * i + 4 * g) = f4(acc[i][j]);
  if (ro.bias && g == 0) {
    float* bs = L.bslab[ro.mat] + (int64_t)split * ro.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[ro.nrow0 + 16 * j + fr] = accb[j][0];
  }
}

}  // namespace v4l

#pragma clang fp contract(fast)  // (the including translation unit's default again)
