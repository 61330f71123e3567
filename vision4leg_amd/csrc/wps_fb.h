// Forward -> loss row -> backward of the transformer stack + pooled heads in ONE launch (round 6; gfx950).
//
// The update used to run a net's layers as wps_layer_fwd_kernel, then a loss launch (critic_loss_kernel / actor_loss_heads_kernel),
// then wps_layer_bwd_kernel, which recomputes both layers from their saved input rows. Nothing in that sequence needs another
// sample: the loss gradient of a row is a function of that row (its head output, its return / action / advantage / stored log
// pi_old) and of scalars known before the pass (the advantage moments of the minibatch: begin_pack_kernel; the mean()'s 1/n), so
// the wave that carried a sample forward can turn around on the spot (torchrl/algo/on_policy/ppo.py:42-123 — the reference's
// autograd graph has the same shape; nets.py:996-1038 is the forward it walks back through). One launch per net-pass then does:
//   layer 0 forward -> layer 1 forward -> pooled heads forward (cooperative, 4 rows) -> the block's loss-gradient rows + its share
//   of the logged statistics -> heads backward (ReLU masks from registers, d(out) rows from LDS) -> per layer {recompute, walk back}
//   (layer 1 from input rows that stayed in registers) -> encoder-side tail.
// Against the three launches: two launch boundaries and the loss launch are gone, layer 1's input rows, the head outputs, the
// loss-gradient rows and the heads' ReLU masks stop making round trips. The work per sample is the same (four layer-forwards).
// Measured and NOT kept: layer 1's backward operands (WpsKeep, ~136 registers) held across the heads so that layer 1 is not
// recomputed — the kernel then spills 131 registers to scratch and is slower than recomputing (update 619 - 632 us against
// 596 - 618; profiles/r6_fused_layers_ab.txt), and its results are a last bit away from the recompute's (another instantiation of
// the layer forward), which the f16 operands' rounding turns into visible tie flips.
// The statistics of the update (losses, log-prob / ratio moments, d log sigma) leave as per-block partial sums (doubles) and are
// finished by fb_loss_finish_kernel, one block beside the weight-grad launches — off the update's chain.
// The arithmetic per element is that of the separate kernels (the same device functions, the same instantiations of the layer
// functions, the same order): activations, gradients and the weight-grad operands are the same bits
// (tests/test_gpu_bench_shapes.py::test_fused_forward_loss_backward_equals_three_launches); only the statistics' summation order
// (per-block partials) differs.
#pragma once
#include "wps.h"

#pragma clang fp contract(on)

namespace v4l {

// per-block partial statistics (doubles): sums, extrema and the clamp count of the block's rows
enum { FBP_LP = 0, FBP_LP2, FBP_SUR, FBP_LPMAX, FBP_LPMIN, FBP_RMAX, FBP_RMIN, FBP_SAT, FBP_DL = 8, FBP_VF = 16, FB_PART = 24 };

struct FbLoss {
  int actor;                  // 0: critic — nn.MSELoss / the clipped value loss (ppo.py:94-112); 1: actor — clipped surrogate (ppo.py:42-70)
  const float *ret, *oldv;    // critic: estimate_returns, old values (rollout slots)
  int clipped;
  float clip;
  ActorArgs aa;               // actor: everything actor_row reads except the mean rows (they are in the block); dmean / mean unused
  const int* rowidx;          // minibatch row -> rollout slot, or null
  float inv_n, gscale;
  float* dout;                // [n][OUT_LD]: the loss-gradient rows (weight-grad operand of the last linear), scaled by gscale
  double* part;               // [blocks][FB_PART]
  float* st;                  // the update's statistics record
};

template <typename T> struct WpsFbLds {
  static constexpr bool LDSW = sizeof(T) == 2;
  static constexpr size_t f_b = WpsFwdLds<T>::main_b, b_b = WpsBwdLds<T>::main_b;
  static constexpr size_t main_b = f_b > b_b ? f_b : b_b;
  static constexpr size_t red_b = WpsBwdLds<T>::red_b;
  static constexpr size_t so_b = 16 * 16 * 4;                    // the block's head outputs [16][16] fp32
  static constexpr size_t rec_b = (size_t)WPS_WPB * FB_PART * 8;  // per row: its statistics record
  static constexpr size_t bytes = main_b + (size_t)WPS_P_TOTAL * 4 + red_b + so_b + rec_b;
  static_assert(bytes <= 160 * 1024, "one block per CU");
};

// actor_row (elem.h) on mean values the caller holds — the same expressions in the same order; stored log pi_old only
__device__ __forceinline__ ActorRow fb_actor_row(const ActorArgs& p, const ActorDims& D, const float (&mu)[8], int slot, float amean,
                                                 float astd) {
  ActorRow o;
  float lp = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    if (a < p.A) {
      const float act = p.acts[(int64_t)slot * p.A + a];
      const float d = act - mu[a];
      const float var = D.sg[a] * D.sg[a];
      lp += -(d * d) / (2.f * var) - D.lsg[a] - HALF_LOG_2PI - 0.f;
      o.z2[a] = d * d / var;
      o.dm[a] = d / var;
    } else {
      o.z2[a] = 0.f; o.dm[a] = 0.f;
    }
  }
  const float lpo = p.logp_old[slot];
  const float ratio = expf(lp - lpo);
  const float an = (p.adv[slot] - amean) / (astd + 1e-5f);
  const float pre = ratio * an;
  const float clp = fminf(fmaxf(ratio, 1.f - p.clip), 1.f + p.clip) * an;
  o.lp = lp; o.ratio = ratio;
  o.sur = fminf(pre, clp);
  o.dlp = (pre <= clp) ? -p.inv_n * an * ratio : 0.f;
  return o;
}

// TOK0_IN: the proprio branch's data-grad chain inside this launch (4 rows per block) instead of beside the layers' weight-grads
template <typename T, bool TOK0_IN>
__global__ __launch_bounds__(256) WPS_EU_ATTR void wps_layer_fb_kernel(InfLayerStack fst, InfHeadPair fhd, WpsBwdStack stk, BwdHead hd,
                                                                      BwdTail tl, WpsTailExtra tx, FbLoss lo, int n) {
  constexpr int NL = 2, VIS = 0, NMT = WPS_NMT_DEF, ROFF = 0;
  typedef WpsFbLds<T> LY;
  typedef WpsFwdLds<T> LF;
  typedef WpsBwdLds<T> LB;
  typedef typename Frag<T>::type frag_t;
  constexpr bool LDSW = LY::LDSW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* wl = reinterpret_cast<T*>(smem);
  float* prm = reinterpret_cast<float*>(smem + LY::main_b);
  float* red = prm + WPS_P_TOTAL;                                   // [WPS_WPB][4][64]
  float* so = red + WPS_WPB * 4 * TD;                               // [16][16]
  double* rec = reinterpret_cast<double*>(so + 16 * 16);            // [WPS_WPB][FB_PART]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4, qr = g * 4;
  const int s0 = blockIdx.x * WPS_WPB;
  const int ns = min(WPS_WPB, n - s0);
  const int smp = s0 + wave;
  const bool live = smp < n;
  const int64_t srow = live ? smp : 0;
  const int64_t row0 = srow * NTOK;
  const bool ok[2] = {live, live && fr == 0};
  const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  BLK_LOG_BEGIN();
  // ======================================================================================================== forward
  {  // layer 0's weights start their trip first
    const WpsPrm pp = wps_prm_of(fst.l[0].n[0]);
    wps_stage<T, LDSW>(fst.l[0].n[0].win, &pp, wl, prm, tid);
  }
  float4 xr[2][4];
  wps_load_rows<VIS>(fst.l[0].n[0].xin + row0 * TD, lane, ok, xr);
  const frag_t E0 = wps_sel<T>(0, lane), E1 = wps_sel<T>(1, lane);
  WpsKeep<T> K;
  const WpsBwdLayer& wb1 = stk.l[0];
  const WpsOut wo1 = wps_out<T>(reinterpret_cast<T*>(wb1.wg) + srow * WPS_WG_STRIDE, reinterpret_cast<T*>(wb1.tk) + srow * WPS_TK_ELEMS, live);
  {
    const InfLayer& w = fst.l[0].n[0];
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    float4 xo[2][4] WPS_Z;
    wps_layer_fwd<T, LDSW, false, false, VIS>(w, LDSW ? wl : reinterpret_cast<const T*>(w.win), prm, xr, lane, ok, row0, srow, xo,
                                              (const WpsOut*)nullptr, E0, E1, (WpsKeep<T>*)nullptr);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) xr[mt][nt] = xo[mt][nt];
  }
  float4 x1k[2][4];  // layer 1's input rows stay in registers for its recompute (they never go to HBM)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) x1k[mt][nt] = xr[mt][nt];
  {
    const InfLayer& w = fst.l[1].n[0];
    __syncthreads();  // every wave is done with layer 0's weights
    {
      const WpsPrm pp = wps_prm_of(w);
      wps_stage<T, LDSW>(w.win, &pp, wl, prm, tid);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    float4 xo[2][4] WPS_Z;
    wps_layer_fwd<T, LDSW, false, false, VIS>(w, LDSW ? wl : reinterpret_cast<const T*>(w.win), prm, xr, lane, ok, row0, srow, xo,
                                              (const WpsOut*)nullptr, E0, E1, (WpsKeep<T>*)nullptr);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) xr[mt][nt] = xo[mt][nt];
  }
  // ======================================================================================================== heads forward
  // (wps_layer_fwd_kernel's HEAD part; the post-ReLU activations stay in registers as the backward's masks: store_h's lane ->
  // element map is the one the heads' backward reads its masks with)
  float4 mk0[4], mk1[4];
  {
    const InfHead& h = fhd.n[0];
    float* pooled = reinterpret_cast<float*>(smem);                    // [16][LDP] fp32 (rows >= ns: zeros)
    T* h1 = reinterpret_cast<T*>(pooled + 16 * LF::LDP);               // [16][LDF]
    T* h2 = h1 + 16 * LF::LDF;
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
    for (int i = tid; i < 16 * LF::LDP; i += 256) pooled[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float4 m = fr == 0 ? xr[1][nt] : xr[0][nt];
      const float4 mv = float4{rowsum16(m.x) * (1.f / 16.f), rowsum16(m.y) * (1.f / 16.f), rowsum16(m.z) * (1.f / 16.f),
                               rowsum16(m.w) * (1.f / 16.f)};
      if (fr == 0 && live) {
        const float4 t0 = xr[0][nt];
        *reinterpret_cast<float4*>(pooled + wave * LF::LDP + nt * 16 + qr) = t0;
        *reinterpret_cast<float4*>(pooled + wave * LF::LDP + TD + nt * 16 + qr) = mv;
        if (h.s_pooled != nullptr) {
          *reinterpret_cast<float4*>(h.s_pooled + (int64_t)smp * 128 + nt * 16 + qr) = t0;
          *reinterpret_cast<float4*>(h.s_pooled + (int64_t)smp * 128 + TD + nt * 16 + qr) = mv;
        }
      }
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto store_h = [&](T* dst, const float4 (&bias)[4], float* save, float4 (&mk)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float4 bb = bias[j];
        const float v0 = fmaxf(acc[0][j][0] + bb.x, 0.f), v1 = fmaxf(acc[0][j][1] + bb.y, 0.f);
        const float v2 = fmaxf(acc[0][j][2] + bb.z, 0.f), v3 = fmaxf(acc[0][j][3] + bb.w, 0.f);
        mk[j] = float4{v0, v1, v2, v3};
        st4(dst + fr * LF::LDF + n4, v0, v1, v2, v3);
        if (fr < ns) st4(save + (int64_t)(s0 + fr) * 256 + n4, v0, v1, v2, v3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 4>(acc, pooled, LF::LDP, (const T*)h.w0, 128, nt4, lane, ring0);
    GemmRing<T, 4, 8> ring1 = gemm_prefetch<T, 4, 8>((const T*)h.w1, 256, nt4, lane);
    store_h(h1, hb0, h.s_h0, mk0);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, h1, LF::LDF, (const T*)h.w1, 256, nt4, lane, ring1);
    store_h(h2, hb1, h.s_h1, mk1);
    __syncthreads();
    if (wave == 0) {  // last linear: one 16-column tile
      const int nt0[1] = {0};
      f32x4 a1[1][1];
      zero_acc(a1);
      block_gemm<T, 1, 1, 8>(a1, h2, LF::LDF, (const T*)h.w2, 256, nt0, lane);
      float ov[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = qr + r;
        ov[r] = c < h.nout ? a1[0][0][r] + hb2[r] : 0.f;
        if (fr < ns) h.out[(int64_t)(s0 + fr) * OUT_LD + c] = ov[r];
      }
      *reinterpret_cast<float4*>(so + fr * 16 + qr) = float4{ov[0], ov[1], ov[2], ov[3]};
    }
  }
  // ======================================================================================================== the block's loss rows
  // rings of the heads' backward: requested before the rows are computed (they arrive meanwhile)
  const int nt2[2] = {wave * 2, wave * 2 + 1};
  GemmRing<T, 4, 2> ring2 = gemm_prefetch<T, 4, 2>((const T*)hd.w2t, 64, nt4, lane);
  GemmRing<T, 4, 8> ringb1 = gemm_prefetch<T, 4, 8>((const T*)hd.w1t, 256, nt4, lane);
  float* dt = reinterpret_cast<float*>(smem);                 // [16][LDX]: d(out) rows, zero padded to 64 columns
  T* dh1 = reinterpret_cast<T*>(dt + 16 * LB::LDX);           // [16][LDF]
  T* dh0 = dh1 + 16 * LB::LDF;
  float* dpool = reinterpret_cast<float*>(dh0 + 16 * LB::LDF);  // [16][LDP]
  __syncthreads();  // the head outputs are in `so`; the forward heads' scratch is dead
  for (int i = tid; i < 16 * LB::LDX; i += 256) dt[i] = 0.f;
  float drow[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) drow[a] = 0.f;
  if (tid < ns) {
    const int i = s0 + tid, slot = lo.rowidx ? lo.rowidx[i] : i;
    double* rc = rec + tid * FB_PART;
#pragma unroll
    for (int k = 0; k < FB_PART; ++k) rc[k] = 0.0;
    int sat = 0;
    if (lo.actor) {
      const ActorArgs& p = lo.aa;
      const ActorDims D = actor_dims(p);
      float mu[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) mu[a] = so[tid * 16 + a];
      const ActorRow o = fb_actor_row(p, D, mu, slot, p.st[ST_ADV_MEAN], p.st[ST_ADV_STD]);
      actor_dmean_row(o, p.gscale, drow, sat);
      rc[FBP_LP] = o.lp; rc[FBP_LP2] = (double)o.lp * o.lp; rc[FBP_SUR] = o.sur;
      rc[FBP_LPMAX] = o.lp; rc[FBP_LPMIN] = o.lp; rc[FBP_RMAX] = o.ratio; rc[FBP_RMIN] = o.ratio;
#pragma unroll
      for (int a = 0; a < 8; ++a) rc[FBP_DL + a] = a < p.A ? o.dlp * (o.z2[a] - 1.f) : 0.f;
    } else {
      float l, gq;
      critic_row(so[tid * 16], lo.ret[slot], lo.clipped ? lo.oldv[slot] : 0.f, lo.clipped, lo.clip, lo.inv_n, l, gq);
      drow[0] = grad_out(gq, lo.gscale, sat);
      rc[FBP_VF] = l;
    }
    rc[FBP_SAT] = (double)sat;
    float4* dg = reinterpret_cast<float4*>(lo.dout + (int64_t)i * OUT_LD);
    dg[0] = float4{drow[0], drow[1], drow[2], drow[3]};
    dg[1] = float4{drow[4], drow[5], drow[6], drow[7]};
    dg[2] = float4{0.f, 0.f, 0.f, 0.f};
    dg[3] = float4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();  // dt is zero everywhere, the rows' records are written
  if (tid < ns) {
    *reinterpret_cast<float4*>(dt + tid * LB::LDX) = float4{drow[0], drow[1], drow[2], drow[3]};
    *reinterpret_cast<float4*>(dt + tid * LB::LDX + 4) = float4{drow[4], drow[5], drow[6], drow[7]};
  }
  if (tid >= 64 && tid < 64 + FB_PART) {  // the block's partial record: its rows in a fixed order (wave 1: off wave 0's chain)
    const int k = tid - 64;
    double s = rec[k];
    for (int r = 1; r < ns; ++r) {
      const double v = rec[r * FB_PART + k];
      s = (k == FBP_LPMAX || k == FBP_RMAX) ? fmax(s, v) : (k == FBP_LPMIN || k == FBP_RMIN) ? fmin(s, v) : s + v;
    }
    lo.part[(int64_t)blockIdx.x * FB_PART + k] = s;
  }
  // ======================================================================================================== heads backward
  float4 dy[2][4];
  float4 upw[4];
  wps_unpool_weights<VIS>((const float*)nullptr, lane, ok, upw);  // (mean pooling: 1/16)
  {
    GemmRing<T, 2, 8> ringb0 = gemm_prefetch<T, 2, 8>((const T*)hd.w0t, 256, nt2, lane);
    __syncthreads();  // the d(out) rows are in dt
    f32x4 acc[1][4];
    auto masked = [&](const float4 (&m)[4], T* dst, float* save) {  // ReLU mask from the forward's activation, rows < ns
      const bool okr = fr < ns;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[0][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[0][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[0][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[0][j][3] : 0.f;
        st4(dst + fr * LB::LDF + n4, d0, d1, d2, d3);
        if (okr) st4(save + (int64_t)(s0 + fr) * 256 + n4, d0, d1, d2, d3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 2>(acc, dt, LB::LDX, (const T*)hd.w2t, 64, nt4, lane, ring2);
    masked(mk1, dh1, hd.o_dh1);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh1, LB::LDF, (const T*)hd.w1t, 256, nt4, lane, ringb1);
    masked(mk0, dh0, hd.o_dh0);
    __syncthreads();
    {
      f32x4 a2[1][2];
      zero_acc(a2);
      block_gemm<T, 1, 2, 8>(a2, dh0, LB::LDF, (const T*)hd.w0t, 256, nt2, lane, ringb0);
#pragma unroll
      for (int j = 0; j < 2; ++j) st4(dpool + fr * LB::LDP + nt2[j] * 16 + qr, a2[0][j][0], a2[0][j][1], a2[0][j][2], a2[0][j][3]);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bool tok0 = mt == 0 && fr == 0;
        const float4 v = *reinterpret_cast<const float4*>(dpool + wave * LB::LDP + (tok0 ? 0 : TD) + nt * 16 + qr);
        const float4 sc = tok0 ? float4{1.f, 1.f, 1.f, 1.f} : upw[nt];
        dy[mt][nt] = ok[mt] ? float4{v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w} : float4{0.f, 0.f, 0.f, 0.f};
      }
  }
  // ======================================================================================================== layers backward
  auto ln_partials = [&](const WpsBwdLayer& w) {  // the block's LayerNorm parameter-gradient partials, waves summed in a fixed order
    const int k = tid >> 6, cidx = tid & 63;
    float sacc = 0.f;
#pragma unroll
    for (int wv = 0; wv < WPS_WPB; ++wv) sacc += red[(wv * 4 + k) * TD + cidx];
    float* dst = k == 0 ? w.gp2 : k == 1 ? w.bp2 : k == 2 ? w.gp1 : w.bp1;
    dst[(int64_t)blockIdx.x * TD + cidx] = sacc;
  };
  auto store_dx = [&](const WpsBwdLayer& w) {
    if (w.o_dx != nullptr) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        if (ok[mt])
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<float4*>(w.o_dx + (row0 + ROFF + mt * 16 + fr) * TD + nt * 16 + qr) = dy[mt][nt];
    }
  };
  {  // ---- layer 1: recomputed from its input rows (registers), then walked back
    const WpsBwdLayer& w = wb1;
    __syncthreads();  // the heads' scratch is dead
    {
      {
        const WpsPrm pp = WpsPrm{w.bin, w.bo, w.b1, w.b2, w.g1, w.be1, w.g2, w.be2};
        wps_stage<T, LDSW>(w.w, &pp, wl, prm, tid);
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      InfLayer none;
      none.win = none.wo = none.w1 = none.w2 = nullptr;
      none.bin = none.bo = none.b1 = none.b2 = none.g1 = none.be1 = none.g2 = none.be2 = nullptr;
      none.xin = nullptr; none.xout = nullptr;
      none.s_qkv = none.s_P = none.s_xh1 = none.s_rs1 = none.s_xh2 = none.s_rs2 = nullptr;
      none.s_xin = none.s_ctx = none.s_x1 = none.s_f = nullptr;
      float4 xo[2][4] WPS_Z;
      wps_layer_fwd<T, LDSW, true, false, VIS>(none, LDSW ? wl : reinterpret_cast<const T*>(w.w), prm, x1k, lane, ok, row0, srow, xo, &wo1, E0, E1, &K);
      __syncthreads();  // every wave is done with the forward weights
    }
    wps_stage<T, LDSW>(w.wt, (const WpsPrm*)nullptr, wl, prm, tid);  // (prm holds layer 1's parameters)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    wps_layer_bwd<T, LDSW, false, NMT>(w, LDSW ? wl : reinterpret_cast<const T*>(w.wt), prm, K, dy, lane, ok, row0, wo1, E0, E1, red + wave * 4 * TD);
    __syncthreads();
    ln_partials(w);
    store_dx(w);
  }
  {  // ---- layer 0: recompute from its input rows, then walk back (wps_layer_bwd_kernel's loop body)
    const WpsBwdLayer& w = stk.l[1];
    const WpsOut wo = wps_out<T>(reinterpret_cast<T*>(w.wg) + srow * WPS_WG_STRIDE, reinterpret_cast<T*>(w.tk) + srow * WPS_TK_ELEMS, live);
    __syncthreads();  // layer 1's transposed weights are dead
    {
      const WpsPrm pp = WpsPrm{w.bin, w.bo, w.b1, w.b2, w.g1, w.be1, w.g2, w.be2};
      wps_stage<T, LDSW>(w.w, &pp, wl, prm, tid);
    }
    wps_load_rows<VIS>(w.xin + row0 * TD, lane, ok, xr);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
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
    __syncthreads();  // every wave is done with the forward weights
    wps_stage<T, LDSW>(w.wt, (const WpsPrm*)nullptr, wl, prm, tid);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    wps_layer_bwd<T, LDSW, false, NMT>(w, LDSW ? wl : reinterpret_cast<const T*>(w.wt), prm, K, dy, lane, ok, row0, wo, E0, E1, red + wave * 4 * TD);
    __syncthreads();
    ln_partials(w);
    store_dx(w);
  }
  // ======================================================================================================== TAIL (as wps_layer_bwd_kernel)
  {
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
        const int patch = mt * 16 + fr - 1;  // the depth patch of this row's token (token t = patch + 1)
        const bool okt = ok[mt] && patch >= 0;
        const int64_t o = okt ? ((int64_t)smp * 16 + patch) * TD + nt * 16 + qr : 0;
        const float4 m = *reinterpret_cast<const float4*>(tl.s_c3 + o);
        if (okt)
          *reinterpret_cast<float4*>(tl.o_dc3 + o) = float4{m.x > 0.f ? acc[mt][0] : 0.f, m.y > 0.f ? acc[mt][1] : 0.f,
                                                            m.z > 0.f ? acc[mt][2] : 0.f, m.w > 0.f ? acc[mt][3] : 0.f};
      }
    }
    BLK_LOG_END();
    if constexpr (!TOK0_IN) return;
    // token 0: (dx_in o [x0 > 0]) -> state_projector' -> [e1 > 0] -> dhc -> fc2' -> [e0 > 0] -> de0, cooperatively (4 rows)
    float* dtt = reinterpret_cast<float*>(smem);                 // [16][LDX]
    T* dh = reinterpret_cast<T*>(dtt + 16 * LB::LDX);            // [16][LDF]
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
    for (int i = tid; i < 16 * LB::LDX; i += 256) dtt[i] = 0.f;
    __syncthreads();
    if (live && fr == 0) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 x = xr[0][nt], d = dy[0][nt];
        *reinterpret_cast<float4*>(dtt + wave * LB::LDX + nt * 16 + qr) =
            float4{x.x > 0.f ? d.x : 0.f, x.y > 0.f ? d.y : 0.f, x.z > 0.f ? d.z : 0.f, x.w > 0.f ? d.w : 0.f};
      }
    }
    __syncthreads();
    f32x4 acc[1][4];
    auto masked = [&](const float4 (&m)[4], T* dst, float* save) {
      const bool okr = fr < ns;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = nt4[j] * 16 + qr;
        const float d0 = m[j].x > 0.f ? acc[0][j][0] : 0.f, d1 = m[j].y > 0.f ? acc[0][j][1] : 0.f;
        const float d2 = m[j].z > 0.f ? acc[0][j][2] : 0.f, d3 = m[j].w > 0.f ? acc[0][j][3] : 0.f;
        if (dst != nullptr) st4(dst + fr * LB::LDF + n4, d0, d1, d2, d3);
        if (okr) st4(save + (int64_t)(s0 + fr) * 256 + n4, d0, d1, d2, d3);
      }
    };
    zero_acc(acc);
    block_gemm<T, 1, 4, 2>(acc, dtt, LB::LDX, (const T*)tl.wpt, 64, nt4, lane, ring_pr);
    GemmRing<T, 4, 8> ring_f2 = gemm_prefetch<T, 4, 8>((const T*)tl.wf2t, 256, nt4, lane);
    masked(tm_e1, dh, tl.o_dhc);
    __syncthreads();
    zero_acc(acc);
    block_gemm<T, 1, 4, 8>(acc, dh, LB::LDF, (const T*)tl.wf2t, 256, nt4, lane, ring_f2);
    masked(tm_e0, (T*)nullptr, tl.o_de0);
  }
}

// Finishes the statistics the fused launch left as per-block partials: one block, a fixed summation order, the expressions of
// critic_loss_body / actor_loss_body's closing threads (elem.h). Thread t takes block t's record (one contiguous 192-byte read;
// blocks t + 256, ... behind it), the records meet in LDS and thread k adds column k over the 256 rows in order: every load of the
// launch is in flight at once (a loop over the blocks per column was 23 us of dependent L2 round trips). Dynamic LDS: FbFinishLds.
struct FbFinishLds { static constexpr int LD = FB_PART + 1; static constexpr size_t bytes = (size_t)256 * LD * 8; };
__global__ __launch_bounds__(256) void fb_loss_finish_kernel(FbLoss lo, int nblk, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
  double* rows = reinterpret_cast<double*>(smem_f);  // [256][LD]
  __shared__ double tot[FB_PART];
  const int tid = threadIdx.x;
  auto is_max = [](int k) { return k == FBP_LPMAX || k == FBP_RMAX; };
  auto is_min = [](int k) { return k == FBP_LPMIN || k == FBP_RMIN; };
  {
    double acc[FB_PART];
#pragma unroll
    for (int k = 0; k < FB_PART; ++k) acc[k] = is_max(k) ? -INFINITY : is_min(k) ? INFINITY : 0.0;
    for (int b = tid; b < nblk; b += 256) {
      const double2* src = reinterpret_cast<const double2*>(lo.part + (int64_t)b * FB_PART);
      double v[FB_PART];
#pragma unroll
      for (int q = 0; q < FB_PART / 2; ++q) { const double2 t = src[q]; v[2 * q] = t.x; v[2 * q + 1] = t.y; }
#pragma unroll
      for (int k = 0; k < FB_PART; ++k) acc[k] = is_max(k) ? fmax(acc[k], v[k]) : is_min(k) ? fmin(acc[k], v[k]) : acc[k] + v[k];
    }
#pragma unroll
    for (int k = 0; k < FB_PART; ++k) rows[tid * FbFinishLds::LD + k] = acc[k];
  }
  __syncthreads();
  // column k over the 256 rows: 8 runs of 32 rows (thread (run, k); the 32 LDS reads of a run are independent: all in flight),
  // then the 8 runs in order
  __shared__ double run[8][32];
  {
    const int k = tid & 31, r8 = tid >> 5;
    if (k < FB_PART) {
      double v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = rows[(r8 * 32 + j) * FbFinishLds::LD + k];
      double sacc = v[0];
#pragma unroll
      for (int j = 1; j < 32; ++j) sacc = is_max(k) ? fmax(sacc, v[j]) : is_min(k) ? fmin(sacc, v[j]) : sacc + v[j];
      run[r8][k] = sacc;
    }
  }
  __syncthreads();
  if (tid < FB_PART) {
    const int k = tid;
    double sacc = run[0][k];
#pragma unroll
    for (int r = 1; r < 8; ++r) sacc = is_max(k) ? fmax(sacc, run[r][k]) : is_min(k) ? fmin(sacc, run[r][k]) : sacc + run[r][k];
    tot[k] = sacc;
  }
  __syncthreads();
  float* st = lo.st;
  if (!lo.actor) {
    if (tid == 0) {
      st[ST_VF_LOSS] = (float)(tot[FBP_VF] * (double)lo.inv_n);
      if (lo.gscale != 1.f) st[ST_F16_SAT] += (float)tot[FBP_SAT];
    }
    return;
  }
  const ActorArgs& p = lo.aa;
  const int A = p.A;
  if (tid < A) {  // d(loss)/d(log sigma): the rows' sum + the entropy term; zero where the clamp is active (actor_loss_body)
    const int a = tid;
    const float raw = p.logstd[a];
    float gsum = (float)tot[FBP_DL + a];
    gsum += -p.ent_coef * p.inv_n * (float)n;
    p.dlogstd[a] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? gsum : 0.f;
  }
  if (tid == 0) {
    const ActorDims D = actor_dims(p);
    const double lpm = tot[FBP_LP] / n;
    st[ST_PI_LOSS] = (float)(-(tot[FBP_SUR] / n) - (double)p.ent_coef * D.ent);
    st[ST_LP_MEAN] = (float)lpm;
    st[ST_LP_STD] = (float)sqrt(fmax(0.0, (tot[FBP_LP2] - n * lpm * lpm) / (double)(n - 1)));
    st[ST_LP_MAX] = (float)tot[FBP_LPMAX]; st[ST_LP_MIN] = (float)tot[FBP_LPMIN];
    st[ST_RATIO_MAX] = (float)tot[FBP_RMAX]; st[ST_RATIO_MIN] = (float)tot[FBP_RMIN];
    if (p.gscale != 1.f) st[ST_F16_SAT] += (float)tot[FBP_SAT];
    double m = 0.0; float mxl = -INFINITY, mnl = INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < A) { m += D.ls[a]; mxl = fmaxf(mxl, D.ls[a]); mnl = fminf(mnl, D.ls[a]); }
    m /= A;
    double q = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < A) q += (D.ls[a] - m) * (D.ls[a] - m);
    st[ST_LS_MEAN] = (float)m;
    st[ST_LS_STD] = A > 1 ? (float)sqrt(q / (A - 1)) : NAN;
    st[ST_LS_MAX] = mxl; st[ST_LS_MIN] = mnl;
  }
}

}  // namespace v4l

#pragma clang fp contract(fast)
