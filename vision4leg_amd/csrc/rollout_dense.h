// Rollout-step kernels of the NatureCNN nets (ppo_nature_cnn: networks/nets.py:194-262 + base.py:345-385; its vision-only
// variant: nets.py:133-191 + base.py:304-342) behind rollout_encoder2_kernel<ENC_FUSE | ENC_FLAT>.
//
// rollout_cnn_kernel ran one block per (sample, net) and pushed the whole dense part through it as matrix-VECTOR products:
// every block streamed 1.1 MB of weights (visual projector 512 KB, head 392 KB, proprio MLP 192 KB) through one CU, and only
// row 0 of each 16-row MFMA fragment carried data: 55 us per env step at E = 16. A rollout step is a batch of E <= 64 rows,
// so here the dense layers are GEMMs over ALL E rows: a weight byte is fetched once per step (per net), and the wide layers
// are spread over 16 CUs per net, one 16-column tile each.
//   rollout_encoder2_kernel<ENC_FUSE>  E + ceil(E/32) blocks   conv stack per sample -> featv[E][1024]; proprio MLP -> cat[:, 256:512]
//   rollout_linear_kernel<32>          16 blocks               visual projector: cat[:, 0:256] = relu(featv . Wpr^T + b)
//   rollout_linear_kernel<16>          16 x 2 nets             head fc0: h0[net] = relu(cat . W0^T + b)
//   rollout_head_kernel                2 blocks                fc1 -> last linear -> sampling / value read-out -> filing
// (vision-only: encoder<ENC_FLAT>, rollout_linear_kernel<32> per net on featv, rollout_head_kernel). bf16 operands, fp32
// accumulation, k order = the packed k order of the general kernels (NHWC flatten for the layer that reads conv3).
#pragma once
#include <type_traits>
#include "infer.h"

namespace v4l {

// H below: the 16-bit operand type of the step (__bf16 | _Float16); hx8<H>: one MFMA fragment of it
template <typename H> using hx8 = typename Frag<H>::type;

template <typename H> struct RollLin {
  const void* w[2];       // PK_FRAG pack [N/16][K/32][64] fragments, per net
  const float* b[2];
  const H* x[2];     // [ceil16(E)][32*KS] in A-fragment order (act_frag_off)
  H* y[2];           // ks_out > 0: columns of a fragment-order [.][32*ks_out] operand; 0: row-major [E][ldy]
  int ks_out, ldy;
};
// y[:, tile*16 .. +16] = relu(x . W[tile]^T + b): one wave per (column tile, net); the tile's KS weight fragments and one
// row tile's KS activation fragments (each one contiguous 1 KB read) are all in flight at once
template <typename H, int KS>
__global__ __launch_bounds__(64) void rollout_linear_kernel(RollLin<H> a, int E) {
  const int lane = threadIdx.x, tile = blockIdx.x, net = blockIdx.y;
  const int fr = lane & 15, g = lane >> 4;
  ROLL_STAMP(112);
  const hx8<H>* W = reinterpret_cast<const hx8<H>*>(a.w[net]) + (size_t)tile * KS * 64 + lane;
  hx8<H> wf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wf[ks] = W[ks * 64];
  const float4 bb = *reinterpret_cast<const float4*>(a.b[net] + tile * 16 + g * 4);
  const int MT = (E + 15) >> 4;
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + fr;
    // rows >= E were never written: those lanes re-read the last real row's chunk (their results are not stored)
    const hx8<H>* X = reinterpret_cast<const hx8<H>*>(a.x[net]) + (size_t)mt * KS * 64 + g * 16 + min(fr, E - 1 - mt * 16);
    hx8<H> xa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[ks] = X[ks * 64];
#ifdef V4L_INFER_TIMING
    ROLL_STAMP(113 + 4 * mt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ROLL_STAMP(114 + 4 * mt);
#endif
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      mma_k32(acc0, wf[ks], xa[ks]);
      mma_k32(acc1, wf[ks + 1], xa[ks + 1]);
    }
    if (row < E) {
      const int n4 = tile * 16 + g * 4;
      H* dst = a.ks_out > 0 ? a.y[net] + act_frag_off(row, n4, a.ks_out) : a.y[net] + (size_t)row * a.ldy + n4;
      st4(dst, fmaxf((acc0[0] + acc1[0]) + bb.x, 0.f), fmaxf((acc0[1] + acc1[1]) + bb.y, 0.f),
          fmaxf((acc0[2] + acc1[2]) + bb.z, 0.f), fmaxf((acc0[3] + acc1[3]) + bb.w, 0.f));
    }
#ifdef V4L_INFER_TIMING
    ROLL_STAMP(115 + 4 * mt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ROLL_STAMP(116 + 4 * mt);
#endif
  }
}

template <typename H> struct RollHead {
  const void *wb[2], *wo[2];   // PK_FRAG packs: [16][8][64] (256 -> 256), [1][8][64] (256 -> out, padded to 16 columns)
  const float *bb[2], *bo[2];
  const H* x[2];          // [E][256]
  float* out[2];               // [E][OUT_LD] fp32 head outputs (what the general path leaves in the workspace)
  int nout[2];
};
struct RollHeadLds { static constexpr int LDH = 256 + 8; static constexpr size_t bytes = (size_t)2 * 64 * LDH * 2 + (2 * 64 * 16 + 16) * 4; };
// One block per net, 8 waves, every row of the step: Linear(256,256)+ReLU -> last Linear -> the explore / value epilogue of
// rollout_stack_kernel (same expressions and summation order), thread i = env i.
template <typename H>
__global__ __launch_bounds__(512) void rollout_head_kernel(RollHead<H> a, InfFinish fin, int E) {
  constexpr int LDH = RollHeadLds::LDH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  H* xs = reinterpret_cast<H*>(smem);      // [64][LDH] input rows
  H* hs = xs + 64 * LDH;                          // [64][LDH] hidden rows
  float* so = reinterpret_cast<float*>(hs + 64 * LDH); // [64][16] head outputs
  float* eps_s = so + 64 * 16;                         // [64][A <= 16] exploration noise
  float* lsd_s = eps_s + 64 * 16;                      // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, net = blockIdx.x;
  const int fr = lane & 15, g = lane >> 4;
  const int MT = (E + 15) >> 4, A = fin.A;
  const long long t_step = fin.ctl->t;
  hx8<H> wo[8], wb[2][8];
  if (wave < 4) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wo[ks] = reinterpret_cast<const hx8<H>*>(a.wo[net])[ks * 64 + lane];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wb[j][ks] = reinterpret_cast<const hx8<H>*>(a.wb[net])[((wave + 8 * j) * 8 + ks) * 64 + lane];
  for (int idx = tid; idx < MT * 16 * 32; idx += 512) {
    const int row = idx >> 5, c = (idx & 31) * 8;
    *reinterpret_cast<hx8<H>*>(xs + row * LDH + c) = *reinterpret_cast<const hx8<H>*>(a.x[net] + (size_t)min(row, E - 1) * 256 + c);
  }
  if (net == 0) {
    for (int idx = tid; idx < E * A; idx += 512) eps_s[idx] = fin.eps[idx];
    if (tid < 16) lsd_s[tid] = fin.logstd[min(tid, A - 1)];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tile = wave + 8 * j;
    const float4 bb = *reinterpret_cast<const float4*>(a.bb[net] + tile * 16 + g * 4);
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        mma_k32(acc, wb[j][ks], *reinterpret_cast<const hx8<H>*>(xs + (mt * 16 + fr) * LDH + ks * 32 + g * 8));
      st4(hs + (mt * 16 + fr) * LDH + tile * 16 + g * 4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f),
          fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
    }
  }
  __syncthreads();
  if (wave < MT) {  // last linear: row tile `wave`, the one (padded) column tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      mma_k32(acc, wo[ks], *reinterpret_cast<const hx8<H>*>(hs + (wave * 16 + fr) * LDH + ks * 32 + g * 8));
    const int row = wave * 16 + fr, nout = a.nout[net];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = g * 4 + r;
      const float v = c < nout ? acc[r] + a.bo[net][min(c, nout - 1)] : 0.f;
      so[row * 16 + c] = v;
      if (row < E) a.out[net][(int64_t)row * OUT_LD + c] = v;
    }
  }
  __syncthreads();
  if (tid < E) {  // GaussianContPolicyBase.explore (continuous_policy.py:85-125) / the collector's value read-out, env i = tid
    const int i = tid;
    if (net == 0) {
      float e = 0.f, lp = 0.f;
      for (int k = 0; k < A; ++k) {
        const float mu = so[i * 16 + k];
        const float ls = fminf(fmaxf(lsd_s[k], LOG_SIG_MIN), LOG_SIG_MAX);
        const float sg = expf(ls);
        e += 0.5f + HALF_LOG_2PI + logf(sg);
        float act = fmaf(sg, eps_s[i * A + k], mu);
        if (fin.tanh_action) act = tanhf(act);  // TanhNormal (distribution.py:61-80), as in act_finish_kernel
        fin.action[(int64_t)i * A + k] = act;
        fin.mean[(int64_t)i * A + k] = mu;
        fin.stdv[(int64_t)i * A + k] = sg;
        if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + k] = act;
        const float d = (fin.tanh_action ? tanh_pre(act) : act) - mu;
        lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
        if (fin.tanh_action) lp -= tanh_corr(act);
      }
      fin.ent[i] = e;
      if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + i] = lp;
    } else {
      const float v = so[i * 16];
      fin.value[i] = v;
      if (fin.values_roll != nullptr) fin.values_roll[t_step * E + i] = v;
    }
  }
  __syncthreads();
  if (tid == 0) {  // the last block to get here advances the step cursor: every block read it at entry
    __threadfence();
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&fin.ctl->done), 1ull);
    if (done == (unsigned long long)gridDim.x - 1) {
      fin.ctl->done = 0;
      fin.ctl->t = t_step + 1;
    }
  }
}

// ------------------------------------------------------------------------------------------ state MLP nets
// Net / GaussianContPolicyBasicBias on proprioception only (networks/nets.py:16-55, starter/ppo_state.py): shared base MLP
// (S -> 256 -> 256, ReLU) + per-net head (256 -> 256 -> 256 -> out). rollout_mlp_kernel ran one block per (sample, net) with
// row-major weights requested at the start of each of its five phases (five cold round trips) and a serial epilogue with
// global loads in its loop: 18 us per env step at E = 1. Here ONE block per net takes all E <= 64 rows as MFMA row tiles,
// wave w owns column tile w of every 256-wide layer, and the fragment-order weights are requested ahead: the first three
// layers' at entry, the fourth's once the first is done with its registers, the last one's after the second.
struct RollMlp2 {
  const void *wf1, *wf2; const float *bf1, *bf2;       // shared base (the policy's): [16][4][64], [16][8][64] fragments
  const void *w0[2], *w1[2], *w2[2]; const float *b0[2], *b1[2], *b2[2];
  float* out[2]; int nout[2];
  int S, Sp;
};
struct RollMlp2Lds {
  static constexpr int LDI = 128 + 8, LDH = 256 + 8;
  static constexpr size_t bytes = (size_t)64 * LDI * 2 + (size_t)2 * 64 * LDH * 2 + (3 * 64 * 16 + 3 * 16) * 4;
};
template <typename H>
__global__ __launch_bounds__(1024) void rollout_mlp2_kernel(const float* __restrict__ obs, int E, RollMlp2 a, InfFinish fin,
                                                            float* __restrict__ state_roll) {
  constexpr int LDI = RollMlp2Lds::LDI, LDH = RollMlp2Lds::LDH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  H* xin = reinterpret_cast<H*>(smem);       // [64][LDI] proprio rows
  H* ha = xin + 64 * LDI;                          // [64][LDH]
  H* hb = ha + 64 * LDH;                           // [64][LDH]
  float* so = reinterpret_cast<float*>(hb + 64 * LDH);  // [64][16] head outputs
  float* eps_s = so + 64 * 16;                          // [64][A]
  float* lt_s = eps_s + 64 * 16;                        // [64][16] log-prob terms
  float* sg_s = lt_s + 64 * 16;                         // [16] | lsg [16] | b2 [16]
  float* lsg_s = sg_s + 16;
  float* b2_s = lsg_s + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, net = blockIdx.x;
  const int fr = lane & 15, g = lane >> 4, MT = (E + 15) >> 4, A = fin.A;
  const long long t_step = fin.t_plus1 > 0 ? fin.t_plus1 - 1 : fin.ctl->t;
  auto frag = [&](const void* W, int ks_per_tile, int t, int ks) {
    return reinterpret_cast<const hx8<H>*>(W)[((size_t)t * ks_per_tile + ks) * 64 + lane];
  };
  // observation rows first (the first GEMM waits for them), then the weights in the order of use
  for (int idx = tid; idx < MT * 16 * 128; idx += 1024) {
    const int r = idx >> 7, c = idx & 127;
    const float x = (r < E && c < a.S) ? obs[(int64_t)r * a.S + c] : 0.f;
    xin[r * LDI + c] = (H)x;
    if (net == 0 && r < E && c < a.Sp) state_roll[((int64_t)t_step * E + r) * a.Sp + c] = x;
  }
  hx8<H> r1[4], r2[8], r3[8];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) r1[ks] = frag(a.wf1, 4, wave, ks);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) r2[ks] = frag(a.wf2, 8, wave, ks);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) r3[ks] = frag(a.w0[net], 8, wave, ks);
  const int n4 = wave * 16 + g * 4;
  const float4 bb1 = *reinterpret_cast<const float4*>(a.bf1 + n4), bb2 = *reinterpret_cast<const float4*>(a.bf2 + n4);
  const float4 bb3 = *reinterpret_cast<const float4*>(a.b0[net] + n4), bb4 = *reinterpret_cast<const float4*>(a.b1[net] + n4);
  if (net == 0) {
    for (int idx = tid; idx < E * A; idx += 1024) eps_s[idx] = fin.eps[idx];
    if (tid < 16) {
      const float ls = fminf(fmaxf(fin.logstd[min(tid, A - 1)], LOG_SIG_MIN), LOG_SIG_MAX);
      const float sg = expf(ls);
      sg_s[tid] = sg;
      lsg_s[tid] = logf(sg);
    }
  }
  if (tid < 16) b2_s[tid] = tid < a.nout[net] ? a.b2[net][tid] : 0.f;
  __syncthreads();
  // y[:, tile] = relu(x . W^T + b) for every row tile; x rows in LDS (stride ldx), weights held in registers
  auto layer = [&](const hx8<H>* w, auto ks_tag, const H* x, int ldx, const float4 bb, H* y) {
    constexpr int KS = decltype(ks_tag)::value;
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        mma_k32(acc, w[ks], *reinterpret_cast<const hx8<H>*>(x + (mt * 16 + fr) * ldx + ks * 32 + g * 8));
      st4(y + (mt * 16 + fr) * LDH + n4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f), fmaxf(acc[2] + bb.z, 0.f),
          fmaxf(acc[3] + bb.w, 0.f));
    }
  };
  layer(r1, std::integral_constant<int, 4>(), xin, LDI, bb1, ha);
  hx8<H> r4[8], r5[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) r4[ks] = frag(a.w1[net], 8, wave, ks);
  __syncthreads();
  layer(r2, std::integral_constant<int, 8>(), ha, LDH, bb2, hb);
  if (wave < 4) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) r5[ks] = frag(a.w2[net], 8, 0, ks);
  }
  __syncthreads();
  layer(r3, std::integral_constant<int, 8>(), hb, LDH, bb3, ha);
  __syncthreads();
  layer(r4, std::integral_constant<int, 8>(), ha, LDH, bb4, hb);
  __syncthreads();
  if (wave < MT) {  // last linear: row tile `wave`, the one (padded) column tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      mma_k32(acc, r5[ks], *reinterpret_cast<const hx8<H>*>(hb + (wave * 16 + fr) * LDH + ks * 32 + g * 8));
    const int row = wave * 16 + fr, nout = a.nout[net];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = g * 4 + r;
      const float v = c < nout ? acc[r] + b2_s[c] : 0.f;
      so[row * 16 + c] = v;
      if (row < E) a.out[net][(int64_t)row * OUT_LD + c] = v;
    }
  }
  __syncthreads();
  // explore / value epilogue (act_finish_kernel's expressions): a (row, k) pair per thread, then the row's thread sums the
  // log-prob terms over k = 0 .. A-1 in order
  if (net == 0) {
    for (int idx = tid; idx < E * A; idx += 1024) {
      const int i = idx / A, k = idx - i * A;
      const float mu = so[i * 16 + k], sg = sg_s[k];
      const float act = fmaf(sg, eps_s[idx], mu);
      const float d = act - mu;
      lt_s[i * 16 + k] = -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
      fin.action[idx] = act;
      fin.mean[idx] = mu;
      fin.stdv[idx] = sg;
      if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E) * A + idx] = act;
    }
    __syncthreads();
    if (tid < E) {
      float e = 0.f, lp = 0.f;
      for (int k = 0; k < A; ++k) { e += 0.5f + HALF_LOG_2PI + lsg_s[k]; lp += lt_s[tid * 16 + k]; }
      fin.ent[tid] = e;
      if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + tid] = lp;
    }
  } else if (tid < E) {
    const float v = so[tid * 16];
    fin.value[tid] = v;
    if (fin.values_roll != nullptr) fin.values_roll[t_step * E + tid] = v;
  }
  __syncthreads();
  if (fin.t_plus1 > 0) {  // (InfFinish::t_plus1)
    if (tid == 0 && net == 0) fin.ctl->t = t_step + 1;
  } else if (tid == 0) {  // the second of the two blocks advances the step cursor: both read it at entry
    __threadfence();
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&fin.ctl->done), 1ull);
    if (done == 1ull) {
      fin.ctl->done = 0;
      fin.ctl->t = t_step + 1;
    }
  }
}

// ------------------------------------------------------------------------------------------ the dense part in ONE launch
// The three launches above cost ~6.5 us each on the device (a cold start, one dependent round trip, the kernel boundary),
// three times per env step. Here the same tiles run as 32 single-wave blocks of one launch (tile = blockIdx & 15, net =
// blockIdx >> 4) that hand their results over through device-side counters: every block requests the weight fragments of
// ALL its stages at entry (they arrive while the earlier stages run), then
//   stage 0 (fuse net, the 16 blocks of net 0)  visual projector tile      -> cat (fragment order)   -> stage[0] += 1
//   stage 1  wait stage[0]; fc0 tile of this net                            -> h0[net]               -> stage[1 + net] += 1
//   stage 2  wait stage[1 + net]; fc1 tile                                   -> h1[net]               -> stage[3 + net] += 1
//   stage 3 (tile 0 only) wait stage[3 + net]; last linear + explore / value epilogue + filing; the last of the two
//           advances the step cursor and the launch sequence number.
// Counters are monotonic (target = 16 (seq + 1)), released / acquired at agent scope (the blocks sit on different XCDs:
// tools/probe/flag_hop.hip measured 1.7 - 2.5 us per hand-over). 32 one-wave blocks are always co-resident, producers have
// the lower block indices, and every spin is bounded: a lost hand-over must never hang the GPU. A wait that runs out sets
// ctl->err (sticky) and the block carries on — but nothing computed from a missed hand-over is ever used silently: the policy's
// finishing block turns the step's actions into NaN whenever ctl->err is set (the collector's "non-finite action" check,
// collector/on_policy.py:102-107, then stops the epoch), and v4l_actor_check reports and clears the flag for the host.
// seq: eager launches get it from the host as an argument (InfFinish::seq_plus1) — a late-dispatched block can therefore not
// pick up the NEXT launch's target after the finishing block has bumped ctl->seq; graph replays read ctl->seq, which in that
// mode only moves once BOTH finishing blocks are done — each of them has waited for all 16 tiles of its net, so every block
// of the launch has passed its entry by then.
template <typename H> struct RollDense {
  const void* wpr; const float* bpr;                     // fuse net: visual projector (the shared encoder's = the policy's)
  const void *w0[2], *w1[2], *w2[2];                     // PK_FRAG packs per net
  const float *b0[2], *b1[2], *b2[2];
  const H* featv;                                   // [ceil16(E)][1024] fragment order (act_frag_off, KS 32)
  H* cat;                                           // fuse net: [ceil16(E)][512] (KS 16)
  H *h0[2], *h1[2];                                 // [ceil16(E)][256] (KS 8)
  float* out[2]; int nout[2];
};
__device__ __forceinline__ void dense_wait(ActCtl* ctl, int idx, unsigned target) {
  if ((threadIdx.x & 63) == 0) {
    long long spin = 0;
    // relaxed polls (an acquire per poll invalidates caches 32 waves x every iteration), ONE acquire fence after the loop
    while ((int)(__hip_atomic_load(&ctl->stage[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spin > (1ll << 22)) { __hip_atomic_store(&ctl->err, 1u + (unsigned)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the whole wave's later loads see what the producers released
}
__device__ __forceinline__ void dense_signal(ActCtl* ctl, int idx) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this wave's stores are out before the count moves
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&ctl->stage[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one 16-column tile of relu(x . W^T + b) for every row tile, x and y in fragment order
template <typename H, int KS>
__device__ __forceinline__ void dense_x(hx8<H> (&xa)[KS], const H* x, int mt, int E, int lane) {
  const int fr = lane & 15, g = lane >> 4;
  const hx8<H>* X = reinterpret_cast<const hx8<H>*>(x) + (size_t)mt * KS * 64 + g * 16 + min(fr, E - 1 - mt * 16);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) xa[ks] = X[ks * 64];
}
// PRE: the caller already requested row tile 0's fragments into xa (ahead of the weights: loads return in order)
template <typename H, int KS, bool PRE = false>
__device__ __forceinline__ void dense_tile(const hx8<H> (&wf)[KS], const H* x, const float4 bb, H* y, int ks_out,
                                           int tile, int E, int lane, hx8<H> (&xa)[KS]) {
  const int fr = lane & 15, g = lane >> 4, MT = (E + 15) >> 4;
  for (int mt = 0; mt < MT; ++mt) {
    if (!PRE || mt > 0) dense_x<H, KS>(xa, x, mt, E, lane);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      mma_k32(acc0, wf[ks], xa[ks]);
      mma_k32(acc1, wf[ks + 1], xa[ks + 1]);
    }
    const int row = mt * 16 + fr;
    if (row < E)
      st4(y + act_frag_off(row, tile * 16 + g * 4, ks_out), fmaxf((acc0[0] + acc1[0]) + bb.x, 0.f),
          fmaxf((acc0[1] + acc1[1]) + bb.y, 0.f), fmaxf((acc0[2] + acc1[2]) + bb.z, 0.f), fmaxf((acc0[3] + acc1[3]) + bb.w, 0.f));
  }
}
template <typename H, bool FUSE>
__global__ __launch_bounds__(64) void rollout_dense_kernel(RollDense<H> a, InfFinish fin, int E) {
  constexpr int KS0 = FUSE ? 16 : 32;
  __shared__ float so[16 * 16];
  __shared__ float eps_s[64 * 16];
  __shared__ float sg_s[16], lsg_s[16], lt_s[16 * 16];
  const int lane = threadIdx.x, tile = blockIdx.x & 15, net = blockIdx.x >> 4;
  const int fr = lane & 15, g = lane >> 4, MT = (E + 15) >> 4;
  ActCtl* ctl = fin.ctl;
  ROLL_STAMP(100);
  const unsigned seq = fin.seq_plus1 > 0 ? fin.seq_plus1 - 1u : ctl->seq;
  const unsigned target = 16u * (seq + 1u);
  const long long t_step = fin.t_plus1 > 0 ? fin.t_plus1 - 1 : ctl->t;
  auto frags = [&](const void* W, int ks_per_tile, int t, int ks) {
    return reinterpret_cast<const hx8<H>*>(W)[((size_t)t * ks_per_tile + ks) * 64 + lane];
  };
  // request order = arrival order: the first stage's activation fragments (the encoder launch left them), its weights,
  // then the later stages' weights and the small operands
  hx8<H> wp[FUSE ? 32 : 1], w0[KS0], w1[8], w2[8], xv[32], xs0[KS0], xs1[8];
  const bool first = !FUSE || net == 0;  // this block's first stage reads featv
  if (first) dense_x<H, 32>(xv, a.featv, 0, E, lane);
  if constexpr (FUSE) {
    if (net == 0) {
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) wp[ks] = frags(a.wpr, 32, tile, ks);
    }
  }
#pragma unroll
  for (int ks = 0; ks < KS0; ++ks) w0[ks] = frags(a.w0[net], KS0, tile, ks);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) w1[ks] = frags(a.w1[net], 8, tile, ks);
  if (tile == 0) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w2[ks] = frags(a.w2[net], 8, 0, ks);
  }
  const int n4 = tile * 16 + g * 4;
  const float4 bpv = FUSE ? *reinterpret_cast<const float4*>(a.bpr + n4) : float4{0.f, 0.f, 0.f, 0.f};
  const float4 b0v = *reinterpret_cast<const float4*>(a.b0[net] + n4), b1v = *reinterpret_cast<const float4*>(a.b1[net] + n4);
  if (tile == 0 && net == 0) {  // explore operands of the policy's finishing block: E x A normals, A log-stds
    const int A = fin.A;
    for (int idx = lane; idx < E * A; idx += 64) eps_s[idx] = fin.eps[idx];
    if (lane < 16) {
      const float ls = fminf(fmaxf(fin.logstd[min(lane, A - 1)], LOG_SIG_MIN), LOG_SIG_MAX);
      const float sg = expf(ls);
      sg_s[lane] = sg;
      lsg_s[lane] = logf(sg);
    }
  }
  ROLL_STAMP(101);
  if constexpr (FUSE) {
    if (net == 0) {
      dense_tile<H, 32, true>(wp, a.featv, bpv, a.cat, 16, tile, E, lane, xv);
      ROLL_STAMP(102);
      dense_signal(ctl, 0);
    }
    ROLL_STAMP(103);
    dense_wait(ctl, 0, target);
  }
  ROLL_STAMP(104);
  if constexpr (FUSE) dense_tile<H, KS0>(w0, a.cat, b0v, a.h0[net], 8, tile, E, lane, xs0);
  else dense_tile<H, 32, true>(w0, a.featv, b0v, a.h0[net], 8, tile, E, lane, xv);
  ROLL_STAMP(105);
  dense_signal(ctl, 1 + net);
  ROLL_STAMP(106);
  dense_wait(ctl, 1 + net, target);
  ROLL_STAMP(107);
  dense_tile<H, 8>(w1, a.h0[net], b1v, a.h1[net], 8, tile, E, lane, xs1);
  ROLL_STAMP(108);
  dense_signal(ctl, 3 + net);
  if (tile != 0) return;
  ROLL_STAMP(109);
  dense_wait(ctl, 3 + net, target);
  ROLL_STAMP(110);
  const int nout = a.nout[net], A = fin.A;
  float b2v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) b2v[r] = a.b2[net][min(g * 4 + r, nout - 1)];
  hx8<H> xl[4][8];  // every row tile's fragments of h1 in one round trip (E <= 64: at most 4 tiles)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
    if (mt < MT) dense_x<H, 8>(xl[mt], a.h1[net], mt, E, lane);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {  // last linear (one padded column tile) + the epilogue of rollout_head_kernel, 16 rows at a time
    if (mt >= MT) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) mma_k32(acc, w2[ks], xl[mt][ks]);
    const int row = mt * 16 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = g * 4 + r;
      const float v = c < nout ? acc[r] + b2v[r] : 0.f;
      so[fr * 16 + c] = v;
      if (row < E) a.out[net][(int64_t)row * OUT_LD + c] = v;
    }
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order
    // GaussianContPolicyBase.explore / the value read-out with act_finish_kernel's expressions; the per-dimension terms are
    // evaluated (row, k) pair per lane, the log-prob is then summed over k = 0 .. A-1 in order by the row's lane
    if (net == 0) {
      // a hand-over of this or an earlier launch timed out (any block, any net): the step's inputs may be stale -> no action
      const bool lost = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      for (int idx = lane; idx < 16 * A; idx += 64) {
        const int r = idx / A, k = idx - r * A, i = mt * 16 + r;
        const float mu = so[r * 16 + k], sg = sg_s[k];
        const float act = lost ? __builtin_nanf("") : fmaf(sg, eps_s[min(i, E - 1) * A + k], mu);
        const float d = act - mu;
        lt_s[r * 16 + k] = -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
        if (i < E) {
          fin.action[(int64_t)i * A + k] = act;
          fin.mean[(int64_t)i * A + k] = mu;
          fin.stdv[(int64_t)i * A + k] = sg;
          if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + k] = act;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int i = mt * 16 + lane;
      if (lane < 16 && i < E) {
        float e = 0.f, lp = 0.f;
        for (int k = 0; k < A; ++k) { e += 0.5f + HALF_LOG_2PI + lsg_s[k]; lp += lt_s[lane * 16 + k]; }
        fin.ent[i] = e;
        if (fin.logp_roll != nullptr) fin.logp_roll[t_step * E + i] = lp;
      }
    } else {
      const int i = mt * 16 + lane;
      if (lane < 16 && i < E) {
        const float v = so[lane * 16];
        fin.value[i] = v;
        if (fin.values_roll != nullptr) fin.values_roll[t_step * E + i] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  ROLL_STAMP(111);
  if (fin.t_plus1 > 0 && fin.seq_plus1 > 0) {  // eager: nobody in this launch reads the cursor or the sequence number
    if (lane == 0 && net == 0) { ctl->t = t_step + 1; ctl->seq = seq + 1u; }
  } else if (lane == 0) {  // the second of the two finishing blocks closes the step
    __threadfence();
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->done), 1ull);
    if (done == 1ull) {
      ctl->done = 0;
      ctl->t = t_step + 1;
      ctl->seq = ctl->seq + 1u;
    }
  }
}

}  // namespace v4l
