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
#include "infer.h"

namespace v4l {

struct RollLin {
  const void* w[2];       // PK_FRAG pack [N/16][K/32][64] fragments, per net
  const float* b[2];
  const __bf16* x[2];     // [E][ldx]
  __bf16* y[2];           // [E][ldy] (+ column offset folded into the pointer)
  int ldx, ldy;
};
// y[:, tile*16 .. +16] = relu(x . W[tile]^T + b): one wave per (column tile, net); the tile's KS weight fragments and one
// row tile's KS activation fragments are all in flight at once (the launch is one dependent round trip + KS MFMAs long)
template <int KS>
__global__ __launch_bounds__(64) void rollout_linear_kernel(RollLin a, int E) {
  const int lane = threadIdx.x, tile = blockIdx.x, net = blockIdx.y;
  const int fr = lane & 15, g = lane >> 4;
  const bf16x8* W = reinterpret_cast<const bf16x8*>(a.w[net]) + (size_t)tile * KS * 64 + lane;
  bf16x8 wf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wf[ks] = W[ks * 64];
  const float4 bb = *reinterpret_cast<const float4*>(a.b[net] + tile * 16 + g * 4);
  const int MT = (E + 15) >> 4;
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + fr;
    const __bf16* xr = a.x[net] + (size_t)min(row, E - 1) * a.ldx + g * 8;
    bf16x8 xa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) mma_k32(acc, wf[ks], xa[ks]);
    if (row < E)
      st4(a.y[net] + (size_t)row * a.ldy + tile * 16 + g * 4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f),
          fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
  }
}

struct RollHead {
  const void *wb[2], *wo[2];   // PK_FRAG packs: [16][8][64] (256 -> 256), [1][8][64] (256 -> out, padded to 16 columns)
  const float *bb[2], *bo[2];
  const __bf16* x[2];          // [E][256]
  float* out[2];               // [E][OUT_LD] fp32 head outputs (what the general path leaves in the workspace)
  int nout[2];
};
struct RollHeadLds { static constexpr int LDH = 256 + 8; static constexpr size_t bytes = (size_t)2 * 64 * LDH * 2 + (2 * 64 * 16 + 16) * 4; };
// One block per net, 8 waves, every row of the step: Linear(256,256)+ReLU -> last Linear -> the explore / value epilogue of
// rollout_stack_kernel (same expressions and summation order), thread i = env i.
__global__ __launch_bounds__(512) void rollout_head_kernel(RollHead a, InfFinish fin, int E) {
  constexpr int LDH = RollHeadLds::LDH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* xs = reinterpret_cast<__bf16*>(smem);      // [64][LDH] input rows
  __bf16* hs = xs + 64 * LDH;                          // [64][LDH] hidden rows
  float* so = reinterpret_cast<float*>(hs + 64 * LDH); // [64][16] head outputs
  float* eps_s = so + 64 * 16;                         // [64][A <= 16] exploration noise
  float* lsd_s = eps_s + 64 * 16;                      // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, net = blockIdx.x;
  const int fr = lane & 15, g = lane >> 4;
  const int MT = (E + 15) >> 4, A = fin.A;
  const long long t_step = fin.ctl->t;
  bf16x8 wo[8], wb[2][8];
  if (wave < 4) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wo[ks] = reinterpret_cast<const bf16x8*>(a.wo[net])[ks * 64 + lane];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wb[j][ks] = reinterpret_cast<const bf16x8*>(a.wb[net])[((wave + 8 * j) * 8 + ks) * 64 + lane];
  for (int idx = tid; idx < MT * 16 * 32; idx += 512) {
    const int row = idx >> 5, c = (idx & 31) * 8;
    *reinterpret_cast<bf16x8*>(xs + row * LDH + c) = *reinterpret_cast<const bf16x8*>(a.x[net] + (size_t)min(row, E - 1) * 256 + c);
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
        mma_k32(acc, wb[j][ks], *reinterpret_cast<const bf16x8*>(xs + (mt * 16 + fr) * LDH + ks * 32 + g * 8));
      st4(hs + (mt * 16 + fr) * LDH + tile * 16 + g * 4, fmaxf(acc[0] + bb.x, 0.f), fmaxf(acc[1] + bb.y, 0.f),
          fmaxf(acc[2] + bb.z, 0.f), fmaxf(acc[3] + bb.w, 0.f));
    }
  }
  __syncthreads();
  if (wave < MT) {  // last linear: row tile `wave`, the one (padded) column tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      mma_k32(acc, wo[ks], *reinterpret_cast<const bf16x8*>(hs + (wave * 16 + fr) * LDH + ks * 32 + g * 8));
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
        const float act = fmaf(sg, eps_s[i * A + k], mu);
        fin.action[(int64_t)i * A + k] = act;
        fin.mean[(int64_t)i * A + k] = mu;
        fin.stdv[(int64_t)i * A + k] = sg;
        if (fin.acts_roll != nullptr) fin.acts_roll[(t_step * E + i) * A + k] = act;
        const float d = act - mu;
        lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
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

}  // namespace v4l
