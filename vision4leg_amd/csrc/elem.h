// Non-GEMM kernels of the PPO hot path: observation ingest, 17-token single-head attention (fwd/bwd),
// residual + LayerNorm (fwd/bwd), token pooling, Gaussian-policy head, PPO losses, advantage statistics,
// global-norm + Adam, weight packing and GAE. All fp32 (GAE fp64) on the VALU; one wave = 64 lanes.
#pragma once
#include "common.h"
#include "../../include/v4l_hip.h"

namespace v4l {

constexpr int NTOK = 17;   // 1 proprio token + 4x4 depth patches (torchrl/networks/base.py:544,617-622)
constexpr int TD = 64;     // token_dim (torchrl/networks/base.py:504)
constexpr int OUT_LD = 16; // row stride of head outputs / their grads (A=6 or 1, zero padded)
constexpr float LOG_SIG_MAX = 2.f, LOG_SIG_MIN = -5.f;  // torchrl/policies/continuous_policy.py:8-9
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

// --------------------------------------------------------------------------------- ingest
// Splits reference observation rows [n][S + C*H*W] (torchrl/networks/nets.py:997-1000) into the two
// device-resident arrays the kernels read: proprio [slot][Sp] fp32 (zero padded to Sp) and the depth
// stack [slot][C*H*W] in the contraction operand type. One block per row.
template <typename ImgT>
__global__ __launch_bounds__(256) void ingest_kernel(const float* __restrict__ obs, int n, int S, int Sp, int img_elems,
                                                     float* __restrict__ state, ImgT* __restrict__ image, int64_t slot0,
                                                     const long long* __restrict__ step_counter) {
  const int r = blockIdx.x;
  if (r >= n) return;
  if (step_counter != nullptr) slot0 = (int64_t)(*step_counter) * n;  // device-side rollout cursor (actor graph)
  const float* src = obs + (int64_t)r * (S + img_elems);
  float* sdst = state + (slot0 + r) * (int64_t)Sp;
  ImgT* idst = image + (slot0 + r) * (int64_t)img_elems;
  for (int i = threadIdx.x; i < Sp; i += 256) sdst[i] = i < S ? src[i] : 0.f;
  const float* isrc = src + S;
  for (int i = threadIdx.x; i < img_elems; i += 256) idst[i] = Op<ImgT>::from_f32(isrc[i]);
}

// --------------------------------------------------------------------------------- attention
// nn.MultiheadAttention(64, 1 head) core on packed qkv rows [n*17][192] (q|k|v), per sample:
//   P = softmax(q k^T / sqrt(64)),  ctx = P v.   NT tokens per sample: 17 (LocoTransformer) or 16 (vision-only Transformer).          (torch nn/functional.py multi_head_attention_forward;
// built by the reference at torchrl/networks/nets.py:948-955). One wave per sample, 4 samples per block.
constexpr int ATT_LD = TD + 1;  // +1 float: conflict-free both for lane=d and lane=(i,j) access
constexpr int ATT_PLD = 20;

// x as a contraction of compute mode `bf` sees it (0: fp32, 1: bf16, 2: f16 — ModeOf<T>::value)
__device__ __forceinline__ float rbf(float x, int bf) { return bf == 1 ? (float)(__bf16)x : bf == 2 ? (float)(_Float16)x : x; }
template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int n, float* __restrict__ P,
                                                       float* __restrict__ ctx, int bf) {
  // bf: bf16 compute mode — the operands of the two products (q, k; P, v) are rounded to bf16 like the MFMA tiles of the
  // fused kernels round theirs (attn_tile, csrc/infer.h); accumulation, scale and softmax stay fp32
  // one sample per block: the 289 scores and the 17x64 context are spread over all 256 threads (the per-sample
  // dependency chain, not throughput, is what bounds this kernel)
  __shared__ float q[NT * ATT_LD], k[NT * ATT_LD], v[NT * ATT_LD], p[NT * ATT_PLD];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const float* src = qkv + (int64_t)b * NT * 3 * TD;
  for (int idx = tid; idx < NT * 3 * TD; idx += 256) {
    const int t = idx / (3 * TD), c = idx - t * 3 * TD;
    const int part = c >> 6, d = c & 63;
    (part == 0 ? q : part == 1 ? k : v)[t * ATT_LD + d] = rbf(src[idx], bf);
  }
  __syncthreads();
  for (int pr = tid; pr < NT * NT; pr += 256) {
    const int i = pr / NT, j = pr - i * NT;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < TD; ++d) s = fmaf(q[i * ATT_LD + d], k[j * ATT_LD + d], s);
    p[i * ATT_PLD + j] = s * 0.125f;
  }
  __syncthreads();
  if (tid < NT) {
    float mx = -INFINITY;
    for (int j = 0; j < NT; ++j) mx = fmaxf(mx, p[tid * ATT_PLD + j]);
    float e[NT], sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { e[j] = expf(p[tid * ATT_PLD + j] - mx); sum += e[j]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float pv = e[j] * inv;
      p[tid * ATT_PLD + j] = pv;
      P[((int64_t)b * NT + tid) * NT + j] = pv;
    }
  }
  __syncthreads();
  for (int o = tid; o < NT * TD; o += 256) {
    const int i = o >> 6, d = o & 63;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) a = fmaf(rbf(p[i * ATT_PLD + j], bf), v[j * ATT_LD + d], a);
    ctx[(int64_t)b * NT * TD + o] = a;
  }
}

// Backward of the above: given dctx, saved P and qkv -> dqkv (same packed layout).
//   dV = P^T dctx ; dP = dctx V^T ; dS = P o (dP - rowsum(P o dP)) ; dQ = dS K / 8 ; dK = dS^T Q / 8
template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                       const float* __restrict__ dctx, int n,
                                                       float* __restrict__ dqkv, int bf) {
  __shared__ float q[NT * ATT_LD], k[NT * ATT_LD], v[NT * ATT_LD], dc[NT * ATT_LD];
  __shared__ float p[NT * ATT_PLD], ds[NT * ATT_PLD];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const float* src = qkv + (int64_t)b * NT * 3 * TD;
  for (int idx = tid; idx < NT * 3 * TD; idx += 256) {
    const int t = idx / (3 * TD), c = idx - t * 3 * TD;
    const int part = c >> 6, d = c & 63;
    (part == 0 ? q : part == 1 ? k : v)[t * ATT_LD + d] = rbf(src[idx], bf);  // q, k, v, dctx: contraction operands only
  }
  for (int o = tid; o < NT * TD; o += 256) dc[(o >> 6) * ATT_LD + (o & 63)] = rbf(dctx[(int64_t)b * NT * TD + o], bf);
  for (int pr = tid; pr < NT * NT; pr += 256) {
    const int i = pr / NT, j = pr - i * NT;
    p[i * ATT_PLD + j] = P[(int64_t)b * NT * NT + pr];
  }
  __syncthreads();
  float* dst = dqkv + (int64_t)b * NT * 3 * TD;
  for (int o = tid; o < NT * TD; o += 256) {  // dV[j][d]
    const int j = o >> 6, d = o & 63;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) a = fmaf(rbf(p[i * ATT_PLD + j], bf), dc[i * ATT_LD + d], a);
    dst[j * 3 * TD + 2 * TD + d] = a;
  }
  for (int pr = tid; pr < NT * NT; pr += 256) {  // dP[i][j]
    const int i = pr / NT, j = pr - i * NT;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < TD; ++d) s = fmaf(dc[i * ATT_LD + d], v[j * ATT_LD + d], s);
    ds[i * ATT_PLD + j] = s;
  }
  __syncthreads();
  if (tid < NT) {
    float rd = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) rd = fmaf(p[tid * ATT_PLD + j], ds[tid * ATT_PLD + j], rd);
#pragma unroll
    for (int j = 0; j < NT; ++j) ds[tid * ATT_PLD + j] = p[tid * ATT_PLD + j] * (ds[tid * ATT_PLD + j] - rd);
  }
  __syncthreads();
  for (int o = tid; o < NT * TD; o += 256) {
    const int t = o >> 6, d = o & 63;
    float aq = 0.f, ak = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      aq = fmaf(rbf(ds[t * ATT_PLD + j], bf), k[j * ATT_LD + d], aq);  // dQ[t][d] = sum_j dS[t][j] K[j][d]
      ak = fmaf(rbf(ds[j * ATT_PLD + t], bf), q[j * ATT_LD + d], ak);  // dK[t][d] = sum_i dS[i][t] Q[i][d]
    }
    dst[t * 3 * TD + d] = aq * 0.125f;
    dst[t * 3 * TD + TD + d] = ak * 0.125f;
  }
}

// --------------------------------------------------------------------------------- residual + LayerNorm
// out = LN(x + y) * gamma + beta over 64 columns, eps 1e-5, biased variance (nn.LayerNorm inside
// nn.TransformerEncoderLayer, post-norm). Saves xhat and rstd for the backward. One wave per row.
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int rows,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ out, float* __restrict__ xhat,
                                                         float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t o = (int64_t)r * TD + lane;
  const float z = x[o] + y[o];
  const float mean = wave_sum(z) * (1.f / TD);
  const float c = z - mean;
  const float var = wave_sum(c * c) * (1.f / TD);
  const float rs = 1.f / sqrtf(var + 1e-5f);
  const float xh = c * rs;
  xhat[o] = xh;
  out[o] = fmaf(xh, gamma[lane], beta[lane]);
  if (lane == 0) rstd[r] = rs;
}

// d(x+y) from dout (in place allowed: dz may alias dout); per-block dgamma/dbeta partials.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     int rows, float* __restrict__ dz, float* __restrict__ gpart,
                                                     float* __restrict__ bpart) {
  // each block leaves its partial dgamma/dbeta in gpart/bpart[blockIdx.x][64]; wgrad_reduce_kernel sums them in a
  // fixed order (no atomics: the whole update is run-to-run deterministic). Each wave walks many rows: four rows are
  // kept in flight per iteration to overlap their load -> reduce -> store chains
  __shared__ float sg[4][TD], sb[4][TD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float g = gamma[lane];
  float ag = 0.f, ab = 0.f;
  for (int r0 = (blockIdx.x * 4 + w) * 4; r0 < rows; r0 += gridDim.x * 16) {
    float d[4], xh[4], rs[4], dxh[4], c1[4], c2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = r0 + u < rows;
      const int64_t o = (int64_t)(ok ? r0 + u : 0) * TD + lane;
      const float dd = dout[o], xx = xhat[o];
      rs[u] = rstd[ok ? r0 + u : 0];
      d[u] = ok ? dd : 0.f;
      xh[u] = ok ? xx : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ag = fmaf(d[u], xh[u], ag);
      ab += d[u];
      dxh[u] = d[u] * g;
      c1[u] = dxh[u];
      c2[u] = dxh[u] * xh[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c1[u] += __shfl_xor(c1[u], o, 64);
        c2[u] += __shfl_xor(c2[u], o, 64);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r0 + u < rows)
        dz[(int64_t)(r0 + u) * TD + lane] = rs[u] * (dxh[u] - c1[u] * (1.f / TD) - xh[u] * (c2[u] * (1.f / TD)));
  }
  sg[w][lane] = ag;
  sb[w][lane] = ab;
  __syncthreads();
  if (w == 0) {
    gpart[blockIdx.x * TD + lane] = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
    bpart[blockIdx.x * TD + lane] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
  }
}

// --------------------------------------------------------------------------------- token pooling
// [state token | mean of the 16 depth tokens] -> [n][128]  (torchrl/networks/nets.py:1015-1021,1034); mx: the max over the
// depth tokens instead (max_pool=True, nets.py:1022-1023: `.max(dim=0)[0]`)
__global__ __launch_bounds__(128) void pool_fwd_kernel(const float* __restrict__ x, int n, float* __restrict__ pooled, int mx) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b >= n) return;
  const float* xb = x + (int64_t)b * NTOK * TD;
  float o;
  if (t < TD) o = xb[t];
  else {
    const int d = t - TD;
    float s = mx ? -INFINITY : 0.f;
#pragma unroll
    for (int i = 1; i < NTOK; ++i) s = mx ? fmaxf(s, xb[i * TD + d]) : s + xb[i * TD + d];
    o = mx ? s : s * (1.f / 16.f);
  }
  pooled[(int64_t)b * 2 * TD + t] = o;
}
// mx: the gradient of a max goes to the (first) token that attained it (torch.max(dim) backward), found again from the
// layer stack's output rows x
__global__ __launch_bounds__(64) void pool_bwd_kernel(const float* __restrict__ dpooled, int n, float* __restrict__ dx,
                                                      const float* __restrict__ x, int mx) {
  const int b = blockIdx.x, d = threadIdx.x;
  if (b >= n) return;
  const float ds = dpooled[(int64_t)b * 2 * TD + d];
  const float dg = dpooled[(int64_t)b * 2 * TD + TD + d];
  float* o = dx + (int64_t)b * NTOK * TD;
  o[d] = ds;
  if (!mx) {
    const float dm = dg * (1.f / 16.f);
#pragma unroll
    for (int i = 1; i < NTOK; ++i) o[i * TD + d] = dm;
    return;
  }
  const float* xb = x + (int64_t)b * NTOK * TD;
  int arg = 1;
  float best = xb[TD + d];
  for (int i = 2; i < NTOK; ++i) {
    const float v = xb[i * TD + d];
    if (v > best) { best = v; arg = i; }
  }
  for (int i = 1; i < NTOK; ++i) o[i * TD + d] = i == arg ? dg : 0.f;
}

// vision-only Transformer (torchrl/networks/nets.py:884-889: out[0 : 1 + 16].mean(dim=0) over a 16-token sequence is the
// mean of all tokens; max_pool=True: their max) -> [n][64]. srows: rows between two samples' first tokens (ntok when the token
// rows are dense; 17 with x pointing at row 1 of a 17-row slot: the wave-per-sample kernels' layout)
__global__ __launch_bounds__(64) void pool_all_fwd_kernel(const float* __restrict__ x, int n, int ntok, int srows,
                                                          float* __restrict__ pooled, int mx) {
  const int b = blockIdx.x, d = threadIdx.x;
  if (b >= n) return;
  const float* xb = x + (int64_t)b * srows * TD;
  float s = mx ? -INFINITY : 0.f;
  for (int i = 0; i < ntok; ++i) s = mx ? fmaxf(s, xb[i * TD + d]) : s + xb[i * TD + d];
  pooled[(int64_t)b * TD + d] = mx ? s : s * (1.f / (float)ntok);
}
__global__ __launch_bounds__(64) void pool_all_bwd_kernel(const float* __restrict__ dpooled, int n, int ntok, int srows,
                                                          float* __restrict__ dx, const float* __restrict__ x, int mx) {
  const int b = blockIdx.x, d = threadIdx.x;
  if (b >= n) return;
  const float dg = dpooled[(int64_t)b * TD + d];
  float* o = dx + (int64_t)b * srows * TD;
  if (!mx) {
    const float dm = dg * (1.f / (float)ntok);
    for (int i = 0; i < ntok; ++i) o[i * TD + d] = dm;
    return;
  }
  const float* xb = x + (int64_t)b * srows * TD;
  int arg = 0;
  float best = xb[d];
  for (int i = 1; i < ntok; ++i) {
    const float v = xb[i * TD + d];
    if (v > best) { best = v; arg = i; }
  }
  for (int i = 0; i < ntok; ++i) o[i * TD + d] = i == arg ? dg : 0.f;
}

// --------------------------------------------------------------------------------- rollout step (actor)
// Device-side cursor of the rollout: env step t of the epoch owns rollout slots [t*E, (t+1)*E).
// done: blocks of the step's last kernel that have finished. seq / stage / err: the hand-over counters of rollout_dense_kernel
// (csrc/rollout_dense.h) — monotonic, never reset; the actor's control block starts zeroed (v4l_actor_bind)
struct ActCtl { long long t; unsigned long long done; unsigned seq, err; unsigned stage[6]; };
__global__ void act_set_kernel(ActCtl* c, long long t) { c->t = t; c->done = 0; }
__global__ __launch_bounds__(256) void act_begin_kernel(const ActCtl* __restrict__ c, int E, int* __restrict__ rowidx) {
  const long long t = c->t;
  for (int i = threadIdx.x; i < E; i += 256) rowidx[i] = (int)(t * E + i);
}
// TanhNormal (policies/distribution.py:5-80): the pre-tanh value of a stored action, `log((1 + a) / (1 - a)) / 2`, and the
// per-dimension change-of-variables term `log(1 - a * a + epsilon)`, epsilon = 1e-6 — the reference's expressions
constexpr float TANH_EPS = 1e-6f;
__device__ __forceinline__ float tanh_pre(float a) { return logf((1.f + a) / (1.f - a)) / 2.f; }
__device__ __forceinline__ float tanh_corr(float a) { return logf(1.f - a * a + TANH_EPS); }

// GaussianContPolicyBase.explore (continuous_policy.py:85-125) + the value read-out of the collector
// (collector/on_policy.py:95-100) for one env step: action = mean + std * eps (== Normal(mean,std).sample() given the
// same standard-normal draws), entropy, value; also files action and value into the rollout arrays and advances t.
__global__ __launch_bounds__(256) void act_finish_kernel(ActCtl* c, const float* __restrict__ meanp,
                                                         const float* __restrict__ logstd, const float* __restrict__ valuep,
                                                         const float* __restrict__ eps, int E, int A,
                                                         float* __restrict__ acts_roll, float* __restrict__ values_roll,
                                                         float* __restrict__ logp_roll, float* __restrict__ action,
                                                         float* __restrict__ mean, float* __restrict__ stdv,
                                                         float* __restrict__ ent, float* __restrict__ value, int tanh_action) {
  const long long t = c->t;
  for (int i = threadIdx.x; i < E; i += 256) {
    float e = 0.f, lp = 0.f;
    for (int a = 0; a < A; ++a) {
      const float ls = fminf(fmaxf(logstd[a], LOG_SIG_MIN), LOG_SIG_MAX);
      const float sg = expf(ls);
      e += 0.5f + HALF_LOG_2PI + logf(sg);
      const float mu = meanp[(int64_t)i * OUT_LD + a];
      float act = fmaf(sg, eps[(int64_t)i * A + a], mu);
      if (tanh_action) act = tanhf(act);  // TanhNormal.rsample (distribution.py:61-80); eps = 0: eval_act's tanh(mean)
      action[(int64_t)i * A + a] = act;
      mean[(int64_t)i * A + a] = mu;
      stdv[(int64_t)i * A + a] = sg;
      if (acts_roll != nullptr) acts_roll[(t * E + i) * A + a] = act;
      // log pi(a|s) of the acting policy, with the expression actor_loss_kernel uses for the frozen target policy:
      // the policy that acts during an epoch IS that epoch's target policy (ppo.py:34 copies it before the updates).
      // tanh policies: through the STORED action like target_pf.update(obs, actions) does (distribution.py:38-51)
      const float d = (tanh_action ? tanh_pre(act) : act) - mu;
      lp += -(d * d) / (2.f * sg * sg) - logf(sg) - HALF_LOG_2PI;
      if (tanh_action) lp -= tanh_corr(act);
    }
    ent[i] = e;
    const float v = valuep[(int64_t)i * OUT_LD];
    value[i] = v;
    if (values_roll != nullptr) values_roll[t * E + i] = v;
    if (logp_roll != nullptr) logp_roll[t * E + i] = lp;
  }
  __syncthreads();
  if (threadIdx.x == 0) c->t = t + 1;
}

// --------------------------------------------------------------------------------- block reductions
struct Red4 { double s, s2; float mx, mn; };
// Block reduction (blockDim.x = 64 * nw, nw <= 16) of (sum, sum of squares, max, min); result valid on all threads.
__device__ __forceinline__ Red4 block_red4(double s, double s2, float mx, float mn) {
  __shared__ double rs[16], rs2[16];
  __shared__ float rmx[16], rmn[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    s2 += __shfl_xor(s2, o, 64);
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    mn = fminf(mn, __shfl_xor(mn, o, 64));
  }
  __syncthreads();  // protect reuse across consecutive calls
  if (lane == 0) { rs[w] = s; rs2[w] = s2; rmx[w] = mx; rmn[w] = mn; }
  __syncthreads();
  Red4 r;
  r.s = 0.0; r.s2 = 0.0; r.mx = -INFINITY; r.mn = INFINITY;
  const int nw = blockDim.x >> 6;
  for (int k = 0; k < nw; ++k) {  // fixed order: deterministic
    r.s += rs[k]; r.s2 += rs2[k];
    r.mx = fmaxf(r.mx, rmx[k]); r.mn = fminf(r.mn, rmn[k]);
  }
  return r;
}

// Layout of the 24-float statistics record one PPO minibatch update produces. [0..17] are the 18 logger
// keys of torchrl/algo/on_policy/ppo.py:77-92,122-123,142-145 in that order; the rest is internal.
enum {
  ST_ADV_MEAN = 0, ST_ADV_STD, ST_ADV_MAX, ST_ADV_MIN, ST_VF_LOSS, ST_GN_VF, ST_PI_LOSS,
  ST_LP_MEAN, ST_LP_STD, ST_LP_MAX, ST_LP_MIN, ST_LS_MEAN, ST_LS_STD, ST_LS_MAX, ST_LS_MIN,
  ST_RATIO_MAX, ST_RATIO_MIN, ST_GN_PF,
  // per-shard moments for the data-parallel exchange: sum a, M2 = sum (a - mean_shard)^2, count, count * mean_shard^2
  // (the global variance is rebuilt as (sum M2_i + sum n_i mean_i^2 - N mean^2) / (N - 1): the only cancellation left is
  // between shard means, not between a raw sum of squares and N mean^2)
  ST_ADV_SUM = 18, ST_ADV_M2, ST_ADV_CNT, ST_ADV_NM2,
  ST_NONFINITE = 22,  // how many of the 18 logged scalars of this update are NaN / Inf (the on-device form of the
                      // collector's "NaN detected" check, collector/on_policy.py:102-107: the host reads one number per epoch)
  ST_F16_SAT = 23,    // V4L_F16: how many loss-gradient elements of this update were clamped at +-V4L_F16_GRAD_CLAMP
  ST_SIZE = 24
};

// advs.mean(), advs.std() (Bessel), max, min of the minibatch (ppo.py:142-145). Single block.
__device__ __forceinline__ void adv_stats_body(const float* __restrict__ adv, const int* rowidx, int n, float* __restrict__ st) {
  double s = 0.0; float mx = -INFINITY, mn = INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float a = adv[rowidx ? rowidx[i] : i];
    s += a; mx = fmaxf(mx, a); mn = fminf(mn, a);
  }
  Red4 r = block_red4(s, 0.0, mx, mn);
  const double mean = r.s / n;
  double q = 0.0, sq = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double a = adv[rowidx ? rowidx[i] : i];
    q += (a - mean) * (a - mean);
    sq += a * a;
  }
  Red4 r2 = block_red4(q, sq, 0.f, 0.f);
  if (threadIdx.x == 0) {
    st[ST_ADV_MEAN] = (float)mean;
    st[ST_ADV_STD] = (float)sqrt(r2.s / (double)(n - 1));
    st[ST_ADV_MAX] = r.mx;
    st[ST_ADV_MIN] = r.mn;
    st[ST_ADV_SUM] = (float)r.s;
    st[ST_ADV_M2] = (float)r2.s;
    st[ST_ADV_CNT] = (float)n;
    st[ST_ADV_NM2] = (float)(n * mean * mean);
  }
}
// the same statistics from values a block already holds in registers (thread t: elements t, t + blockDim, ... — the order
// adv_stats_body's loops add them in, so the bits are the same)
__device__ __forceinline__ void adv_stats_regs(const float (&a)[4], const bool (&ok)[4], int n, float* __restrict__ st) {
  double s = 0.0; float mx = -INFINITY, mn = INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (ok[j]) { s += a[j]; mx = fmaxf(mx, a[j]); mn = fminf(mn, a[j]); }
  Red4 r = block_red4(s, 0.0, mx, mn);
  const double mean = r.s / n;
  double q = 0.0, sq = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (ok[j]) { const double d = a[j]; q += (d - mean) * (d - mean); sq += d * d; }
  Red4 r2 = block_red4(q, sq, 0.f, 0.f);
  if (threadIdx.x == 0) {
    st[ST_ADV_MEAN] = (float)mean;
    st[ST_ADV_STD] = (float)sqrt(r2.s / (double)(n - 1));
    st[ST_ADV_MAX] = r.mx;
    st[ST_ADV_MIN] = r.mn;
    st[ST_ADV_SUM] = (float)r.s;
    st[ST_ADV_M2] = (float)r2.s;
    st[ST_ADV_CNT] = (float)n;
    st[ST_ADV_NM2] = (float)(n * mean * mean);
  }
}
__global__ __launch_bounds__(256) void adv_stats_kernel(const float* __restrict__ adv, const int* __restrict__ rowidx, int n,
                                                        float* __restrict__ st) {
  adv_stats_body(adv, rowidx, n, st);
}
// Data-parallel: mean / unbiased std of the GLOBAL minibatch from the all-reduced shard moments.
__global__ void adv_stats_finalize_kernel(float* st) {
  const double s = st[ST_ADV_SUM], m2 = st[ST_ADV_M2], c = st[ST_ADV_CNT], nm2 = st[ST_ADV_NM2];
  const double mean = s / c;
  st[ST_ADV_MEAN] = (float)mean;
  st[ST_ADV_STD] = (float)sqrt(fmax(0.0, (m2 + (nm2 - c * mean * mean)) / (c - 1.0)));
}
// The scalars that ride in the tail of a gradient bucket through its all-reduce (8 floats behind the gradients):
//   critic bucket: [sum a, M2, count, count * mean^2, vf_loss share, 0, 0, 0]   policy bucket: [policy_loss share, 0 ...]
// pack = 1: record -> tail (before the collective); pack = 0: tail -> record (after it). A share is the shard's value / world:
// the sum over ranks is the big-batch mean the reference would log.
// V4L_BUCKET_TAIL (= 8) comes from include/v4l_hip.h
__global__ void bucket_tail_kernel(float* __restrict__ st, float* __restrict__ tail, int which, int pack, float inv_world) {
  const int t = threadIdx.x;
  if (t >= V4L_BUCKET_TAIL) return;
  if (which == 1) {  // critic
    const int src[5] = {ST_ADV_SUM, ST_ADV_M2, ST_ADV_CNT, ST_ADV_NM2, ST_VF_LOSS};
    if (pack) tail[t] = t < 5 ? st[src[t]] : 0.f;   // vf_loss is already a share: critic_loss_kernel scales by 1 / (n * world)
    else if (t < 5) st[src[t]] = tail[t];
  } else {
    if (pack) tail[t] = t == 0 ? st[ST_PI_LOSS] * inv_world : 0.f;
    else if (t == 0) st[ST_PI_LOSS] = tail[0];
  }
}

// Communicator self-test (v4l_trainer_comm_selftest): a rank-dependent pattern of small integers — exact in fp32 whatever
// order a collective adds them in — into a gradient bucket and the record fields its tail carries, and the check of the
// all-reduced result against the sum every rank can compute on its own.
__device__ __forceinline__ float comm_pattern(int64_t i, int rank) { return (float)((i * 7 + (int64_t)rank * 13) % 251); }
__global__ __launch_bounds__(256) void comm_pattern_kernel(float* __restrict__ g, int64_t n, int rank, float* __restrict__ st,
                                                           int which) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] = comm_pattern(i, rank);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (which == 1) {
      st[ST_ADV_SUM] = (float)(rank + 1); st[ST_ADV_M2] = (float)(2 * rank + 1); st[ST_ADV_CNT] = 64.f;
      st[ST_ADV_NM2] = (float)(rank + 3); st[ST_VF_LOSS] = (float)(rank + 5);
    } else {
      st[ST_PI_LOSS] = (float)(rank + 1);
    }
  }
}
__global__ __launch_bounds__(256) void comm_check_kernel(const float* __restrict__ g, int64_t n, int world,
                                                         const float* __restrict__ st, int which, int* __restrict__ bad) {
  int mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float want = 0.f;
    for (int r = 0; r < world; ++r) want += comm_pattern(i, r);
    mine += g[i] != want;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float w = (float)world, tri = 0.5f * w * (w - 1.f);  // sum of the ranks
    if (which == 1) {
      mine += st[ST_ADV_SUM] != tri + w;
      mine += st[ST_ADV_M2] != 2.f * tri + w;
      mine += st[ST_ADV_CNT] != 64.f * w;
      mine += st[ST_ADV_NM2] != tri + 3.f * w;
      mine += st[ST_VF_LOSS] != tri + 5.f * w;
    } else {
      mine += fabsf(st[ST_PI_LOSS] - (tri + w) / w) > 1e-5f * (tri + w) / w;  // shares: mean over the ranks
    }
  }
  if (mine) atomicAdd(bad, mine);
}

// (per-statement contraction for the loss rows: the statistics block and the blocks that run the heads' chain beside it
// evaluate the same row in different kernels / instantiations and must produce the same bits — see csrc/wps.h)
#pragma clang fp contract(on)
// nn.MSELoss()(values, est_rets) and its gradient (ppo.py:94-123; clipped_value_loss=False path, and the
// clipped variant of ppo.py:105-112 when clip > 0). values is the critic output [n][OUT_LD], column 0.
// One row: loss term l and d(loss)/d(value) g. (Shared by the statistics block and by the blocks that run the heads'
// data-grads beside it, csrc/wps.h: both must see the same bits.)
// A loss-gradient element as the backward's contractions get it: x gscale (1, or V4L_F16's power of two: exact) and, scaled, kept
// inside half's range (a NaN stays a NaN). sat counts the clamped ones.
__device__ __forceinline__ float grad_out(float g, float gscale, int& sat) {
  if (gscale == 1.f) return g;
  const float x = g * gscale;
  const bool over = fabsf(x) > V4L_F16_GRAD_CLAMP;
  sat += over ? 1 : 0;
  return over ? copysignf(V4L_F16_GRAD_CLAMP, x) : x;
}
__device__ __forceinline__ void critic_row(float v, float r, float ov, int clipped, float clip, float inv_n, float& l, float& g) {
  if (!clipped) {
    const float d = v - r;
    l = d * d;
    g = 2.f * d * inv_n;
  } else {
    const float dv = v - ov;
    const float vc = ov + fminf(fmaxf(dv, -clip), clip);
    const float l1 = (v - r) * (v - r), l2 = (vc - r) * (vc - r);
    // 0.5 * max(l1, l2).mean(); torch.max splits ties evenly between both arguments
    const float g1 = 2.f * (v - r);
    const float g2 = (dv >= -clip && dv <= clip) ? 2.f * (vc - r) : 0.f;
    l = 0.5f * fmaxf(l1, l2);
    g = 0.5f * inv_n * (l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * (g1 + g2)));
  }
}
__device__ __forceinline__ void critic_loss_body(const float* __restrict__ values, const float* __restrict__ ret,
                                                 const float* __restrict__ oldv, const int* __restrict__ rowidx, int n,
                                                 float inv_n, int clipped, float clip, float* __restrict__ dvalues,
                                                 float* __restrict__ st, float gscale) {
  double s = 0.0;
  int sat = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int slot = rowidx ? rowidx[i] : i;
    const float v = values[(int64_t)i * OUT_LD], r = ret[slot];
    float g, l;
    critic_row(v, r, clipped ? oldv[slot] : 0.f, clipped, clip, inv_n, l, g);
    g = grad_out(g, gscale, sat);
    s += l;
#pragma unroll
    for (int c = 0; c < OUT_LD; ++c) dvalues[(int64_t)i * OUT_LD + c] = c == 0 ? g : 0.f;
  }
  Red4 r = block_red4(s, (double)sat, 0.f, 0.f);
  if (threadIdx.x == 0) {
    st[ST_VF_LOSS] = (float)(r.s * (double)inv_n);
    if (gscale != 1.f) st[ST_F16_SAT] += (float)r.s2;
  }
}
__global__ __launch_bounds__(1024) void critic_loss_kernel(const float* __restrict__ values, const float* __restrict__ ret,
                                                          const float* __restrict__ oldv, const int* __restrict__ rowidx,
                                                          int n, float inv_n, int clipped, float clip,
                                                          float* __restrict__ dvalues, float* __restrict__ st, float gscale) {
  critic_loss_body(values, ret, oldv, rowidx, n, inv_n, clipped, clip, dvalues, st, gscale);
}


// Clipped-surrogate + entropy loss of PPO.update_actor (ppo.py:42-92) and its gradient w.r.t. the policy
// mean [n][OUT_LD] and logstd [A]. logp_old comes from the frozen target policy's mean/logstd on the same
// minibatch. Advantages are normalised with the minibatch statistics in st (ppo.py:148). Single block.
// inv_n is 1/(global batch) so that data-parallel ranks sum to the big-batch gradient.
struct ActorArgs {
  const float *mean, *logstd, *tmean, *tlogstd, *logp_old, *acts, *adv;
  const int* rowidx;
  int n, A;
  float inv_n, clip, ent_coef;
  float *dmean, *dlogstd, *st;
  float gscale;     // the dmean rows leave multiplied by this (v4l_net_grad_scale: 1, or V4L_F16's power of two); d log sigma never does
  int tanh_action;  // TanhNormal policies: log-probs of the stored (post-tanh) actions through atanh, distribution.py:38-51
};
struct ActorDims { float ls[8], sg[8], lsg[8], tls[8], tsg[8], tlsg[8]; float ent; };
__device__ __forceinline__ ActorDims actor_dims(const ActorArgs& p) {
  ActorDims d;
  d.ent = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    d.ls[a] = d.sg[a] = d.lsg[a] = d.tls[a] = d.tsg[a] = d.tlsg[a] = 0.f;
    if (a < p.A) {
      d.ls[a] = fminf(fmaxf(p.logstd[a], LOG_SIG_MIN), LOG_SIG_MAX);
      d.sg[a] = expf(d.ls[a]); d.lsg[a] = logf(d.sg[a]);
      d.tls[a] = p.tlogstd != nullptr ? fminf(fmaxf(p.tlogstd[a], LOG_SIG_MIN), LOG_SIG_MAX) : 0.f;
      d.tsg[a] = expf(d.tls[a]); d.tlsg[a] = logf(d.tsg[a]);
      d.ent += 0.5f + HALF_LOG_2PI + d.lsg[a];
    }
  }
  return d;
}
// one minibatch row i (rollout slot `slot`): log pi, ratio, surrogate term, d(loss)/d(log pi) and the per-dimension factors of
// d(log pi)/d(mean) (dm) and d(log pi)/d(log sigma) + 1 (z2)
// TANH: compiled in only for TanhNormal policies (as a run-time flag the atanh / correction logs were evaluated speculatively for
// every policy: the single-block statistics launch went from 14 to 36 us)
struct ActorRow { float lp, ratio, sur, dlp, dm[8], z2[8]; };
template <bool TANH>
__device__ __forceinline__ ActorRow actor_row(const ActorArgs& p, const ActorDims& D, int i, int slot, float amean, float astd) {
  ActorRow o;
  float lp = 0.f, lpo = 0.f, mu[8], tmu[8];
  {  // the padded [OUT_LD] rows are 64-byte aligned: two 16-byte loads per row instead of A strided scalar ones
    const float4 m0 = *reinterpret_cast<const float4*>(p.mean + (int64_t)i * OUT_LD);
    const float4 m1 = *reinterpret_cast<const float4*>(p.mean + (int64_t)i * OUT_LD + 4);
    mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w; mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
    if (p.logp_old == nullptr) {
      const float4 t0 = *reinterpret_cast<const float4*>(p.tmean + (int64_t)i * OUT_LD);
      const float4 t1 = *reinterpret_cast<const float4*>(p.tmean + (int64_t)i * OUT_LD + 4);
      tmu[0] = t0.x; tmu[1] = t0.y; tmu[2] = t0.z; tmu[3] = t0.w; tmu[4] = t1.x; tmu[5] = t1.y; tmu[6] = t1.z; tmu[7] = t1.w;
    }
  }
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    if (a < p.A) {
      const float act = p.acts[(int64_t)slot * p.A + a];
      const float x = TANH ? tanh_pre(act) : act;
      const float corr = TANH ? tanh_corr(act) : 0.f;
      const float d = x - mu[a];
      const float var = D.sg[a] * D.sg[a];
      lp += -(d * d) / (2.f * var) - D.lsg[a] - HALF_LOG_2PI - corr;
      o.z2[a] = d * d / var;
      o.dm[a] = d / var;
      if (p.logp_old == nullptr) {
        const float dt = x - tmu[a];
        lpo += -(dt * dt) / (2.f * D.tsg[a] * D.tsg[a]) - D.tlsg[a] - HALF_LOG_2PI - corr;
      }
    } else {
      o.z2[a] = 0.f; o.dm[a] = 0.f;
    }
  }
  if (p.logp_old != nullptr) lpo = p.logp_old[slot];  // stored when the action was taken (== the target policy's)
  const float ratio = expf(lp - lpo);
  const float an = (p.adv[slot] - amean) / (astd + 1e-5f);
  const float pre = ratio * an;
  const float clp = fminf(fmaxf(ratio, 1.f - p.clip), 1.f + p.clip) * an;
  o.lp = lp; o.ratio = ratio;
  o.sur = fminf(pre, clp);
  // d(-mean(min(pre,clp)))/dlogp: the clipped branch has zero slope outside the clip range
  o.dlp = (pre <= clp) ? -p.inv_n * an * ratio : 0.f;
  return o;
}
// the row's d(loss)/d(mean) as the backward gets it (statistics block and chain blocks: the same bits)
__device__ __forceinline__ void actor_dmean_row(const ActorRow& o, float gscale, float (&dmr)[8], int& sat) {
#pragma unroll
  for (int a = 0; a < 8; ++a) dmr[a] = grad_out(o.dlp * o.dm[a], gscale, sat);
}
template <bool TANH>
__device__ __forceinline__ void actor_loss_body(const ActorArgs& p) {
  __shared__ float sdl[16][8];
  const int n = p.n, A = p.A;
  const ActorDims D = actor_dims(p);
  float dl[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) dl[a] = 0.f;
  const float amean = p.st[ST_ADV_MEAN], astd = p.st[ST_ADV_STD];
  double s_lp = 0.0, s_lp2 = 0.0, s_sur = 0.0;
  int sat = 0;
  float lp_mx = -INFINITY, lp_mn = INFINITY, r_mx = -INFINITY, r_mn = INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int slot = p.rowidx ? p.rowidx[i] : i;
    const ActorRow o = actor_row<TANH>(p, D, i, slot, amean, astd);
    s_sur += o.sur;
    {
      float4* drow = reinterpret_cast<float4*>(p.dmean + (int64_t)i * OUT_LD);
      float dmr[8];
      actor_dmean_row(o, p.gscale, dmr, sat);
      drow[0] = float4{dmr[0], dmr[1], dmr[2], dmr[3]};  // dm[a >= A] == 0
      drow[1] = float4{dmr[4], dmr[5], dmr[6], dmr[7]};
      drow[2] = float4{0.f, 0.f, 0.f, 0.f};
      drow[3] = float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < A) dl[a] += o.dlp * (o.z2[a] - 1.f);
    s_lp += o.lp; s_lp2 += (double)o.lp * o.lp;
    lp_mx = fmaxf(lp_mx, o.lp); lp_mn = fminf(lp_mn, o.lp);
    r_mx = fmaxf(r_mx, o.ratio); r_mn = fminf(r_mn, o.ratio);
  }
  Red4 r1 = block_red4(s_lp, s_lp2, lp_mx, lp_mn);
  Red4 r2 = block_red4(s_sur, (double)sat, r_mx, r_mn);
  // reduce dlogstd over the block
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    float v = wave_sum(dl[a]);
    if (lane == 0) sdl[w][a] = v;
  }
  __syncthreads();
  float* st = p.st;
  if (threadIdx.x < A) {
    const int a = threadIdx.x;
    const float raw = p.logstd[a];
    float g = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) g += sdl[k][a];
    // entropy term: -ent_coef * mean_b(sum_a log sigma_a + const); each local sample carries weight inv_n
    g += -p.ent_coef * p.inv_n * (float)n;
    p.dlogstd[a] = (raw >= LOG_SIG_MIN && raw <= LOG_SIG_MAX) ? g : 0.f;
  }
  if (threadIdx.x == 0) {
    const double lpm = r1.s / n;
    st[ST_PI_LOSS] = (float)(-(r2.s / n) - (double)p.ent_coef * D.ent);
    st[ST_LP_MEAN] = (float)lpm;
    st[ST_LP_STD] = (float)sqrt(fmax(0.0, (r1.s2 - n * lpm * lpm) / (double)(n - 1)));
    st[ST_LP_MAX] = r1.mx; st[ST_LP_MIN] = r1.mn;
    st[ST_RATIO_MAX] = r2.mx; st[ST_RATIO_MIN] = r2.mn;
    if (p.gscale != 1.f) st[ST_F16_SAT] += (float)r2.s2;
    double m = 0.0; float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < A) { m += D.ls[a]; mx = fmaxf(mx, D.ls[a]); mn = fminf(mn, D.ls[a]); }
    m /= A;
    double q = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < A) q += (D.ls[a] - m) * (D.ls[a] - m);
    st[ST_LS_MEAN] = (float)m;
    st[ST_LS_STD] = A > 1 ? (float)sqrt(q / (A - 1)) : NAN;
    st[ST_LS_MAX] = mx; st[ST_LS_MIN] = mn;
  }
}
template <bool TANH>
__global__ __launch_bounds__(1024) void actor_loss_kernel(ActorArgs p) { actor_loss_body<TANH>(p); }
#pragma clang fp contract(fast)

// Gaussian head post-processing for the policy API (continuous_policy.py:85-146,486-492): from the padded
// mean [n][OUT_LD] and logstd [A] produce contiguous mean/std [n][A], clamped log_std [A], ent [n] and, when
// acts != null, log_prob [n] of those actions.
__global__ __launch_bounds__(256) void gauss_head_kernel(const float* __restrict__ meanp, const float* __restrict__ logstd,
                                                         const float* __restrict__ acts, int n, int A,
                                                         float* __restrict__ mean, float* __restrict__ stdv,
                                                         float* __restrict__ logstd_c, float* __restrict__ ent,
                                                         float* __restrict__ logp, int tanh_action = 0,
                                                         const float* __restrict__ pre_tanh = nullptr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float e = 0.f, lp = 0.f;
  for (int a = 0; a < A; ++a) {
    const float ls = fminf(fmaxf(logstd[a], LOG_SIG_MIN), LOG_SIG_MAX);
    const float sg = expf(ls), lsg = logf(sg);
    e += 0.5f + HALF_LOG_2PI + lsg;
    if (i == 0) logstd_c[a] = ls;
    if (i < n) {
      const float mu = meanp[(int64_t)i * OUT_LD + a];
      mean[(int64_t)i * A + a] = mu;
      stdv[(int64_t)i * A + a] = sg;
      if (acts != nullptr) {
        const float act = acts[(int64_t)i * A + a];
        const float x = !tanh_action ? act : (pre_tanh != nullptr ? pre_tanh[(int64_t)i * A + a] : tanh_pre(act));
        const float d = x - mu;
        lp += -(d * d) / (2.f * sg * sg) - lsg - HALF_LOG_2PI;
        if (tanh_action) lp -= tanh_corr(act);
      }
    }
  }
  if (i < n) {
    ent[i] = e;
    if (acts != nullptr) logp[i] = lp;
  }
}

// Copy column 0 of a padded head output to a contiguous [n] vector (critic values for the collector).
__global__ __launch_bounds__(256) void col0_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[(int64_t)i * OUT_LD];
}

// --------------------------------------------------------------------------------- grad norm + Adam
// sum of squares of a flat gradient buffer -> part[blockIdx.x] (<= GRAD_NORM_PARTS blocks; clip_adam_kernel adds them in
// order). 16-byte loads, one or two per thread: the 388 K-float buffer is one short burst over the whole chip, not 24
// dependent trips of 64 blocks (12 us -> ~4 us).
constexpr int GRAD_NORM_PARTS = 256;
constexpr int ADAM_MAX_PARTS = 16384;  // partials clip_adam_kernel can add up (64 per thread; the NatureCNN nets have 9.4 K reduce blocks)
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  double s = 0.0;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(g)[i];
    s += ((double)x.x * x.x + (double)x.y * x.y) + ((double)x.z * x.z + (double)x.w * x.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float x = g[(n4 << 2) + threadIdx.x]; s += (double)x * x; }
  Red4 r = block_red4(s, 0.0, 0.f, 0.f);
  if (threadIdx.x == 0) part[blockIdx.x] = (float)r.s;
}

// One (tensor) segment of an optimiser's parameter set: parameters live in caller-owned tensors (pointer
// table), gradients and both Adam moments in flat buffers at offset goff.
struct ParamSeg { float* p; int64_t goff; int64_t n; int64_t blk0; };  // blk0: first block of this segment

// torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam (eps after sqrt(v_hat),
// no weight decay == AdamW with wd 0) — torchrl/algo/on_policy/ppo.py:73-75,118-120, a2c.py:30-40.
// Operation order follows torch/optim/adam.py::_single_tensor_adam. The clip coefficient is derived from the
// squared norm accumulated in *sumsq; thread 0 of block 0 also publishes the pre-clip norm to *norm_out.
// Device-resident control block of the update loop. Everything that changes from one minibatch update to the
// next (which rows, Adam's bias corrections, where the statistics go) lives here and is advanced by kernels, so
// the launch sequence of an update is *identical* every time and can be replayed as one hipGraph.
struct UpdCtl {
  int upd_index;        // minibatch number inside the epoch: row of rowidx_all / stats_all
  int pad0;
  long long step;       // optimiser steps taken so far (both Adams step once per update)
  double lr_pf, lr_vf;
  float beta1, beta2;
  float step_size[2];   // lr / (1 - beta1^step) for [0] pf, [1] vf — refreshed by upd_begin_body (block 0 of begin_pack_kernel)
  float bc2_sqrt;       // sqrt(1 - beta2^step)
  float pad1;
};
__global__ void ctl_set_kernel(UpdCtl* c, int upd_index, long long step, double lr_pf, double lr_vf, float b1, float b2) {
  c->upd_index = upd_index; c->step = step; c->lr_pf = lr_pf; c->lr_vf = lr_vf; c->beta1 = b1; c->beta2 = b2;
}
// the Adam step / bias corrections of the update being opened (double precision, like torch/optim/adam.py::_single_tensor_adam)
__device__ __forceinline__ void upd_begin_step(UpdCtl* c) {
  if (threadIdx.x == 0) {
    const long long step = c->step + 1;
    c->step = step;
    const double bc1 = 1.0 - pow((double)c->beta1, (double)step);
    const double bc2 = 1.0 - pow((double)c->beta2, (double)step);
    c->step_size[0] = (float)(c->lr_pf / bc1);
    c->step_size[1] = (float)(c->lr_vf / bc1);
    c->bc2_sqrt = (float)sqrt(bc2);
  }
}
// Opens update #upd_index: selects its rows (identity when rowidx_all is null), clears the statistics record and
// advances the Adam step / bias corrections.
// adv != null: also the advantage statistics of the selected rows (adv_stats_kernel's work, one launch less).
__device__ __forceinline__ void upd_begin_body(UpdCtl* c, const int* __restrict__ rowidx_all, int n, int* rowidx_cur,
                                               float* stats_cur, const float* __restrict__ adv) {
  const int u = c->upd_index;
  if (adv != nullptr && n <= 4 * (int)blockDim.x) {
    // (round 5) the rows' indices and advantages stay in registers: index -> advantage is the only dependent pair of round
    // trips left in this block's chain (it was index -> store -> barrier -> reload -> advantage -> second pass)
    int ri[4];
    float av[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = (int)threadIdx.x + j * (int)blockDim.x;
      ok[j] = i < n;
      ri[j] = rowidx_all ? rowidx_all[(int64_t)u * n + (ok[j] ? i : 0)] : i;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = adv[ok[j] ? ri[j] : 0];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ok[j]) rowidx_cur[(int)threadIdx.x + j * (int)blockDim.x] = ri[j];
    if (threadIdx.x < ST_SIZE) stats_cur[threadIdx.x] = 0.f;
    upd_begin_step(c);  // (thread 0's double-precision pow() runs while the loads above are in flight)
    __syncthreads();  // the cleared record is in place before thread 0 fills its part
    adv_stats_regs(av, ok, n, stats_cur);
    return;
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) rowidx_cur[i] = rowidx_all ? rowidx_all[(int64_t)u * n + i] : i;
    if (threadIdx.x < ST_SIZE) stats_cur[threadIdx.x] = 0.f;
    if (adv != nullptr) {
      __syncthreads();  // rowidx_cur and the cleared record are visible to the whole block
      adv_stats_body(adv, rowidx_cur, n, stats_cur);
    }
  }
  upd_begin_step(c);
}
__global__ __launch_bounds__(256) void clip_adam_kernel(const ParamSeg* __restrict__ segs, int nseg,
                                                        const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ part, int npart,
                                                        const float* __restrict__ extra, int nextra, float max_norm, float eps, const UpdCtl* __restrict__ ctl,
                                                        int which, float* __restrict__ norm_out, UpdCtl* close_ctl,
                                                        const float* stats_cur, float* __restrict__ stats_all) {
  const ParamSeg sg = segs[find_desc(segs, nseg, (int64_t)blockIdx.x)];
  const float beta1 = ctl->beta1, beta2 = ctl->beta2;
  const float step_size = ctl->step_size[which], bc2_sqrt = ctl->bc2_sqrt;
  const int lane_ = threadIdx.x & 63;
  // this thread's element: its four operands are requested before the norm is summed (they do not depend on it; behind the
  // barrier below they would be one more round trip at the end of the block's chain — round 5)
  const int64_t i = ((int64_t)blockIdx.x - sg.blk0) * 256 + threadIdx.x;
  const bool mine = i < sg.n;
  const int64_t o = sg.goff + (mine ? i : 0);
  const float g_in = g[o], m_in = m[o], v_in = v[o], p_in = sg.p[mine ? i : 0];
  // the norm's partials (grad_sumsq_kernel's, or one per wgrad_reduce block: npart <= ADAM_MAX_PARTS), strided over the block's
  // threads with every load in flight at once, then lanes -> waves in a fixed order: the same bits in every block
  __shared__ float wpart[4];
  float pv[ADAM_MAX_PARTS / 256];
#pragma unroll
  for (int i = 0; i < ADAM_MAX_PARTS / 256; ++i) {
    const int k = (int)threadIdx.x + 256 * i;
    pv[i] = part[k < npart ? k : 0];
  }
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < ADAM_MAX_PARTS / 256; ++i) ps += (int)threadIdx.x + 256 * i < npart ? pv[i] : 0.f;
  if ((int)threadIdx.x < nextra) { const float x = extra[threadIdx.x]; ps = fmaf(x, x, ps); }  // gradients no reduce block wrote (log sigma)
  ps = wave_sum(ps);
  if (lane_ == 0) wpart[threadIdx.x >> 6] = ps;
  __syncthreads();
  const float tot = sqrtf((wpart[0] + wpart[1]) + (wpart[2] + wpart[3]));
  const float coef = fminf(max_norm / (tot + 1e-6f), 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out != nullptr) *norm_out = tot;
  if (blockIdx.x == 0 && close_ctl != nullptr) {
    // last launch of an update: publish the statistics record (this block just completed it with the policy's gradient
    // norm) and move on to the next minibatch (no separate closing launch)
    __syncthreads();
    const int u = close_ctl->upd_index;
    if (threadIdx.x < 64) {  // wave 0: count the non-finite logger scalars (losses, norms, ratios: a NaN anywhere reaches them)
      const float sv = threadIdx.x < 18 ? stats_cur[threadIdx.x] : 0.f;
      const unsigned long long badm = __ballot(!(fabsf(sv) <= 3.0e38f));
      if (threadIdx.x == 0) const_cast<float*>(stats_cur)[ST_NONFINITE] = (float)__popcll(badm);
    }
    __syncthreads();
    if (stats_all != nullptr && threadIdx.x < ST_SIZE) stats_all[(int64_t)u * ST_SIZE + threadIdx.x] = stats_cur[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) close_ctl->upd_index = u + 1;
  }
  if (!mine) return;
  const float gr = g_in * coef;
  float mm = m_in, vv = v_in;
  mm = mm + (gr - mm) * (1.f - beta1);           // exp_avg.lerp_(grad, 1 - beta1)
  vv = vv * beta2 + (1.f - beta2) * (gr * gr);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  sg.p[i] = p_in - step_size * (mm / denom);     // param.addcdiv_(exp_avg, denom, value=-step_size)
  m[o] = mm;
  v[o] = vv;
}

// --------------------------------------------------------------------------------- weight packing
// Builds the contraction-ready copies of a weight tensor (operand type T, zero padded to the GEMM tiles).
enum { PK_NT = 0, PK_T = 1, PK_CONV_NHWC = 2, PK_CONV_NHWC_T = 3, PK_CONV_DGRAD = 4, PK_FRAG = 5, PK_FRAGP = 6, PK_FRAGPT = 7, PK_FRAGT = 8 };
struct PackDesc {
  const float* src;  // PyTorch-layout weight
  int64_t dst_off;   // element offset in the packed buffer
  int kind;
  int R, Cc;         // packed rows / cols (padded)
  int N, K;          // source: [N][K] (Linear) or [N][Cin*taps] (Conv)
  int Cin, taps, KW; // conv
  int s, py, px, TW; // dgrad class
  int64_t blk0;
};
// 8 consecutive packed elements (one row: every packed row length is a multiple of 8) per thread, one 16 / 32-byte store
constexpr int PACK_PER_BLOCK = 256 * 8;
template <typename T>
__device__ __forceinline__ void pack_body(const PackDesc* __restrict__ descs, int nd, T* __restrict__ dst, int64_t blk) {
  const PackDesc d = descs[find_desc(descs, nd, blk)];
  const V4L_GLOBAL float* src = as_global(d.src);
  const int64_t e0 = ((blk - d.blk0) * 256 + threadIdx.x) * 8;
  if (e0 >= (int64_t)d.R * d.Cc) return;
  const int r = (int)(e0 / d.Cc), c0 = (int)(e0 - (int64_t)r * d.Cc);
  float val[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = c0 + q;
    const int64_t e = e0 + q;
    float v = 0.f;
    switch (d.kind) {
      case PK_FRAG: {  // MFMA fragment order [column tile n/16][k-step k/32][lane = (k%32)/8*16 + n%16][k%8]: the 64 lanes
        // of a wave read one fragment as ONE contiguous 64 x sizeof(fragment) block (8 whole cache lines for bf16)
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), blk = (int)(e >> 9), ksteps = d.Cc >> 5;
        const int tile = blk / ksteps, ks = blk - tile * ksteps;
        const int n = tile * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
        if (n < d.N && k < d.K) {
          if (d.Cin > 0) { const int tap = k / d.Cin, ci = k - tap * d.Cin; v = src[(int64_t)n * d.K + ci * d.taps + tap]; }  // NHWC k order
          else v = src[(int64_t)n * d.K + k];
        }
      } break;
      case PK_FRAGT: {  // PK_FRAG of the TRANSPOSED weight (data-grad GEMMs): rows = input features (NHWC order when
        // d.Cin > 0, like PK_CONV_NHWC_T), contraction index = output feature
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), blk = (int)(e >> 9), ksteps = d.Cc >> 5;
        const int tile = blk / ksteps, ks = blk - tile * ksteps;
        const int row = tile * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + j;
        if (row < d.K && k < d.N) {
          if (d.Cin > 0) { const int tap = row / d.Cin, ci = row - tap * d.Cin; v = src[(int64_t)k * d.K + ci * d.taps + tap]; }
          else v = src[(int64_t)k * d.K + row];
        }
      } break;
      case PK_FRAGP:     // fragment order with the wave-per-sample kernels' k permutation (csrc/wps.h): slot (g = lane >> 4, j)
      case PK_FRAGPT: {  // holds k = 32 ks + 16 (j >> 2) + 4 g + (j & 3). FRAGP: rows = output features n, k = input feature
        // (forward GEMMs); FRAGPT: rows = input features, k = output feature (data-grad GEMMs: the transposed weight)
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), blk = (int)(e >> 9), ksteps = d.Cc >> 5;
        const int tile = blk / ksteps, ks = blk - tile * ksteps;
        const int row = tile * 16 + (lane & 15), k = ks * 32 + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        if (d.kind == PK_FRAGP) { if (row < d.N && k < d.K) v = src[(int64_t)row * d.K + k]; }
        else if (row < d.K && k < d.N) v = src[(int64_t)k * d.K + row];
      } break;
      // (d.px: the source's K axis starts at packed column / row px — the vision-only Transformer's first head layer inside
      // the 128-wide pooled operand of the wave-per-sample kernels, csrc/wps.h; 0 everywhere else)
      case PK_NT: if (r < d.N && c >= d.px && c - d.px < d.K) v = src[(int64_t)r * d.K + (c - d.px)]; break;
      case PK_T: if (r >= d.px && r - d.px < d.K && c < d.N) v = src[(int64_t)c * d.K + (r - d.px)]; break;
      case PK_CONV_NHWC:  // dst[n][tap*Cin+ci] = W[n][ci][tap]
        if (r < d.N && c < d.K) { const int tap = c / d.Cin, ci = c - tap * d.Cin; v = src[(int64_t)r * d.K + ci * d.taps + tap]; }
        break;
      case PK_CONV_NHWC_T:  // dst[tap*Cin+ci][n] = W[n][ci][tap]
        if (r < d.K && c < d.N) { const int tap = r / d.Cin, ci = r - tap * d.Cin; v = src[(int64_t)c * d.K + ci * d.taps + tap]; }
        break;
      case PK_CONV_DGRAD: {  // dst[ci][(a*TW+bb)*N + n] = W[n][ci][py+s*a][px+s*bb]
        const int kk = d.TW * d.TW * d.N;
        if (r < d.Cin && c < kk) {
          const int tap = c / d.N, n = c - tap * d.N;
          const int a = tap / d.TW, bb = tap - a * d.TW;
          const int ky = d.py + d.s * a, kx = d.px + d.s * bb;
          v = src[(int64_t)n * d.K + r * d.taps + ky * d.KW + kx];
        }
      } break;
    }
    val[q] = v;
  }
  T* o = dst + d.dst_off + e0;
  st4(o, val[0], val[1], val[2], val[3]);
  st4(o + 4, val[4], val[5], val[6], val[7]);
}
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const PackDesc* __restrict__ descs, int nd, T* __restrict__ dst) {
  pack_body<T>(descs, nd, dst, (int64_t)blockIdx.x);
}
// The first launch of an update: block 0 opens the update (upd_begin_body), the others refresh the critic's operand-type
// weight copies — two independent latency chains that used to be two launches back to back.
template <typename T>
__global__ __launch_bounds__(256) void begin_pack_kernel(UpdCtl* c, const int* __restrict__ rowidx_all, int n, int* rowidx_cur,
                                                         float* stats_cur, const float* __restrict__ adv,
                                                         const PackDesc* __restrict__ descs, int nd, T* __restrict__ dst) {
  if (blockIdx.x == 0) upd_begin_body(c, rowidx_all, n, rowidx_cur, stats_cur, adv);
  else pack_body<T>(descs, nd, dst, (int64_t)blockIdx.x - 1);
}

// --------------------------------------------------------------------------------- GAE
// torchrl/replay_buffers/on_policy.py:17-45 — fp64, same expression order as the numpy code, FMA contraction
// off, so the result is bit-identical to the reference's (then optionally cast once to fp32, as ppo.py:138,140
// does). Two phases: (1) everything that does not depend on the recursion — delta_t, the carry coefficient
// ((1-term)*gamma)*tau and the time-limit factor — is computed fully parallel over [T,E]; (2) the 3-flop
// recursion runs one lane per env over t with the loads of 8 steps in flight (they do not depend on A).
// tl_stride_e: 0 when _time_limits is [T,1] (broadcast over envs), 1 when [T,E,1].
__global__ __launch_bounds__(256) void gae_prep_kernel(const double* __restrict__ rewards, const double* __restrict__ values,
                                                       const double* __restrict__ terminals,
                                                       const double* __restrict__ time_limits, int tl_stride_e,
                                                       const double* __restrict__ last_value, int T, int E, double gamma,
                                                       double tau, int use_tl, double* __restrict__ delta,
                                                       double* __restrict__ coef, double* __restrict__ tlm) {
#pragma clang fp contract(off)
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (int64_t)T * E) return;
  const int t = (int)(o / E), e = (int)(o - (int64_t)t * E);
  const double c = (1.0 - terminals[o]) * gamma;
  const double vnext = (t == T - 1) ? last_value[e] : values[o + E];
  double d = rewards[o] + c * vnext;
  d = d - values[o];
  delta[o] = d;
  coef[o] = c * tau;
  tlm[o] = use_tl ? (1.0 - time_limits[tl_stride_e ? o : t]) : 1.0;
}

__global__ void gae_scan_kernel(const double* __restrict__ delta, const double* __restrict__ coef,
                                const double* __restrict__ tlm, const double* __restrict__ values, int T, int E,
                                int use_tl, double* __restrict__ advs, double* __restrict__ rets,
                                float* __restrict__ advs32, float* __restrict__ rets32) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double A = 0.0;
  for (int t0 = T - 1; t0 >= 0; t0 -= 8) {
    double d[8], c[8], m[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 - j;
      const int64_t o = (int64_t)(t < 0 ? 0 : t) * E + e;
      d[j] = delta[o]; c[j] = coef[o]; m[j] = tlm[o]; v[j] = values[o];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 - j;
      if (t < 0) break;
      const int64_t o = (int64_t)t * E + e;
      A = d[j] + c[j] * A;
      if (use_tl) A = A * m[j];
      const double r = A + v[j];
      advs[o] = A;
      rets[o] = r;
      if (advs32 != nullptr) { advs32[o] = (float)A; rets32[o] = (float)r; }
    }
  }
}

// discount_reward (PPO(gae=False)): torchrl/replay_buffers/on_policy.py:47-71 — fp64, the numpy expression order, FMA
// contraction off: bit-identical to the reference. One lane per env walks t = T-1 .. 0 with 8 steps' loads in flight.
//   time-limit filter:  R = (r + (((1 - term) * gamma) * R) * (1 - tl)) + tl * V;   else:  R = r + ((1 - term) * gamma) * R
//   advs = R - V, estimate_returns = R.
__global__ void discount_scan_kernel(const double* __restrict__ rewards, const double* __restrict__ values,
                                     const double* __restrict__ terminals, const double* __restrict__ time_limits,
                                     int tl_stride_e, const double* __restrict__ last_value, int T, int E, double gamma,
                                     int use_tl, double* __restrict__ advs, double* __restrict__ rets,
                                     float* __restrict__ advs32, float* __restrict__ rets32) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double R = last_value[e];
  for (int t0 = T - 1; t0 >= 0; t0 -= 8) {
    double r[8], c[8], m[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 - j < 0 ? 0 : t0 - j;
      const int64_t o = (int64_t)t * E + e;
      r[j] = rewards[o]; v[j] = values[o];
      c[j] = (1.0 - terminals[o]) * gamma;
      m[j] = use_tl ? time_limits[tl_stride_e ? o : t] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 - j;
      if (t < 0) break;
      const int64_t o = (int64_t)t * E + e;
      double x = c[j] * R;
      if (use_tl) {
        x = x * (1.0 - m[j]);
        x = r[j] + x;
        R = x + m[j] * v[j];
      } else {
        R = r[j] + x;
      }
      const double a = R - v[j];
      advs[o] = a;
      rets[o] = R;
      if (advs32 != nullptr) { advs32[o] = (float)a; rets32[o] = (float)R; }
    }
  }
}

// Running observation normaliser of the vectorised env: NormObsWithImg.observation (vision4leg/get_env.py:58-67) /
// NormObs.observation (torchrl/env/base_wrapper.py:119-122) on the [E][S] proprio block of one env step —
// Normalizer.update_estimate (base_wrapper.py:77-84: merge the batch mean / variance over the E envs into the running
// statistics with update_mean_var_count, :44-61) when `update`, then Normalizer.filt (:93-96) with the new statistics.
// fp64, the numpy expression order (axis-0 reductions run row after row), FMA contraction off: bit-identical to the
// reference; out32 is the fp32 cast the collector's torch.Tensor(ob) makes (collector/on_policy.py:93). Block 0 owns
// the statistics (one lane per proprio dimension: its E raw values are one coalesced column walk, nothing is shared
// between lanes but the count); blocks 1.. move the step's depth stack into the same fp32 observation rows (the
// reference's per-step np.hstack of a 16 K-float image row), converting from fp64 when the env hands that over.
struct ObsNorm {
  const double* raw; int64_t ld_raw;       // [E][S] raw proprio rows
  double *mean, *var, *count;              // [S], [S], [1] running statistics (updated in place)
  double clip; int E, S, update;
  float* out32; int64_t ld32;              // normalised rows as fp32 (nullable)
  double* out64; int64_t ld64;             // ... and as fp64 (nullable)
  const void* img; int img_f64; int64_t ld_img, img_elems;  // [E][img_elems] fp32 / fp64 depth stack (nullable)
  float* img_out; int64_t ld_img_out;
};

__global__ __launch_bounds__(256) void obs_norm_kernel(ObsNorm p) {
#pragma clang fp contract(off)
  if (blockIdx.x != 0) {
    const int64_t n = (int64_t)p.E * p.img_elems, step = (int64_t)(gridDim.x - 1) * 256;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x; i < n; i += step) {
      const int64_t e = i / p.img_elems, c = i - e * p.img_elems;
      const float v = p.img_f64 ? (float)((const double*)p.img)[e * p.ld_img + c] : ((const float*)p.img)[e * p.ld_img + c];
      p.img_out[e * p.ld_img_out + c] = v;
    }
    return;
  }
  const double cnt = *p.count, bc = (double)p.E;
  __syncthreads();  // every lane holds the old count before lane 0 replaces it
  for (int d = threadIdx.x; d < p.S; d += 256) {
    const double* col = p.raw + d;
    double m = p.mean[d], v = p.var[d];
    // the column's E values, 8 loads in flight at a time (the adds below are sequential by contract, the loads are not)
    auto walk = [&](auto&& f) {
      for (int e0 = 0; e0 < p.E; e0 += 8) {
        double c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = col[(int64_t)(e0 + j < p.E ? e0 + j : e0) * p.ld_raw];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (e0 + j < p.E) f(e0 + j, c[j]);
      }
    };
    if (p.update) {
      double s = 0.0;
      walk([&](int e, double c) { s = e == 0 ? c : s + c; });
      const double bm = s / bc;
      double q = 0.0;
      walk([&](int e, double c) {
        const double x = c - bm;
        q = e == 0 ? x * x : q + x * x;
      });
      const double bv = q / bc;
      const double delta = bm - m, tot = cnt + bc;
      const double new_mean = m + delta * bc / tot;
      const double m_a = v * cnt, m_b = bv * bc;
      const double M2 = m_a + m_b + delta * delta * cnt * bc / tot;
      m = new_mean;
      v = M2 / tot;
      p.mean[d] = m;
      p.var[d] = v;
    }
    const double den = sqrt(v) + 1e-4;
    walk([&](int e, double c) {
      double y = (c - m) / den;
      y = y < -p.clip ? -p.clip : y;
      y = y > p.clip ? p.clip : y;
      if (p.out64 != nullptr) p.out64[e * p.ld64 + d] = y;
      if (p.out32 != nullptr) p.out32[e * p.ld32 + d] = (float)y;
    });
  }
  if (p.update && threadIdx.x == 0) *p.count = cnt + bc;
}

}  // namespace v4l
