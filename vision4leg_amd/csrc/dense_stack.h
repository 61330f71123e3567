// The dense stack of the NatureCNN nets in training — networks/nets.py:194-262 (ImpalaFuseEncoder... NatureFuseEncoder: visual
// projector over conv3's flatten, concat with the proprio MLP's output) + base.py:345-385 / nets.py:16-55 (the 256-256 head) and the
// vision-only variant nets.py:133-191 (head straight over the flatten) — as ONE launch per direction.
//
// Round 3 ran it as gemm_nt_deep_kernel launches: 4 forward and 5 data-grad launches per net-pass, each 5 - 8 us for
// 0.03 - 0.5 GFLOP (tools/update_timeline.py, profiles/r4_update_timeline_cnn.txt: 73 of 260 us). Chaining the same blocks
// through device-side counters inside one launch (tried in round 4) is SLOWER (forward 38 -> 74 us, backward 35 -> 114 us): an
// agent-scope release / acquire per block is an L2 write-back / invalidate of the block's XCD, 400 of them per launch. Here
// a block keeps its 16 MT rows ON CHIP through the whole stack instead: activations live in LDS in the operand type, every
// stage streams its weight as whole MFMA fragments (the PK_FRAG / PK_FRAGT packs the rollout kernels and gemm_nt_deep read: one
// contiguous 1 KB read per fragment, up to 16 k-steps x 4 tiles = 64 KB in flight per wave), wave w owns column tiles
// 4w .. 4w+3 of every 256-wide slice, and only what the backward pass / the weight-grads read is written to HBM (the same
// buffers as before).
// The floor is the per-CU fetch rate: 896 KB (forward) / 1 056 KB (backward) of weights + ~150 / 290 KB of rows, masks and saves
// through one CU at the measured 31.8 B/clk = 14 / 18 us; measured at B = 1024 (64 blocks of 16 rows): 17.4 / 24.1 us,
// against 37.6 / 35 us for the 4 + 5 launches (NatureCNN update 521 -> 469 us, profiles/r4_update_timeline_cnn.txt). What
// mattered on the way: 4 k-steps in flight per wave gave 38 / 53 us (10 B/clk per CU: latency-bound), 16 k-steps and the next
// stage's ring requested ahead of the epilogue 24.8 / 39.5 us, 16 instead of 32 rows per block and two slices' rings in
// flight in the backward the rest.
// Arithmetic: the k order of every output element (ascending k, one accumulator, K = 32 per MFMA step), the operand rounding
// points (fp32 -> T when a row is staged) and the epilogue expressions are gemm_nt_deep_kernel's, so the results are the same
// bits (tests/test_gpu_parity.py::test_fused_dense_stack_equals_layer_by_layer).
#pragma once
#include "infer.h"

namespace v4l {

// acc[mt][j] += (A[row tile mt] . W[column tile nt[j]]^T)^T over KS k-steps; A: LDS rows (AT = T, or float converted at the
// fragment load), W: fragment-order pack [tiles][KS][64 lanes] of 8-element fragments. The first PD k-steps of the weight
// fragments are requested ahead of time (ds_prefetch) — before the rows are staged, or before the previous stage's epilogue —
// and a load is issued per step consumed: one CU pulls the whole stack's weights through its own L1, so the bytes in flight
// per wave set the rate (PD = 4: 10 B/clk per CU, measured; the bf16 kernels have the registers for 16).
template <typename T> struct DsPd { static constexpr int value = sizeof(T) == 2 ? 16 : 4; };
template <typename T, int NTW, int KS> struct DsRing {
  static constexpr int PD = KS < DsPd<T>::value ? KS : DsPd<T>::value;
  typename Frag<T>::type fb[PD][NTW];
};
template <typename T, int NTW, int KS>
__device__ __forceinline__ DsRing<T, NTW, KS> ds_prefetch(const void* __restrict__ Wf, const int (&nt)[NTW], int lane) {
  typedef typename Frag<T>::type frag_t;
  DsRing<T, NTW, KS> ring;
#pragma unroll
  for (int d = 0; d < DsRing<T, NTW, KS>::PD; ++d)
#pragma unroll
    for (int j = 0; j < NTW; ++j) ring.fb[d][j] = (reinterpret_cast<const frag_t*>(Wf) + (size_t)nt[j] * KS * 64 + lane)[d * 64];
  return ring;
}
template <typename T, int MT, int NTW, int KS, typename AT>
__device__ __forceinline__ void ds_gemm(f32x4 (&acc)[MT][NTW], const AT* sA, int lda, const void* __restrict__ Wf,
                                        const int (&nt)[NTW], int lane, DsRing<T, NTW, KS>& ring) {
  typedef typename Frag<T>::type frag_t;
  constexpr int PD = DsRing<T, NTW, KS>::PD;
  const int fr = lane & 15, fg = (lane >> 4) * 8;
  const frag_t* W[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) W[j] = reinterpret_cast<const frag_t*>(Wf) + (size_t)nt[j] * KS * 64 + lane;
  frag_t (&fb)[PD][NTW] = ring.fb;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    frag_t cur[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) cur[j] = fb[ks % PD][j];
    if (ks + PD < KS) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) fb[ks % PD][j] = W[j][(ks + PD) * 64];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      frag_t fa;
      if constexpr (sizeof(AT) == sizeof(T)) fa = *reinterpret_cast<const frag_t*>(sA + (mt * 16 + fr) * lda + ks * 32 + fg);
      else fa = afrag<T>(reinterpret_cast<const float*>(sA) + (mt * 16 + fr) * lda + ks * 32 + fg);
#pragma unroll
      for (int j = 0; j < NTW; ++j) mma_k32(acc[mt][j], cur[j], fa);
    }
  }
}

// rows per block: 16. A block's time is its CU pulling the stack's ~1 MB of weights plus its rows' activations, masks and saves
// through one L1 (measured: 32 rows 24.8 / 39.5 us forward / backward); more rows per block would only add to that, fewer
// blocks than CUs leave the others idle anyway (64 blocks at B = 1024)
template <typename T> struct DsCfg { static constexpr int MT = 1; };

// ------------------------------------------------------------------------------------------------------------ forward
struct DsFwd {
  const float* c3;          // [n][1024] conv3's NHWC flatten (post-ReLU), fp32
  float* cat;               // fuse net: [n][512] = [visual projector's output | proprio MLP's output (already there)]
  const void *wp, *w0, *w1, *w2;   // PK_FRAG packs: projector [16][32], head fc0 [16][16 | 32], fc1 [16][8], last [1][8]
  const float *bp, *b0, *b1, *b2;
  float *h0, *h1, *out;     // [n][256], [n][256], [n][OUT_LD]
  int n, nout;
};
template <typename T> struct DsFwdLds {
  static constexpr int MT = DsCfg<T>::MT, P = InfLd<T>::PAD;
  static constexpr int LDI = 1024 + P, LDC = 512 + P, LDH = 256 + P;
  static constexpr size_t in_b = (size_t)16 * MT * LDI * sizeof(T), cat_b = (size_t)16 * MT * LDC * sizeof(T),
                          h_b = (size_t)16 * MT * LDH * sizeof(T);
  static constexpr size_t bytes = in_b + cat_b + 2 * h_b;
};
// FUSE: c3 -> projector -> [cat] -> fc0 -> fc1 -> last;   !FUSE (vision-only net): c3 -> fc0 -> fc1 -> last
template <typename T, bool FUSE>
__global__ __launch_bounds__(256) void dense_stack_fwd_kernel(DsFwd a) {
  typedef DsFwdLds<T> LY;
  constexpr int MT = LY::MT, LDI = LY::LDI, LDC = LY::LDC, LDH = LY::LDH, R = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sIn = reinterpret_cast<T*>(smem);
  T* sCat = reinterpret_cast<T*>(smem + LY::in_b);
  T* sH0 = reinterpret_cast<T*>(smem + LY::in_b + LY::cat_b);
  T* sH1 = reinterpret_cast<T*>(smem + LY::in_b + LY::cat_b + LY::h_b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, qr = (lane >> 4) * 4;
  const int r0 = blockIdx.x * R, n = a.n;
  const int nt4[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  const int nt1[1] = {0};
  DsRing<T, 4, 32> ring_in = ds_prefetch<T, 4, 32>(FUSE ? a.wp : a.w0, nt4, lane);  // ahead of the rows
  // rows >= n re-read the last row: finite operands, their results are never stored
  // (thread t stages columns 4t .. 4t+3 of every row: 8 rows' loads in flight at a time)
#pragma unroll 1
  for (int rb = 0; rb < R; rb += 8) {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(a.c3 + (int64_t)min(r0 + rb + i, n - 1) * 1024 + tid * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) st4(sIn + (rb + i) * LDI + tid * 4, v[i].x, v[i].y, v[i].z, v[i].w);
  }
  if constexpr (FUSE) {
    float4 v[R / 4];  // R x 64 float4: thread t takes rows (t >> 6) + 4 i, columns 256 + 4 (t & 63)
#pragma unroll
    for (int i = 0; i < R / 4; ++i)
      v[i] = *reinterpret_cast<const float4*>(a.cat + (int64_t)min(r0 + (tid >> 6) + 4 * i, n - 1) * 512 + 256 + (tid & 63) * 4);
#pragma unroll
    for (int i = 0; i < R / 4; ++i) st4(sCat + ((tid >> 6) + 4 * i) * LDC + 256 + (tid & 63) * 4, v[i].x, v[i].y, v[i].z, v[i].w);
  }
  __syncthreads();
  f32x4 acc[MT][4];
  // y = relu(acc + b) -> LDS rows (operand type) and the fp32 rows the backward pass reads
  auto finish = [&](const float* __restrict__ bias, T* dst, int ldd, float* __restrict__ save, int lds) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n4 = nt4[j] * 16 + qr;
      const float4 bv = *reinterpret_cast<const float4*>(bias + n4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float v0 = fmaxf(acc[mt][j][0] + bv.x, 0.f), v1 = fmaxf(acc[mt][j][1] + bv.y, 0.f);
        const float v2 = fmaxf(acc[mt][j][2] + bv.z, 0.f), v3 = fmaxf(acc[mt][j][3] + bv.w, 0.f);
        st4(dst + (mt * 16 + fr) * ldd + n4, v0, v1, v2, v3);
        const int row = r0 + mt * 16 + fr;
        if (row < n) st4(save + (int64_t)row * lds + n4, v0, v1, v2, v3);
      }
    }
  };
  zero_acc(acc);
  if constexpr (FUSE) {
    ds_gemm<T, MT, 4, 32>(acc, sIn, LDI, a.wp, nt4, lane, ring_in);
    DsRing<T, 4, 16> ring_0 = ds_prefetch<T, 4, 16>(a.w0, nt4, lane);  // ahead of the epilogue's stores and the barrier
    finish(a.bp, sCat, LDC, a.cat, 512);
    __syncthreads();
    zero_acc(acc);
    ds_gemm<T, MT, 4, 16>(acc, sCat, LDC, a.w0, nt4, lane, ring_0);
  } else {
    ds_gemm<T, MT, 4, 32>(acc, sIn, LDI, a.w0, nt4, lane, ring_in);
  }
  DsRing<T, 4, 8> ring_1 = ds_prefetch<T, 4, 8>(a.w1, nt4, lane);
  finish(a.b0, sH0, LDH, a.h0, 256);
  __syncthreads();
  zero_acc(acc);
  ds_gemm<T, MT, 4, 8>(acc, sH0, LDH, a.w1, nt4, lane, ring_1);
  DsRing<T, 1, 8> ring_2 = ds_prefetch<T, 1, 8>(a.w2, nt1, lane);
  finish(a.b1, sH1, LDH, a.h1, 256);
  __syncthreads();
  if (wave < MT) {  // last linear: one (padded) column tile, row tile `wave`; columns >= nout of the padded row are zeros
    f32x4 o[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
    ds_gemm<T, 1, 1, 8>(o, sH1 + wave * 16 * LDH, LDH, a.w2, nt1, lane, ring_2);
    const int row = r0 + wave * 16 + fr;
    if (row < n) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = qr + r;
        v[r] = c < a.nout ? o[0][0][r] + a.b2[c < a.nout ? c : 0] : 0.f;
      }
      st4(a.out + (int64_t)row * OUT_LD + qr, v[0], v[1], v[2], v[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ backward
struct DsBwd {
  const float* dout;        // [n][OUT_LD] d(loss)/d(head output)
  const void *w2t, *w1t, *w0t, *wpt, *wf2t;  // PK_FRAGT packs: [16][2], [16][8], [32 | 64][8], projector [64][8], proprio fc2 [16][8]
  const float *h1, *h0;     // [n][256] post-ReLU head activations (ReLU masks)
  const float* cat;         // fuse net: [n][512] post-ReLU [visual projector | proprio MLP] outputs
  const float* c3;          // [n][1024] conv3's post-ReLU flatten
  const float* e0;          // fuse net: [n][256] proprio MLP's first activation
  float *dh1, *dh0;         // [n][256] masked data-grads (dY operands of the head's weight-grads)
  float* dcat;              // fuse net: [n][512] grad w.r.t. the concat, UN-masked (its readers apply the concat's ReLU mask)
  float* dc3;               // [n][1024] grad w.r.t. conv3's pre-activation (masked)
  float* de0;               // fuse net: [n][256] masked
  int n;
};
template <typename T> struct DsBwdLds {
  static constexpr int MT = DsCfg<T>::MT, P = InfLd<T>::PAD;
  static constexpr int LDX = 64 + 4, LDH = 256 + P, LDC = 512 + P;
  static constexpr size_t dt_b = (size_t)16 * MT * LDX * 4, h_b = (size_t)16 * MT * LDH * sizeof(T),
                          cat_b = (size_t)16 * MT * LDC * sizeof(T);
  static constexpr size_t bytes = dt_b + 2 * h_b + cat_b;
};
// FUSE: dout -> W2' -> dh1 -> W1' -> dh0 -> W0' -> dcat -> { projector' -> dc3 ; proprio fc2' -> de0 }
// !FUSE: dout -> W2' -> dh1 -> W1' -> dh0 -> W0' -> dc3
template <typename T, bool FUSE>
__global__ __launch_bounds__(256) void dense_stack_bwd_kernel(DsBwd a) {
  typedef DsBwdLds<T> LY;
  constexpr int MT = LY::MT, LDX = LY::LDX, LDH = LY::LDH, LDC = LY::LDC, R = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dt = reinterpret_cast<float*>(smem);                       // [R][LDX] dout rows, zero beyond OUT_LD and for rows >= n
  T* sD1 = reinterpret_cast<T*>(smem + LY::dt_b);                   // [R][LDH] dh1
  T* sD0 = reinterpret_cast<T*>(smem + LY::dt_b + LY::h_b);         // [R][LDH] dh0
  T* sDc = reinterpret_cast<T*>(smem + LY::dt_b + 2 * LY::h_b);     // [R][LDC] dcat, ReLU-masked by the concat
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, qr = (lane >> 4) * 4;
  const int r0 = blockIdx.x * R, n = a.n;
  // After the first stage (K = 64) the stack is a sequence of 256-column slices with K = 256: W1', then W0' (2 slices of the
  // concat's 512 features / 4 of the flatten's 1024), then — fuse net — the projector' (4 slices) and the proprio fc2'. A slice's
  // 32 weight fragments per wave (all 8 k-steps of its 4 tiles) form one ring; TWO slices' rings are in flight at any time.
  constexpr int NS0 = FUSE ? 2 : 4, NS = 1 + NS0 + (FUSE ? 5 : 0);
  auto slice_w = [&](int q) -> const void* {
    return q == 0 ? a.w1t : q <= NS0 ? a.w0t : q <= NS0 + 4 ? a.wpt : a.wf2t;
  };
  auto slice_ch = [&](int q) { return q == 0 ? 0 : q <= NS0 ? q - 1 : q <= NS0 + 4 ? q - 1 - NS0 : 0; };
  auto prefetch = [&](int q) {
    const int ch = slice_ch(q);
    const int nt[4] = {ch * 16 + wave * 4, ch * 16 + wave * 4 + 1, ch * 16 + wave * 4 + 2, ch * 16 + wave * 4 + 3};
    return ds_prefetch<T, 4, 8>(slice_w(q), nt, lane);
  };
  const int nt0[4] = {wave * 4, wave * 4 + 1, wave * 4 + 2, wave * 4 + 3};
  DsRing<T, 4, 2> ring_2 = ds_prefetch<T, 4, 2>(a.w2t, nt0, lane);
  DsRing<T, 4, 8> ring[2] = {prefetch(0), prefetch(1)};
  for (int idx = tid; idx < R * 16; idx += 256) {
    const int r = idx >> 4, c4 = (idx & 15) * 4, row = r0 + r;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < n && c4 < OUT_LD) v = *reinterpret_cast<const float4*>(a.dout + (int64_t)row * OUT_LD + c4);
    *reinterpret_cast<float4*>(dt + r * LDX + c4) = v;
  }
  __syncthreads();
  f32x4 acc[MT][4];
  // columns of slice `ch` owned by this wave: ReLU mask from the saved activation `m` -> optional LDS rows (operand type,
  // `dst`), optional fp32 rows in HBM (`save`; `save_masked`: with the mask, else the raw sums)
  auto finish = [&](int ch, const float* __restrict__ m, int ldm, T* dst, int ldd, float* __restrict__ save, int lds,
                    bool save_masked) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = r0 + mt * 16 + fr;
      const int64_t mrow = min(row, n - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n4 = (ch * 16 + wave * 4 + j) * 16 + qr;
        const float4 mk = *reinterpret_cast<const float4*>(m + mrow * ldm + n4);
        const float u0 = acc[mt][j][0], u1 = acc[mt][j][1], u2 = acc[mt][j][2], u3 = acc[mt][j][3];
        const float d0 = mk.x > 0.f ? u0 : 0.f, d1 = mk.y > 0.f ? u1 : 0.f, d2 = mk.z > 0.f ? u2 : 0.f, d3 = mk.w > 0.f ? u3 : 0.f;
        if (dst != nullptr) st4(dst + (mt * 16 + fr) * ldd + n4, d0, d1, d2, d3);
        if (row < n) {
          if (save_masked) st4(save + (int64_t)row * lds + n4, d0, d1, d2, d3);
          else st4(save + (int64_t)row * lds + n4, u0, u1, u2, u3);
        }
      }
    }
  };
  zero_acc(acc);
  ds_gemm<T, MT, 4, 2>(acc, dt, LDX, a.w2t, nt0, lane, ring_2);
  finish(0, a.h1, 256, sD1, LDH, a.dh1, 256, true);
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (q == 0 || q == 1 || q == NS0 + 1) __syncthreads();  // the operand rows of this stage are complete
    const int ch = slice_ch(q);
    const int nt[4] = {ch * 16 + wave * 4, ch * 16 + wave * 4 + 1, ch * 16 + wave * 4 + 2, ch * 16 + wave * 4 + 3};
    const T* sA = q == 0 ? sD1 : q <= NS0 ? sD0 : q <= NS0 + 4 ? sDc : sDc + 256;
    const int lda = q <= NS0 ? LDH : LDC;
    zero_acc(acc);
    ds_gemm<T, MT, 4, 8>(acc, sA, lda, slice_w(q), nt, lane, ring[q & 1]);
    if (q + 2 < NS) ring[q & 1] = prefetch(q + 2);
    if (q == 0) finish(0, a.h0, 256, sD0, LDH, a.dh0, 256, true);
    else if (q <= NS0) {
      if constexpr (FUSE) finish(ch, a.cat, 512, sDc, LDC, a.dcat, 512, false);
      else finish(ch, a.c3, 1024, (T*)nullptr, 0, a.dc3, 1024, true);
    } else if (q <= NS0 + 4) finish(ch, a.c3, 1024, (T*)nullptr, 0, a.dc3, 1024, true);
    else finish(0, a.e0, 256, (T*)nullptr, 0, a.de0, 256, true);
  }
}

}  // namespace v4l
