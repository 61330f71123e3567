// Host-side plans and kernel-launch drivers: one v4l_net = one reference network (state MLP, NatureCNN fuse
// net or LocoTransformer), one v4l_trainer = the PPO minibatch update over (pf, vf, target_pf).
#pragma once
#include <string>
#include <vector>

#include "../../include/v4l_hip.h"
#include "elem.h"
#include "gemm.h"

namespace v4l { struct RowsChain; struct FbLoss; }

namespace v4l {

struct ParamInfo {
  std::string name;
  int ndim;
  int64_t shape[4];
  int64_t numel;
  int64_t goff;  // offset in the flat grad / moment buffers
};

struct Lin {         // nn.Linear (or the 1x1 up-conv): y = x W^T + b, W [N][K]
  int w = -1, b = -1;
  int N = 0, K = 0;
  int Np = 0, Kp = 0;     // forward pack [Np][Kp]
  int64_t pk = 0;
  int Rt = 0, Ct = 0;     // transposed pack [Rt = pad16(K)][Ct = pad64(N)] for the data-grad
  int64_t pkt = 0;
  int64_t pkf = -1;       // fragment-order pack [Np/16][Kp/32][64 lanes][8] (rollout kernels, gemm_nt_deep_kernel), -1: none
  int64_t pkft = -1;      // the same of W^T: [Rt/16][Ct/32][64][8] (data-grads through gemm_nt_deep_kernel), -1: none
  int64_t pkp = -1, pkpt = -1;  // k-permuted fragment-order packs of W / W^T (wave-per-sample layer kernels, csrc/wps.h), -1: none
  int64_t pko = -1, pkto = -1;  // vision-only Transformer, first head layer: [N][128] / [128][N] row-major packs with the 64
                                // real K entries at offset 64 (the pooled operand's second half), -1: none
  int cin = 0, taps = 0;  // > 0: input is an NHWC flatten, packed k = tap*cin + c  (PyTorch k = c*taps + tap)
  bool need_dgrad = true;
  std::string tag_fwd, tag_wgrad, tag_dgrad;  // profiler labels
};

struct Conv {        // nn.Conv2d, square kernel, no padding. Activations NHWC, conv1 reads the CHW depth stack.
  int w = -1, b = -1;
  int Cin = 0, Cout = 0, KH = 0, stride = 0, IH = 0, OH = 0;
  int K = 0, Kp = 0, Np = 0;
  int64_t pk = 0;
  int64_t pkf = -1;       // fragment-order pack (rollout_encoder2_kernel), -1: none
  bool chw = false;       // conv1
  int ncls = 0;           // stride*stride parity classes of the gather-form data-grad
  int Kd = 0, Kdp = 0, Rd = 0;
  int64_t pkd[4] = {0, 0, 0, 0};
};

struct LNp { int g = -1, b = -1; };
struct TLayer { Lin inproj, outproj, ff1, ff2; LNp ln1, ln2; };

// fused layer kernels keep xin (copy of the layer input), ctx, x1, f in the contraction type T inside these fp32-sized slots
struct LayerWs { int64_t qkv, P, ctx, xh1, rs1, x1, f, xh2, rs2, xin; };
// backward: every layer keeps its own gradient tensors alive until the grouped weight-grad launch at the end
struct LayerBw { int64_t dz2, df, dx1, dz1, dctx, dqkv; };
struct Layout {
  int n = 0;
  int64_t c1 = 0, c2 = 0, c3 = 0;
  std::vector<int64_t> eh;       // encoder MLP activations
  int64_t vis = 0;               // CNN: concat buffer [n][visual_dim + enc_last]
  std::vector<int64_t> x;        // LOCO: token tensors x[0..L]
  int64_t xfin = 0, xhF = 0, rsF = 0, dxfin = 0;    // pytorch_encoder: the final LayerNorm's output, xhat / rstd, grad w.r.t. its output
  int64_t x0raw = 0, xh0 = 0, rs0 = 0, dx0raw = 0;  // token_norm: the encoder's tokens (x[0] = their LayerNorm), xhat / rstd, grad w.r.t. them
  std::vector<LayerWs> lw;
  int64_t ytmp = 0;              // [R][64] sub-layer output before add+LN
  int64_t pooled = 0;
  std::vector<int64_t> hh;       // head hidden activations
  int64_t out = 0, dout = 0;     // [n][OUT_LD]
  // backward: one buffer per gradient tensor (nothing is overwritten before the deferred weight-grads ran)
  std::vector<int64_t> dhh;      // grad w.r.t. head hidden pre-activations
  std::vector<int64_t> deh;      // grad w.r.t. encoder-MLP hidden pre-activations
  std::vector<LayerBw> lb;
  std::vector<int64_t> dxl;      // LOCO: grad w.r.t. x[l], l = 0..L
  int64_t dhc = 0;               // [n][maxwidth] hand-off between two stacks (head -> encoder / concat)
  int64_t dpool = 0, dc3 = 0, dc2 = 0, dc1 = 0;
  std::vector<int64_t> wps_wg, wps_tk;  // per layer: fragment-order weight-grad operand blocks of the wave-per-sample kernels
  int64_t slab = 0;              // weight-grad partial slabs
  int64_t total = 0;
};

}  // namespace v4l

struct v4l_net {
  v4l_net_cfg cfg;
  int Sp = 0;
  int ntok = v4l::NTOK;  // tokens per sample: 17 (LocoTransformer), 16 (vision-only Transformer)
  // c1 / c2 of the training forward in the operand type (round 5): asked for by v4l_net_forward(train = 1) — the trainer's and
  // the tests' forward / backward pairs —, granted by forward_t when the persistent training encoder writes them and the fused
  // conv backward will read them (bf16, shipped conv geometry, no test taps), remembered for the backward of the same pass
  bool want_acts16 = false;
  const float* acts16_c1 = nullptr;  // the c1 slot (= the workspace) whose c1 / c2 currently hold the operand type; null: none
  bool vis_only() const { return cfg.kind == V4L_NET_CNN_VIS || cfg.kind == V4L_NET_LOCO_VIS; }
  bool is_tf() const { return cfg.kind == V4L_NET_LOCO || cfg.kind == V4L_NET_LOCO_VIS; }
  std::vector<v4l::ParamInfo> params;
  int64_t total_params = 0;
  // layers
  v4l::Conv conv[3];
  v4l::Lin upconv, proj;            // LOCO: depth_up_conv, state_projector ; CNN: proj = visual_projector
  std::vector<v4l::Lin> enc;        // encoder.base / Net.base MLP
  std::vector<v4l::TLayer> layers;
  v4l::LNp fin_ln;                  // cfg.pytorch_encoder: nn.TransformerEncoder's final norm
  v4l::LNp tok_ln, stok_ln;         // cfg.token_norm: token_ln (applied to every token), state_token_ln (a parameter nobody uses)
  std::vector<v4l::Lin> head;       // append fcs + last
  int logstd = -1;
  int64_t packed_elems = 0;
  std::vector<v4l::PackDesc> packs;  // src filled at bind
  std::vector<int> pack_param;       // param index per desc
  int64_t pack_blocks = 0;
  // binding
  std::vector<float*> p;             // device pointers per param
  void* packed = nullptr;
  v4l::PackDesc* d_packs = nullptr;
  v4l::ParamSeg* d_segs = nullptr;
  v4l::RedDesc* d_red = nullptr;
  float* d_sq = nullptr;             // sum of squares of what each wgrad_reduce block wrote (the gradient norm's partials)
  int64_t sq_cap() const { return total_params / 64 + 2 * MAX_RED + 64; }
  int red_blocks = 0;                // blocks of the last wgrad_reduce launch (0: no partials available)
  const float* red_grads = nullptr;  // the gradient buffer that launch wrote: the partials describe THIS buffer only
  static constexpr int MAX_RED = 96;
  static constexpr int MAX_TNP = 64;
  v4l::TnProb* d_tnp = nullptr;
  static constexpr int MAX_WIDE = 16;
  v4l::TnWide* d_wide = nullptr;
  std::vector<v4l::TnWide> wide, wide_cached;  // whole-output weight-grad problems of the fused transformer layers
  double wide_flops = 0;
  double tnp_flops = 0;
  std::vector<v4l::TnProb> tnp, tnp_cached;   // deferred dense weight-grad problems of the current / last backward
  std::vector<v4l::RedDesc> red, red_cached;  // weight-grad reduce descriptors of the current / last backward
  int64_t slab_cap = 0;
  float grad_unscale = 1.f;          // 1 / the gradient scale of the backward pass in flight (V4L_F16; 1 otherwise)
  float grad_scale_next = 0.f;       // v4l_net_set_grad_scale: the next backward pass's scale (0: the rule of v4l_net_grad_scale)
  int64_t seg_blocks = 0;
  bool bound = false;
  int gen = 0;  // bind generation: bumped by every v4l_net_bind; captured graphs of trainers / actors are keyed on it
  // auxiliary stream + fork/join events for sibling-kernel concurrency (created at bind; null = serial)
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t aux2 = nullptr;  // third branch of the weight-grad section (V4L_PAR_WGRAD=3)
  hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;

  int build();
  v4l::Layout layout(int n) const;
  int64_t table_bytes() const;
  int64_t slab_floats(int n) const;
  // enc_ws != null: reuse the encoder output another net (same encoder parameters and shapes) left in ITS workspace
  // stage: 0 = whole net, 1 = encoder only (up to the token / concat tensor), 2 = trunk + head only
  template <typename T> int forward_t(const float* state, const T* image, const int* rowidx, int n, float* ws, hipStream_t s,
                                      const float* enc_ws = nullptr, int stage = 0);
  // round 4: the pooled heads' data-grads ran beside the loss statistics (heads_ext() handed the trainer their operands) for
  // the backward pass over (heads_ext_ws, heads_ext_n) that follows; backward_t consumes the mark
  float* heads_ext_ws = nullptr;
  int heads_ext_n = 0;
  bool wps_bwd_plain() const;   // backward_t will run the (non-vision) wave-per-sample backward
  int heads_ext(float* ws, int n, v4l::RowsChain* out);
  // round 6: the trainer handed over the loss of the pass (rows, scalars, where the statistics go) — backward_t then runs the
  // layers' forward, the loss rows and the backward as ONE launch (csrc/wps_fb.h) over a workspace whose forward stopped behind
  // the encoder (forward_t stage 1); backward_t consumes the mark
  const v4l::FbLoss* fb_loss = nullptr;
  bool fb_ok() const;         // ... which serves the plain shipped LocoTransformer
  bool fused_layers() const;  // the transformer layers run as fused forward / backward launches (csrc/infer.h, bwd.h)
  bool wps_layers() const;
  bool wps_max_pool() const;  // max_pool=True pooled inside the wave-per-sample pair
  bool wps_opt() const;       // token_norm / use_pytorch_encoder / another proprio MLP around the wave-per-sample layers
  bool wps_opt_vis() const;   // the same for the vision-only Transformer (native 16-token kernels, 17-row slots)
  bool wps_tail_shape() const;  // the proprio MLP the wave-per-sample backward continues into (256-256)
  bool wps_vis() const;  // vision-only Transformer on the wave-per-sample kernels (dummy token row, csrc/wps.h)    // ... as wave-per-sample launches (csrc/wps.h)
  template <typename T> int backward_t(const float* state, const T* image, const int* rowidx, int n, float* ws, float* grads, hipStream_t s);
};

struct v4l_actor {
  v4l_net *pf = nullptr, *vf = nullptr;
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int E = 0;
  float* ws = nullptr;
  v4l::ActCtl* ctl = nullptr;
  int* rowidx = nullptr;
  hipGraphExec_t gexec = nullptr;
  const void* key[16] = {};
  bool warm = false, bound = false;
  long long t_host = -1;  // the env step index as the host counts it (v4l_actor_seek sets it, eager steps advance it); -1: unknown
  // host mirror of ActCtl::seq (launches of rollout_dense_kernel so far; v4l_actor_bind zeroes both): eager launches pass it as
  // an argument. dense_in_graph: the captured step contains that kernel, so every replay advances the device's count too.
  unsigned dense_seq = 0;
  bool dense_in_graph = false;
  // v4l_actor_step_split: the depth stacks of the step in flight as bf16 rows (null: fp32 observation rows)
  const void* img16 = nullptr;
  int64_t ld_img16 = 0;
};

struct GraphKey { v4l_rollout ro; v4l_ppo_hyper hp; int n; int gen[3]; };

struct v4l_trainer {
  v4l_net *pf = nullptr, *vf = nullptr, *tpf = nullptr;
  float *g_pf = nullptr, *m_pf = nullptr, *v_pf = nullptr, *g_vf = nullptr, *m_vf = nullptr, *v_vf = nullptr;
  float* ws = nullptr;
  int64_t ws_floats = 0;
  // device control block + per-update staging (caller-allocated, v4l_trainer_ctl_bytes)
  v4l::UpdCtl* ctl = nullptr;
  float* stats_cur = nullptr;
  int* rowidx_cur = nullptr;
  float* norm_part = nullptr;
  int n_max = 0;
  const int* rowidx_all = nullptr;
  float* stats_all = nullptr;
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // hipGraph of one minibatch update
  GraphKey gkey = {};
  hipGraphExec_t gexec = nullptr;
  hipGraphExec_t gexec_run = nullptr;  // hipGraph of a whole run of updates (v4l_trainer_update_run), gexec_run_count of them
  int gexec_run_count = 0;
  bool warm = false;
  bool bound = false;
  // data parallel: RCCL communicator of this trainer's process group (null: single GPU, or the host drives the exchange)
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
};
