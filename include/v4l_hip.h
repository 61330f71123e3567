/* v4l_hip.h — C ABI of libv4l_hip.so, the MI355X (gfx950) implementation of the vision4leg PPO hot path.
 *
 * The reference (Mehooz/vision4leg) has no FFI for this path: the boundary is the Python object protocol of
 * its in-tree `torchrl` package. Each entry point below therefore replaces the arithmetic
 * behind one reference Python symbol, cited as `file:line` relative to the reference checkout. The Python
 * shell in vision4leg_amd/torchrl binds these with ctypes and keeps the reference class/method names.
 *
 * Conventions
 *   - every function returns int: 0 ok, -1 bad argument / unsupported configuration, -2 HIP error;
 *     v4l_last_error() returns the message (thread-local, valid until the next failing call).
 *   - plain pointers and sizes only. Pointers named *_dev are device (HBM) pointers owned by the caller;
 *     the library never allocates or frees device memory for the caller's data and keeps no pointer
 *     beyond what v4l_net_bind / v4l_trainer_bind register (those must stay valid until rebind/destroy).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). Calls only enqueue work.
 *   - one handle = one device, driven from one host thread at a time (the reference is single-threaded,
 *     torchrl/algo/rl_algo.py:97-168).
 *   - all activations/params/grads are fp32; `compute` selects the contraction operand type:
 *     V4L_F32 = exact fp32 MFMA (parity mode), V4L_BF16 = bf16 operands, fp32 accumulate, V4L_F16 = IEEE half operands,
 *     fp32 accumulate (round 6: bf16's MFMA rate and bytes, three more significand bits — 8 x closer to the fp32 reference;
 *     half's 5 exponent bits are why its backward runs on scaled loss-gradient rows: v4l_net_grad_scale).
 *
 * Environment switches read by the library: 21 (round 4: 37 — the ones no test referenced went in round 5, their fast-path value
 * is now compiled in; V4L_VIS17 came with the native 16-token kernels). tests/test_cpu.py keeps this list and the source in step. Three configure a run, the others are DIAGNOSTIC: every one of them backs a bit-equality / cross-check test
 * under tests/ that compares a fused kernel or a launch schedule against the general one. Read when first needed unless (per call).
 *   run:   V4L_TRACE (launch / graph log on stderr), V4L_RCCL_LIB (RCCL library to dlopen; default librccl.so.1), V4L_ROCTX=1 (roctx
 *          ranges phase|op|kernel around every launch call for `rocprofv3 --marker-trace`)
 *   V4L_PAR=0                 no auxiliary streams (sibling kernels run serially)
 *   V4L_PAR_WGRAD=0|1|2       weight-grad launch schedule of a backward pass (per call; 2 = default: conv data-grads first,
 *                             then dW3 on the main stream next to the dense weight-grads on the auxiliary stream)
 *   V4L_SPLIT_REDUCE=0        wgrad_reduce as one launch behind the join instead of two inside the forked section (same bits) (per call)
 *   V4L_NO_DENSE_STACK        the NatureCNN nets' visual projector + head and their data-grads as one gemm_nt_deep launch per
 *                             linear instead of one launch per direction (csrc/dense_stack.h; same bits) (per call)
 *   V4L_NO_FUSED_CONV_BWD     conv-stack backward layer by layer (per call)
 *   V4L_ACTS_F32              the training encoder saves conv1 / conv2 activations in fp32 even where the fused conv backward would
 *                             take them in the operand type (round 5; same bits, 19 KB more per sample each way) (per call)
 *   V4L_VIS17                 the vision-only Transformer on the 17-row wave-per-sample instantiation (dummy row 0, masked key) instead
 *                             of the native 16-token one (round 5; agrees to rounding: cross-checked) (per call)
 *   V4L_NO_FB                 the LocoTransformer's layers + heads of an update pass as three launches (forward, loss, backward)
 *                             instead of the fused forward-loss-backward launch (round 6, csrc/wps_fb.h; cross-checked) (per call)
 *   V4L_NO_LAYER_STACK        one launch per transformer layer instead of one per direction (per call)
 *   V4L_NO_WPS_LAYERS         transformer layers on the block-cooperative kernels instead of the wave-per-sample ones (per call)
 *   V4L_LAYER_TAPS            the wave-per-sample layer kernels and the fused conv backward also write every intermediate into
 *                             its v4l_net_ws_offset slot (production keeps them on chip) (per call)
 *   V4L_CONV_BWD_BLOCKS, V4L_CONV3_WGRAD_BLOCKS   persistent-block counts (tests force ragged and many-samples-per-block shapes)
 *   V4L_GEMM_DEEP_MIN_M       smallest row count the general path's dense layers send to gemm_nt_deep_kernel (default 256) (per call)
 *   V4L_ROLLOUT_DENSE_SPLIT   NatureCNN nets' rollout step: the dense layers as three launches instead of one launch with
 *                             device-side hand-overs (per call)
 *   V4L_WPS_HEAD_IN, V4L_WPS_TOK0_IN, V4L_WPS_HEAD_EXT_CRITIC   where the pooled heads' / the proprio branch's data-grad chains run:
 *                             inside wps_layer_bwd_kernel (4 rows per block) or beside the loss statistics / the layers'
 *                             weight-grads (16 - 64 rows per block) (per call)
 * Read by the Python shell, not by the library: V4L_COMPUTE=f16|bf16|f32, V4L_GRAPH=0, V4L_EPOCH_GRAPH=0 (one hipGraph replay per
 * update instead of one per epoch), V4L_DP_COMM=auto|torch|rccl,
 * V4L_CAST_THREADS, V4L_COLLECT_SPLIT=0 (fp32 observation rows over PCIe instead of fp32 proprio + 16-bit depth rows),
 * V4L_COLLECT_HOST_STEP=0 (the collector's env step as torch cast + RolloutActor.step_host_split instead of ONE v4l_actor_step_rows call),
 * V4L_SPLIT_VIA_COPY, V4L_GUARD, V4L_FORCE_DP_PHASES, V4L_LIB (diagnostic builds). Everything except COMPUTE / GRAPH / DP_COMM /
 * CAST_THREADS / RCCL_LIB / TRACE / ROCTX is diagnostic: the Python shell warns once at load time when one is set
 * (_lib.diagnostic_switches).
 */
#ifndef V4L_HIP_H
#define V4L_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define V4L_F32 0
#define V4L_BF16 1
#define V4L_F16 2
/* V4L_F16: a backward pass over n rows runs on loss-gradient rows multiplied by 2^(V4L_F16_SCALE_LOG2 + ceil(log2 n)) — the
   mean()'s 1/n taken out and 16 x on top: the rows' bulk sits inside half's normal range and the largest element two to three
   decades under its overflow (profiles/r6_f16_attribution.txt) — and elements beyond +-V4L_F16_GRAD_CLAMP are clamped there
   (counted in the record's slot 23). The weight-grad reduction multiplies the scale out again: gradients leave the library
   unscaled in every mode. A power of two: scaling and unscaling are exact in fp32. */
#define V4L_F16_SCALE_LOG2 4
#define V4L_F16_GRAD_CLAMP 32768.0f

#define V4L_NET_MLP 0  /* networks.Net + MLPBase            torchrl/networks/nets.py:16-55 (ppo_state.py)      */
#define V4L_NET_CNN 1  /* networks.ImpalaEncoderProjNet + NatureFuseEncoder   nets.py:194-262, base.py:345-385 */
#define V4L_NET_LOCO 2 /* networks.LocoTransformer + LocoTransformerEncoder   nets.py:909-1038, base.py:497-626 */
/* vision-only variants: the observation row is the 4x64x64 depth stack alone, state_dim = 0,
 * n_enc_hidden = 0 */
#define V4L_NET_CNN_VIS 3  /* networks.NatureEncoderProjNet + NatureEncoder(flatten)   nets.py:133-191, base.py:304-342
                              (starter/ppo_nature_cnn_vision_only.py:80-97) */
#define V4L_NET_LOCO_VIS 4 /* networks.Transformer + TransformerEncoder (depth only: 16 tokens, mean-pooled)
                              nets.py:784-906, base.py:388-494 (starter/ppo_locotransformer_vision_only.py:77-97) */

#define V4L_MAX_HIDDEN 4
#define V4L_STATS 24 /* floats per update record; [0..17] = the 18 logger keys of ppo.py:77-92,122-123,142-145;
                        [18..21] shard moments of the advantages (data-parallel exchange); [22] = how many of the 18 are
                        NaN / Inf (device-side form of the collector's NaN check, collector/on_policy.py:102-107; a half
                        operand that overflowed anywhere in a V4L_F16 pass ends up here through the losses / gradient norms);
                        [23] = V4L_F16: loss-gradient elements of this update clamped at +-V4L_F16_GRAD_CLAMP (0 otherwise) */
#define V4L_OUT_LD 16 /* row stride of head outputs (action mean / value), zero padded */

typedef struct v4l_net_cfg {
  int kind;            /* V4L_NET_*                                                                      */
  int compute;         /* V4L_F32 | V4L_BF16 | V4L_F16                                                   */
  int state_dim;       /* S: proprio length (state_input_dim, starter/ppo_locotransformer.py:81)         */
  int out_dim;         /* A for a policy, 1 for a value net                                              */
  int in_channels;     /* 4 (depth stack); ignored for V4L_NET_MLP                                       */
  int img_hw;          /* 64                                                                             */
  int n_enc_hidden;    /* encoder.hidden_shapes / MLPBase hidden_shapes (base.py:8-44)                   */
  int enc_hidden[V4L_MAX_HIDDEN];
  int visual_dim;      /* V4L_NET_CNN: NatureFuseEncoder.visual_dim (base.py:358-362)                    */
  int token_dim;       /* V4L_NET_LOCO: 64                                                               */
  int n_layers;        /* V4L_NET_LOCO: len(transformer_params) (nets.py:948-955)                        */
  int ff_dim;          /* V4L_NET_LOCO: dim_feedforward                                                  */
  int n_head_hidden;   /* append_hidden_shapes (nets.py:35-50, 224-243, 973-992)                         */
  int head_hidden[V4L_MAX_HIDDEN];
  int has_logstd;      /* 1: Gaussian policy with a state-independent logstd parameter                   */
  int tanh_action;     /* Gaussian policy with tanh_action=True (policies/distribution.py:5-80 TanhNormal,
                          continuous_policy.py:85-146): action = tanh(mean + std * eps), log-prob of a stored action through
                          atanh(action) with the -log(1 - a^2 + 1e-6) correction. The fused rollout step kernels sample through
                          tanh in their epilogues (round 5; no shipped config sets it) */
  int max_pool;        /* V4L_NET_LOCO / V4L_NET_LOCO_VIS: token pooling by max instead of mean (max_pool=True,
                          nets.py:1022-1030, 884-889); fused like the mean (round 5; no shipped config sets it) */
  int token_norm;      /* V4L_NET_LOCO / V4L_NET_LOCO_VIS: LayerNorm(token_dim) over every token in front of the transformer layers
                          (token_norm=True: nets.py:815-818, 879-880, 1007-1008; parameters token_ln.* and the never-used
                          state_token_ln.*); round 5: fused rollout step, update with the layers on the wave-per-sample kernels
                          and token_ln as a launch of its own; no shipped config sets it */
  int pytorch_encoder; /* V4L_NET_LOCO / V4L_NET_LOCO_VIS: the layers are an nn.TransformerEncoder WITH a final LayerNorm
                          (use_pytorch_encoder=True: nets.py:955-963, 884-885, 1012-1013; parameters
                          visual_trans_encoder.layers.N.*, visual_trans_encoder.norm.*); round 5: as token_norm, with the
                          final norm + pooling + heads as launches of their own */
} v4l_net_cfg;

typedef struct v4l_net v4l_net;         /* host-side plan of one network (no device memory)  */
typedef struct v4l_trainer v4l_trainer; /* host-side plan of the PPO update over (pf, vf, target_pf) */
typedef struct v4l_actor v4l_actor;     /* host-side plan of one rollout step: policy sample + value at batch E */

const char* v4l_last_error(void);
int v4l_version(void);
/* sizeof(v4l_net_cfg / v4l_ppo_hyper / v4l_rollout) as THIS library was compiled: a binding whose struct mirrors differ (a
   stale libv4l_hip.so next to newer host code, or the other way round) must refuse to run instead of reading fields at the
   wrong offsets. which: 0 net_cfg, 1 ppo_hyper, 2 rollout; anything else: -1 */
int v4l_abi_sizeof(int which);

/* ---- network plan: replaces the nn.Module constructors' shape bookkeeping (nets.py / base.py above) ---- */
int v4l_net_create(const v4l_net_cfg* cfg, v4l_net** out);
void v4l_net_destroy(v4l_net* net);
int v4l_net_num_params(const v4l_net* net);
/* name = the reference state_dict key (SURVEY.md §8b "Checkpoint names"); shape in PyTorch layout */
int v4l_net_param_info(const v4l_net* net, int i, const char** name, int* ndim, int64_t shape[4], int64_t* numel,
                       int64_t* grad_offset);
int64_t v4l_net_total_params(const v4l_net* net);   /* length of a flat grad / Adam-moment buffer (floats) */
int64_t v4l_net_packed_bytes(const v4l_net* net);   /* bytes of the packed (contraction-ready) weight buffer */
int64_t v4l_net_table_bytes(const v4l_net* net);    /* bytes of the small device descriptor table */
int64_t v4l_net_ws_floats(const v4l_net* net, int n, int train); /* workspace floats for a batch of n */
int v4l_net_state_ld(const v4l_net* net);           /* Sp: row stride of the ingested proprio array */
/* Register where the parameters live (params_dev[i] = device pointer of parameter i, PyTorch layout) and the
 * caller-allocated packed-weight and descriptor-table buffers. Uploads the descriptor tables (async). */
int v4l_net_bind(v4l_net* net, float* const* params_dev, void* packed_dev, void* table_dev, void* stream);
/* Refresh the packed weights from the current parameter values (after optimiser steps / load_state_dict). */
int v4l_net_pack(v4l_net* net, void* stream);

/* Split reference observation rows [n][S + C*H*W] fp32 (vision4leg/envs/utilities/env_utils.py:27-51,
 * nets.py:997-1000) into state_dev[slot0+i][Sp] fp32 and image_dev[slot0+i][C*H*W] (fp32 or bf16 per
 * `compute`). image_dev may be NULL for V4L_NET_MLP. */
int v4l_ingest(const v4l_net* net, const float* obs_dev, int n, float* state_dev, void* image_dev, int64_t slot0,
               void* stream);

/* Forward pass == nn.Module.forward of nets.py:52-55 / 247-262 / 996-1038 on rows rowidx[i] (or i when
 * rowidx_dev is NULL) of the ingested arrays. The head output (action mean or value) is left in the
 * workspace; v4l_net_out_ptr gives its address: [n][V4L_OUT_LD] fp32. train=1: a v4l_net_backward over this workspace
 * follows — with bf16 compute and the shipped conv geometry the conv1 / conv2 activation slots ("c1", "c2" of
 * v4l_net_ws_offset) then hold the operand type, not fp32 (bit-identical gradients; V4L_LAYER_TAPS / V4L_ACTS_F32 keep
 * fp32). train=0 leaves every slot as documented. */
int v4l_net_forward(v4l_net* net, const float* state_dev, const void* image_dev, const int* rowidx_dev, int n,
                    float* ws_dev, int train, void* stream);
float* v4l_net_out_ptr(const v4l_net* net, float* ws_dev, int n, int train);
float* v4l_net_dout_ptr(const v4l_net* net, float* ws_dev, int n);
/* Backward of the same pass: reads d(out) at v4l_net_dout_ptr ([n][V4L_OUT_LD]), accumulates parameter
 * gradients into grads_dev (flat, offsets from v4l_net_param_info; caller zeroes it). Replaces
 * loss.backward() of ppo.py:72,117 for everything below the head output.
 * The d(out) rows are expected MULTIPLIED by v4l_net_grad_scale(net, n) — 1 unless compute == V4L_F16 (see V4L_F16_SCALE_LOG2;
 * the trainer's loss kernels do it, a caller that fills d(out) itself must) — and the gradients come out unscaled.
 * v4l_net_grad_scale is the rule for MEAN-loss gradients (rows of size ~1/n). A caller whose d(out) rows have another size
 * announces its own power of two with v4l_net_set_grad_scale before the backward call (consumed by it; ignored unless V4L_F16):
 * the rows then arrive multiplied by THAT, sized so that the largest element sits around 2^12 .. 2^13 (the Python shell's
 * HipNet.backward does this from max |d(out)|). */
float v4l_net_grad_scale(const v4l_net* net, int n);
int v4l_net_set_grad_scale(v4l_net* net, float scale);
int v4l_net_backward(v4l_net* net, const float* state_dev, const void* image_dev, const int* rowidx_dev, int n,
                     float* ws_dev, float* grads_dev, void* stream);

/* GaussianContPolicyBase.{forward,explore,update} post-processing (policies/continuous_policy.py:85-146,
 * 486-492): contiguous mean/std [n][A], clamped log_std [A], ent [n], and log_prob [n] of acts_dev ([n][A])
 * when acts_dev != NULL. meanp_dev is v4l_net_out_ptr of the policy net. */
int v4l_gauss_head(const float* meanp_dev, const float* logstd_dev, const float* acts_dev, int n, int A,
                   float* mean_dev, float* std_dev, float* logstd_c_dev, float* ent_dev, float* logp_dev,
                   void* stream);
/* the same for a tanh_action policy (TanhNormal, policies/distribution.py:38-51): log_prob of the post-tanh actions acts_dev
 * through z = log((1 + a) / (1 - a)) / 2 — or through pre_tanh_dev [n][A] when given (explore(return_log_probs=True) hands the
 * pre-tanh draw over) — minus log(1 - a^2 + 1e-6) per dimension. mean / std / entropy are the Normal's, as in the reference. */
int v4l_gauss_head_tanh(const float* out_dev, const float* logstd_dev, const float* acts_dev, const float* pre_tanh_dev, int n,
                        int A, float* mean_dev, float* std_dev, float* logstd_clamped_dev, float* ent_dev, float* logp_dev,
                        void* stream);
/* column 0 of a padded head output -> contiguous [n] (vf(x) for the collector, collector/on_policy.py:99-100) */
int v4l_col0(const float* src_dev, int n, float* dst_dev, void* stream);

/* OnPolicyReplayBufferBase.generalized_advantage_estimation (torchrl/replay_buffers/on_policy.py:17-45), fp64,
 * bit-identical to the numpy loop. Arrays are [T][E] (the trailing 1 of the reference's [T,E,1] dropped);
 * time_limits is [T] when tl_per_env == 0 (the collector's `[False]` rows, collector/on_policy.py:122-124) or
 * [T][E]; use_time_limit mirrors `time_limit_filter`. advs32/rets32 (optional) receive the fp32 casts of
 * ppo.py:138,140. */
int v4l_gae(const double* rewards_dev, const double* values_dev, const double* terminals_dev,
            const double* time_limits_dev, int tl_per_env, const double* last_value_dev, int T, int E, double gamma,
            double tau, int use_time_limit, double* scratch_dev /* 3*T*E doubles */, double* advs_dev, double* rets_dev,
            float* advs32_dev, float* rets32_dev, void* stream);

/* OnPolicyReplayBufferBase.discount_reward (torchrl/replay_buffers/on_policy.py:47-71; PPO(gae=False),
 * algo/on_policy/on_rl_algo.py:29-33), fp64, bit-identical to the numpy loop: R_t = r_t + (1 - term_t) gamma R_{t+1} from
 * R_T = last_value (time-limit filter: R_t = (r_t + (1 - term_t) gamma R_{t+1} (1 - tl_t)) + tl_t V_t); advs = R - V,
 * estimate_returns = R. Array shapes as v4l_gae. */
int v4l_discount_reward(const double* rewards_dev, const double* values_dev, const double* terminals_dev,
                        const double* time_limits_dev, int tl_per_env, const double* last_value_dev, int T, int E, double gamma,
                        int use_time_limit, double* advs_dev, double* rets_dev, float* advs32_dev, float* rets32_dev,
                        void* stream);

/* Running observation normaliser of the vectorised env on the device: what
 * NormObsWithImg.observation (vision4leg/get_env.py:58-67) / NormObs.observation (torchrl/env/base_wrapper.py:119-122)
 * do on the host per env step. raw_dev: the step's [E][S] fp64 proprio rows (row stride ld_raw). update != 0 (the
 * wrapper's training mode): Normalizer.update_estimate (base_wrapper.py:77-84) merges the batch mean / variance over the
 * E envs into mean_dev / var_dev [S] and count_dev [1] (update_mean_var_count, :44-61; a fresh Normalizer is mean 0,
 * var 1, count 1e-4, :64-72). Then Normalizer.filt (:93-96): clip((raw - mean) / (sqrt(var) + 1e-4), -clip, clip), in
 * fp64 and bit-identical to the numpy code, written as fp64 (out64_dev) and / or as the fp32 cast the collector makes
 * (out32_dev); either may point into the [E][S + C*H*W] observation rows v4l_actor_step reads (row stride ld_out*).
 * image_dev (optional): the step's [E][image_elems] depth stack, fp32 or fp64 (image_f64), copied / cast into
 * image_out_dev — the np.hstack of get_env.py:64-67 without the host pass. */
int v4l_obs_norm(const double* raw_dev, int64_t ld_raw, int E, int S, double* mean_dev, double* var_dev, double* count_dev,
                 double clip, int update, float* out32_dev, int64_t ld_out32, double* out64_dev, int64_t ld_out64,
                 const void* image_dev, int image_f64, int64_t ld_image, int64_t image_elems, float* image_out_dev,
                 int64_t ld_image_out, void* stream);

/* ---- rollout step: what VecOnPolicyCollector.take_actions asks of the networks per env step
 * (torchrl/collector/on_policy.py:90-100): out = pf.explore(ob) and values = vf(ob) for the E observation rows of the
 * step — as ONE captured launch sequence: ingest of the rows into rollout slots [t*E,(t+1)*E), the policy forward, the
 * value forward (shared_encoder=1: on the policy's encoder output, the two nets hold the same encoder parameters),
 * action = mean + std*eps, and the filing of action/value into the rollout arrays. The step cursor t lives on the
 * device (v4l_actor_seek sets it, every step advances it). obs_dev: [E][S+C*H*W] fp32 at a FIXED address the caller
 * refreshes each step; eps_dev: [E][A] standard-normal draws (the caller's generator). Outputs: [E][A] / [E]. ---- */
int v4l_actor_create(v4l_net* pf, v4l_net* vf, int E, v4l_actor** out);
void v4l_actor_destroy(v4l_actor* a);
int64_t v4l_actor_ws_floats(const v4l_actor* a);
int64_t v4l_actor_ctl_bytes(const v4l_actor* a);
/* ctl_dev: v4l_actor_ctl_bytes() bytes; bind zeroes its control block on `stream` (step cursor, hand-over counters, error
 * flag) — issue the first step on the same stream or after synchronising it. */
int v4l_actor_bind(v4l_actor* a, float* ws_dev, void* ctl_dev, void* stream);
/* *err_out = 0, or 1 + the hand-over counter a block of the NatureCNN rollout step gave up waiting for since the last check
 * (the step then filed NaN actions — collector/on_policy.py:102-107 "NaN detected" — instead of numbers computed from stale
 * activations). Synchronises the stream, clears the flag. */
int v4l_actor_check(v4l_actor* a, int* err_out, void* stream);
int v4l_actor_seek(v4l_actor* a, int64_t t, void* stream);
int v4l_actor_step(v4l_actor* a, const float* obs_dev, const float* eps_dev, float* state_roll_dev, void* image_roll_dev,
                   float* acts_roll_dev, float* values_roll_dev, float* logp_roll_dev, float* action_dev, float* mean_dev,
                   float* std_dev, float* ent_dev, float* value_dev, int shared_encoder, int use_graph, void* stream);

/* The same step with the observation handed over SPLIT: proprio rows [E][S] fp32 (NULL when S = 0) and the depth stacks
 * [E][C*H*W] already in bf16 — the type the bf16 kernels round the image to at ingest anyway, so the result is bit-identical
 * to v4l_actor_step on fp32 rows holding the same values, with half the bytes crossing PCIe when the pointers are pinned host
 * memory (collector/on_policy.py:90-93 uploads one fp32 row per env). Eager launches only. v4l_actor_split_supported: 1 when
 * this actor's step runs on kernels that take the split form (bf16 compute, image nets on the fused rollout step). */
int v4l_actor_split_supported(const v4l_actor* a, int shared_encoder);
int v4l_actor_step_split(v4l_actor* a, const float* proprio_dev, const void* image16_dev, const float* eps_dev,
                         float* state_roll_dev, void* image_roll_dev, float* acts_roll_dev, float* values_roll_dev,
                         float* logp_roll_dev, float* action_dev, float* mean_dev, float* std_dev, float* ent_dev,
                         float* value_dev, int shared_encoder, void* stream);

/* The collector's whole env step (collector/on_policy.py:90-100) as ONE host call, no interpreter in the loop (round 6):
 * `rows_host` [E][ld] float64 — the rows the env wrappers hand over, pageable memory is fine — are cast on a persistent pool of
 * `threads` host threads (AVX-512 where the CPU has it) into the caller's PINNED staging blocks (`proprio_pinned` [E][S] fp32,
 * NULL when S = 0; `image16_pinned` [E][C*H*W] in the net's 16-bit operand type: float64 -> float32 -> operand type, both
 * round-to-nearest-even, i.e. torch.Tensor(ob) followed by the kernels' ingest cast), the two launches of v4l_actor_step_split
 * read them in place, and the call watches `action_pinned` [E][A] and `value_pinned` [E] (pinned, armed with NaN) until every
 * slot holds a number or `poll_seconds` have passed. Returns 0: both outputs have arrived — every block has finished reading
 * the staging blocks, they may be overwritten; 1: not waited for (poll_seconds <= 0) or timed out — synchronise the stream
 * before touching the buffers; < 0: error. v4l_host_cast_rows is the cast alone (no GPU involved; compute V4L_F32: image_out
 * is fp32); v4l_host_cast_simd: 1 when the AVX-512 path is in use. */
int v4l_host_cast_rows(const double* rows_host, int64_t ld, int E, int S, int64_t img_elems, float* proprio_out, void* image_out,
                       int compute, int threads);
int v4l_host_cast_simd(void);
int v4l_actor_step_rows(v4l_actor* a, const double* rows_host, int64_t ld, float* proprio_pinned, void* image16_pinned,
                        const float* eps_dev, float* state_roll_dev, void* image_roll_dev, float* acts_roll_dev,
                        float* values_roll_dev, float* logp_roll_dev, float* action_pinned, float* mean_dev, float* std_dev,
                        float* ent_dev, float* value_pinned, int shared_encoder, int threads, double poll_seconds, void* stream);

/* ---- PPO minibatch update: replaces PPO.update / update_critic / update_actor (ppo.py:42-153), the two
 * clip_grad_norm_(…, 0.5) calls and the two Adam steps (a2c.py:30-40). pf and vf may share encoder parameters
 * (same device pointers in both tables): the critic step runs first and the actor forward sees the updated
 * encoder, exactly as ppo.py:150-151. ---- */
typedef struct v4l_ppo_hyper {
  float clip_para;      /* ppo.py:17 */
  float entropy_coeff;  /* a2c.py:19 */
  float max_grad_norm;  /* 0.5, ppo.py:73-74,118-119 */
  float beta1, beta2, eps; /* Adam: 0.9, 0.999, 1e-5 (a2c.py:30-40) */
  int clipped_value_loss;  /* ppo.py:105-112 */
  int world_size;       /* data-parallel ranks; losses are scaled by 1/(n*world_size) */
} v4l_ppo_hyper;

typedef struct v4l_rollout {
  const float* state_dev;  /* [slots][Sp]   */
  const void* image_dev;   /* [slots][C*H*W] fp32|bf16, NULL for V4L_NET_MLP */
  const float* acts_dev;   /* [slots][A]    */
  const float* advs_dev;   /* [slots]       */
  const float* rets_dev;   /* [slots]  estimate_returns */
  const float* values_dev; /* [slots]  old values (only for clipped_value_loss) */
  const float* logp_old_dev; /* [slots] or NULL. log pi_old(a|s) recorded when the action was taken (v4l_actor_step).
                              * NULL: the frozen target policy is evaluated on every minibatch, as ppo.py:55-57 does.
                              * Non-NULL: that forward pass is skipped — same numbers (the acting policy of an epoch is
                              * the epoch's target policy, ppo.py:34), one third less forward work per update. */
} v4l_rollout;

int v4l_trainer_create(v4l_net* pf, v4l_net* vf, v4l_net* target_pf, v4l_trainer** out);
void v4l_trainer_destroy(v4l_trainer* tr);
int64_t v4l_trainer_ws_floats(const v4l_trainer* tr, int n);
int64_t v4l_trainer_ctl_bytes(const v4l_trainer* tr, int n); /* device control block + per-update staging */
/* grads/moments: flat fp32 buffers of v4l_net_total_params(pf|vf) floats each (moments zero-initialised by the
 * caller == fresh torch.optim.Adam state). ctl_dev: v4l_trainer_ctl_bytes(n_max) bytes. Re-binding drops the graph. */
int v4l_trainer_bind(v4l_trainer* tr, float* g_pf_dev, float* m_pf_dev, float* v_pf_dev, float* g_vf_dev,
                     float* m_vf_dev, float* v_vf_dev, float* ws_dev, int64_t ws_floats, void* ctl_dev, int n_max,
                     void* stream);
/* Opens a run of updates (normally: one epoch = opt_epochs x minibatches): update #u will train on rows
 * rowidx_all_dev[u][0..n) (NULL = rows 0..n-1 every time) and write its V4L_STATS-float record to
 * stats_all_dev[u] (may be NULL). steps_done = Adam steps already taken (0 for fresh optimisers); lr_* are the
 * learning rates of update_linear_schedule (algo/utils.py:28-32). The update index, the Adam step count and the
 * bias corrections then live on the device and advance by themselves, one per update. */
int v4l_trainer_begin(v4l_trainer* tr, const int* rowidx_all_dev, float* stats_all_dev, double lr_pf, double lr_vf,
                      int64_t steps_done, const v4l_ppo_hyper* hp, void* stream);
/* The next minibatch update, == PPO.update (ppo.py:125-153). use_graph=1: the launch sequence (~20 kernels) is
 * captured once per (rollout pointers, n, hyper-parameters) as a hipGraph and replayed; needs a non-default stream.
 * The first update of a configuration always runs eagerly. */
int v4l_trainer_update_next(v4l_trainer* tr, const v4l_rollout* ro, int n, const v4l_ppo_hyper* hp, int use_graph,
                            void* stream);
/* The next `count` minibatch updates, == the epoch loop of PPO.update_per_epoch (ppo.py:28-40: opt_epochs x minibatches) behind
 * one v4l_trainer_begin. use_graph=1: ALL count updates are ONE hipGraph (count x ~22 kernel nodes; the row indices, the update
 * index, Adam's step count and bias corrections and the learning rates live on the device, so the node sequence is the same for
 * every update and every epoch), captured once per (rollout pointers, n, hyper-parameters, count) and replayed with one
 * hipGraphLaunch per epoch (round 6; BASELINE configs[4] "hipGraph-captured PPO epoch"). The call that meets a new configuration
 * runs it through v4l_trainer_update_next (first update eagerly, the others as single-update replays). */
int v4l_trainer_update_run(v4l_trainer* tr, const v4l_rollout* ro, int n, const v4l_ppo_hyper* hp, int count, int use_graph,
                           void* stream);
/* Phases of the same update for a host-driven data-parallel schedule: critic_grads -> all-reduce(critic bucket) ->
 * critic_step -> actor_grads -> all-reduce(policy bucket) -> actor_step (buckets and their tails: see v4l_sync_grads).
 * v4l_trainer_stats_cur = the record being filled. */
int v4l_trainer_critic_grads(v4l_trainer* tr, const v4l_rollout* ro, int n, const v4l_ppo_hyper* hp, void* stream);
int v4l_trainer_critic_step(v4l_trainer* tr, const v4l_ppo_hyper* hp, void* stream);
int v4l_trainer_actor_grads(v4l_trainer* tr, const v4l_rollout* ro, int n, const v4l_ppo_hyper* hp, void* stream);
int v4l_trainer_actor_step(v4l_trainer* tr, const v4l_ppo_hyper* hp, void* stream);
float* v4l_trainer_stats_cur(const v4l_trainer* tr);
/* Convenience: begin(rowidx_dev as the only row, stats_dev, lr, step-1) + one eager update_next. step >= 1. */
int v4l_trainer_update(v4l_trainer* tr, const v4l_rollout* ro, const int* rowidx_dev, int n, const v4l_ppo_hyper* hp,
                       double lr_pf, double lr_vf, int64_t step, float* stats_dev, void* stream);
/* copy_model_params_from_to(pf, target_pf) (torchrl/algo/utils.py:23-25, ppo.py:34) + repack of the target */
int v4l_trainer_sync_target(v4l_trainer* tr, void* stream);

/* ---- data parallel: one process per GPU, every rank owns an env shard and its own rollout; gradients are
 * summed with ONE in-place RCCL all-reduce per optimiser step on the flat gradient buffer. The reference has no
 * counterpart (single process); the exchange sits where `vf_loss.backward()` / `policy_loss.backward()` end
 * (torchrl/algo/on_policy/ppo.py:71-72,116-117), before clip_grad_norm_, so that N ranks with batch n reproduce one process
 * with batch N*n. RCCL is loaded at run time (dlopen librccl.so.1); nothing here is needed on a single GPU.
 *   g_pf_dev / g_vf_dev of v4l_trainer_bind must have V4L_BUCKET_TAIL (8) floats of room behind total_params: the tail
 *   carries the shard's advantage moments (global-minibatch normalisation, ppo.py:148) and loss shares through the same
 *   collective. With a communicator attached and hyper.world_size > 1, v4l_trainer_update_next issues both all-reduces
 *   itself on the update's stream (they are captured into its hipGraph); v4l_sync_grads is the same step for hosts that
 *   drive the four phases (critic_grads / critic_step / actor_grads / actor_step) themselves. */
#define V4L_COMM_ID_BYTES 128
#define V4L_BUCKET_TAIL 8
int v4l_comm_available(void); /* 0 when RCCL can be loaded in this process (else -1 + message): ranks should agree on this
                                 before any of them enters v4l_trainer_comm_init — the rendezvous blocks until all arrive */
int v4l_comm_unique_id(char* id_out /* [V4L_COMM_ID_BYTES], rank 0; broadcast it to the other ranks by any means */);
int v4l_trainer_comm_init(v4l_trainer* tr, const char* id /* [V4L_COMM_ID_BYTES] */, int rank, int world);
int v4l_trainer_comm_destroy(v4l_trainer* tr);
/* rank / size as the attached communicator reports them (ncclCommUserRank / ncclCommCount); 0 / 1 without one */
int v4l_trainer_comm_info(const v4l_trainer* tr, int* rank_out, int* world_out);
int v4l_sync_grads(v4l_trainer* tr, int which /* 1 = critic bucket, 0 = policy bucket */, void* stream);
/* Self-test of the attached communicator through the calls an update makes (v4l_sync_grads on both buckets, eagerly and —
 * use_graph = 1 — as a captured hipGraph replayed twice): a rank-dependent integer pattern is all-reduced and every element
 * checked on the device against the sum each rank can compute alone. *mismatches_out = elements wrong on this rank (0 = pass).
 * Synchronises the stream; overwrites the gradient buckets and the current statistics record (call it between updates).
 * The host makes the in-graph RCCL schedule its default only when every rank passes (algo/on_policy/ppo.py shell). */
int v4l_trainer_comm_selftest(v4l_trainer* tr, int use_graph, int64_t* mismatches_out, void* stream);
/* for a host that runs the collective itself (e.g. torch.distributed): statistics record -> bucket tail (pack = 1, before
 * the all-reduce of total_params + V4L_BUCKET_TAIL floats) and back (pack = 0) */
int v4l_trainer_bucket_tail(v4l_trainer* tr, int which, int pack, int world, void* stream);

/* ---- built-in per-kernel timer (HIP events on the launch stream; used by bench.py for the roofline numbers).
 * v4l_prof_collect synchronises the device, writes "phase|op|kernel\tcalls\ttotal_us\talgorithmic_flops\n" lines
 * into buf (NUL terminated, truncated to cap) and returns the untruncated length. */
int v4l_prof_enable(int on);
int64_t v4l_prof_collect(char* buf, int64_t cap);
/* (diagnostic builds compiled with -DV4L_INFER_TIMING additionally export a clock64 phase-stamp reader used by
 * tools/probe/stamps*.py; it is not part of this ABI and absent from the shipped library.) */

/* ---- introspection for tests: float offset of a named activation inside a workspace laid out for n rows
 * ("c1","c2","c3","eh<i>","x<l>","qkv<l>","P<l>","ctx<l>","mid<l>","ff<l>","xin<l>","xh1_<l>","rs1_<l>","pooled","hh<i>",
 * "out","dout", and after a backward pass "dz2_<l>","df<l>","dz1_<l>","dqkv<l>","dx<l>","dhh<i>","deh<i>","dpool","dc1".."dc3");
 * -1 if unknown. Tensors the fused kernels only ever use as MFMA operands hold the operand type T in an fp32-sized slot. */
int64_t v4l_net_ws_offset(const v4l_net* net, int n, const char* name);

#ifdef __cplusplus
}
#endif
#endif /* V4L_HIP_H */
