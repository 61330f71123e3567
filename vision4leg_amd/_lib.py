"""ctypes binding of libv4l_hip.so (C ABI declared in include/v4l_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("V4L_LIB", os.path.join(_HERE, "libv4l_hip.so"))  # V4L_LIB: diagnostic builds only
SRC_DIR = os.path.join(_HERE, "csrc")

V4L_F32, V4L_BF16, V4L_F16 = 0, 1, 2
V4L_NET_MLP, V4L_NET_CNN, V4L_NET_LOCO, V4L_NET_CNN_VIS, V4L_NET_LOCO_VIS = 0, 1, 2, 3, 4
V4L_MAX_HIDDEN = 4
V4L_STATS = 24
ST_NONFINITE = 22   # record slot: number of NaN / Inf among the 18 logged scalars of an update
ST_F16_SAT = 23     # record slot: V4L_F16 — loss-gradient elements of the update clamped at +-V4L_F16_GRAD_CLAMP
V4L_OUT_LD = 16
V4L_BUCKET_TAIL = 8   # scalars behind the gradients of an all-reduce bucket
V4L_COMM_ID_BYTES = 128

# order of the 18 logger keys inside a stats record (torchrl/algo/on_policy/ppo.py:77-92,122-123,142-145)
STAT_KEYS = [
  "advs/mean", "advs/std", "advs/max", "advs/min", "Training/vf_loss", "grad_norm/vf",
  "Training/policy_loss", "logprob/mean", "logprob/std", "logprob/max", "logprob/min",
  "log_std/mean", "log_std/std", "log_std/max", "log_std/min", "ratio/max", "ratio/min",
  "grad_norm/pf",
]


class NetCfg(C.Structure):
  _fields_ = [
    ("kind", C.c_int), ("compute", C.c_int), ("state_dim", C.c_int), ("out_dim", C.c_int),
    ("in_channels", C.c_int), ("img_hw", C.c_int), ("n_enc_hidden", C.c_int),
    ("enc_hidden", C.c_int * V4L_MAX_HIDDEN), ("visual_dim", C.c_int), ("token_dim", C.c_int),
    ("n_layers", C.c_int), ("ff_dim", C.c_int), ("n_head_hidden", C.c_int),
    ("head_hidden", C.c_int * V4L_MAX_HIDDEN), ("has_logstd", C.c_int), ("tanh_action", C.c_int), ("max_pool", C.c_int),
    ("token_norm", C.c_int), ("pytorch_encoder", C.c_int),
  ]


class PPOHyper(C.Structure):
  _fields_ = [
    ("clip_para", C.c_float), ("entropy_coeff", C.c_float), ("max_grad_norm", C.c_float),
    ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
    ("clipped_value_loss", C.c_int), ("world_size", C.c_int),
  ]


class Rollout(C.Structure):
  _fields_ = [
    ("state_dev", C.c_void_p), ("image_dev", C.c_void_p), ("acts_dev", C.c_void_p),
    ("advs_dev", C.c_void_p), ("rets_dev", C.c_void_p), ("values_dev", C.c_void_p), ("logp_old_dev", C.c_void_p),
  ]


def build_command(out=LIB_PATH):
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
          os.path.join(SRC_DIR, "v4l_hip.hip"), "-o", out]


def build(force=False):
  """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
  srcs = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)]
  srcs.append(os.path.join(_HERE, "..", "include", "v4l_hip.h"))
  if not force and os.path.exists(LIB_PATH) and all(
      os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
    return LIB_PATH
  subprocess.run(build_command(), check=True)
  return LIB_PATH


_P = C.c_void_p
_SIGS = {
  "v4l_last_error": (C.c_char_p, []),
  "v4l_version": (C.c_int, []),
  "v4l_abi_sizeof": (C.c_int, [C.c_int]),
  "v4l_net_create": (C.c_int, [C.POINTER(NetCfg), C.POINTER(_P)]),
  "v4l_net_destroy": (None, [_P]),
  "v4l_net_num_params": (C.c_int, [_P]),
  "v4l_net_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
  "v4l_net_total_params": (C.c_int64, [_P]),
  "v4l_net_packed_bytes": (C.c_int64, [_P]),
  "v4l_net_table_bytes": (C.c_int64, [_P]),
  "v4l_net_ws_floats": (C.c_int64, [_P, C.c_int, C.c_int]),
  "v4l_net_state_ld": (C.c_int, [_P]),
  "v4l_net_bind": (C.c_int, [_P, C.POINTER(_P), _P, _P, _P]),
  "v4l_net_pack": (C.c_int, [_P, _P]),
  "v4l_ingest": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int64, _P]),
  "v4l_net_forward": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, _P]),
  "v4l_net_out_ptr": (_P, [_P, _P, C.c_int, C.c_int]),
  "v4l_net_dout_ptr": (_P, [_P, _P, C.c_int]),
  "v4l_net_grad_scale": (C.c_float, [_P, C.c_int]),
  "v4l_net_set_grad_scale": (C.c_int, [_P, C.c_float]),
  "v4l_net_backward": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P]),
  "v4l_gauss_head": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
  "v4l_gauss_head_tanh": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
  "v4l_col0": (C.c_int, [_P, C.c_int, _P, _P]),
  "v4l_gae": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                        _P, _P, _P, _P, _P, _P]),
  "v4l_discount_reward": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_int, _P, _P, _P, _P, _P]),
  "v4l_obs_norm": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, _P, _P, C.c_double, C.c_int, _P, C.c_int64, _P, C.c_int64,
                             _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_int64, _P]),
  "v4l_actor_create": (C.c_int, [_P, _P, C.c_int, C.POINTER(_P)]),
  "v4l_actor_destroy": (None, [_P]),
  "v4l_actor_ws_floats": (C.c_int64, [_P]),
  "v4l_actor_ctl_bytes": (C.c_int64, [_P]),
  "v4l_actor_bind": (C.c_int, [_P, _P, _P, _P]),
  "v4l_actor_seek": (C.c_int, [_P, C.c_int64, _P]),
  "v4l_actor_check": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
  "v4l_actor_step": (C.c_int, [_P] + [_P] * 12 + [C.c_int, C.c_int, _P]),
  "v4l_actor_split_supported": (C.c_int, [_P, C.c_int]),
  "v4l_actor_step_split": (C.c_int, [_P] + [_P] * 13 + [C.c_int, _P]),
  "v4l_trainer_create": (C.c_int, [_P, _P, _P, C.POINTER(_P)]),
  "v4l_trainer_destroy": (None, [_P]),
  "v4l_trainer_ws_floats": (C.c_int64, [_P, C.c_int]),
  "v4l_trainer_ctl_bytes": (C.c_int64, [_P, C.c_int]),
  "v4l_trainer_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int, _P]),
  "v4l_trainer_begin": (C.c_int, [_P, _P, _P, C.c_double, C.c_double, C.c_int64, C.POINTER(PPOHyper), _P]),
  "v4l_trainer_update_next": (C.c_int, [_P, C.POINTER(Rollout), C.c_int, C.POINTER(PPOHyper), C.c_int, _P]),
  "v4l_host_cast_rows": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int64, _P, _P, C.c_int, C.c_int]),
  "v4l_host_cast_simd": (C.c_int, []),
  "v4l_actor_step_rows": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                    C.c_double, _P]),
  "v4l_trainer_update_run": (C.c_int, [_P, C.POINTER(Rollout), C.c_int, C.POINTER(PPOHyper), C.c_int, C.c_int, _P]),
  "v4l_trainer_critic_grads": (C.c_int, [_P, C.POINTER(Rollout), C.c_int, C.POINTER(PPOHyper), _P]),
  "v4l_trainer_critic_step": (C.c_int, [_P, C.POINTER(PPOHyper), _P]),
  "v4l_trainer_actor_grads": (C.c_int, [_P, C.POINTER(Rollout), C.c_int, C.POINTER(PPOHyper), _P]),
  "v4l_trainer_actor_step": (C.c_int, [_P, C.POINTER(PPOHyper), _P]),
  "v4l_trainer_stats_cur": (_P, [_P]),
  "v4l_trainer_update": (C.c_int, [_P, C.POINTER(Rollout), _P, C.c_int, C.POINTER(PPOHyper), C.c_double,
                                   C.c_double, C.c_int64, _P, _P]),
  "v4l_trainer_sync_target": (C.c_int, [_P, _P]),
  "v4l_comm_available": (C.c_int, []),
  "v4l_trainer_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
  "v4l_comm_unique_id": (C.c_int, [C.c_char_p]),
  "v4l_trainer_comm_init": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int]),
  "v4l_trainer_comm_destroy": (C.c_int, [_P]),
  "v4l_sync_grads": (C.c_int, [_P, C.c_int, _P]),
  "v4l_trainer_comm_selftest": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), _P]),
  "v4l_trainer_bucket_tail": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
  "v4l_net_ws_offset": (C.c_int64, [_P, C.c_int, C.c_char_p]),
  "v4l_prof_enable": (C.c_int, [C.c_int]),
  "v4l_prof_collect": (C.c_int64, [C.c_char_p, C.c_int64]),
}

_lib = None


def exported_symbols():
  """Every entry point include/v4l_hip.h declares (checked by the CPU test-suite)."""
  return sorted(_SIGS)


# V4L_* variables that configure a run; every OTHER V4L_* variable is a diagnostic switch that swaps kernels or schedules
# (include/v4l_hip.h lists them) — fine for A/B probes and the cross-check tests, a trap when one is left in a shell by accident
_CONFIG_ENV = {"V4L_COMPUTE", "V4L_GRAPH", "V4L_DP_COMM", "V4L_CAST_THREADS", "V4L_RCCL_LIB", "V4L_TRACE", "V4L_ROCTX", "V4L_LIB"}


def diagnostic_switches():
  """The diagnostic V4L_* switches present in the environment right now (name -> value)."""
  return {k: v for k, v in sorted(os.environ.items()) if k.startswith("V4L_") and k not in _CONFIG_ENV}


def lib():
  global _lib
  if _lib is None:
    active = diagnostic_switches()
    if active:  # said once, when the library is loaded: a stray switch must not change kernels under a user silently
      import warnings
      warnings.warn("vision4leg_amd: diagnostic switches are set and change which kernels / schedules run: %s"
                    % ", ".join("%s=%s" % kv for kv in active.items()), RuntimeWarning, stacklevel=2)
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
        "vision4leg_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the HIP hot path)" % LIB_PATH)
    l = C.CDLL(LIB_PATH)
    rebuild = "rebuild it (python -c 'import __graft_entry__ as g; g.build()')"
    # a stale .so is exactly what the guards below are for, and it lacks the newest symbols: resolve the version / ABI entries
    # first and say "rebuild" instead of dying on a bare AttributeError('undefined symbol') in the binding loop
    try:
      l.v4l_version.restype, l.v4l_version.argtypes = _SIGS["v4l_version"]
      l.v4l_abi_sizeof.restype, l.v4l_abi_sizeof.argtypes = _SIGS["v4l_abi_sizeof"]
    except AttributeError as e:
      raise RuntimeError("vision4leg_amd: %s predates this Python binding (%s) — %s" % (LIB_PATH, e, rebuild)) from None
    # the struct mirrors above must be the structs this library was compiled with (a stale .so next to newer host code reads
    # fields at the wrong offsets otherwise)
    for which, (name, mirror) in enumerate((("v4l_net_cfg", NetCfg), ("v4l_ppo_hyper", PPOHyper), ("v4l_rollout", Rollout))):
      have = l.v4l_abi_sizeof(which)
      if have != C.sizeof(mirror):
        raise RuntimeError("vision4leg_amd: %s was compiled with sizeof(%s) = %d, the Python binding expects %d — %s"
                           % (LIB_PATH, name, have, C.sizeof(mirror), rebuild))
    for name, (res, args) in _SIGS.items():
      try:
        fn = getattr(l, name)
      except AttributeError:
        raise RuntimeError("vision4leg_amd: %s does not export %s — %s" % (LIB_PATH, name, rebuild)) from None
      fn.restype = res
      fn.argtypes = args
    _lib = l
  return _lib


def check(rc, what=""):
  if rc != 0:
    msg = lib().v4l_last_error()
    raise RuntimeError("libv4l_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
