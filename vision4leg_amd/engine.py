"""Host-side glue between the torchrl-shaped Python modules and libv4l_hip.so.

torch is used for device memory (tensors own the parameters, workspaces and rollout arrays) and for the
current HIP stream; all arithmetic is enqueued through the C ABI. Nothing here falls back to torch ops.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib
from ._lib import (V4L_BF16, V4L_F16, V4L_F32, V4L_NET_CNN, V4L_NET_LOCO, V4L_NET_MLP, V4L_OUT_LD, V4L_STATS,
                   NetCfg, PPOHyper, Rollout, check)


def default_compute():
  """Contraction operand type: V4L_COMPUTE = f16 (default since round 6: IEEE half operands — bf16's speed, 8 x closer to the
  fp32 reference; scaled backward, include/v4l_hip.h V4L_F16) | bf16 (8 exponent bits: for observations / activations that
  can exceed half's 65 504) | f32 (the exact-fp32 parity mode)."""
  v = os.environ.get("V4L_COMPUTE", "f16").lower()
  if v in ("f32", "fp32", "float32"):
    return V4L_F32
  if v in ("bf16", "bfloat16"):
    return V4L_BF16
  if v in ("f16", "fp16", "float16", "half"):
    return V4L_F16
  raise ValueError("V4L_COMPUTE must be 'bf16', 'f16' or 'f32', got %r" % v)


COMPUTE_NAMES = {V4L_F32: "f32", V4L_BF16: "bf16", V4L_F16: "f16"}


def operand_dtype(compute):
  """torch dtype of the contraction operands (and of the stored depth stacks) in a compute mode."""
  return {V4L_F32: torch.float32, V4L_BF16: torch.bfloat16, V4L_F16: torch.float16}[compute]


def _stream():
  return torch.cuda.current_stream().cuda_stream


# ---- guard bands (V4L_GUARD=1, tests): every device buffer the library writes into is carved out of a larger allocation with
# a canary-filled band on both sides; check_guards() reports any band a kernel wrote into. The kernels index HBM with
# hand-computed offsets (53 of them) and device-side AddressSanitizer does not build for this library in finite time (DESIGN.md
# section 5), so this is the GPU-side out-of-bounds WRITE check the test-suite runs.
GUARD_BYTES = 64 * 1024
_CANARY = 0xA5
_guards = []


def _buf(count, dtype, device, zero=False):
  """A [count] device tensor of dtype; with V4L_GUARD=1 a view between two canary bands of GUARD_BYTES."""
  count = int(count)
  if os.environ.get("V4L_GUARD", "0") == "0":
    return torch.zeros(count, dtype=dtype, device=device) if zero else torch.empty(count, dtype=dtype, device=device)
  nbytes = count * torch.empty((), dtype=dtype).element_size()
  pad = (-nbytes) % 256  # keeps the tail band 256-byte aligned
  raw = torch.full((GUARD_BYTES + nbytes + pad + GUARD_BYTES,), _CANARY, dtype=torch.uint8, device=device)
  view = raw[GUARD_BYTES:GUARD_BYTES + nbytes].view(dtype)
  if zero:
    view.zero_()
  _guards.append((raw, nbytes))
  return view


def release_guard(view):
  """Forget the canary bands of a guarded buffer that is being replaced (HipNet.workspace keeps one workspace): without this
  V4L_GUARD=1 keeps every replaced allocation alive for the life of the process."""
  if view is None or not _guards:
    return
  want = view.data_ptr() - GUARD_BYTES
  _guards[:] = [(raw, n) for raw, n in _guards if raw.data_ptr() != want]


def check_guards(reset=False):
  """-> list of (buffer index, which band, first corrupted byte offset) for every canary band that was written into."""
  bad = []
  for i, (raw, nbytes) in enumerate(_guards):
    for name, band in (("head", raw[:GUARD_BYTES]), ("tail", raw[GUARD_BYTES + nbytes:])):
      hit = (band != _CANARY).nonzero()
      if hit.numel():
        bad.append((i, name, int(hit[0].item())))
  if reset:
    _guards.clear()
  return bad


def _ptr(t):
  return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_gpu(t, what):
  if not t.is_cuda:
    raise RuntimeError(
      "vision4leg_amd: %s must live on the GPU (got device %s). The HIP engine has no CPU path; "
      "move the module and its inputs with .to('cuda')." % (what, t.device))


class HipNet:
  """One v4l_net plan bound to the parameters of a top-level module (state_dict keys = reference names)."""

  def __init__(self, module, cfg):
    self.module = module
    self.cfg = cfg
    self.L = _lib.lib()
    h = C.c_void_p()
    check(self.L.v4l_net_create(C.byref(cfg), C.byref(h)), "v4l_net_create")
    self.h = h
    self.kind = cfg.kind
    self.compute = cfg.compute
    self.out_dim = cfg.out_dim
    self.state_dim = cfg.state_dim
    self.img_elems = 0 if cfg.kind == V4L_NET_MLP else cfg.in_channels * cfg.img_hw * cfg.img_hw
    self.Sp = self.L.v4l_net_state_ld(h)
    self.total_params = self.L.v4l_net_total_params(h)
    self.param_names, self.param_shapes, self.grad_offsets = [], [], []
    name, ndim, numel, goff = C.c_char_p(), C.c_int(), C.c_int64(), C.c_int64()
    shape = (C.c_int64 * 4)()
    for i in range(self.L.v4l_net_num_params(h)):
      check(self.L.v4l_net_param_info(h, i, C.byref(name), C.byref(ndim), shape, C.byref(numel), C.byref(goff)))
      self.param_names.append(name.value.decode())
      self.param_shapes.append(tuple(shape[d] for d in range(ndim.value)))
      self.grad_offsets.append(goff.value)
    self._ptrs = None
    self._tlist = None
    self._vsum = None
    self._versions = None
    self._dirty = True
    self._ws = {}
    self._stage = {}
    self.device = None

  def __del__(self):
    try:
      if getattr(self, "h", None):
        self.L.v4l_net_destroy(self.h)
        self.h = None
    except Exception:
      pass

  # ---- parameter binding -------------------------------------------------------------------
  def _tensors(self):
    sd = self.module.state_dict(keep_vars=True)
    ts = []
    for name, shp in zip(self.param_names, self.param_shapes):
      if name not in sd:
        raise RuntimeError("vision4leg_amd: module has no parameter %r expected by the HIP plan" % name)
      t = sd[name]
      if tuple(t.shape) != shp:
        raise RuntimeError("vision4leg_amd: parameter %s has shape %s, HIP plan expects %s"
                           % (name, tuple(t.shape), shp))
      if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("vision4leg_amd: parameter %s must be contiguous float32" % name)
      _require_gpu(t, "parameter " + name)
      ts.append(t)
    return ts

  def ensure_bound(self, fast=False):
    """Re-register parameter pointers / detect in-place parameter changes. fast=True (per-step callers) checks the
    cached tensor list only: versions of all tensors and the storage address of the first and last one."""
    if fast and self._tlist is not None:
      ts = self._tlist
      if ts[0].data_ptr() == self._ptrs[0] and ts[-1].data_ptr() == self._ptrs[-1]:
        vers = 0
        for t in ts:
          vers += t._version
        if vers != self._vsum:
          self._vsum = vers
          self._versions = [t._version for t in ts]
          self._dirty = True
        return ts
    ts = self._tensors()
    self._tlist = ts
    ptrs = [t.data_ptr() for t in ts]
    if ptrs != self._ptrs:
      self.device = ts[0].device
      with torch.cuda.device(self.device):
        self._packed = _buf(self.L.v4l_net_packed_bytes(self.h), torch.uint8, self.device)
        self._table = _buf(self.L.v4l_net_table_bytes(self.h), torch.uint8, self.device)
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        check(self.L.v4l_net_bind(self.h, arr, _ptr(self._packed), _ptr(self._table), _stream()), "v4l_net_bind")
      self._ptrs = ptrs
      self._dirty = True
      self._ws.clear()
      self._stage.clear()
    vers = [t._version for t in ts]
    if vers != self._versions:
      self._versions = vers
      self._dirty = True
    self._vsum = sum(vers)
    return ts

  def mark_dirty(self):
    """Parameters were changed behind torch's back (an optimiser step inside the library, or a write through
    `param.data`, which has its own version counter: copy_model_params_from_to, soft_update_from_to, init.*,
    dist.broadcast(p.data))."""
    self._dirty = True

  def pack_if_needed(self, fast=False):
    """Refresh the operand-type weight copies. Module-level calls (fast=False: pf(x), vf(x), explore, update) repack
    unconditionally — one ~5 us launch — because `.data` writes are invisible to the version check; per-step callers
    (fast=True: RolloutActor) rely on the version sum and on mark_dirty() / module.mark_params_changed()."""
    self.ensure_bound(fast)
    if self._dirty or not fast:
      check(self.L.v4l_net_pack(self.h, _stream()), "v4l_net_pack")
      self._dirty = False

  # ---- buffers -----------------------------------------------------------------------------
  def ws_floats(self, n):
    return self.L.v4l_net_ws_floats(self.h, n, 1)

  def workspace(self, n):
    ws = self._ws.get(n)
    if ws is None:
      for old in self._ws.values():
        release_guard(old)
      ws = _buf(self.ws_floats(n), torch.float32, self.device)
      self._ws = {n: ws}  # keep one
    return ws

  def image_dtype(self):
    return operand_dtype(self.compute)

  def grad_scale(self, n):
    """The library's rule for MEAN-loss gradient rows (size ~1/n): what they are multiplied by before a backward pass over n rows
    (1 unless compute == f16); the gradients come out unscaled (include/v4l_hip.h v4l_net_grad_scale)."""
    return float(self.L.v4l_net_grad_scale(self.h, int(n)))

  @staticmethod
  def f16_scale_for(amax):
    """The power of two that puts a largest |d(out)| element of `amax` into [2^12, 2^13): a factor of 8 under half's overflow for
    the row itself, room for the backward's growth (LayerNorm's 1/sigma), and the bulk far above half's smallest normal."""
    if not (amax > 0.0) or not np.isfinite(amax):
      return 1.0
    return float(2.0 ** min(30, max(-14, int(np.floor(np.log2(8192.0 / amax))))))

  def alloc_rollout(self, slots, device):
    state = _buf(slots * self.Sp, torch.float32, device, zero=True).view(slots, self.Sp)
    image = None
    if self.img_elems:
      image = _buf(slots * self.img_elems, self.image_dtype(), device, zero=True).view(slots, self.img_elems)
    return state, image

  def ingest(self, obs, state, image, slot0=0):
    """obs: [n][S + C*H*W] float32 cuda tensor in the reference's row layout."""
    _require_gpu(obs, "observation batch")
    if obs.dtype != torch.float32 or not obs.is_contiguous():
      obs = obs.contiguous().float()
    n, d = obs.shape
    if d != self.state_dim + self.img_elems:
      raise RuntimeError("vision4leg_amd: observation rows have %d columns, net expects %d (+%d image)"
                         % (d, self.state_dim, self.img_elems))
    check(self.L.v4l_ingest(self.h, _ptr(obs), n, _ptr(state), _ptr(image), slot0, _stream()), "v4l_ingest")

  def stage(self, obs):
    """Ingest a batch of reference observation rows into per-net staging arrays; returns (state, image, n)."""
    obs = obs.reshape(-1, obs.shape[-1])
    n = obs.shape[0]
    st = self._stage.get(n)
    if st is None:
      st = self.alloc_rollout(n, obs.device)
      self._stage = {n: st}
    self.ingest(obs, st[0], st[1])
    return st[0], st[1], n

  # ---- compute -----------------------------------------------------------------------------
  def forward(self, state, image, n, rowidx=None, ws=None, train=False):
    """Runs the net; returns the padded head output [n][V4L_OUT_LD] as a view into the workspace."""
    self.pack_if_needed()
    if ws is None:
      ws = self.workspace(n)
    check(self.L.v4l_net_forward(self.h, _ptr(state), _ptr(image), _ptr(rowidx), n, _ptr(ws), int(train), _stream()),
          "v4l_net_forward")
    off = self.ws_offset(n, "out")
    return ws[off:off + n * V4L_OUT_LD].view(n, V4L_OUT_LD)

  def backward(self, state, image, n, dout, grads, rowidx=None, ws=None, scale=None):
    """dout: [n][V4L_OUT_LD] (zero padded), unscaled. grads: flat float32 buffer, every element written.
    f16 compute: the rows enter the backward multiplied by a power of two (`scale`; default: chosen from max |dout| — one
    device-to-host read — so that any row size is safe; last_grad_scale keeps it) and the gradients come out unscaled."""
    if ws is None:
      ws = self.workspace(n)
    off = self.ws_offset(n, "dout")
    gs = 1.0
    if self.compute == V4L_F16:
      gs = float(scale) if scale is not None else self.f16_scale_for(float(dout.abs().max().item()))
      check(self.L.v4l_net_set_grad_scale(self.h, gs), "v4l_net_set_grad_scale")
    self.last_grad_scale = gs
    ws[off:off + n * V4L_OUT_LD].view(n, V4L_OUT_LD).copy_(dout if gs == 1.0 else dout * gs)
    check(self.L.v4l_net_backward(self.h, _ptr(state), _ptr(image), _ptr(rowidx), n, _ptr(ws), _ptr(grads), _stream()),
          "v4l_net_backward")

  def ws_offset(self, n, name):
    off = self.L.v4l_net_ws_offset(self.h, n, name.encode())
    if off < 0:
      raise KeyError(name)
    return off

  def ws_view(self, n, name, rows, cols, ws=None):
    ws = self.workspace(n) if ws is None else ws
    off = self.ws_offset(n, name)
    return ws[off:off + rows * cols].view(rows, cols)

  def grad_view(self, grads, name):
    i = self.param_names.index(name)
    shp = self.param_shapes[i]
    o = self.grad_offsets[i]
    return grads[o:o + int(np.prod(shp))].view(shp)

  def value(self, obs):
    """vf(x): [n][1]"""
    st, im, n = self.stage(obs)
    out = self.forward(st, im, n)
    v = torch.empty(n, dtype=torch.float32, device=obs.device)
    check(self.L.v4l_col0(_ptr(out), n, _ptr(v), _stream()), "v4l_col0")
    return v.view(n, 1)

  def gaussian(self, obs, logstd, acts=None, tanh_action=False):
    """Policy head: mean/std [n][A], clamped log_std [A], ent [n][1] and log_prob [n][1] of acts (optional)."""
    st, im, n = self.stage(obs)
    out = self.forward(st, im, n)
    return self.gauss_head(out, logstd, n, acts, tanh_action=tanh_action)

  def gauss_head(self, out, logstd, n, acts=None, tanh_action=False, pre_tanh=None):
    A, dev = self.out_dim, out.device
    mean = torch.empty(n, A, dtype=torch.float32, device=dev)
    std = torch.empty(n, A, dtype=torch.float32, device=dev)
    lsc = torch.empty(A, dtype=torch.float32, device=dev)
    ent = torch.empty(n, dtype=torch.float32, device=dev)
    logp = None
    if acts is not None:
      acts = acts.reshape(n, A).contiguous().float()
      logp = torch.empty(n, dtype=torch.float32, device=dev)
    if tanh_action:  # TanhNormal.log_prob (policies/distribution.py:38-51)
      if pre_tanh is not None:
        pre_tanh = pre_tanh.reshape(n, A).contiguous().float()
      check(self.L.v4l_gauss_head_tanh(_ptr(out), _ptr(logstd), _ptr(acts), _ptr(pre_tanh), n, A, _ptr(mean), _ptr(std),
                                       _ptr(lsc), _ptr(ent), _ptr(logp), _stream()), "v4l_gauss_head_tanh")
    else:
      check(self.L.v4l_gauss_head(_ptr(out), _ptr(logstd), _ptr(acts), n, A, _ptr(mean), _ptr(std), _ptr(lsc),
                                  _ptr(ent), _ptr(logp), _stream()), "v4l_gauss_head")
    return mean, std, lsc, ent.view(n, 1), (logp.view(n, 1) if logp is not None else None)


class HipTrainer:
  """PPO minibatch update over (pf, vf, target_pf): owns the flat grad/Adam buffers, the shared workspace, the
  device control block and the side stream the update graph is captured on. Mirrors PPO.update of
  torchrl/algo/on_policy/ppo.py:125-153."""

  def __init__(self, pf_net, vf_net, tpf_net, batch, clip_para, entropy_coeff, max_grad_norm=0.5,
               betas=(0.9, 0.999), eps=1e-5, clipped_value_loss=False, world_size=1):
    self.pf, self.vf, self.tpf = pf_net, vf_net, tpf_net
    self.L = _lib.lib()
    for net in (pf_net, vf_net, tpf_net):
      net.ensure_bound()
    self.device = pf_net.device
    h = C.c_void_p()
    check(self.L.v4l_trainer_create(pf_net.h, vf_net.h, tpf_net.h, C.byref(h)), "v4l_trainer_create")
    self.h = h
    dev = self.device
    z = lambda n: _buf(n, torch.float32, dev, zero=True)
    # gradient buckets: [total_params | V4L_BUCKET_TAIL scalars that ride through the data-parallel all-reduce]
    self.g_pf_bucket = z(pf_net.total_params + _lib.V4L_BUCKET_TAIL)
    self.g_vf_bucket = z(vf_net.total_params + _lib.V4L_BUCKET_TAIL)
    self.g_pf, self.m_pf, self.v_pf = self.g_pf_bucket[:pf_net.total_params], z(pf_net.total_params), z(pf_net.total_params)
    self.g_vf = self.g_vf_bucket[:vf_net.total_params]
    self.m_vf, self.v_vf = z(vf_net.total_params), z(vf_net.total_params)
    self.batch = 0
    self.stream = torch.cuda.Stream(device=dev)  # hipGraph capture needs a non-default stream
    self._alloc_ws(batch)
    self.hp = PPOHyper(clip_para, entropy_coeff, max_grad_norm, betas[0], betas[1], eps,
                       int(bool(clipped_value_loss)), int(world_size))
    self.step = 0  # Adam steps taken (both optimisers step once per update)
    self._one_stats = torch.zeros(V4L_STATS, dtype=torch.float32, device=dev)
    self.has_comm = False

  def comm_init(self, comm_id, rank, world):
    """Attach an RCCL communicator (v4l_trainer_comm_init): from then on update_next() issues the two all-reduces of an
    update itself, on its stream, inside the captured graph. comm_id: the 128 bytes rank 0 got from comm_unique_id()."""
    check(self.L.v4l_trainer_comm_init(self.h, bytes(comm_id), int(rank), int(world)), "v4l_trainer_comm_init")
    self.has_comm = True

  @staticmethod
  def comm_available():
    """True when this process can load RCCL (v4l_comm_available); ranks agree on it before the communicator rendezvous."""
    return _lib.lib().v4l_comm_available() == 0

  def comm_info(self):
    """(rank, ranks) of the attached communicator as RCCL reports them; (0, 1) without one."""
    world, rank = C.c_int(1), C.c_int(0)
    check(self.L.v4l_trainer_comm_info(self.h, C.byref(rank), C.byref(world)), "v4l_trainer_comm_info")
    return rank.value, world.value

  def comm_world(self):
    """Ranks of the attached communicator as RCCL reports them (ncclCommCount), 1 without one."""
    world = C.c_int(1)
    rank = C.c_int(0)
    check(self.L.v4l_trainer_comm_info(self.h, C.byref(rank), C.byref(world)), "v4l_trainer_comm_info")
    return world.value

  def comm_selftest(self, graph=True):
    """v4l_trainer_comm_selftest on the current stream: -> number of wrong elements on this rank (0 = the communicator
    all-reduces both buckets correctly, eagerly and as a replayed graph). Overwrites the gradient buckets."""
    bad = C.c_int64(-1)
    check(self.L.v4l_trainer_comm_selftest(self.h, int(graph), C.byref(bad), _stream()), "v4l_trainer_comm_selftest")
    return bad.value

  def comm_destroy(self):
    check(self.L.v4l_trainer_comm_destroy(self.h), "v4l_trainer_comm_destroy")
    self.has_comm = False

  @staticmethod
  def comm_unique_id():
    buf = C.create_string_buffer(_lib.V4L_COMM_ID_BYTES)
    check(_lib.lib().v4l_comm_unique_id(buf), "v4l_comm_unique_id")
    return buf.raw

  def bucket_tail(self, which, pack, world):
    check(self.L.v4l_trainer_bucket_tail(self.h, int(which), int(pack), int(world), _stream()), "v4l_trainer_bucket_tail")

  def __del__(self):
    try:
      if getattr(self, "h", None):
        self.L.v4l_trainer_destroy(self.h)
        self.h = None
    except Exception:
      pass

  def _alloc_ws(self, n):
    if n <= self.batch:
      return
    self.batch = n
    self.ws = _buf(self.L.v4l_trainer_ws_floats(self.h, n), torch.float32, self.device)
    self.ctl = _buf(self.L.v4l_trainer_ctl_bytes(self.h, n), torch.uint8, self.device, zero=True)
    check(self.L.v4l_trainer_bind(self.h, _ptr(self.g_pf), _ptr(self.m_pf), _ptr(self.v_pf), _ptr(self.g_vf),
                                  _ptr(self.m_vf), _ptr(self.v_vf), _ptr(self.ws), self.ws.numel(), _ptr(self.ctl), n,
                                  _stream()), "v4l_trainer_bind")

  @staticmethod
  def rollout(state, image, acts, advs, rets, values=None, logp_old=None):
    """logp_old: [slots] log pi_old(a|s) recorded at action time (RolloutActor) -> the update skips the frozen
    target policy's forward pass; None -> it is evaluated per minibatch like the reference does."""
    ro = Rollout(state.data_ptr(), image.data_ptr() if image is not None else None, acts.data_ptr(),
                 advs.data_ptr(), rets.data_ptr(), values.data_ptr() if values is not None else None,
                 logp_old.data_ptr() if logp_old is not None else None)
    ro._keep = (state, image, acts, advs, rets, values, logp_old)
    return ro

  def sync_target(self):
    """copy_model_params_from_to(pf, target_pf) + repack of the frozen target (ppo.py:34)."""
    for net in (self.pf, self.tpf):
      net.ensure_bound()
    check(self.L.v4l_trainer_sync_target(self.h, _stream()), "v4l_trainer_sync_target")
    self.tpf._dirty = False

  def _pre(self, n):
    self._alloc_ws(n)
    for net in (self.pf, self.vf, self.tpf):
      net.ensure_bound()
    if self.tpf._dirty:
      self.tpf.pack_if_needed()

  def _after_steps(self):
    # the library repacks pf/vf itself at the start of every grads phase; other users of the nets must repack
    for net in (self.pf, self.vf):
      net.mark_dirty()

  def begin(self, rowidx_all, stats_all, lr_pf, lr_vf):
    """Open a run of updates on rows rowidx_all[u] (int32 [U][n] device tensor or None) -> stats_all[u]."""
    self._run = (rowidx_all, stats_all)  # keep alive
    check(self.L.v4l_trainer_begin(self.h, _ptr(rowidx_all), _ptr(stats_all), float(lr_pf), float(lr_vf), self.step,
                                   C.byref(self.hp), _stream()), "v4l_trainer_begin")

  def update_next(self, ro, n, graph=True):
    self._pre(n)
    check(self.L.v4l_trainer_update_next(self.h, C.byref(ro), n, C.byref(self.hp), int(graph), _stream()),
          "v4l_trainer_update_next")
    self.step += 1
    self._after_steps()

  def update_run(self, ro, n, count, graph=True):
    """The next `count` updates behind one begin(): with graph=True ONE hipGraph replay for all of them (v4l_trainer_update_run)."""
    self._pre(n)
    check(self.L.v4l_trainer_update_run(self.h, C.byref(ro), n, C.byref(self.hp), int(count), int(graph), _stream()),
          "v4l_trainer_update_run")
    self.step += int(count)
    self._after_steps()

  def update(self, ro, rowidx, n, lr_pf, lr_vf, stats):
    """One eager PPO.update on minibatch rows rowidx (int32 device tensor or None); stats: [V4L_STATS] floats."""
    self._pre(n)
    self.begin(rowidx, stats, lr_pf, lr_vf)
    self.update_next(ro, n, graph=False)

  # phases, for the data-parallel schedule (all-reduce between grads and step); begin() first
  def stats_cur(self):
    if getattr(self, "_stats_cur", None) is None or self._stats_cur_ctl is not self.ctl:
      off = 256  # layout of the control buffer: [UpdCtl | stats_cur | rowidx_cur]
      self._stats_cur = self.ctl[off:off + 4 * V4L_STATS].view(torch.float32)
      self._stats_cur_ctl = self.ctl
    return self._stats_cur

  def critic_grads(self, ro, n):
    self._pre(n)
    check(self.L.v4l_trainer_critic_grads(self.h, C.byref(ro), n, C.byref(self.hp), _stream()), "v4l_trainer_critic_grads")

  def critic_step(self):
    check(self.L.v4l_trainer_critic_step(self.h, C.byref(self.hp), _stream()), "v4l_trainer_critic_step")
    self.vf.mark_dirty(); self.pf.mark_dirty()

  def actor_grads(self, ro, n):
    check(self.L.v4l_trainer_actor_grads(self.h, C.byref(ro), n, C.byref(self.hp), _stream()), "v4l_trainer_actor_grads")

  def actor_step(self):
    check(self.L.v4l_trainer_actor_step(self.h, C.byref(self.hp), _stream()), "v4l_trainer_actor_step")
    self.step += 1
    self._after_steps()


class HipActor:
  """One rollout step for E envs as a single captured launch sequence: pf.explore + vf on a shared encoder pass
  (reference protocol: torchrl/collector/on_policy.py:90-100), with the step's observation rows, action and value
  filed straight into the HBM-resident rollout arrays. The step cursor lives on the device."""

  def __init__(self, pf_net, vf_net, E, rollout=None, shared_encoder=True, graph=True):
    self.pf, self.vf, self.E = pf_net, vf_net, E
    self.L = _lib.lib()
    for net in (pf_net, vf_net):
      net.ensure_bound()
    dev = self.device = pf_net.device
    h = C.c_void_p()
    check(self.L.v4l_actor_create(pf_net.h, vf_net.h, E, C.byref(h)), "v4l_actor_create")
    self.h = h
    A = pf_net.out_dim
    self.ws = _buf(self.L.v4l_actor_ws_floats(h), torch.float32, dev)
    self.ctl = _buf(self.L.v4l_actor_ctl_bytes(h), torch.uint8, dev, zero=True)
    self.obs = torch.zeros(E, pf_net.state_dim + pf_net.img_elems, dtype=torch.float32, device=dev)
    self.eps = torch.zeros(E, A, dtype=torch.float32, device=dev)
    z = lambda *shape: _buf(int(np.prod(shape)), torch.float32, dev, zero=True).view(*shape)
    self.action, self.mean, self.std, self.ent, self.value = z(E, A), z(E, A), z(E, A), z(E, 1), z(E, 1)
    self.shared_encoder, self.graph = bool(shared_encoder), bool(graph)
    self.stream = torch.cuda.Stream(device=dev)
    self.attach(rollout)
    check(self.L.v4l_actor_bind(h, _ptr(self.ws), _ptr(self.ctl), _stream()), "v4l_actor_bind")

  def __del__(self):
    try:
      if getattr(self, "h", None):
        self.L.v4l_actor_destroy(self.h)
        self.h = None
    except Exception:
      pass

  def attach(self, rollout):
    """rollout = (state [slots][Sp], image [slots][C*H*W] | None, acts [slots][A] | None, values [slots] | None
    [, logp [slots] | None]); None: private E-slot scratch, nothing is filed and the cursor is rewound every step."""
    self.own = rollout is None
    if rollout is None:
      st, im = self.pf.alloc_rollout(self.E, self.device)
      rollout = (st, im, None, None, None)
    if len(rollout) == 4:
      rollout = tuple(rollout) + (None,)
    self.rollout = rollout
    st, im, acts, vals, logp = rollout
    self._obs_ptr = self.obs.data_ptr()
    self._args = (_ptr(self.obs), _ptr(self.eps), _ptr(st), _ptr(im), _ptr(acts), _ptr(vals), _ptr(logp), _ptr(self.action),
                  _ptr(self.mean), _ptr(self.std), _ptr(self.ent), _ptr(self.value), int(self.shared_encoder),
                  int(self.graph))
    self._out = {"action": self.action, "mean": self.mean, "std": self.std, "ent": self.ent, "value": self.value}

  def seek(self, t):
    check(self.L.v4l_actor_seek(self.h, int(t), _stream()), "v4l_actor_seek")

  # ---- are the operand-type weight copies current? Every step asks both nets (version counters of ~60 parameter tensors each:
  # ~6 us per net and step in the interpreter). A caller that owns the loop and knows the parameters stand still — the collector's
  # train_one_epoch: nothing steps an optimiser between two env steps — asks once and freezes the answer for the loop.
  _frozen = False

  def _refresh_packs(self):
    if not self._frozen:
      self.pf.pack_if_needed(fast=True)
      self.vf.pack_if_needed(fast=True)

  def freeze_params(self, on):
    """on=True: check the parameters now (repacking if they changed) and skip the per-step check until freeze_params(False).
    The caller promises not to change pf / vf parameters in between (a change would go unnoticed until the freeze ends)."""
    self._frozen = False
    if on:
      self._refresh_packs()
    self._frozen = bool(on)

  def check(self):
    """v4l_actor_check: raise if a device-side hand-over of the rollout step timed out since the last check (the affected
    steps filed NaN actions rather than numbers computed from stale activations). One stream synchronise; once per epoch."""
    err = C.c_int(0)
    check(self.L.v4l_actor_check(self.h, C.byref(err), _stream()), "v4l_actor_check")
    if err.value:
      raise RuntimeError("vision4leg_amd: a block of the rollout step gave up waiting for hand-over counter %d "
                         "(stale activations; the step's actions were replaced by NaN)" % (err.value - 1))

  def draw_noise(self, n_steps):
    """Draw the standard normals of the next n_steps env steps with ONE generator call ([n_steps][E][A], torch's
    generator on the current stream) instead of one 5 us launch per step; the following non-deterministic eager steps
    consume one [E][A] slice each. Same distribution and generator as the per-step draws, a different position in its
    stream — call it only where reproducing a per-step seeded sequence does not matter."""
    E, A = self.eps.shape
    if getattr(self, "_bulk_buf", None) is None or self._bulk_buf.shape[0] != n_steps:
      self._bulk_buf = torch.empty(n_steps, E, A, dtype=torch.float32, device=self.device)
    self._bulk_buf.normal_()
    self._bulk, self._bulk_t = self._bulk_buf, 0

  def step(self, obs, deterministic=False):
    """obs: [E][S+C*H*W] float32 cuda rows of this env step. Returns a dict of views of fixed output buffers
    (valid until the next step): action/mean/std [E][A], ent/value [E][1]. deterministic: no draw, action == mean
    (the eval_act / deployment protocol, policies/continuous_policy.py:78-83)."""
    _require_gpu(obs, "observation batch")
    if self.graph:  # capture needs a non-default stream
      cur = torch.cuda.current_stream(self.device)
      self.stream.wait_stream(cur)
      with torch.cuda.stream(self.stream):
        self._step(obs, deterministic)
      cur.wait_stream(self.stream)
    else:
      self._step(obs, deterministic)
    return self._out

  def step_host(self, obs_pinned, deterministic=False):
    """The collector's per-step call without any copy of its own: `obs_pinned` is a PINNED host float32 tensor
    [E][S+C*H*W] that the rollout kernels read in place over PCIe (pinned host memory is mapped into the device's address
    space), and the action AND the value land in pinned host buffers the same way. One launch pair; the call returns when both
    outputs of every env have arrived (no stream synchronise unless polling is off / times out), i.e. when every block that
    reads `obs_pinned` has finished: the caller may overwrite the observation buffer as soon as this returns. Returns the
    [E][A] action as a numpy view of that buffer (valid until the next step). Eager launches only."""
    if self.graph:
      raise RuntimeError("vision4leg_amd: step_host drives eager launches (construct the actor with graph=False)")
    if (not obs_pinned.is_pinned() or obs_pinned.dtype != torch.float32 or not obs_pinned.is_contiguous()
        or obs_pinned.numel() != self.obs.numel()):
      raise RuntimeError("vision4leg_amd: step_host needs a pinned, contiguous float32 [E][S+C*H*W] host tensor")
    self._host_outputs()
    if getattr(self, "_args_host_of", None) is not self._args:  # first call, or attach() rebuilt the argument tuple
      self._args_host = (self._args[:7] + (C.c_void_p(self._act_host.data_ptr()),) + self._args[8:11]
                         + (C.c_void_p(self._val_host.data_ptr()),) + self._args[12:])
      self._args_host_of = self._args
    self._refresh_packs()
    args = (C.c_void_p(obs_pinned.data_ptr()),) + self._args_host[1:]
    bulk = getattr(self, "_bulk", None)
    if not deterministic and bulk is not None:
      args = args[:1] + (C.c_void_p(bulk.data_ptr() + self._bulk_t * bulk.stride(0) * 4),) + args[2:]
      self._bulk_t += 1
      if self._bulk_t >= bulk.shape[0]:
        self._bulk = None
    elif not deterministic:
      self.eps.normal_()
      self._eps_zero = False
    elif not getattr(self, "_eps_zero", False):
      self.eps.zero_()
      self._eps_zero = True
    if self.own:
      self.seek(0)
    self._arm_action()
    check(self.L.v4l_actor_step(self.h, *args, _stream()), "v4l_actor_step")
    return self._await_action()

  # ---- completion of a host step: the policy blocks write the [E][A] action and the value blocks the [E] value straight into
  # pinned host memory, so the host can watch them ARRIVE instead of asking the runtime for a stream synchronise (whose wake-up
  # costs 20-50 us on ROCm 7.2: tools/probe/collector_pipe.py). Both buffers are armed with NaN before the launch; the step is
  # complete for the host when no NaN is left in EITHER — every (env, net) block writes its output last, after it has read its
  # observation row, so at that point no block of the step still reads the caller's pinned observation buffer (the one-launch
  # state-MLP / NatureCNN step kernels run policy and value blocks side by side: watching the action alone would not cover the
  # value blocks' reads; round 6, advisor finding). A net that really produces NaN never disarms its buffer: after POLL_SECONDS
  # of wall time the call falls back to the synchronise and returns what is there (the collector's non-finite check then raises,
  # as it always did). V4L_STEP_POLL=0 (read when the actor takes its first host step): always synchronise.
  POLL_SECONDS = 200e-6

  def _host_outputs(self):
    if getattr(self, "_act_host", None) is None:
      self._act_host = torch.zeros(self.action.shape, dtype=torch.float32).pin_memory()
      self._val_host = torch.zeros(self.value.shape, dtype=torch.float32).pin_memory()

  def _arm_action(self):
    if getattr(self, "_act_np", None) is None or self._act_np_of is not self._act_host:
      self._act_np, self._val_np, self._act_np_of = self._act_host.numpy(), self._val_host.numpy(), self._act_host
      self._poll = os.environ.get("V4L_STEP_POLL", "1") != "0"
    if self._poll:
      self._act_np.fill(np.nan)
      self._val_np.fill(np.nan)

  def _await_action(self):
    a, v = self._act_np, self._val_np
    if self._poll:
      isnan, now = np.isnan, time.perf_counter
      t_end = now() + self.POLL_SECONDS
      while True:
        # (the value is looked at only once the action is there: one small array per look while waiting)
        if not isnan(a).any() and not isnan(v).any():
          return a
        if now() > t_end:
          break
    torch.cuda.current_stream(self.device).synchronize()
    return a

  def split_supported(self):
    """True when this actor's step can take the observation split (v4l_actor_step_split): bf16 compute, an image net on the
    fused rollout step, eager launches."""
    return (not self.graph) and bool(self.L.v4l_actor_split_supported(self.h, int(self.shared_encoder)))

  def split_device_buffers(self):
    """HBM landing buffers of the pipelined observation hand-over ([E][max(S,1)] float32, [E][C*H*W] bfloat16)."""
    if getattr(self, "_split_dev", None) is None:
      self._split_dev = (torch.empty(self.E, max(self.pf.state_dim, 1), dtype=torch.float32, device=self.device),
                         torch.empty(self.E, self.pf.img_elems, dtype=self.pf.image_dtype(), device=self.device))
    return self._split_dev

  def step_host_split(self, prop_pinned, img16_pinned, deterministic=False, via_copy=None, on_device=False):
    """step_host with the observation split: `prop_pinned` [E][S] float32 (None when the net has no proprio input) and
    `img16_pinned` [E][C*H*W] bfloat16, both PINNED host tensors the rollout kernels read in place — the depth stack crosses
    PCIe in the type the kernels round it to anyway (half the bytes of fp32 rows; same results bit for bit). Returns the [E][A]
    action as a numpy view of a pinned buffer (valid until the next step).
    via_copy (V4L_SPLIT_VIA_COPY=1; default off): the rows first go to HBM with two asynchronous copies on the launch stream
    and the kernels read them there — measured SLOWER (90 us per step against 72 us for the in-place read, 89 us for fp32 rows
    in place: each copy costs ~10 us of launch / completion latency on top of its bytes; tools/probe/collector_stages.py,
    profiles/r4_collector_stages.txt), kept as a switch for hosts whose PCIe reads from the GPU side are slower."""
    if self.graph:
      raise RuntimeError("vision4leg_amd: step_host_split drives eager launches (construct the actor with graph=False)")
    S = self.pf.state_dim
    # on_device: the caller already moved the rows into split_device_buffers() on this stream (the collector's pipelined
    # hand-over: cast a row chunk, start its DMA, cast the next chunk under it) — same kernels, reading HBM
    there = (lambda t: t.is_cuda) if on_device else (lambda t: t.is_pinned())
    ok = (there(img16_pinned) and img16_pinned.dtype == self.pf.image_dtype() and img16_pinned.is_contiguous()
          and tuple(img16_pinned.shape) == (self.E, self.pf.img_elems))
    if S:
      ok = ok and (prop_pinned is not None and there(prop_pinned) and prop_pinned.dtype == torch.float32
                   and prop_pinned.is_contiguous() and tuple(prop_pinned.shape) == (self.E, max(S, 1) if on_device else S))
    if not ok:
      raise RuntimeError("vision4leg_amd: step_host_split needs pinned, contiguous [E][S] float32 and [E][C*H*W] %s host tensors"
                         % str(self.pf.image_dtype()))
    self._host_outputs()
    self._refresh_packs()
    a = self._args  # (obs, eps, st, im, acts, vals, logp, action, mean, std, ent, value, shared_encoder, graph)
    eps = a[1]
    bulk = getattr(self, "_bulk", None)
    if not deterministic and bulk is not None:
      eps = C.c_void_p(bulk.data_ptr() + self._bulk_t * bulk.stride(0) * 4)
      self._bulk_t += 1
      if self._bulk_t >= bulk.shape[0]:
        self._bulk = None
    elif not deterministic:
      self.eps.normal_()
      self._eps_zero = False
    elif not getattr(self, "_eps_zero", False):
      self.eps.zero_()
      self._eps_zero = True
    if self.own:
      self.seek(0)
    if via_copy is None:
      via_copy = (not on_device) and os.environ.get("V4L_SPLIT_VIA_COPY", "0") == "1"
    if via_copy:
      dprop, dimg = self.split_device_buffers()
      if S:
        dprop.copy_(prop_pinned, non_blocking=True)
      dimg.copy_(img16_pinned, non_blocking=True)
      prop_pinned, img16_pinned = dprop, dimg
    self._arm_action()
    check(self.L.v4l_actor_step_split(self.h, C.c_void_p(prop_pinned.data_ptr() if S else 0), C.c_void_p(img16_pinned.data_ptr()),
                                      eps, a[2], a[3], a[4], a[5], a[6], C.c_void_p(self._act_host.data_ptr()), a[8], a[9], a[10],
                                      C.c_void_p(self._val_host.data_ptr()), a[12], _stream()), "v4l_actor_step_split")
    return self._await_action()

  def step_host_rows(self, rows, deterministic=False, threads=8):
    """The collector's env step as ONE library call (v4l_actor_step_rows, csrc/host_step.h): `rows` is the numpy float64
    [E][S+C*H*W] array the env wrappers hand over (C-contiguous, any memory); the library casts it on its own thread pool into
    this actor's pinned staging blocks (fp32 proprio | 16-bit depth stack: the same two roundings as torch.Tensor(ob) followed by
    the kernels' ingest), issues the step's two launches on them and watches the pinned action / value buffers fill. No torch
    call and no Python loop per step. Returns the [E][A] action as a numpy view of pinned memory (valid until the next step).
    Needs split_supported(); eager launches only."""
    if self.graph:
      raise RuntimeError("vision4leg_amd: step_host_rows drives eager launches (construct the actor with graph=False)")
    S, img = self.pf.state_dim, self.pf.img_elems
    if (rows.dtype != np.float64 or rows.ndim != 2 or rows.shape != (self.E, S + img) or not rows.flags.c_contiguous):
      raise RuntimeError("vision4leg_amd: step_host_rows needs a C-contiguous float64 [E][S+C*H*W] array")
    self._host_outputs()
    if getattr(self, "_rows_pins", None) is None:
      self._rows_pins = (torch.empty(self.E, max(S, 1), dtype=torch.float32).pin_memory(),
                         torch.empty(self.E, img, dtype=self.pf.image_dtype()).pin_memory())
      self._poll = os.environ.get("V4L_STEP_POLL", "1") != "0"
      self._act_np, self._val_np, self._act_np_of = self._act_host.numpy(), self._val_host.numpy(), self._act_host
    self._refresh_packs()
    a = self._args  # (obs, eps, st, im, acts, vals, logp, action, mean, std, ent, value, shared_encoder, graph)
    eps = a[1]
    bulk = getattr(self, "_bulk", None)
    if not deterministic and bulk is not None:
      eps = C.c_void_p(bulk.data_ptr() + self._bulk_t * bulk.stride(0) * 4)
      self._bulk_t += 1
      if self._bulk_t >= bulk.shape[0]:
        self._bulk = None
    elif not deterministic:
      self.eps.normal_()
      self._eps_zero = False
    elif not getattr(self, "_eps_zero", False):
      self.eps.zero_()
      self._eps_zero = True
    if self.own:
      self.seek(0)
    # (the arguments that do not change from step to step are converted once: 20 ctypes conversions, `rows.ctypes` and four
    # data_ptr() calls per step were ~4 us of interpreter time)
    fixed = getattr(self, "_rows_fixed", None)
    if fixed is None or fixed[0] is not a:
      prop, img16 = self._rows_pins
      fixed = self._rows_fixed = (a, C.c_void_p(prop.data_ptr()), C.c_void_p(img16.data_ptr()),
                                  C.c_void_p(self._act_host.data_ptr()), C.c_void_p(self._val_host.data_ptr()))
    rc = self.L.v4l_actor_step_rows(self.h, C.c_void_p(rows.__array_interface__["data"][0]), rows.shape[1], fixed[1], fixed[2], eps,
                                    a[2], a[3], a[4], a[5], a[6], fixed[3], a[8], a[9], a[10], fixed[4], a[12], int(threads),
                                    self.POLL_SECONDS if self._poll else 0.0, _stream())
    if rc == 1:  # not waited for / timed out: the stream's completion is the step's
      torch.cuda.current_stream(self.device).synchronize()
    elif rc != 0:
      check(rc, "v4l_actor_step_rows")
    return self._act_np

  def _step(self, obs, deterministic=False):
    self._refresh_packs()
    args = self._args
    if obs.data_ptr() != self._obs_ptr:
      if self.graph or obs.dtype != torch.float32 or not obs.is_contiguous() or obs.numel() != self.obs.numel():
        self.obs.copy_(obs.reshape(self.obs.shape), non_blocking=True)  # a captured graph reads the fixed buffer
      else:
        args = (C.c_void_p(obs.data_ptr()),) + args[1:]  # eager launches read the caller's rows in place
    bulk = getattr(self, "_bulk", None)
    if not deterministic and bulk is not None and not self.graph:
      # this step's [E][A] slice of the draws made by draw_noise(): the kernel reads it in place
      args = args[:1] + (C.c_void_p(bulk.data_ptr() + self._bulk_t * bulk.stride(0) * 4),) + args[2:]
      self._bulk_t += 1
      if self._bulk_t >= bulk.shape[0]:
        self._bulk = None
    elif not deterministic:
      self.eps.normal_()  # torch's generator: the same standard-normal draws Normal(mean, std).sample() would use
      self._eps_zero = False
    elif not getattr(self, "_eps_zero", False):
      self.eps.zero_()  # action = mean + std * 0
      self._eps_zero = True
    if self.own:
      self.seek(0)
    check(self.L.v4l_actor_step(self.h, *args, _stream()), "v4l_actor_step")


def _gae_buffers(rewards, want32, out):
  T, E = rewards.shape
  dev = rewards.device
  key = (T, E, bool(want32), str(dev))
  if out is not None and out.get("key") == key:
    return out["bufs"]
  advs = torch.empty(T, E, dtype=torch.float64, device=dev)
  rets = torch.empty_like(advs)
  a32 = torch.empty(T, E, dtype=torch.float32, device=dev) if want32 else None
  r32 = torch.empty_like(a32) if want32 else None
  scratch = torch.empty(3 * T * E, dtype=torch.float64, device=dev)
  if out is not None:
    out["key"], out["bufs"] = key, (advs, rets, a32, r32, scratch)
  return advs, rets, a32, r32, scratch


def _tl_per_env(time_limits, T, E):
  return int(time_limits is not None and time_limits.dim() == 2 and time_limits.shape[1] == E and E > 1
             or (time_limits is not None and time_limits.numel() == T * E and E == 1))


def gae(rewards, values, terminals, time_limits, last_value, gamma, tau, use_time_limit, want32=True, out=None):
  """HIP GAE on float64 cuda tensors [T][E] (time_limits [T] or [T][E]); returns (advs, rets, advs32, rets32).
  out: a dict that keeps the output / scratch tensors between calls, so their addresses stay stable across epochs
  (the captured update graph is keyed on the rollout pointers). tau must be a number: PPO(gae=False) is discount_reward()."""
  if tau is None:
    raise TypeError("engine.gae: tau is None (use engine.discount_reward for PPO(gae=False))")
  L = _lib.lib()
  T, E = rewards.shape
  advs, rets, a32, r32, scratch = _gae_buffers(rewards, want32, out)
  check(L.v4l_gae(_ptr(rewards), _ptr(values), _ptr(terminals), _ptr(time_limits), _tl_per_env(time_limits, T, E), _ptr(last_value),
                  T, E, float(gamma), float(tau), int(bool(use_time_limit)), _ptr(scratch), _ptr(advs), _ptr(rets), _ptr(a32),
                  _ptr(r32), _stream()), "v4l_gae")
  return advs, rets, a32, r32


def discount_reward(rewards, values, terminals, time_limits, last_value, gamma, use_time_limit, want32=True, out=None):
  """HIP discount_reward (PPO(gae=False), replay_buffers/on_policy.py:47-71) on float64 cuda tensors [T][E]; same conventions
  as gae()."""
  L = _lib.lib()
  T, E = rewards.shape
  advs, rets, a32, r32, _ = _gae_buffers(rewards, want32, out)
  check(L.v4l_discount_reward(_ptr(rewards), _ptr(values), _ptr(terminals), _ptr(time_limits), _tl_per_env(time_limits, T, E),
                              _ptr(last_value), T, E, float(gamma), int(bool(use_time_limit)), _ptr(advs), _ptr(rets), _ptr(a32),
                              _ptr(r32), _stream()), "v4l_discount_reward")
  return advs, rets, a32, r32
