"""PPO with the reference's constructor, attributes and epoch protocol
(torchrl/algo/on_policy/ppo.py:10-161 on top of a2c.py:13-44, on_rl_algo.py:11-34, rl_algo.py:19-168).

What changes is *where* the work runs: `update` enqueues one fused critic-then-actor minibatch update on the
MI355X (forward, loss, backward, global-norm clip and Adam for both optimisers, all hand-written HIP behind
libv4l_hip.so) and `process_epoch_samples` runs the last-value forward and the fp64 GAE kernel. With a
`DeviceOnPolicyReplayBuffer` the observations never leave HBM and the 18 logger scalars of all 48 updates are
read back once per epoch instead of 18 `.item()` syncs per update.

Data parallel (one process per GPU, torch.distributed "nccl" == RCCL): every rank owns an env shard and its own
rollout; gradients are summed with ONE all-reduce per optimiser step on the flat gradient buffer (the critic's
carries the 3 advantage-normalisation scalars in its tail), then clipped and applied identically on every rank.
"""
import copy
import os
import os.path as osp
import pathlib
import pickle
import time
from collections import deque

import numpy as np
import torch

from .. import utils as atu
from ... import replay_buffers as rb
from .... import _lib
from ....engine import HipTrainer


class _HipAdam:
    """Optimiser facade: the Adam state lives in the trainer's flat device buffers; this object only carries
    the learning rate so `update_linear_schedule` and user code can read/write `param_groups[0]['lr']`."""

    def __init__(self, params, lr, eps=1e-5, betas=(0.9, 0.999)):
        self.param_groups = [{"params": list(params), "lr": lr, "eps": eps, "betas": betas, "weight_decay": 0}]

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self):
        pass


def _dist_world():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size()
    return 1


def exchange_comm_id(dist, device, available, unique_id, id_bytes=_lib.V4L_COMM_ID_BYTES):
    """Bootstrap of the library's own RCCL communicator over an existing process group (any backend): every rank first
    reports whether it can load RCCL (`available()` -> bool) and the ranks AGREE (a min all-reduce) before anyone enters
    ncclCommInitRank — a rank that cannot load the library would otherwise leave the others hanging in the rendezvous.
    Then rank 0's `unique_id()` (128 bytes) is broadcast. -> the id bytes, or None when some rank cannot take part (all ranks
    get None together and fall back to torch.distributed's all-reduce)."""
    on_dev = dist.get_backend() == "nccl"
    kw = {"device": device} if on_dev else {}
    ok = torch.tensor([1 if available() else 0], dtype=torch.int32, **kw)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return None
    idt = torch.zeros(id_bytes, dtype=torch.uint8, **kw)
    if dist.get_rank() == 0:
        raw = bytes(unique_id())
        if len(raw) != id_bytes:
            raise RuntimeError("vision4leg_amd: communicator id has %d bytes, expected %d" % (len(raw), id_bytes))
        idt.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    dist.broadcast(idt, src=0)
    return bytes(idt.cpu().numpy().tobytes())


def physical_device_index(device, environ=None):
    """The logical device index resolved through the visibility strings (HIP_ / CUDA_VISIBLE_DEVICES select among what
    ROCR_VISIBLE_DEVICES left): '0' and '0,1' both give 0 for logical device 0; an entry that is not a number (a uuid) is kept
    as it is. The tie-breaker of device_identity for runtimes that report one uuid / bus id for several partitions of a GPU."""
    env = os.environ if environ is None else environ
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
    for name in (("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"), ("ROCR_VISIBLE_DEVICES",)):
        val = next((env.get(n) for n in name if env.get(n)), None)
        if val is None or not isinstance(idx, int):
            continue
        entries = [e.strip() for e in val.split(",")]
        if idx < len(entries):
            idx = int(entries[idx]) if entries[idx].lstrip("-").isdigit() else entries[idx]
    return idx


def device_identity(device):
    """A 63-bit fingerprint of (host, physical GPU) — equal on two ranks exactly when they drive the same device. Where the
    runtime reports a uuid or PCI ids, (hostname, uuid, PCI ids, physical index) go in — the index RESOLVED through the visibility
    strings, never the strings themselves: two ranks that reach one physical GPU through different visibility strings ('0' and
    '0,1') must get the same fingerprint, or the 'ranks share a GPU' guard is bypassed and ncclCommInitRank is entered with a
    duplicate device; two partitions of one GPU (CPX / SR-IOV guests reporting the same uuid and bus id) must NOT (round-5
    advisor finding). The logical index and the *_VISIBLE_DEVICES strings are the fallback when the runtime reports neither."""
    import hashlib
    import socket
    props = torch.cuda.get_device_properties(device)
    uuid = str(getattr(props, "uuid", "") or "")
    pci = (getattr(props, "pci_domain_id", None), getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None))
    if uuid.strip("0-") or any(v is not None for v in pci):
        ident = [socket.gethostname(), uuid, pci, physical_device_index(device)]
    else:
        ident = [socket.gethostname(), (os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"),
                                        os.environ.get("CUDA_VISIBLE_DEVICES"), torch.device(device).index)]
    return int.from_bytes(hashlib.sha1(repr(ident).encode()).digest()[:8], "little") >> 1


class PPO:
    def __init__(self, pf, vf, plr=3e-4, vlr=3e-4, optimizer_class=None, entropy_coeff=0.001, clip_para=0.2,
                 opt_epochs=10, clipped_value_loss=False, shuffle=True, tau=None, gae=True, env=None,
                 replay_buffer=None, collector=None, logger=None, grad_clip=None, discount=0.99, num_epochs=3000,
                 batch_size=128, device="cpu", save_interval=100, eval_interval=1, save_dir=None, **kwargs):
        if optimizer_class is not None and optimizer_class is not torch.optim.Adam:
            raise NotImplementedError("vision4leg_amd: PPO runs Adam on the HIP engine; optimizer_class must be Adam")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("vision4leg_amd: PPO needs a GPU device (got %s); there is no CPU path" % self.device)
        # ---- reference attribute surface
        self.target_pf = copy.deepcopy(pf)  # before .to(device), as ppo.py:21
        self.env = env
        self.continuous = True
        self.replay_buffer = replay_buffer
        self.collector = collector
        self.discount = discount
        self.num_epochs = num_epochs
        self.epoch_frames = getattr(collector, "epoch_frames", None)
        self.batch_size = batch_size
        self.training_update_num = 0
        self.grad_clip = grad_clip
        self.logger = logger
        self.episode_rewards = deque(maxlen=30)
        self.training_episode_rewards = deque(maxlen=30)
        self.save_interval = save_interval
        self.save_dir = save_dir
        if save_dir is not None:
            pathlib.Path(save_dir).mkdir(parents=True, exist_ok=True)
        self.best_eval = None
        self.eval_interval = eval_interval
        self.explore_time = 0
        self.train_time = 0
        self.start = time.time()
        self.shuffle = shuffle
        self.tau = tau
        self.gae = gae
        self.pf = pf
        self.vf = vf
        self.to(self.device)
        self.plr, self.vlr = plr, vlr
        self.optimizer_class = torch.optim.Adam
        self.pf_optimizer = _HipAdam(self.pf.parameters(), lr=plr)
        self.vf_optimizer = _HipAdam(self.vf.parameters(), lr=vlr)
        self.entropy_coeff = entropy_coeff
        self.clip_para = clip_para
        self.opt_epochs = opt_epochs
        self.clipped_value_loss = clipped_value_loss
        self.sample_key = ["obs", "acts", "advs", "estimate_returns", "values"]
        self.current_epoch = 0
        # ---- engine
        self.world_size = _dist_world()
        # the data-parallel update (four phases + one all-reduce per optimiser step) also runs on a 1-rank process group
        # when asked to: that is how the RCCL path is exercised on a single-GPU box (tests/test_gpu_parity.py)
        self.dp_phases = self.world_size > 1 or (os.environ.get("V4L_FORCE_DP_PHASES", "0") != "0"
                                                   and torch.distributed.is_available()
                                                   and torch.distributed.is_initialized())
        if self.world_size > 1:
            for p in list(self.pf.parameters()) + list(self.vf.parameters()):
                torch.distributed.broadcast(p.data, src=0)
            atu.copy_model_params_from_to(self.pf, self.target_pf)
            for net in self.networks:  # `.data` writes do not bump the Parameters' version counters
                net.mark_params_changed()
        self.use_graph = os.environ.get("V4L_GRAPH", "1") != "0"
        with torch.cuda.device(self.device):
            self.trainer = HipTrainer(self.pf.hip, self.vf.hip, self.target_pf.hip, batch_size, clip_para,
                                      entropy_coeff, max_grad_norm=0.5, clipped_value_loss=clipped_value_loss,
                                      world_size=self.world_size)
            # The exchange. Preferred: the library's own RCCL communicator, whose two all-reduces are issued by update_next()
            # INSIDE the captured update graph (no host round trip between backward, all-reduce and Adam). It becomes the
            # schedule only after every rank has passed a self-test of that communicator through the very calls an update
            # makes (_library_comm); otherwise all ranks together keep torch.distributed's all-reduce between four eager
            # phases per update (_update_phases). V4L_DP_COMM = auto (default) | rccl (same, but a failing self-test raises)
            # | torch (never try the library's communicator).
            self.dp_in_library, self.dp_comm_note = False, "single process"
            if self.dp_phases:
                want = os.environ.get("V4L_DP_COMM", "auto").lower()
                if want not in ("auto", "rccl", "torch"):
                    raise ValueError("V4L_DP_COMM must be auto, rccl or torch (got %r)" % want)
                if want == "torch":
                    self.dp_comm_note = "torch.distributed all-reduce between phases (V4L_DP_COMM=torch)"
                else:
                    ok, why = self._library_comm()
                    if ok:
                        self.dp_in_library = True
                        self.dp_phases = False  # update_next() carries the collectives itself
                        self.dp_comm_note = "library RCCL communicator inside the update graph (%s)" % why
                    elif want == "rccl":
                        raise RuntimeError("vision4leg_amd: V4L_DP_COMM=rccl but the library's communicator is not usable: " + why)
                    else:
                        self.dp_comm_note = "torch.distributed all-reduce between phases (library RCCL declined: %s)" % why
        if isinstance(replay_buffer, rb.DeviceOnPolicyReplayBuffer):
            replay_buffer.attach(self.pf.hip, self.device)
        elif replay_buffer is not None:
            replay_buffer.gae_device = self.device
        self._stage = None

    # ---- data-parallel bootstrap ---------------------------------------------------------------------
    def _agree(self, ok):
        """True only if `ok` holds on EVERY rank (a MIN all-reduce over the host's process group)."""
        dist = torch.distributed
        kw = {"device": self.device} if dist.get_backend() == "nccl" else {}
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, **kw)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def _library_comm(self):
        """Bring up the library's RCCL communicator over the host's process group and prove it before relying on it.
        -> (True, what was checked) with the communicator attached on every rank, or (False, why) with none attached on
        any rank. Every decision is taken by all ranks together, so no rank ever waits in a rendezvous alone:
          1. one GPU per rank (RCCL refuses two ranks on a device — the 2-process / 1-GPU test setup);
          2. every rank can load librccl (`exchange_comm_id`), then rank 0's unique id is broadcast and the ranks meet in
             ncclCommInitRank;
          3. v4l_trainer_comm_selftest: both gradient buckets [gradients | tail] carry a rank-dependent integer pattern
             through v4l_sync_grads — eagerly and as a captured graph replayed twice, as update_next() will run it — and
             are checked element by element on the device;
          4. the same pattern through torch.distributed's all-reduce must give the same buffer, and the communicator's
             own rank / size (ncclCommUserRank / ncclCommCount) must equal the process group's."""
        dist = torch.distributed
        tr = self.trainer
        rank, world = dist.get_rank(), dist.get_world_size()
        kw = {"device": self.device} if dist.get_backend() == "nccl" else {}
        mine = torch.tensor([device_identity(self.device)], dtype=torch.int64, **kw)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        if len({int(t.item()) for t in every}) < world:
            return False, "ranks share a GPU; RCCL needs one device per rank"
        comm_id = exchange_comm_id(dist, self.device, HipTrainer.comm_available, HipTrainer.comm_unique_id)
        if comm_id is None:
            return False, "a rank cannot load librccl"
        err = ""
        try:
            tr.comm_init(comm_id, rank, world)
        except RuntimeError as e:
            err = str(e)
        if not self._agree(not err):
            if tr.has_comm:
                tr.comm_destroy()
            return False, "ncclCommInitRank failed on a rank" + (": " + err if err else "")
        try:
            with torch.cuda.stream(tr.stream):
                bad = tr.comm_selftest(graph=self.use_graph)
                if bad:
                    err = "%d elements of the all-reduced self-test pattern are wrong on rank %d" % (bad, rank)
                crank, cworld = tr.comm_info()
                if (crank, cworld) != (rank, world):
                    err = "communicator reports rank %d of %d, the process group %d of %d" % (crank, cworld, rank, world)
                # the critic bucket still holds the library's all-reduced pattern: torch.distributed must produce the same
                n = tr.vf.total_params
                i = torch.arange(n, dtype=torch.int64, device=self.device)
                ref = ((i * 7 + rank * 13) % 251).to(torch.float32)
                got = tr.g_vf.clone()
            tr.stream.synchronize()
            dist.all_reduce(ref)
            if not err and not torch.equal(ref, got):
                err = "library all-reduce and torch.distributed all-reduce disagree on rank %d" % rank
        except RuntimeError as e:
            err = str(e)
        tr.g_vf_bucket.zero_()
        tr.g_pf_bucket.zero_()
        if not self._agree(not err):
            tr.comm_destroy()
            return False, "self-test failed on a rank" + (": " + err if err else "")
        return True, "self-test passed on %d ranks: eager + %s all-reduce of both buckets, cross-checked against " \
                     "torch.distributed" % (world, "graph-replayed" if self.use_graph else "eager-only")

    # ---- reference plumbing ------------------------------------------------------------------------
    @property
    def networks(self):
        return [self.pf, self.vf, self.target_pf]

    @property
    def snapshot_networks(self):
        return [("pf", self.pf), ("vf", self.vf)]

    @property
    def target_networks(self):
        return []

    def to(self, device):
        for net in self.networks:
            net.to(device)

    def start_epoch(self):
        pass

    def finish_epoch(self):
        return {}

    def pretrain(self):
        pass

    def snapshot(self, prefix, epoch):
        """state_dict checkpoints with the reference's file names (rl_algo.py:84-95)."""
        norm = getattr(self.env, "_obs_normalizer", None)
        if norm is not None:
            if hasattr(norm, "to_reference"):
                # the device-resident normaliser (vision4leg_amd.torchrl.env): write what the reference's viewers unpickle
                # (starter/locotransformer_viewer.py:125-147) — an instance of the REFERENCE's Normalizer class — whenever that
                # class is importable (it is under overlay.install(): the reference's torchrl.env is untouched)
                import importlib
                try:
                    ref_cls = importlib.import_module("torchrl.env.base_wrapper").Normalizer
                except ImportError:  # stand-alone use of this package (no reference tree): the pickle then needs this package
                    ref_cls = None
                if ref_cls is not None and ref_cls is not type(norm):
                    norm = norm.to_reference(ref_cls)  # a failing conversion must not silently write an incompatible pickle
            with open(osp.join(prefix, "_obs_normalizer_{}.pkl".format(epoch)), "wb") as f:
                pickle.dump(norm, f)
        for name, network in self.snapshot_networks:
            torch.save(network.state_dict(), osp.join(prefix, "model_{}_{}.pth".format(name, epoch)))

    # ---- epoch ---------------------------------------------------------------------------------------
    def process_epoch_samples(self):
        """last_value = vf(next_obs[T-1]) * (1 - terminal) then GAE — or, gae=False, discounted rewards (on_rl_algo.py:23-34)."""
        sample = self.replay_buffer.last_sample(["next_obs", "terminals", "time_limits"])
        last_ob = sample["next_obs"]
        if not isinstance(last_ob, torch.Tensor):
            last_ob = torch.from_numpy(np.ascontiguousarray(last_ob, dtype=np.float32))
        last_value = self.vf(last_ob.to(self.device, torch.float32)).cpu().numpy()
        last_value = last_value * (1 - sample["terminals"])
        if self.gae:
            self.replay_buffer.generalized_advantage_estimation(last_value, self.discount, self.tau)
        else:  # on_rl_algo.py:29-33
            self.replay_buffer.discount_reward(last_value, self.discount)

    def update_per_epoch(self):
        with torch.cuda.device(self.device):
            self.process_epoch_samples()
            atu.update_linear_schedule(self.pf_optimizer, self.current_epoch, self.num_epochs, self.plr)
            atu.update_linear_schedule(self.vf_optimizer, self.current_epoch, self.num_epochs, self.vlr)
            self.trainer.sync_target()  # copy_model_params_from_to(pf, target_pf), ppo.py:34
            if isinstance(self.replay_buffer, rb.DeviceOnPolicyReplayBuffer):
                self._update_epoch_resident()
                return
            for _ in range(self.opt_epochs):
                for batch in self.replay_buffer.one_iteration(self.batch_size, self.sample_key, self.shuffle):
                    infos = self.update(batch)
                    self.logger.add_update_info(infos)

    def _update_epoch_resident(self):
        buf = self.replay_buffer
        state, image, acts, advs, rets, vals, logp = buf.device_rollout()
        ro = HipTrainer.rollout(state, image, acts, advs, rets, vals, logp)
        batches = []
        for _ in range(self.opt_epochs):
            for b in buf.one_iteration(self.batch_size, self.sample_key, self.shuffle):
                batches.append(b["rowidx"])
        rowidx = torch.from_numpy(np.stack(batches)).to(self.device)  # one upload per epoch
        stats = torch.zeros(len(batches), _lib.V4L_STATS, dtype=torch.float32, device=self.device)
        self.run_updates(ro, rowidx, stats)
        host = stats.cpu().numpy()  # the only device->host sync of the epoch's updates
        bad = np.nonzero(host[:, _lib.ST_NONFINITE] > 0)[0]
        if len(bad):  # the device-side tripwire (collector/on_policy.py:102-107 "NaN detected. BOOM")
            raise FloatingPointError("vision4leg_amd: non-finite training statistics in minibatch update %d of epoch %d: %s"
                                     % (int(bad[0]), self.current_epoch,
                                        {k: float(host[bad[0], j]) for j, k in enumerate(_lib.STAT_KEYS)}))
        self._note_f16_saturation(float(host[:, _lib.ST_F16_SAT].sum()))
        for row in host:
            self.logger.add_update_info({k: float(row[j]) for j, k in enumerate(_lib.STAT_KEYS)})

    f16_saturated = 0.0  # V4L_COMPUTE=f16: loss-gradient elements clamped at +-V4L_F16_GRAD_CLAMP so far (include/v4l_hip.h, record slot 23)

    def _note_f16_saturation(self, count):
        """f16 compute mode: the loss-gradient rows enter the backward scaled by a power of two and clamped inside half's range.
        A clamped element is a per-sample gradient clip the reference does not have: counted, and reported once."""
        if count > 0:
            if self.f16_saturated == 0:
                import warnings
                warnings.warn("vision4leg_amd: %d loss-gradient element(s) of an f16 update exceeded +-%g after scaling and were "
                              "clamped (record slot %d; PPO.f16_saturated keeps the total). Returns / advantages of that size "
                              "want V4L_COMPUTE=bf16." % (int(count), 32768.0, _lib.ST_F16_SAT), RuntimeWarning)
            self.f16_saturated += count

    def run_updates(self, ro, rowidx, stats):
        """All minibatch updates of an epoch: rowidx [U][n] int32 (device), stats [U][V4L_STATS] (device).
        The whole run is ONE hipGraph replay on the trainer's stream (round 6; one replay per update with V4L_EPOCH_GRAPH=0) — on one
        GPU, and data-parallel with the library's own communicator (V4L_DP_COMM=rccl: both all-reduces are nodes of that graph). With torch.distributed doing the exchange
        (the data-parallel default) an update is four eager phases with one all-reduce per optimiser step between them."""
        tr = self.trainer
        U, n = rowidx.shape
        # the captured graph is keyed on these addresses: keep them stable from epoch to epoch
        if getattr(self, "_rowidx_buf", None) is None or self._rowidx_buf.shape != rowidx.shape:
            self._rowidx_buf = torch.empty_like(rowidx)
            self._stats_buf = torch.zeros(U, _lib.V4L_STATS, dtype=torch.float32, device=self.device)
        self._rowidx_buf.copy_(rowidx)
        cur = torch.cuda.current_stream(self.device)
        tr.stream.wait_stream(cur)
        with torch.cuda.stream(tr.stream):
            tr.begin(self._rowidx_buf, self._stats_buf, self.pf_optimizer.lr, self.vf_optimizer.lr)
            if not self.dp_phases and self.use_graph and os.environ.get("V4L_EPOCH_GRAPH", "1") != "0":
                # the whole epoch loop (ppo.py:28-40) as ONE hipGraph replay (round 6); V4L_EPOCH_GRAPH=0: one replay per update
                self.training_update_num += U
                tr.update_run(ro, n, U, graph=True)
            else:
                for _ in range(U):
                    self.training_update_num += 1
                    if not self.dp_phases:
                        tr.update_next(ro, n, graph=self.use_graph)
                    else:
                        self._update_phases(ro, n)
            stats.copy_(self._stats_buf)
        cur.wait_stream(tr.stream)

    def _update_phases(self, ro, n):
        """The exchange done by torch.distributed (V4L_DP_COMM=torch): four eager phases, the buckets [gradients | tail]
        all-reduced between them."""
        dist = torch.distributed
        tr = self.trainer
        tr.critic_grads(ro, n)
        tr.bucket_tail(1, 1, self.world_size)
        dist.all_reduce(tr.g_vf_bucket)  # sum over env shards: critic grads + advantage moments + vf_loss share
        tr.bucket_tail(1, 0, self.world_size)
        tr.critic_step()
        tr.actor_grads(ro, n)
        tr.bucket_tail(0, 1, self.world_size)
        dist.all_reduce(tr.g_pf_bucket)
        tr.bucket_tail(0, 0, self.world_size)
        tr.actor_step()

    def update(self, batch):
        """One minibatch update from a reference-style host batch (ppo.py:125-153). Returns the 18-key info."""
        if "rowidx" in batch:
            raise RuntimeError("device-resident batches are consumed by update_per_epoch")
        self.training_update_num += 1
        dev = self.device
        with torch.cuda.device(dev):
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            obs = up(batch["obs"])
            n = obs.shape[0]
            net = self.pf.hip
            net.ensure_bound()
            if self._stage is None or self._stage[0].shape[0] != n:
                self._stage = net.alloc_rollout(n, dev)
            net.ingest(obs, self._stage[0], self._stage[1])
            acts = up(batch["acts"]).reshape(n, -1)
            advs = up(batch["advs"]).reshape(n)
            rets = up(batch["estimate_returns"]).reshape(n)
            vals = up(batch["values"]).reshape(n)
            ro = HipTrainer.rollout(self._stage[0], self._stage[1], acts, advs, rets, vals)
            stats = torch.zeros(1, _lib.V4L_STATS, dtype=torch.float32, device=dev)
            tr = self.trainer
            if not self.dp_phases:
                tr.update(ro, None, n, self.pf_optimizer.lr, self.vf_optimizer.lr, stats)
            else:
                tr._pre(n)
                tr.begin(None, stats, self.pf_optimizer.lr, self.vf_optimizer.lr)
                self._update_phases(ro, n)
            host = stats[0].cpu().numpy()
        if host[_lib.ST_NONFINITE] > 0:  # the device-side tripwire, as in _update_epoch_resident
            raise FloatingPointError("vision4leg_amd: non-finite training statistics in minibatch update %d: %s"
                                     % (self.training_update_num, {k: float(host[j]) for j, k in enumerate(_lib.STAT_KEYS)}))
        self._note_f16_saturation(float(host[_lib.ST_F16_SAT]))
        return {k: float(host[j]) for j, k in enumerate(_lib.STAT_KEYS)}

    # ---- outer loop (rl_algo.py:97-168) ----------------------------------------------------------------
    def train(self):
        self.pretrain()
        total_frames = getattr(self, "pretrain_frames", 0)
        self.start_epoch()
        for epoch in range(self.num_epochs):
            self.current_epoch = epoch
            self.start_epoch()
            t0 = time.time()
            training_epoch_info = self.collector.train_one_epoch()
            for reward in training_epoch_info["train_rewards"]:
                self.training_episode_rewards.append(reward)
            self.explore_time += time.time() - t0
            t0 = time.time()
            self.update_per_epoch()
            torch.cuda.synchronize(self.device)
            self.train_time += time.time() - t0
            finish_epoch_info = self.finish_epoch()
            total_frames += self.epoch_frames
            if epoch % self.eval_interval == 0:
                t0 = time.time()
                eval_infos = self.collector.eval_one_epoch()
                eval_time = time.time() - t0
                infos = {}
                for reward in eval_infos["eval_rewards"]:
                    self.episode_rewards.append(reward)
                mean_eval = np.mean(eval_infos["eval_rewards"])
                if self.best_eval is None or mean_eval > self.best_eval:
                    self.best_eval = mean_eval
                    self.snapshot(self.save_dir, "best")
                    print("Best Saved: {:.5f},  EPoch: {}".format(mean_eval, epoch))
                del eval_infos["eval_rewards"]
                infos["Running_Average_Rewards"] = np.mean(self.episode_rewards)
                infos["Train_Epoch_Reward"] = training_epoch_info["train_epoch_reward"]
                infos["Running_Training_Average_Rewards"] = np.mean(self.training_episode_rewards)
                infos["Explore_Time"] = self.explore_time
                infos["Train___Time"] = self.train_time
                infos["Eval____Time"] = eval_time
                self.explore_time = 0
                self.train_time = 0
                infos.update(eval_infos)
                infos.update(finish_epoch_info)
                self.logger.add_epoch_info(epoch, total_frames, time.time() - self.start, infos)
                self.start = time.time()
            if epoch % self.save_interval == 0:
                self.snapshot(self.save_dir, epoch)
        self.snapshot(self.save_dir, "finish")
        self.collector.terminate()
