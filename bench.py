#!/usr/bin/env python
"""bench.py — PPO env-steps/s of the hot path (encoder + update, simulator excluded) on MI355X.

One "step" = one PPO epoch of everything the reference does outside env.step:
  (i)   T rollout steps at batch E: ingest of the E observation rows + pf.explore + vf forward (+ storing them)
  (ii)  last-value forward + GAE over [T,E]
  (iii) opt_epochs x (T*E/B) minibatch updates (critic fwd/bwd/clip/Adam, actor fwd + frozen-target fwd/bwd/clip/Adam)
  (iv)  LR schedule + target-policy sync
on a synthetic epoch (distributions of BASELINE.md §3) that is resident in HBM when the timed region starts.
value = E*T*n_gpus / t_step. Multi-GPU: one process per GPU (torchrun), each rank owns an env shard (weak scaling),
gradients cross ranks with one RCCL all-reduce per optimiser step.

Prints ONE JSON line on rank 0. Extra objects: "roofline" (dominant kernel, from HIP-event timings collected by the
library's built-in profiler on the launch stream) and "cpu_baseline" (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]: ppo_locotransformer.py thin-goal, 1 MI355X, 32 envs (headline: the metric's MFMA
    # clause and BASELINE.md's 10x target are quoted on the LocoTransformer)
    "loco": dict(kind="loco", S=93, A=6, E=32, T=512, B=1024, enc=[256, 256], head=[256, 256], layers=2, ff=256,
                 name="ppo_locotransformer thin-goal: LocoTransformer S=93 A=6, E=32 envs x T=512, B=1024, 3 opt epochs"),
    # configs[1]: ppo_nature_cnn.py, 16 envs
    "cnn": dict(kind="cnn", S=93, A=6, E=16, T=1024, B=1024, enc=[256, 256], head=[256, 256], visual_dim=256,
                name="ppo_nature_cnn thin-goal: NatureCNN fuse net S=93 A=6, E=16 envs x T=1024, B=1024, 3 opt epochs"),
    # configs[0]: ppo_state.py (the reference's CPU-runnable plumbing case)
    "mlp": dict(kind="mlp", S=93, A=6, E=1, T=16384, B=1024, enc=[256, 256], head=[256, 256],
                name="ppo_state: state-only MLP S=93 A=6, E=1 env x T=16384, B=1024, 3 opt epochs"),
    # configs[4] per-GPU shard: 64 envs/GPU
    "loco64": dict(kind="loco", S=93, A=6, E=64, T=256, B=1024, enc=[256, 256], head=[256, 256], layers=2, ff=256,
                   name="ppo_locotransformer challenge/mountain shard: E=64 envs x T=256 per GPU, B=1024"),
    # the vision-only starters (config/mpc_vision_only/*/thin-goal.json: 8192 frames per epoch)
    "loco_vis": dict(kind="loco_vis", S=0, A=6, E=32, T=256, B=1024, enc=[], head=[256, 256], layers=2, ff=256,
                     name="ppo_locotransformer_vision_only: Transformer over 16 depth tokens, A=6, E=32 envs x T=256, B=1024"),
    "cnn_vis": dict(kind="cnn_vis", S=0, A=6, E=32, T=256, B=1024, enc=[], head=[256, 256],
                    name="ppo_nature_cnn_vision_only: NatureCNN -> 1024 -> head, A=6, E=32 envs x T=256, B=1024"),
    # an OPTION variant of the headline net (nets.py:1022-1030 max_pool=True; no shipped config sets it): reference goldens;
    # on the fused kernels since round 5 — this line shows an option costs nothing against the headline
    "loco_max": dict(kind="loco_max", S=93, A=6, E=32, T=512, B=1024, enc=[256, 256], head=[256, 256], layers=2, ff=256,
                     name="ppo_locotransformer with max_pool=True (option variant, fused kernels), E=32 x T=512, B=1024"),
    # token_norm=True / use_pytorch_encoder=True (nets.py:815-818, 955-963): the update's layers on the wave-per-sample kernels,
    # token_ln / the final norm + pooling + heads as launches of their own (round 5); the rollout step layer by layer
    "loco_tn": dict(kind="loco_tn", S=93, A=6, E=32, T=512, B=1024, enc=[256, 256], head=[256, 256], layers=2, ff=256,
                    name="ppo_locotransformer with token_norm=True (option variant), E=32 x T=512, B=1024"),
    "loco_pe": dict(kind="loco_pe", S=93, A=6, E=32, T=512, B=1024, enc=[256, 256], head=[256, 256], layers=2, ff=256,
                    name="ppo_locotransformer with use_pytorch_encoder=True (option variant), E=32 x T=512, B=1024"),
}
# algorithmic MFLOP per env-step incl. rollout inference, and F_pf: the forward pass of the frozen
# target policy that each of the 3 sample-visits skips when log pi_old is recorded at action time (§8d's declared saving)
# vision-only nets, same accounting: F_pf = 2 * (3 612 672 conv + 65 536 up-conv + 2 * 819 200 layer (16 tokens) + 83 456 head)
# = 10.800 MFLOP, F_vf = 10.798; NatureCNN: F_pf = 2 * (3 612 672 + 329 216) = 7.884, F_vf = 7.881
MFLOP_PER_ENV_STEP = {"loco": 258.9, "loco64": 258.9, "cnn": 191.4, "mlp": 10.2, "loco_vis": 248.4, "cnn_vis": 181.3, "loco_max": 258.9,
                      "loco_tn": 258.9, "loco_pe": 258.9}
MFLOP_TARGET_FWD = {"loco": 11.258, "loco64": 11.258, "cnn": 8.325, "mlp": 0.444, "loco_vis": 10.800, "cnn_vis": 7.884, "loco_max": 11.258,
                    "loco_tn": 11.258, "loco_pe": 11.258}
OPT_EPOCHS = 3
LAST_ALLREDUCE = None
PEAK = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # dense TFLOP/s, MI355X_MICROARCH.md (bf16 = f16 MFMA / f32 MFMA)


def launcher_argv(n, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself with when it was not started by a launcher: one process
    per GPU of this node over RCCL, rendezvous on 127.0.0.1 (the driver's own multi-GPU command line, spelled out)."""
    if port is None:
        port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: by --gpus — 1, 2, 4 GPUs: loco (BASELINE configs[2] / [3]: 32 envs per GPU); 8 GPUs: loco64 "
                         "(configs[4]: 64 envs per GPU)")
    ap.add_argument("--compute", default=os.environ.get("V4L_COMPUTE", "f16"), choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity check, h2d and reference-protocol legs")
    ap.add_argument("--no-reference-protocol", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="time (ii)-(iv) only (the reference's Train___Time)")
    ap.add_argument("--breakdown", default=None, help="write the per-op HIP-event breakdown to this file")
    a = ap.parse_args(argv)
    if a.workload is None:
        a.workload = "loco64" if a.gpus >= 8 else "loco"
    return a


class Epoch:
    """Device-resident synthetic epoch + the nets/trainer that consume it."""

    def __init__(self, wl, compute, dev, world):
        os.environ["V4L_COMPUTE"] = compute
        from vision4leg_amd import recipes as util
        import vision4leg_amd.torchrl.networks as networks
        import vision4leg_amd.torchrl.policies as policies
        from vision4leg_amd.torchrl.algo import PPO
        self.wl, self.dev = wl, dev
        case = dict(wl, seed=0)
        torch.manual_seed(0)
        pf, vf = util.build_nets(networks, policies, case)

        class Coll: epoch_frames = wl["E"] * wl["T"]
        self.agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=OPT_EPOCHS, tau=0.95, shuffle=True,
                         entropy_coeff=0.005, collector=Coll(), device=dev, discount=0.99, num_epochs=1500,
                         batch_size=wl["B"])
        self.pf, self.vf = self.agent.pf, self.agent.vf
        E, T, S, A = wl["E"], wl["T"], wl["S"], wl["A"]
        D = util.obs_dim(case)
        g = torch.Generator(device=dev).manual_seed(1234 + (torch.distributed.get_rank() if world > 1 else 0))
        rn = lambda *s: torch.randn(*s, device=dev, generator=g)
        # observation rows in the reference's layout [T*E, S + 4*64*64] fp32 (what the collector would upload)
        self.obs = torch.empty(T * E, D, device=dev)
        self.obs[:, :S] = rn(T * E, S).clamp_(-10, 10)
        if D > S:
            self.obs[:, S:] = rn(T * E, D - S).clamp_(-2.5, 2.8)
        self.rewards = rn(T, E).double()
        self.terminals = (torch.rand(T, E, device=dev, generator=g) < 0.01).double()
        self.time_limits = (torch.rand(T, E, device=dev, generator=g) < 0.002).double()
        net = self.pf.hip
        net.ensure_bound()
        self.state, self.image = net.alloc_rollout(T * E, dev)
        self.acts = torch.zeros(T * E, A, device=dev)
        self.values = torch.zeros(T * E, device=dev)
        # log pi_old(a|s): recorded by the rollout step (the acting policy of an epoch is that epoch's target policy,
        # ppo.py:34) so the update skips the target forward. V4L_STORED_LOGP=0: evaluate target_pf per minibatch.
        self.logp = None
        self.stats = torch.zeros(OPT_EPOCHS * (T * E // wl["B"]), 24, device=dev)
        self.epoch = 0
        self.actor = None
        if os.environ.get("V4L_ACTOR", "1") != "0":
            self.actor = policies.RolloutActor(self.pf, self.vf, E, graph=os.environ.get("V4L_ACTOR_GRAPH", "0") != "0")
            if os.environ.get("V4L_STORED_LOGP", "1") != "0":
                self.logp = torch.zeros(T * E, device=dev)
            self.actor.attach((self.state, self.image, self.acts, self.values, self.logp))
        if wl.get("skip_rollout"):
            self.rollout()  # populate once so updates have data

    def rollout(self):
        """(i): per env step, E rows: ingest + pf.explore + vf; results stay on the device."""
        E, T = self.wl["E"], self.wl["T"]
        pf, vf, net = self.pf, self.vf, self.pf.hip
        if self.actor is not None:  # fused rollout step (shared encoder pass, one graph replay per env step)
            self.actor.seek(0)
            if os.environ.get("V4L_BULK_NOISE", "1") != "0":
                self.actor.draw_noise(T)  # the epoch's T x E x A exploration normals in one generator call
            # The loop is the collector's (VecOnPolicyCollector.train_one_epoch): nothing steps an optimiser between two env steps,
            # so "did the parameters change?" is asked once for the loop (round 6: the per-step question cost ~12 us of interpreter
            # time, which made this loop host-bound on slower hosts: 13.0 - 15.2 ms for 512 x 24.7 us of kernels).
            if getattr(self, "_obs_steps", None) is None:
                self._obs_steps = [self.obs[t * E:(t + 1) * E] for t in range(T)]
            self.actor.freeze_params(True)
            try:
                for ob in self._obs_steps:
                    self.actor.step(ob)
            finally:
                self.actor.freeze_params(False)
            return
        for t in range(T):  # the reference's call protocol: pf.explore(ob) then vf(ob)
            ob = self.obs[t * E:(t + 1) * E]
            out = pf.explore(ob)
            self.acts[t * E:(t + 1) * E] = out["action"]
            self.values[t * E:(t + 1) * E] = vf(ob).view(E)
            net.ingest(ob, self.state, self.image, slot0=t * E)

    def update(self):
        """(ii)-(iv)"""
        from vision4leg_amd import engine
        from vision4leg_amd.engine import HipTrainer
        from vision4leg_amd.torchrl.algo import utils as atu
        wl, ag = self.wl, self.agent
        E, T, B = wl["E"], wl["T"], wl["B"]
        last_value = self.vf(self.obs[-E:]).view(E).double() * (1 - self.terminals[-1])
        if not hasattr(self, "_gae_out"):
            self._gae_out = {}
        _, _, a32, r32 = engine.gae(self.rewards, self.values.view(T, E).double(), self.terminals, self.time_limits,
                                    last_value, 0.99, 0.95, True, out=self._gae_out)
        ag.current_epoch = self.epoch
        atu.update_linear_schedule(ag.pf_optimizer, ag.current_epoch, ag.num_epochs, ag.plr)
        atu.update_linear_schedule(ag.vf_optimizer, ag.current_epoch, ag.num_epochs, ag.vlr)
        ag.trainer.sync_target()
        rows = B // E
        idx = []
        for _ in range(OPT_EPOCHS):
            perm = np.random.permutation(T)
            for pos in range(0, T, rows):
                sel = perm[pos:pos + rows]
                idx.append((sel[:, None] * E + np.arange(E)[None, :]).reshape(-1))
        rowidx = torch.from_numpy(np.stack(idx).astype(np.int32)).to(self.dev)
        ro = HipTrainer.rollout(self.state, self.image, self.acts, a32.reshape(-1), r32.reshape(-1), self.values, self.logp)
        ag.run_updates(ro, rowidx, self.stats)
        self.epoch += 1

    def step(self, with_rollout=True):
        if with_rollout:
            self.rollout()
        self.update()


def side_leg(wl_name, compute, dev, steps=3, warmup=1):
    """A short timed run of another (workload, compute mode) in the SAME process, after the headline's timed region: the same
    Epoch, the same step, `steps` epochs. Used for the exact-fp32 mode (the mode that meets the north star's literal 1e-3 against
    the fp32 reference) and for the net's option variants."""
    before = os.environ.get("V4L_COMPUTE")
    wl = dict(WORKLOADS[wl_name])
    try:
        ep = Epoch(wl, compute, dev, 1)
        for _ in range(warmup):
            ep.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_roll = 0.0
        for _ in range(steps):
            r0 = time.perf_counter()
            ep.rollout()
            torch.cuda.synchronize()
            t_roll += time.perf_counter() - r0
            ep.update()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        frames = wl["E"] * wl["T"]
        ok = bool(torch.isfinite(ep.stats[:, :18]).all().item())
        out = {"value": round(frames * steps / dt, 1), "unit": "env-steps/s", "dtype": compute, "steps": steps, "warmup": warmup,
               "ms_per_step": round(1e3 * dt / steps, 3), "rollout_inference_ms_per_step": round(1e3 * t_roll / steps, 3),
               "update_only_env_steps_per_s": round(frames * steps / max(dt - t_roll, 1e-9), 1), "stats_finite": ok,
               "workload": wl["name"]}
        del ep
        torch.cuda.empty_cache()
        return out
    finally:
        if before is None:
            os.environ.pop("V4L_COMPUTE", None)
        else:
            os.environ["V4L_COMPUTE"] = before


def dp1_ingraph_leg(workload, compute):
    """The data-parallel schedule on ONE rank (a 1-process torchrun of this file with V4L_FORCE_DP_PHASES=1): RCCL communicator
    owned by the library, self-test, both all-reduces of every update inside the captured graph. Makes regressions of the DP
    schedule visible in the single-GPU line, and times one all-reduce call (world 1: its launch / completion floor)."""
    env = dict(os.environ, V4L_FORCE_DP_PHASES="1", V4L_BENCH_CHILD="1", V4L_COMPUTE=compute)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None); env.pop("MASTER_PORT", None)
    cmd = launcher_argv(1, ["--gpus", "1", "--workload", workload, "--compute", compute, "--steps", "3", "--warmup", "1",
                            "--no-parity", "--no-cpu-baseline"])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        d = json.loads(line[-1])
        return {"value": d["value"], "ms_per_step": d["ms_per_step"], "update_only_env_steps_per_s": d["update_only_env_steps_per_s"],
                "dp_comm": d["config"].get("dp_comm"), "rccl_ranks": d.get("rccl_ranks"), "allreduce": d.get("allreduce")}
    except Exception as e:  # never let a side leg take the headline line down
        return {"error": repr(e)[:300]}


# what one in-graph all-reduce of a gradient bucket (1.55 MB fp32) should cost on N GPUs of one node, and what
# that does to `value`. Assumptions, stated so that the first real multi-GPU run has something to be compared with: xGMI full mesh,
# 7 links x 153 GB/s per GPU; RCCL on a 1.5 MB message is latency-bound — a launch / completion floor (alpha0, measured here on one
# rank by dp1_ingraph: `allreduce.us_per_call`) plus a per-hop cost of the protocol (LL / LL128 hops of a ring: 2 (N - 1) hops; a
# tree: 2 log2 N) of 1.0 - 2.5 us each; the bandwidth term is S / (N x B_link) x 2 (direct reduce-scatter + all-gather over the
# mesh) to 2 (N - 1) / N x S / B_link (one ring on one link).
def dp_model(n_gpus, bucket_bytes, alpha0_us, update_us_1gpu, epoch_ms_1gpu, updates_per_epoch, frames_per_gpu):
    import math
    if n_gpus <= 1:
        lo = hi = alpha0_us
    else:
        b_link = 153e9
        bw_lo = 2.0 * bucket_bytes / (n_gpus * b_link) * 1e6
        bw_hi = 2.0 * (n_gpus - 1) / n_gpus * bucket_bytes / b_link * 1e6
        lo = alpha0_us + 2 * math.log2(n_gpus) * 1.0 + bw_lo
        hi = alpha0_us + 2 * (n_gpus - 1) * 2.5 + bw_hi
    per_epoch_lo, per_epoch_hi = 2 * updates_per_epoch * lo * 1e-3, 2 * updates_per_epoch * hi * 1e-3   # ms, unoverlapped
    return {"allreduce_us_predicted": [round(lo, 1), round(hi, 1)],
            "allreduces_per_epoch": 2 * updates_per_epoch, "bucket_bytes": bucket_bytes,
            "value_predicted": [round(n_gpus * frames_per_gpu / (epoch_ms_1gpu + per_epoch_hi) * 1e3, 0),
                                round(n_gpus * frames_per_gpu / (epoch_ms_1gpu + per_epoch_lo) * 1e3, 0)],
            "efficiency_predicted": [round(epoch_ms_1gpu / (epoch_ms_1gpu + per_epoch_hi), 3),
                                     round(epoch_ms_1gpu / (epoch_ms_1gpu + per_epoch_lo), 3)],
            "assumes": "alpha0 = %.1f us launch/completion floor of one RCCL call (measured on 1 rank or the default 12), 1.0-2.5 us "
                       "per protocol hop (tree 2 log2 N .. ring 2 (N - 1)), xGMI 153 GB/s per link, collectives not overlapped with "
                       "compute (clip_adam waits for them); 1-GPU epoch %.2f ms" % (alpha0_us, epoch_ms_1gpu)}


def host_cpu():
    """CPU model string, physical cores (unique (package, core) pairs) and logical CPUs of this host."""
    model, pairs, phys, core = "unknown", set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                pairs.add((phys, core)); phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"model": model, "physical_cores": len(pairs) or None, "logical_cpus": os.cpu_count(), "usable_cpus": affinity}


def cpu_baseline(wl, compute):
    """The CPU oracle (oracle/ppo_oracle.py, kind 'port': the restatement pinned against the reference) on a bounded
    sample of the same workload (warm-up, then 16 minibatch updates + 64 inference step pairs +
    1 GAE), extrapolated to one epoch. Threads: the best of 8/16/32/64 — torch's intra-op pool thrashes beyond that."""
    from vision4leg_amd import recipes as util
    from oracle import ppo_oracle as orc
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    ncpu = os.cpu_count() or 1
    case = dict(wl, seed=0)
    torch.manual_seed(0)
    pf, vf = util.build_nets(networks, policies, case)
    opf = {k: v.clone() for k, v in pf.state_dict().items()}
    ovf = util.share_encoder(opf, {k: v.clone() for k, v in vf.state_dict().items()}, wl["kind"])
    oracle = orc.PPOOracle(wl["kind"], opf, ovf, {k: v.clone() for k, v in opf.items()}, wl["S"], "f32")
    oracle.sync_target()
    B, E, T = wl["B"], wl["E"], wl["T"]
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    b = util.make_batch(case, B=B)
    args = (t(b["obs"]), t(b["acts"]), t(b["advs"]), t(b["estimate_returns"]), t(b["values"]), 1e-4, 1e-4)
    oracle.update(*args)  # warm-up
    # torch's intra-op pool thrashes with hundreds of threads on these layer sizes: pick the best of a few widths
    best = None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        oracle.update(*args)
        t0 = time.perf_counter()
        oracle.update(*args)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
        if dt > 30:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    n_upd = 16 if best[1] < 1.0 else 4  # ~4 s at 0.23 s per update; fewer on a slow host to stay inside ~30 s
    t0 = time.perf_counter()
    for _ in range(n_upd):
        oracle.update(*args)
    t_upd = (time.perf_counter() - t0) / n_upd
    fwd = orc.FORWARDS[wl["kind"]]
    ob = t(b["obs"][:E])
    pp = {k: v for k, v in opf.items() if k != "logstd"}
    with torch.no_grad():
        fwd(pp, ob, wl["S"]); fwd(ovf, ob, wl["S"])
        n_inf = 64
        t0 = time.perf_counter()
        for _ in range(n_inf):
            fwd(pp, ob, wl["S"]); fwd(ovf, ob, wl["S"])
        t_inf = (time.perf_counter() - t0) / n_inf
    ro = orc.synthetic_rollout(T, E, 1, 1, seed=0, with_images=False)
    t0 = time.perf_counter()
    orc.gae(ro["rewards"], ro["values"], ro["terminals"], ro["time_limits"], ro["last_value"], 0.99, 0.95, True)
    t_gae = time.perf_counter() - t0
    n_mb = OPT_EPOCHS * (T * E // B)
    t_epoch = n_mb * t_upd + T * t_inf + t_gae
    return {
        "value": round(E * T / t_epoch, 2), "unit": "env-steps/s", "cores": cores, "kind": "port", "host": host_cpu(),
        "sample": "%d minibatch updates (B=%d) + %d rollout step pairs (pf+vf fwd, E=%d) + 1 GAE [%dx%d] of the same "
                  "workload, fp32 torch-CPU oracle with %d threads (best of 8/16/32/64 on this host), extrapolated to one epoch (%d updates, %d steps)"
                  % (n_upd, B, n_inf, E, T, E, cores, n_mb, T),
        "update_only_value": round(E * T / (n_mb * t_upd), 2),
        "s_per_update": round(t_upd, 4), "s_per_rollout_step": round(t_inf, 5), "s_gae": round(t_gae, 4),
    }


def parity_check(wl, compute, dev):
    """The checker leg (oracle = test infrastructure, never in the timed region) on THE PATH THE BENCH TIMES, at the
    workload's own E and B: 2 B / E steps of `RolloutActor.step` filing action / value / log pi_old, then 2 stored-log-pi
    updates through `run_updates` (the second one is a hipGraph replay) — against the CPU oracle fed the reference protocol
    (pf.explore / vf per step, frozen-target forward inside every minibatch update), in the bench's compute flavour and
    in fp32 (oracle/bench_path.py; the same function tests/test_gpu_bench_path.py gates at E = 32 / 64, 4 updates)."""
    from oracle import bench_path
    E, B = wl["E"], wl["B"]
    T = max(2, 2 * B // E)
    r = bench_path.run(dict(wl, seed=0), E, T, B, 2, compute, dev, threads=min(16, os.cpu_count() or 1))
    out = {"path": r["path"], "batch": B, "envs": E, "rollout_steps": T, "updates": 2, "finite": r["finite"],
           "graph_replays": r["graph_replays"]}
    for fl in dict.fromkeys((compute, "f32")):
        out["max_rel_err_infos_vs_%s_oracle" % fl] = float("%.3e" % r["infos_vs_%s" % fl])
        out["max_abs_param_diff_vs_%s_oracle" % fl] = float("%.3e" % r["param_max_vs_%s" % fl])
        out["mean_abs_param_diff_vs_%s_oracle" % fl] = float("%.3e" % r["param_mean_vs_%s" % fl])
        out["rollout_mean_rel_vs_%s_oracle" % fl] = float("%.3e" % r["rollout_mean_vs_%s" % fl])
        out["rollout_value_rel_vs_%s_oracle" % fl] = float("%.3e" % r["rollout_value_vs_%s" % fl])
        out["stored_logp_abs_vs_%s_oracle" % fl] = float("%.3e" % r["rollout_logp_abs_vs_%s" % fl])
    out["rel_err_definition"] = "tensor distances: max |a - b| / max |b|; infos: |a - b| / max(1, |b|) per scalar"
    if compute != "f32":
        out["rollout_mean_envelope_p95"] = float("%.3e" % r["rollout_mean_envelope_p95"])
        out["rollout_value_envelope_p95"] = float("%.3e" % r["rollout_value_envelope_p95"])
        out["envelope_note"] = ("the bf16 oracle's own sensitivity: p95 over 8 forward passes with parameters nudged by 1e-7; "
                                "same rounding points <=> the HIP distance is of that size (gated at 3x in tests/test_gpu_bench_path.py)")
        out["oracle_%s_vs_f32_infos_per_update" % compute] = r["oracle_%s_vs_f32_per_update" % compute]
        out["hip_vs_f32_infos_per_update"] = r["infos_vs_f32_per_update"]
    return out


def fast_collector(wl, compute, dev):
    """The critical-path transfers, timed: the product's own `VecOnPolicyCollector` (fast path) + `DeviceOnPolicyReplayBuffer`
    + `PPO.update_per_epoch` over a zero-cost vec env that hands out float64 observation rows the way the reference's env
    wrappers do (collector/on_policy.py:90-100). Per env step: fp64 -> fp32 cast into pinned memory, H2D of E x (S+16384)
    fp32, the two rollout launches, D2H of the [E][A] action (the simulator needs it before it can step) — a synchronous
    vec env cannot overlap any of it. One warm-up epoch, three timed (median reported)."""
    from vision4leg_amd import recipes
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    from vision4leg_amd.torchrl.algo import PPO
    from vision4leg_amd.torchrl.collector import VecOnPolicyCollector
    from vision4leg_amd.torchrl.replay_buffers import DeviceOnPolicyReplayBuffer
    case = dict(wl, seed=0)
    E, T, B = wl["E"], wl["T"], wl["B"]
    torch.manual_seed(0)
    pf, vf = recipes.build_nets(networks, policies, case)
    env = recipes.ZeroCostVecEnv(E, case, p_done=0.0)
    buf = DeviceOnPolicyReplayBuffer(max_replay_buffer_size=E * T, env_nums=E, time_limit_filter=True)
    coll = VecOnPolicyCollector(vf, env=env, eval_env=recipes.ZeroCostVecEnv(E, case), pf=pf, replay_buffer=buf, device=dev,
                                epoch_frames=E * T, max_episode_frames=10 ** 9)

    class Log:
        def add_update_info(self, info): pass
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=OPT_EPOCHS, tau=0.95, shuffle=True,
                entropy_coeff=0.005, collector=coll, replay_buffer=buf, logger=Log(), device=dev, discount=0.99,
                num_epochs=1500, batch_size=B)

    def epoch():
        t0 = time.perf_counter()
        coll.train_one_epoch()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        agent.update_per_epoch()
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1
    epoch()
    runs = []
    for ep in (1, 2, 3):  # three timed epochs, the median one is reported (a single epoch swings +-25 % with the box's host load)
        agent.current_epoch = ep
        runs.append(epoch())
    t_coll, t_upd = sorted(runs, key=lambda r: r[0] + r[1])[1]
    D = recipes.obs_dim(case)
    return {"value": round(E * T / (t_coll + t_upd), 1), "unit": "env-steps/s",
            "collect_ms_per_epoch": round(1e3 * t_coll, 2), "update_ms_per_epoch": round(1e3 * t_upd, 2),
            "collect_us_per_env_step": round(1e6 * t_coll / T, 1),
            "h2d_bytes_per_env_step": (E * (wl["S"] * 4 + (D - wl["S"]) * 2) if coll._split else E * D * 4),
            "observation_rows_over_pcie": ("proprio fp32 + depth stack in the 16-bit operand type (what the kernels round the image to at "
                                           "ingest: bit-identical results, RolloutActor.step_host_split)" if coll._split else "fp32 rows"),
            "host_cast_threads": coll.cast_threads,
            "what": "VecOnPolicyCollector(fast path).train_one_epoch over a zero-cost vec env (float64 rows) + "
                    "PPO.update_per_epoch (last-value forward, GAE, %d stored-log-pi graph updates, one stats read-back): "
                    "everything the reference's train loop does except env.step" % (OPT_EPOCHS * (E * T // B))}


def reference_protocol(wl, compute, dev):
    """One epoch driven the way the UNCHANGED reference collector / PPO drive the modules (collector/on_policy.py:90-155,
    ppo.py:28-40,125-153): per env step `torch.Tensor(ob).to(device)` from pageable float64 rows, `pf.explore`, `vf`,
    D2H of action and value, float64 host `OnPolicyReplayBuffer.add_sample`; then last value + GAE and 48 minibatch
    updates, each gathered from the float64 host arrays and uploaded (5 H2D copies). Simulator time is zero (synthetic
    rows); this is what a maintainer gets from `overlay.install()` without touching the collector."""
    from vision4leg_amd import recipes as util
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    from vision4leg_amd.torchrl.algo import PPO
    from vision4leg_amd.torchrl.replay_buffers import OnPolicyReplayBuffer
    case = dict(wl, seed=0)
    E, T, B, A = wl["E"], wl["T"], wl["B"], wl["A"]
    D = util.obs_dim(case)
    torch.manual_seed(0)
    pf, vf = util.build_nets(networks, policies, case)
    buf = OnPolicyReplayBuffer(max_replay_buffer_size=E * T, env_nums=E, time_limit_filter=True)

    class Coll: epoch_frames = E * T

    class Log:
        def add_update_info(self, info): pass
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=OPT_EPOCHS, tau=0.95, shuffle=True,
                entropy_coeff=0.005, collector=Coll(), replay_buffer=buf, logger=Log(), device=dev, discount=0.99,
                num_epochs=1500, batch_size=B)
    rs = np.random.RandomState(0)
    pool = [np.concatenate([np.clip(rs.randn(E, wl["S"]), -10, 10), np.clip(rs.randn(E, D - wl["S"]), -2.5, 2.8)], 1)
            for _ in range(8)]  # float64 rows as the env wrappers hand them over

    def epoch():
        ob = pool[0]
        for t_ in range(T):
            ob_t = torch.Tensor(ob).to(dev)
            acts = agent.pf.explore(ob_t)["action"].detach().cpu().numpy().reshape(E, A)  # (explore squeezes a batch of 1)
            values = agent.vf(ob_t).detach().cpu().numpy().reshape(E, 1)
            nxt = pool[(t_ + 1) % len(pool)]
            buf.add_sample({"obs": ob, "next_obs": nxt, "acts": acts, "values": values, "rewards": np.ones((E, 1)),
                            "terminals": np.zeros((E, 1), dtype=bool), "time_limits": [False]})
            ob = nxt
        t1 = time.perf_counter()
        agent.update_per_epoch()
        torch.cuda.synchronize()
        return t1
    epoch()  # warm-up (allocations, graph-free eager path)
    t0 = time.perf_counter()
    t1 = epoch()
    t2 = time.perf_counter()
    return {"value": round(E * T / (t2 - t0), 1), "unit": "env-steps/s", "ms_per_epoch": round(1e3 * (t2 - t0), 1),
            "rollout_ms": round(1e3 * (t1 - t0), 1), "update_ms": round(1e3 * (t2 - t1), 1),
            "what": "reference call protocol on the HIP modules: per-step pageable H2D + pf.explore + vf + D2H, float64 host "
                    "buffer, per-minibatch gather + upload; no RolloutActor, no device-resident buffer, no stored log-prob"}


def h2d_per_step(wl, dev, steps=64):
    """Pinned-host -> HBM upload of one env step's E observation rows (E x (S+16384) fp32), what the fast collector pays
    per step when it is NOT overlapped with the simulator: mean ms per step over back-to-back async copies + one sync."""
    from vision4leg_amd import recipes as util
    D = util.obs_dim(dict(wl, seed=0))
    host = [torch.empty(wl["E"], D, dtype=torch.float32).pin_memory() for _ in range(2)]
    devb = [torch.empty(wl["E"], D, dtype=torch.float32, device=dev) for _ in range(2)]
    for i in range(4):
        devb[i & 1].copy_(host[i & 1], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        devb[i & 1].copy_(host[i & 1], non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


PEAK_HBM = 8000.0  # GB/s, MI355X_MICROARCH.md


def algo_bytes(kernel, wl, compute):
    """ALGORITHMIC HBM bytes of one launch of the update's big kernels (DESIGN.md section 4 derives the per-sample figures):
    every operand / result the kernel must read / write once, at the storage width it has in this mode."""
    n, t = wl["B"], (4 if compute == "f32" else 2)
    R = 17 * n                                   # token rows
    tok = 64 * 4                                 # one fp32 token row
    # c1 + c2 of one sample in the operand type (round 5: the trainer's passes save them as T), c3 fp32
    conv_act = (225 * 32 + 36 * 64) * t + 16 * 64 * 4
    blocks = min(n, 256)
    per_launch = {
        # image + proprio row in; c1, c2, c3, tokens, 2 MLP activations out
        "fused_encoder": n * (16384 * t + 128 * 4 + conv_act + 17 * tok + 2 * 256 * 4),
        # x in; x out, qkv, P, xhat1/2, rstd1/2 (fp32) + layer-input copy, ctx, x1, f (T)
        "fused_layer": R * (2 * tok + 192 * 4 + 2 * tok + 8 + (64 + 64 + 64 + 256) * t) + n * 289 * 4,
        # both layers in one launch: the second layer's input rows stay in LDS (one token-row read less)
        "fused_layer_stack": 2 * (R * (2 * tok + 192 * 4 + 2 * tok + 8 + (64 + 64 + 64 + 256) * t) + n * 289 * 4) - R * tok,
        # dy, xhat1/2, rstd, qkv (fp32), P, f (T) in; dz2, df, dz1, dqkv (T) + dx (fp32) out
        "fused_layer_bwd": R * (tok + 2 * tok + 8 + 192 * 4 + 256 * t + (64 + 256 + 64 + 192) * t + tok) + n * 289 * 4,
        # both layers in one launch: the upper layer's dx is the lower layer's dy in LDS (one token-row read less)
        "fused_layer_bwd_stack": 2 * (R * (tok + 2 * tok + 8 + 192 * 4 + 256 * t + (64 + 256 + 64 + 192) * t + tok) + n * 289 * 4) - R * tok,
        # dc3 (fp32), c2, c1 (T) + image (T) in; one dW1 + dW2 slab per block out
        "fused_conv_bwd": n * (16 * 64 * 4 + (36 * 64 + 225 * 32) * t + 16384 * t) + blocks * (32 * 256 + 64 * 512) * 4,
        "fused_conv3_wgrad": n * (16 * 64 * 4 + 36 * 64 * t) + blocks * 64 * 576 * 4,
        # both operands of the 4 linears x 2 layers (T) in; 55 slabs of the 4 weight shapes x 2 layers out
        "gemm_tn_wide": 2 * R * 2 * (64 + 256 + 64 + 192) * t + 2 * 55 * 49152 * 4,
        # wave-per-sample kernels (csrc/wps.h). forward: tokens in, layer 1's input rows out (+ pooled / head activations);
        # backward: both layers' input rows + dout / masks in, 2 x 1024 features x 17 tokens of weight-grad operands (T) + dx,
        # dc3, head / encoder-MLP grads out; weight-grads: those operands in, one slab set per 32 samples out
        "wps_layer_stack_head": R * 2 * tok + n * (128 + 256 + 256 + 16) * 4,
        "wps_layer_bwd_stack": R * 2 * tok + 2 * R * 1024 * t + R * tok + n * (16 * 64 + 16 + 4 * 256 + 2 * 256) * 4,
        "wps_wgrad": 2 * R * 1024 * t + 2 * (-(-n // 32)) * (49152 + 576) * 4,
    }
    # NatureCNN nets' dense stack (csrc/dense_stack.h), fp32 rows. forward: conv3's flatten (+ the proprio half of the concat)
    # in; projector output, h0, h1, padded head output out. backward: dout + the ReLU masks (h1, h0, concat, flatten, e0) in;
    # dh1, dh0, dcat, dc3, de0 out. Weights: every (row-block) streams them; algorithmic = once.
    if wl["kind"] == "cnn":
        w_f = (1024 * 256 + 512 * 256 + 256 * 256 + 256 * 16) * t
        w_b = (64 * 256 + 256 * 256 + 512 * 256 + 1024 * 256 + 256 * 256) * t
        per_launch["dense_stack_fwd"] = n * (1024 + 256 + 256 + 256 + 256 + 16) * 4 + w_f
        per_launch["dense_stack_bwd"] = n * (16 + 256 + 256 + 512 + 1024 + 256 + 256 + 256 + 512 + 1024 + 256) * 4 + w_b
    elif wl["kind"] == "cnn_vis":
        w_f = (1024 * 256 + 256 * 256 + 256 * 16) * t
        w_b = (64 * 256 + 256 * 256 + 1024 * 256) * t
        per_launch["dense_stack_fwd"] = n * (1024 + 256 + 256 + 16) * 4 + w_f
        per_launch["dense_stack_bwd"] = n * (16 + 256 + 256 + 1024 + 256 + 256 + 1024) * 4 + w_b
    E = wl["E"]
    enc_w = (4 * 64 * 32 + 32 * 16 * 64 + 64 * 9 * 64 + 64 * 64 + 128 * 256 + 256 * 256 + 256 * 64) * t
    layer_w = (64 * 192 + 64 * 64 + 64 * 256 + 256 * 64) * t
    head_w = (128 * 256 + 256 * 256 + 256 * 16) * t
    # rollout step (E rows): observation rows in, rollout image / proprio rows + tokens out, weights streamed once;
    # layer stack: tokens in for both nets, both nets' layer + head weights, action / value / log-prob out
    per_launch["rollout_encoder"] = E * ((wl["S"] + 16384) * 4 + 16384 * t + 128 * 4 + 17 * tok) + enc_w
    per_launch["rollout_layers_head"] = 2 * E * 17 * tok + 2 * (wl.get("layers", 2) * layer_w + head_w) + E * 64
    # NatureCNN nets (csrc/rollout_dense.h): conv3's flatten (T) in, every dense weight once (the projector is the shared
    # encoder's; the heads per net), action / value / log-prob out
    if wl["kind"] == "cnn":
        per_launch["rollout_dense"] = E * (1024 + 256) * t + (1024 * 256 + 2 * (512 * 256 + 256 * 256 + 256 * 16)) * t + E * 64
    elif wl["kind"] == "cnn_vis":
        per_launch["rollout_dense"] = E * 1024 * t + 2 * (1024 * 256 + 256 * 256 + 256 * 16) * t + E * 64
    # state MLP (rollout_mlp2_kernel): proprio rows in, the shared base + each net's head once per net-block
    per_launch["rollout_mlp"] = E * wl["S"] * 4 + 2 * (128 * 256 + 256 * 256 + 2 * 256 * 256 + 256 * 16) * t + E * 64
    for k in sorted(per_launch, key=len, reverse=True):  # longest name first: fused_layer_bwd_* before fused_layer_*
        if kernel == k or kernel.startswith(k + "_"):
            return float(per_launch[k])
    return None


CU_FETCH_B_PER_CLK = 31.8  # what one CU streams from L2 into registers, every CU at once (tools/probe/cu_fetch.hip; DMA to LDS: 59-63)
CU_CLOCK_HZ = 2.4e9


def block_fetch_bytes(kernel, wl, compute):
    """Bytes ONE block pulls through its CU (the weights it streams + its share of the launch's algorithmic bytes) for the
    kernels whose time is their blocks' fetch-and-compute chain: a CU streams at most 31.8 B/clk from L2 into registers
    (DESIGN.md 4.1), so this is the per-block view the chip-level HBM fraction cannot give. -> (bytes, blocks) or None."""
    n, E, t = wl["B"], wl["E"], (4 if compute == "f32" else 2)
    lw = (64 * 192 + 64 * 64 + 64 * 256 + 256 * 64) * t          # one transformer layer's weights
    head_w = (128 * 256 + 256 * 256 + 256 * 16) * t
    conv_w = (4 * 64 * 32 + 32 * 16 * 64 + 64 * 9 * 64 + 64 * 64) * t
    mlp_w = (128 * 256 + 256 * 256 + 256 * 64) * t
    L = wl.get("layers", 2)
    table = {
        "rollout_layers_head": (2 * E, L * lw + head_w),
        "rollout_encoder": (E, conv_w),
        "fused_conv_bwd": (min(n, 256), -(-n // min(n, 256)) * 64 * 576 * t + 4 * 32 * 256 * t),
        "fused_layer_bwd_stack": (-(-n // 4), L * lw + head_w + mlp_w - 128 * 256 * t + 64 * 64 * t),
        "fused_layer_stack_head": (-(-n // 2), L * lw + head_w),
        "wps_layer_stack_head": (-(-n // 4), L * lw + head_w),
        "wps_layer_bwd_stack": (-(-n // 4), 2 * L * lw + head_w + mlp_w - 128 * 256 * t + 64 * 64 * t),
        "fused_encoder": (256, conv_w),
    }
    if wl["kind"] in ("cnn", "cnn_vis"):  # 16 rows per block, each block streams the whole stack's weights
        fuse = wl["kind"] == "cnn"
        table["dense_stack_fwd"] = (-(-n // 16), ((1024 * 256 + 512 * 256 if fuse else 1024 * 256) + 256 * 256 + 256 * 16) * t)
        table["dense_stack_bwd"] = (-(-n // 16), (64 * 256 + 256 * 256 + (512 * 256 + 1024 * 256 + 256 * 256 if fuse else 1024 * 256)) * t)
    if kernel not in table:
        return None
    blocks, w = table[kernel]
    by = algo_bytes(kernel, wl, compute)
    return (w + (by / blocks if by else 0.0), blocks)


def _profiled(L, fn):
    """Run fn() with the library's HIP-event profiler on; -> [(phase|op|kernel, calls, total_us, flops)]."""
    import ctypes as C
    torch.cuda.synchronize()
    L.v4l_prof_enable(1)
    fn()
    buf = C.create_string_buffer(1 << 20)
    L.v4l_prof_collect(buf, len(buf))
    L.v4l_prof_enable(0)
    rows = []
    for line in buf.value.decode().splitlines():
        label, calls, us, flops = line.split("\t")
        rows.append((label, int(calls), float(us), float(flops)))
    return rows


def _per_kernel(rows):
    out = {}
    for label, calls, us, fl in rows:
        k = label.split("|")[-1]
        c0, u0, f0 = out.get(k, (0, 0.0, 0.0))
        out[k] = (c0 + calls, u0 + us, f0 + fl)
    return out


def roofline(ep, compute, breakdown_path):
    """Per-op timings of ONE profiled step (rollout (i) + update (ii)-(iv), the same proportions as the timed region): HIP
    events around every launch, on the launch stream. The dominant kernel of the WHOLE step (largest total time) is priced
    against BOTH ceilings — algorithmic FLOPs / dense MFMA peak and algorithmic HBM bytes / 8 TB/s — and reported against
    the one that bounds it (the larger lower-bound time). `rollout` and `update` carry the dominant kernel of each side."""
    from vision4leg_amd import _lib
    L = _lib.lib()
    wl = ep.wl
    upd = _profiled(L, ep.update)
    roll = _profiled(L, ep.rollout) if ep.actor is not None else []
    global LAST_ALLREDUCE
    ar = [(c, u) for label, c, u, _ in upd if label.split("|")[-1].startswith("allreduce")]
    LAST_ALLREDUCE = ({"calls": sum(c for c, _ in ar), "us_per_call": round(sum(u for _, u in ar) / max(1, sum(c for c, _ in ar)), 2),
                       "what": "HIP events around ncclAllReduce of one gradient bucket on the update's stream (eager profiled pass)"}
                      if ar else None)
    rows = sorted(upd + roll, key=lambda r: -r[2])
    if not rows:
        return None
    total_us = sum(r[2] for r in rows)
    upd_us, roll_us = sum(r[2] for r in upd), sum(r[2] for r in roll)
    if breakdown_path:
        with open(breakdown_path, "w") as f:
            f.write("# one profiled step = rollout (%d env steps x %d envs) + epoch-update (%d updates): phase|op|kernel, calls, "
                    "total_us, avg_us, TFLOP/s, share of the step's kernel time\n" % (wl["T"], wl["E"], ep.stats.shape[0]))
            for label, calls, us, fl in rows:
                f.write("%-70s %6d %12.1f %9.2f %9.2f %6.2f%%\n"
                        % (label, calls, us, us / calls, fl / us * 1e-6 if us > 0 else 0.0, 100 * us / total_us))
            f.write("# total kernel time %.1f us (update %.1f, rollout %.1f)\n" % (total_us, upd_us, roll_us))
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_pass.sh + pmc_summary.py (same command)
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
        except Exception:
            pmc = {}

    def price(kern, calls, us, fl):
        avg_us = us / calls
        # ALGORITHMIC numerator: the library's per-launch FLOP counts are what the kernel PERFORMS; the layer
        # backward (and the fused forward-loss-backward launch) re-run one forward per layer to rebuild activations — that recompute
        # is the kernel's own choice, not work the algorithm asks for, and does not count
        performed = fl / calls
        if kern in ("wps_layer_fb_stack", "wps_layer_bwd_stack"):
            fl = fl - calls * wl.get("layers", 2) * 2.0 * wl["B"] * 872576.0
        tf = fl / us * 1e-6                      # TFLOP/s
        by = algo_bytes(kern, wl, compute)
        gbs = by / avg_us * 1e-3 if by else None  # GB/s
        t_mfma = fl / calls / (PEAK[compute] * 1e12) * 1e6
        t_hbm = by / (PEAK_HBM * 1e9) * 1e6 if by else 0.0
        hbm_bound = bool(by) and t_hbm >= t_mfma
        rec = pmc.get(kern) or {}
        for suffix in ("_stack_head", "_stack", "_head", "_tail"):  # PMC rows are keyed by kernel function (tools/pmc_traffic.py)
            if not rec and kern.endswith(suffix):
                rec = pmc.get(kern[:-len(suffix)]) or {}
        bf = block_fetch_bytes(kern, wl, compute)
        cu = None
        if bf:
            bpc = bf[0] / (avg_us * 1e-6 * CU_CLOCK_HZ)
            cu = {"bytes_per_block": round(bf[0]), "blocks": bf[1], "achieved_B_per_clk": round(bpc, 2),
                  "ceiling_B_per_clk": CU_FETCH_B_PER_CLK, "frac": round(bpc / CU_FETCH_B_PER_CLK, 3),
                  "note": "L2 -> CU fetch of ONE block (its weights + its share of the algorithmic bytes) over the launch time, "
                          "against the measured per-CU streaming rate (tools/probe/cu_fetch.hip, DESIGN.md 4.1)"}
        return {
            "cu_fetch": cu,
            "bound": "hbm" if hbm_bound else "mfma", "kernel": kern,
            "achieved": round(gbs, 1) if hbm_bound else round(tf, 2),
            "peak": PEAK_HBM if hbm_bound else PEAK[compute], "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": round((gbs / PEAK_HBM) if hbm_bound else (tf / PEAK[compute]), 5),
            "traffic": rec.get("hbm_bytes_per_launch"), "traffic_source": rec.get("source"),
            "algorithmic_bytes_per_launch": by, "algorithmic_flops_per_launch": fl / calls, "performed_flops_per_launch": performed,
            "hbm_frac": round(gbs / PEAK_HBM, 5) if gbs else None, "mfma_frac": round(tf / PEAK[compute], 5),
            "avg_launch_us": round(avg_us, 2), "launches": calls, "share_of_kernel_time": round(us / total_us, 4),
        }
    kern, (calls, us, fl) = max(_per_kernel(rows).items(), key=lambda kv: kv[1][1])
    res = price(kern, calls, us, fl)
    res["scope"] = "whole timed step: rollout-side inference + GAE + %d minibatch updates" % ep.stats.shape[0]
    res["all_kernels_tflops"] = round(sum(r[3] for r in rows) / total_us * 1e-6, 2)
    res["kernel_time_split"] = {"update_us": round(upd_us, 1), "rollout_us": round(roll_us, 1)}
    if upd:
        k, (c, u, f) = max(_per_kernel(upd).items(), key=lambda kv: kv[1][1])
        res["update"] = price(k, c, u, f)
    if roll:
        k, (c, u, f) = max(_per_kernel(roll).items(), key=lambda kv: kv[1][1])
        r = price(k, c, u, f)
        nets = 2 if "layer" in k else 1
        blocks = wl["E"] * nets + (wl["E"] + 31) // 32 * (0 if "layer" in k else 1)
        r["workgroups"] = blocks
        r["cu_occupancy"] = round(min(1.0, blocks / 256.0), 4)
        r["env_step_us"] = round(roll_us / wl["T"], 2)
        res["rollout"] = r
    # the transformer block's own kernels (north star: MFMA fraction "for the transformer block"; SURVEY 8d: kernel time x
    # known FLOPs over the block's kernels only): fused layer forward / backward launches and their weight-grad GEMM
    tb = [r for r in upd if r[0].split("|")[1] in ("layer", "layer.wgrad", "attn", "ln1", "ln2")
          or ".self_attn." in r[0] or ".linear1." in r[0] or ".linear2." in r[0]]
    if tb:
        tb_us, tb_fl = sum(r[2] for r in tb), sum(r[3] for r in tb)
        res["transformer_block"] = {
            "tflops": round(tb_fl / tb_us * 1e-6, 2), "mfma_frac": round(tb_fl / tb_us * 1e-6 / PEAK[compute], 5),
            "share_of_kernel_time": round(tb_us / total_us, 4), "kernels": sorted({r[0].split("|")[-1] for r in tb}),
            "note": "2*M*N*K FLOPs PERFORMED by the update's layer launches (incl. the backward's recompute) / their HIP-event time "
                    "(DESIGN.md section 4); the flat key transformer_block_mfma_frac has the ALGORITHMIC numerator"}
        # ALGORITHMIC numerator: forward + data-grad + weight-grad of the layers = 3 x the forward's
        # FLOPs per net-pass, whatever the kernels recompute; a net-pass = one launch of the layers' forward kernel
        passes = sum(r[1] for r in tb if r[0].split("|")[1] == "layer" and "bwd" not in r[0].split("|")[-1])
        alg = passes * 3 * 2.0 * wl["B"] * wl.get("layers", 2) * 872576.0
        res["transformer_block_mfma_frac"] = round(alg / tb_us * 1e-6 / PEAK[compute], 5)
        res["transformer_block_us_per_net_pass"] = round(tb_us / max(passes, 1), 2)
    # flat, driver-visible keys
    n_upd = ep.stats.shape[0]
    res["launches_per_update"] = round(sum(r[1] for r in upd if not r[0].split("|")[-1].startswith("allreduce")) / n_upd, 2)
    res["update_kernel_us"] = round(upd_us / n_upd, 1)
    if roll:
        res["rollout_us_per_env_step"] = round(roll_us / wl["T"], 2)
    traffic, unpriced = 0.0, []
    for k, (c, u, _) in _per_kernel(upd).items():
        rec = pmc.get(k) or {}
        for suffix in ("_stack_head", "_stack", "_head", "_tail"):
            if not rec and k.endswith(suffix):
                rec = pmc.get(k[:-len(suffix)]) or {}
        if not rec and c >= n_upd and u / n_upd >= 1.0 and not k.startswith("allreduce"):
            unpriced.append(k)  # a launch of every update (>= 1 us) without a PMC row: the sum below would silently leave it out
        traffic += (rec.get("hbm_bytes_per_launch") or 0.0) * c / n_upd
    # (round 6: a line printed 0.84 GB while pmc_traffic.json predated the fused layer launch — its 2 x 116 MB were missing.
    # The sum is only reported when every launch of the update has a row.)
    res["hbm_bytes_per_update"] = round(traffic) if traffic and not unpriced else None
    if unpriced:
        res["hbm_bytes_per_update_unpriced_kernels"] = sorted(unpriced)
    res["hbm_bytes_per_update_source"] = "profiles/pmc_traffic.json (PMC FETCH_SIZE x2 / WRITE_SIZE per launch) x this run's launches per update"
    res["method"] = ("HIP events around every launch of one extra (untimed) rollout + epoch-update on the launch stream; "
                     "achieved = algorithmic bytes (bench.py algo_bytes, DESIGN.md section 4) or 2*M*N*K FLOPs per launch / "
                     "average launch time")
    return res


def batch1_latency(ep, policies, dev):
    """the batch-1 deployment call (RolloutActor(env_nums=1).eval_act — the reference ships a
    TensorRT engine for this, a1_hardware/convert_tensor_rt/): device time per call by events over 300 back-to-back
    launches, and the synchronous host round trip incl. the action's D2H copy."""
    actor = policies.RolloutActor(ep.pf, ep.vf, 1)
    x = ep.obs[:1].clone()
    for _ in range(30):
        actor.eval_act(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(300):
        actor.step(x, deterministic=True)
    e1.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        actor.eval_act(x)
    return {"device_us": round(e0.elapsed_time(e1) * 1e3 / 300, 2),
            "host_round_trip_us": round((time.perf_counter() - t0) * 1e6 / 100, 1),
            "what": "policy mean for one observation row [1][S+4*64*64], operands in the net's compute type"}


def main():
    a = parse()
    # stdout carries exactly ONE line (the JSON result of rank 0): everything else any library writes there while the run is
    # going on (RCCL prints a version banner on communicator creation) is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: re-launch ourselves as N ranks (one per GPU) and pass rank 0's line through
        os.dup2(real_stdout, 1)
        raise SystemExit(subprocess.call(launcher_argv(a.gpus, sys.argv[1:])))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with: %s)"
                         % (a.gpus, world, " ".join(launcher_argv(a.gpus, sys.argv[1:]))))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # V4L_FORCE_DP_PHASES=1 under a 1-process torchrun exercises the whole multi-GPU code path (RCCL group, barriers,
    # four-phase update with an all-reduce per optimiser step) on a single GPU
    dist_on = world > 1 or (os.environ.get("V4L_FORCE_DP_PHASES", "0") != "0" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", device_id=dev)
    wl = dict(WORKLOADS[a.workload])
    if a.no_rollout:
        wl["skip_rollout"] = True
    np.random.seed(rank)
    ep = Epoch(wl, a.compute, dev, world)

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        ep.step(not a.no_rollout)
    barrier()
    t0 = time.perf_counter()
    t_roll = 0.0
    for _ in range(a.steps):
        if not a.no_rollout:
            r0 = time.perf_counter()
            ep.rollout()
            torch.cuda.synchronize()
            t_roll += time.perf_counter() - r0
        ep.update()
        torch.cuda.synchronize()  # keeps the rollout/update split honest (one sync per epoch)
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt, t_roll], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt, t_roll = tt[0].item(), tt[1].item()
    frames = wl["E"] * wl["T"] * world
    value = frames * a.steps / dt
    res = {
        "metric": "PPO env-steps/s (encoder+update, sim excluded)", "value": round(value, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.compute, "data": "synthetic",
        "config": {"workload": wl["name"], "net": wl["kind"], "envs_per_gpu": wl["E"], "horizon": wl["T"],
                   "minibatch": wl["B"], "opt_epochs": OPT_EPOCHS, "parallelism": "dp%d" % world,
                   "includes_rollout_inference": not a.no_rollout,
                   "logp_old": "recorded at action time (target_pf forward skipped; SURVEY 8d declared saving)"
                               if ep.logp is not None else "target_pf evaluated per minibatch (ppo.py:55-57)"},
        "update_only_env_steps_per_s": round(frames * a.steps / max(dt - t_roll, 1e-9), 1),
        "rollout_inference_ms_per_step": round(1e3 * t_roll / a.steps, 3),
        "algorithmic_tflops": round(value * (MFLOP_PER_ENV_STEP[a.workload] - (OPT_EPOCHS * MFLOP_TARGET_FWD[a.workload]
                                                                               if ep.logp is not None else 0.0)) * 1e-6, 3),
    }
    if dist_on:
        ag = ep.agent
        res["config"]["dp_comm"] = ag.dp_comm_note  # which exchange runs, and the self-test that decided it (ppo.py::_library_comm)
        res["rccl_ranks"] = ag.trainer.comm_world() if ag.dp_in_library else torch.distributed.get_world_size()
        res["scaling_note"] = "weak scaling: every rank owns E envs; multi-GPU numbers are only as good as the node they ran on"
    stats_ok = bool(torch.isfinite(ep.stats[:, :18]).all().item())
    if not stats_ok:
        raise SystemExit("bench.py: non-finite logger statistics after the timed epochs: the step is invalid")
    res["stats_finite"] = stats_ok
    if rank == 0:
        res["roofline"] = roofline(ep, a.compute, a.breakdown)
        if res["roofline"]:
            n_upd_epoch = OPT_EPOCHS * (wl["E"] * wl["T"] // wl["B"])
            # wall clock of one update inside the timed region (graph replays incl. host gaps): (epoch - rollout) / updates
            res["roofline"]["update_us"] = round(1e6 * (dt - t_roll) / (a.steps * n_upd_epoch), 1)
        upd_per_epoch = OPT_EPOCHS * (wl["E"] * wl["T"] // wl["B"])
        bucket = 4 * (int(ep.vf.hip.total_params) + 8)
        if dist_on:
            res["allreduce"] = LAST_ALLREDUCE
            if world > 1:
                # prediction from THIS run's own compute time: strip the measured collectives, add the modelled ones
                ar_ms = (LAST_ALLREDUCE["us_per_call"] * 2 * upd_per_epoch * 1e-3) if LAST_ALLREDUCE else 0.0
                res["dp_model"] = dp_model(world, bucket, 12.0, None, max(1e3 * dt / a.steps - ar_ms, 1e-3), upd_per_epoch,
                                           wl["E"] * wl["T"])
                res["dp_model"]["allreduce_us_measured"] = LAST_ALLREDUCE["us_per_call"] if LAST_ALLREDUCE else None
        child = os.environ.get("V4L_BENCH_CHILD", "0") != "0"
        if world == 1 and not a.no_parity and not child and a.workload in ("loco", "loco64"):
            # side legs (driver-visible, all after the timed region): the other two compute modes with the same parity check, the DP
            # schedule on one rank, the net's option variants (max_pool; token_norm, use_pytorch_encoder), and the multi-GPU cost model
            notes = {"f32": "compute=f32: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) — the mode whose outputs meet the north star's "
                            "1e-3 against the fp32 reference literally in every case (DESIGN.md section 2)",
                     "f16": "compute=f16: IEEE half operands (v_mfma_f32_16x16x32_f16), fp32 accumulate, loss-gradient rows scaled by a "
                            "power of two — bf16's speed, ~8 x closer to the fp32 reference (DESIGN.md section 2)",
                     "bf16": "compute=bf16: bf16 operands (v_mfma_f32_16x16x32_bf16), fp32 accumulate"}
            for other in ("f32", "f16", "bf16"):
                if other != a.compute:
                    res[other + "_mode"] = dict(side_leg(a.workload, other, dev), parity_check=parity_check(wl, other, dev),
                                                note=notes[other])
                    res[other + "_mode_value"] = res[other + "_mode"]["value"]  # flat: the driver's parser drops nested objects
                else:
                    res[other + "_mode_value"] = value  # the headline's own mode, under the same flat name as the side legs
            res["dp1_ingraph"] = dp1_ingraph_leg(a.workload, a.compute)
            res["dp1_ingraph_value"] = res["dp1_ingraph"].get("value")
            res["option_variant"] = side_leg("loco_max", a.compute, dev, steps=2, warmup=1)
            # (round 5: the other two net options, fused since this round — DESIGN.md section 1)
            res["option_variants"] = {w: side_leg(w, a.compute, dev, steps=2, warmup=1) for w in ("loco_tn", "loco_pe")}
            alpha0 = (res["dp1_ingraph"].get("allreduce") or {}).get("us_per_call") or 12.0
            res["dp_model"] = {str(n): dp_model(n, bucket, alpha0, None, 1e3 * dt / a.steps, upd_per_epoch, wl["E"] * wl["T"])
                               for n in (2, 4, 8)}
        if world == 1 and not a.no_parity:
            res["parity_check"] = parity_check(wl, a.compute, dev)
            res["h2d_ms_per_step"] = round(h2d_per_step(wl, dev), 4)
            res["h2d_note"] = ("pinned upload of E x (S+16384) fp32 observation rows per env step, NOT inside `value` (the "
                               "epoch is resident in HBM when the timed region starts); value_incl_h2d adds T x this, unoverlapped")
            res["value_incl_h2d"] = round(frames * a.steps / (dt + a.steps * wl["T"] * res["h2d_ms_per_step"] * 1e-3), 1)
            if not a.no_rollout and not a.no_reference_protocol:
                res["reference_protocol"] = reference_protocol(wl, a.compute, dev)
            if not a.no_rollout and ep.actor is not None:
                # the headline WITH the critical-path transfers: the product's collector over a zero-cost env
                res["fast_collector"] = fast_collector(wl, a.compute, dev)
                res["value_incl_transfers"] = res["fast_collector"]["value"]
                res["value_incl_transfers_ratio"] = round(res["value_incl_transfers"] / value, 3)
                res["value_incl_transfers_note"] = (
                    "`value` starts with the epoch resident in HBM (the contract); value_incl_transfers is the same epoch "
                    "driven by VecOnPolicyCollector from float64 host rows: per env step a host cast, %d bytes read over PCIe by the kernels, "
                    "2 launches and the action's D2H, serialised by the synchronous vec-env protocol"
                    % res["fast_collector"]["h2d_bytes_per_env_step"])
        if not a.no_cpu_baseline and world == 1:  # the host baseline is an N = 1 measurement (the other ranks would idle)
            res["cpu_baseline"] = cpu_baseline(wl, a.compute)
            res["vs_cpu_baseline"] = round(value / world / res["cpu_baseline"]["value"], 1)
        if world == 1 and ep.actor is not None:
            import vision4leg_amd.torchrl.policies as policies
            res["batch1_inference"] = batch1_latency(ep, policies, dev)
    elif dist_on:
        ep.update()  # keep collectives matched with rank 0's profiled pass
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
