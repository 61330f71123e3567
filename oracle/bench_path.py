"""TEST INFRASTRUCTURE — checker for the configuration bench.py actually times. NOT part of the product.

bench.py's timed step is: `RolloutActor.step` at batch E (rollout_encoder2_kernel + rollout_stack_kernel) filing action,
value and log pi_old(a|s) into the HBM-resident rollout, then `PPO.run_updates` over B-row index minibatches as hipGraph
replays that read the STORED log pi_old instead of evaluating the frozen target policy. This module drives exactly that
through the product's Python shell / C ABI and checks it against the oracle fed the REFERENCE protocol:

  per env step   mean / std / value of `pf.explore(ob)` / `vf(ob)`        (torchrl/collector/on_policy.py:90-100)
                 log pi_old = Normal(mean, std).log_prob(action).sum(-1)  (torchrl/policies/continuous_policy.py:127-146)
  per minibatch  oracle.update with its OWN frozen-target forward         (torchrl/algo/on_policy/ppo.py:34,55-59)

Only tests/ and bench.py's `parity_check` leg import it (never inside a timed region).
"""
import os

import numpy as np
import torch

from . import ppo_oracle as orc


def _rel(a, b):
    """max |a - b| relative to the reference tensor's max-abs (policy means are O(1e-2): no floor at 1)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


ENV_SEEDS = 8


def _forward_envelope(fwd, params, obs, S, ref, scale=1e-7, mode="bf16"):
    """The bf16 oracle's own sensitivity (as tests/test_gpu_parity.py::_bf16_envelope): p95 over ENV_SEEDS runs with the
    parameters nudged by 1e-7 of the forward output's distance to the un-nudged run."""
    errs = []
    for sd in range(1, ENV_SEEDS + 1):
        gen = torch.Generator().manual_seed(sd)
        q = {k: v * (1 + scale * torch.randn(v.shape, generator=gen)) for k, v in params.items()}
        errs.append(_rel(fwd(q, obs, S, mode).reshape(ref.shape), ref))
    return float(np.percentile(errs, 95))


def run(case, E, T, B, U, mode, dev, seed=0, threads=None, envelope=True, traj_seeds=0):
    """case: a recipes case dict (kind, S, A, ...). Rollout of T steps x E envs, then U stored-log-pi updates of B rows.
    -> dict of measured distances. mode: the product's compute mode ("f32" | "bf16" | "f16"); the oracle runs in that flavour
    and (16-bit modes) also in fp32, so the caller can apply the trajectory rule |hip - fp32| <= k |bf16 oracle - fp32|.
    Distances between tensors are relative to the reference tensor's max-abs. envelope (bf16): also the bf16 oracle's own
    sensitivity on the rollout's mean / value (8 nudged forward passes each), the yardstick of the bf16 rollout gate.
    traj_seeds (bf16): that many extra bf16 oracles start from parameters nudged by 1e-7 and run the same U updates; the largest
    distance of their 18 infos to the un-nudged bf16 oracle's, per update, is the yardstick of the trajectory gate
    (`traj_envelope_per_update`): two bf16 evaluations of the same update sequence drift apart like that (the PPO clip makes the
    loss discontinuous in the ratio), an implementation with other rounding points drifts further."""
    os.environ["V4L_COMPUTE"] = mode
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    from vision4leg_amd import recipes
    from vision4leg_amd.engine import HipTrainer
    from vision4leg_amd.torchrl.algo import PPO
    from vision4leg_amd.torchrl.policies import RolloutActor
    if threads:
        torch.set_num_threads(threads)
    kind, S, A = case["kind"], case["S"], case["A"]
    assert B % E == 0 and (T * E) % B == 0, "B must be whole time rows and divide the rollout"
    torch.manual_seed(case.get("seed", 0))
    pf, vf = recipes.build_nets(networks, policies, case)
    pf, vf = pf.to(dev), vf.to(dev)

    class Coll:
        epoch_frames = T * E
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, entropy_coeff=0.005,
                collector=Coll(), device=dev, batch_size=B)
    flavours = list(dict.fromkeys((mode, "f32")))
    oracles = {}
    for fl in flavours:
        opf = {k: v.detach().cpu().clone() for k, v in pf.state_dict().items()}
        ovf = recipes.share_encoder(opf, {k: v.detach().cpu().clone() for k, v in vf.state_dict().items()}, kind)
        oracles[fl] = orc.PPOOracle(kind, opf, ovf, {k: v.clone() for k, v in opf.items()}, S, fl)
        oracles[fl].sync_target()
    nudged = []
    for sd in range(1, (traj_seeds if mode != "f32" else 0) + 1):
        gen = torch.Generator().manual_seed(sd)
        opf = {k: v.detach().cpu() * (1 + 1e-7 * torch.randn(v.shape, generator=gen)) for k, v in pf.state_dict().items()}
        ovf = recipes.share_encoder(opf, {k: v.detach().cpu() * (1 + 1e-7 * torch.randn(v.shape, generator=gen))
                                          for k, v in vf.state_dict().items()}, kind)
        nudged.append(orc.PPOOracle(kind, opf, ovf, {k: v.clone() for k, v in opf.items()}, S, mode))
        nudged[-1].sync_target()

    # ---- (i) the rollout the bench times: T steps of RolloutActor.step at batch E, everything filed on the device
    rs = np.random.RandomState(4242 + seed)
    obs_np = recipes.obs_rows(rs, T * E, case).astype(np.float32)
    obs = torch.from_numpy(obs_np).to(dev)
    net = pf.hip
    net.ensure_bound()
    state, image = net.alloc_rollout(T * E, dev)
    acts = torch.zeros(T * E, A, device=dev)
    vals = torch.zeros(T * E, device=dev)
    logp = torch.zeros(T * E, device=dev)
    actor = RolloutActor(pf, vf, E)
    actor.attach((state, image, acts, vals, logp))
    actor.seek(0)
    means, stds = [], []
    for t in range(T):
        torch.manual_seed(9000 + t)
        out = actor.step(obs[t * E:(t + 1) * E])
        means.append(out["mean"].clone())
        stds.append(out["std"].clone())
    torch.cuda.synchronize()
    mean_h = torch.cat(means).cpu()
    std_h = torch.cat(stds).cpu()
    acts_h, vals_h, logp_h = acts.cpu(), vals.cpu(), logp.cpu()
    res = {"E": E, "T": T, "B": B, "updates": U, "mode": mode,
           "path": "RolloutActor.step (E=%d, %d steps) -> stored log pi_old -> run_updates (B=%d, %d hipGraph-replayed "
                   "updates) vs oracle on the reference protocol (pf.explore / vf per step, target_pf forward per minibatch)"
                   % (E, T, B, U)}
    fwd = orc.FORWARDS[kind]
    ob_cpu = torch.from_numpy(obs_np)
    with torch.no_grad():
        for fl in flavours:
            o = oracles[fl]
            m_o = fwd({k: v for k, v in o.pf.items() if k != "logstd"}, ob_cpu, S, fl)
            m_o, s_o, _ = orc.gaussian(m_o, o.pf["logstd"])
            v_o = fwd(o.vf, ob_cpu, S, fl).reshape(-1)
            lp_o, _ = orc.log_prob_entropy(m_o, s_o, acts_h)
            res["rollout_mean_vs_%s" % fl] = _rel(mean_h, m_o)
            res["rollout_value_vs_%s" % fl] = _rel(vals_h, v_o)
            if fl == mode and fl != "f32" and envelope:
                res["rollout_mean_envelope_p95"] = _forward_envelope(fwd, {k: v for k, v in o.pf.items() if k != "logstd"},
                                                                     ob_cpu, S, m_o, mode=fl)
                res["rollout_value_envelope_p95"] = _forward_envelope(fwd, o.vf, ob_cpu, S, v_o, mode=fl)
            res["rollout_std_vs_%s" % fl] = _rel(std_h, s_o)
            # log pi_old of the FILED action under the oracle's own mean / std: what the reference's target forward yields
            res["rollout_logp_abs_vs_%s" % fl] = float((logp_h - lp_o.reshape(-1)).abs().max())
        # the stored log-prob is Normal(mean, std).log_prob(action) of the kernel's own mean / std
        lp_self = torch.distributions.Normal(mean_h, std_h).log_prob(acts_h).sum(-1)
        res["rollout_logp_abs_vs_own_normal"] = float((logp_h - lp_self).abs().max())

    # ---- (ii)+(iii) the updates the bench times: B-row index minibatches (whole time rows, replay_buffers/on_policy.py:73-92)
    advs = rs.randn(T * E).astype(np.float32)
    rets = rs.randn(T * E).astype(np.float32)
    rows_per = B // E
    idx = []
    while len(idx) < U:
        perm = rs.permutation(T)
        for pos in range(0, T, rows_per):
            sel = perm[pos:pos + rows_per]
            idx.append((sel[:, None] * E + np.arange(E)[None, :]).reshape(-1))
    rows = np.stack(idx[:U]).astype(np.int32)
    ro = HipTrainer.rollout(state, image, acts, torch.from_numpy(advs).to(dev), torch.from_numpy(rets).to(dev), vals, logp)
    stats = torch.zeros(U, 24, device=dev)
    agent.trainer.sync_target()
    agent.run_updates(ro, torch.from_numpy(rows).to(dev), stats)
    torch.cuda.synchronize()
    got = stats.cpu().numpy()
    res["graph_replays"] = bool(agent.use_graph)
    res["finite"] = bool(np.isfinite(got[:, :18]).all())
    keys = recipes.STAT_KEYS
    infos = {}
    for fl in flavours:
        o = oracles[fl]
        per_update = []
        for u in range(U):
            r = rows[u]
            oi = o.update(ob_cpu[r], acts_h[r], torch.from_numpy(advs[r, None]), torch.from_numpy(rets[r, None]),
                          vals_h[r, None], 1e-4, 1e-4)
            per_update.append([oi[k] for k in keys])
        infos[fl] = np.asarray(per_update)
        err = np.abs(got[:, :18] - infos[fl]) / np.maximum(1.0, np.abs(infos[fl]))
        res["infos_vs_%s_per_update" % fl] = [float("%.3e" % e) for e in err.max(axis=1)]
        res["infos_vs_%s" % fl] = float(err.max())
        res["infos_vs_%s_worst_key" % fl] = keys[int(err.max(axis=0).argmax())]
        d = [(pf.state_dict()[k].cpu() - o.pf[k]).abs() for k in o.pf] + [(vf.state_dict()[k].cpu() - o.vf[k]).abs() for k in o.vf]
        res["param_max_vs_%s" % fl] = float(max(x.max().item() for x in d))
        res["param_mean_vs_%s" % fl] = float(sum(x.sum().item() for x in d) / sum(x.numel() for x in d))
        # elements further off than half an Adam step (lr = 1e-4): a gradient element whose two evaluations differ in SIGN moves
        # its parameter by +lr in one and -lr in the other, however small the gradient (Adam normalises it) — counted, and named
        names = list(o.pf) + ["vf." + k for k in o.vf]
        res["param_frac_above_5e-5_vs_%s" % fl] = float(sum((x > 5e-5).sum().item() for x in d) / sum(x.numel() for x in d))
        res["param_worst_key_vs_%s" % fl] = names[int(np.argmax([x.max().item() for x in d]))]
    # the product's final parameters (for HIP-vs-HIP yardsticks in the tests; not a distance, not recorded)
    res["_params"] = torch.cat([pf.state_dict()[k].detach().cpu().reshape(-1) for k in oracles["f32"].pf] +
                               [vf.state_dict()[k].detach().cpu().reshape(-1) for k in oracles["f32"].vf])
    if nudged:
        env = np.zeros(U)
        for o in nudged:
            per_update = []
            for u in range(U):
                r = rows[u]
                oi = o.update(ob_cpu[r], acts_h[r], torch.from_numpy(advs[r, None]), torch.from_numpy(rets[r, None]),
                              vals_h[r, None], 1e-4, 1e-4)
                per_update.append([oi[k] for k in keys])
            err = np.abs(np.asarray(per_update) - infos[mode]) / np.maximum(1.0, np.abs(infos[mode]))
            env = np.maximum(env, err.max(axis=1))
        res["traj_envelope_per_update"] = [float("%.3e" % e) for e in env]
        res["traj_envelope_seeds"] = len(nudged)
    if mode != "f32":
        # how far the bf16 ORACLE is from the fp32 reference trajectory (recorded; the gate is the envelope above)
        err = np.abs(infos[mode] - infos["f32"]) / np.maximum(1.0, np.abs(infos["f32"]))
        res["oracle_%s_vs_f32_per_update" % mode] = [float("%.3e" % e) for e in err.max(axis=1)]
        ob, of = oracles[mode], oracles["f32"]
        d = [(ob.pf[k] - of.pf[k]).abs() for k in of.pf] + [(ob.vf[k] - of.vf[k]).abs() for k in of.vf]
        res["oracle_%s_vs_f32_param_mean" % mode] = float(sum(x.sum().item() for x in d) / sum(x.numel() for x in d))
        res["oracle_%s_vs_f32_param_max" % mode] = float(max(x.max().item() for x in d))
    return res
