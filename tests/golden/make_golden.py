"""Generates tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) on CPU in the build
container, and at the same time pins the oracle (oracle/ppo_oracle.py, oracle/gae_ref.c) and the product's
seeded parameter construction against it. Run: `python tests/golden/make_golden.py` (needs /root/reference).

What a fixture holds (everything else is regenerated from seeds by tests/util.py, identically here and on
the GPU box):
  * per-parameter fingerprints of the reference's seeded initial state_dict (sum, abs-sum) — the product's
    modules must reproduce them exactly, which is what lets the fixture stay small;
  * reference forward outputs (policy mean, value) on a seeded observation batch;
  * after each of two consecutive reference `PPO.update(batch)` calls: the 18 info scalars, fingerprints and
    leading elements of every parameter's change, and the full value of all small tensors;
  * GAE: reference advantages / returns (float64, bit-exact targets) on a seeded rollout.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import util  # noqa: E402  (tests/util.py: shared case definitions)


def import_reference():
    """stub `gym` (only gym.spaces.Box is touched on this path), import the reference's
    in-tree torchrl from /root/reference."""
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box:  # noqa: D401
        pass

    spaces.Box = Box
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces
    sys.path.insert(0, REF)
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo.on_policy.ppo import PPO
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    sys.path.remove(REF)
    return networks, policies, PPO, OnPolicyReplayBuffer, Box


def build_ref_nets(networks, policies, case):
    """The wire-up of starter/ppo_{locotransformer,nature_cnn,state}.py."""
    torch.manual_seed(case["seed"])
    return util.build_nets(networks, policies, case)


def fingerprint(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item()])


def main():
    networks, policies, RefPPO, RefBuffer, Box = import_reference()
    import vision4leg_amd.torchrl.networks as my_networks
    import vision4leg_amd.torchrl.policies as my_policies
    from oracle import ppo_oracle as orc
    from oracle.gae_c import gae_c

    torch.set_num_threads(8)
    only = set(sys.argv[1:])  # optional: regenerate just these cases (e.g. `make_golden.py loco_gen`)
    for name, case in util.CASES.items():
        if only and name not in only:
            continue
        print("==", name, case)
        pf, vf = build_ref_nets(networks, policies, case)
        # ---- the product's containers must construct bit-identical parameters from the same seed
        torch.manual_seed(case["seed"])
        mpf, mvf = util.build_nets(my_networks, my_policies, case)
        for (k, a), (k2, b) in zip(pf.state_dict().items(), mpf.state_dict().items()):
            assert k == k2 and torch.equal(a, b), ("pf init mismatch", k, k2)
        for (k, a), (k2, b) in zip(vf.state_dict().items(), mvf.state_dict().items()):
            assert k == k2 and torch.equal(a, b), ("vf init mismatch", k, k2)
        out = {}
        for k, v in pf.state_dict().items():
            out["init_pf/" + k] = fingerprint(v)
        for k, v in vf.state_dict().items():
            out["init_vf/" + k] = fingerprint(v)

        # ---- forward
        batch = util.make_batch(case)
        obs = torch.tensor(batch["obs"], dtype=torch.float32)
        with torch.no_grad():
            mean, std, log_std = pf(obs)
            value = vf(obs)
        out["fwd_mean"] = mean.numpy()
        out["fwd_value"] = value.numpy()
        # oracle forward vs reference
        opf = {k: v.clone() for k, v in pf.state_dict().items()}
        ovf = util.share_encoder(opf, {k: v.clone() for k, v in vf.state_dict().items()}, case["kind"])
        with torch.no_grad():
            omean = orc.FORWARDS[case["kind"]]({k: v for k, v in opf.items() if k != "logstd"}, obs, case["S"])
            oval = orc.FORWARDS[case["kind"]](ovf, obs, case["S"])
        err_m = (omean - mean).abs().max().item() / max(mean.abs().max().item(), 1e-12)
        err_v = (oval - value).abs().max().item() / max(value.abs().max().item(), 1e-12)
        print("   oracle fwd rel err: mean %.2e value %.2e" % (err_m, err_v))
        assert err_m < 2e-5 and err_v < 2e-5

        # ---- two PPO.update calls on the reference
        class Env: action_space = Box()
        class Coll: epoch_frames = 1
        class Log:
            def add_update_info(self, info): pass
        tmp = tempfile.mkdtemp()
        agent = RefPPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, shuffle=True,
                       entropy_coeff=0.005, env=Env(), replay_buffer=None, collector=Coll(), logger=Log(),
                       device=torch.device("cpu"), discount=0.99, num_epochs=1500, batch_size=case["B"],
                       save_interval=100, eval_interval=10, save_dir=tmp,
                       clipped_value_loss=case.get("clipped_value_loss", False))
        oracle = orc.PPOOracle(case["kind"], opf, ovf, {k: v.clone() for k, v in opf.items()}, case["S"],
                               clipped_value_loss=case.get("clipped_value_loss", False))
        oracle.sync_target()
        small = util.small_param_names(pf.state_dict(), vf.state_dict())
        for u in range(2):
            before_pf = {k: v.clone() for k, v in pf.state_dict().items()}
            before_vf = {k: v.clone() for k, v in vf.state_dict().items()}
            b = util.make_batch(case, update=u)
            info = agent.update({k: b[k] for k in ("obs", "acts", "advs", "estimate_returns", "values")})
            t = lambda a: torch.tensor(a, dtype=torch.float32)
            oinfo = oracle.update(t(b["obs"]), t(b["acts"]), t(b["advs"]), t(b["estimate_returns"]), t(b["values"]),
                                  1e-4, 1e-4)
            out["u%d/info" % u] = np.array([info[k] for k in util.STAT_KEYS])
            for k in util.STAT_KEYS:
                assert abs(info[k] - oinfo[k]) <= 2e-4 * max(1.0, abs(info[k])), (k, info[k], oinfo[k])
            for tag, net, before, onet in (("pf", pf, before_pf, opf), ("vf", vf, before_vf, ovf)):
                for k, v in net.state_dict().items():
                    d = (v - before[k]).double()
                    out["u%d/d%s/%s" % (u, tag, k)] = np.concatenate(
                        [[d.sum().item(), d.abs().sum().item()], d.flatten()[:32].numpy()])
                    if (tag, k) in small:
                        out["u%d/%s/%s" % (u, tag, k)] = v.numpy().copy()
                    # oracle tracks the reference through the update (Adam amplifies tiny grad noise: loose atol)
                    assert torch.allclose(onet[k], v, rtol=0, atol=2e-5), (tag, k, (onet[k] - v).abs().max())
                # clipped grads the reference leaves in .grad
                for k, prm in net.named_parameters():
                    if prm.grad is not None:
                        g = prm.grad.double()
                        out["u%d/g%s/%s" % (u, tag, k)] = np.concatenate(
                            [[g.sum().item(), g.abs().sum().item(), g.norm().item()], g.flatten()[:32].numpy()])
            print("   update %d: ratio max %.4f vf_loss %.4f gn_pf %.4f" %
                  (u, info["ratio/max"], info["Training/vf_loss"], info["grad_norm/pf"]))
        np.savez_compressed(os.path.join(HERE, "ppo_%s.npz" % name), **out)

    if only and not (only & set(util.GAE_CASES) or "gae" in only):
        return
    # ---- GAE
    gout = {}
    for name, g in util.GAE_CASES.items():
        ro = util.make_gae_inputs(g)
        buf = RefBuffer(max_replay_buffer_size=g["T"] * g["E"], env_nums=g["E"], time_limit_filter=g["tl_filter"])
        for t in range(g["T"]):
            buf.add_sample({"rewards": ro["rewards"][t], "values": ro["values"][t], "terminals": ro["terminals"][t],
                            "time_limits": ro["time_limits"][t]})
        buf.generalized_advantage_estimation(ro["last_value"], g["gamma"], g["tau"])
        advs, rets = buf._advs, buf._estimate_returns
        oa, orr = orc.gae(ro["rewards"], ro["values"], ro["terminals"], ro["time_limits"], ro["last_value"],
                          g["gamma"], g["tau"], g["tl_filter"])
        assert np.array_equal(oa, advs) and np.array_equal(orr, rets), name
        T, E = g["T"], g["E"]
        tl = ro["time_limits"].reshape(T, -1)
        ca, cr = gae_c(ro["rewards"].reshape(T, E), ro["values"].reshape(T, E), ro["terminals"].reshape(T, E),
                       tl.reshape(T) if tl.shape[1] == 1 and E > 1 else tl, ro["last_value"], g["gamma"], g["tau"],
                       g["tl_filter"])
        assert np.array_equal(ca, advs.reshape(T, E)) and np.array_equal(cr, rets.reshape(T, E)), name
        gout[name + "/advs"] = advs
        gout[name + "/rets"] = rets
        print("== gae", name, "bit-exact (numpy oracle, C oracle)")
        # PPO(gae=False): discount_reward on the same rollout (replay_buffers/on_policy.py:47-71)
        buf.discount_reward(ro["last_value"], g["gamma"])
        dadvs, drets = buf._advs, buf._estimate_returns
        oa, orr = orc.discount_reward(ro["rewards"], ro["values"], ro["terminals"], ro["time_limits"], ro["last_value"],
                                      g["gamma"], g["tl_filter"])
        assert np.array_equal(oa, dadvs) and np.array_equal(orr, drets), name
        ca, cr = gae_c(ro["rewards"].reshape(T, E), ro["values"].reshape(T, E), ro["terminals"].reshape(T, E),
                       tl.reshape(T) if tl.shape[1] == 1 and E > 1 else tl, ro["last_value"], g["gamma"], None, g["tl_filter"])
        assert np.array_equal(ca, dadvs.reshape(T, E)) and np.array_equal(cr, drets.reshape(T, E)), name
        gout[name + "/dr_advs"] = dadvs
        gout[name + "/dr_rets"] = drets
        print("== discount_reward", name, "bit-exact (numpy oracle, C oracle)")
    np.savez_compressed(os.path.join(HERE, "gae.npz"), **gout)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
