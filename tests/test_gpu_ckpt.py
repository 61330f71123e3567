"""-m gpu: checkpoint cross-load. The fixtures under tests/golden/ckpt/ are `.pth` files the UNMODIFIED
reference wrote with its own `RLAlgo.snapshot` (torchrl/algo/rl_algo.py:84-95) after two reference `PPO.update` calls
(tests/golden/make_golden_ckpt.py) — parameters no seeded construction here reproduces. They go into the HIP modules the way
the reference's viewers load them (starter/locotransformer_viewer.py:125-147: `pf.load_state_dict(torch.load(PATH,
map_location=...))`), and the forward must give what the reference's classes computed from the same files. The reverse
direction (a HIP-trained `PPO.snapshot` file into the reference classes) runs where the reference tree is
(tests/test_overlay_cpu.py); here the HIP-written file is read back with plain torch.load and checked by the oracle."""
import os

import numpy as np
import pytest
import torch

import util
from oracle import ppo_oracle as orc

pytestmark = pytest.mark.gpu

CKPT = os.path.join(util.GOLDEN, "ckpt")
# f32: the north star's gate with margin; bf16: the fixed fp32-reference distance of tests/test_gpu_parity.py (bf16 operands
# against an fp32 computation: ~4e-3 by construction)
TOL = {"f32": 2e-4, "bf16": 2e-2, "f16": 4e-3}


def _build(case, mode, dev, seed):
    os.environ["V4L_COMPUTE"] = mode
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    torch.manual_seed(seed)
    pf, vf = util.build_nets(networks, policies, case)
    return pf.to(dev), vf.to(dev)


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["loco_s84", "mlp_s93"])
def test_reference_checkpoint_loads_into_hip_modules(name, mode, device):
    case = util.CASES[name]
    gold = np.load(os.path.join(util.GOLDEN, "ckpt_%s.npz" % name))
    pf, vf = _build(case, mode, device, seed=case["seed"] + 100)   # NOT the seed the checkpoint started from
    assert list(pf.state_dict().keys()) == list(gold["pf_keys"]) and list(vf.state_dict().keys()) == list(gold["vf_keys"])
    obs = torch.tensor(util.make_batch(case, update=5)["obs"], dtype=torch.float32).to(device)
    # a forward BEFORE the load: the operand-type weight packs of the unrelated seeded parameters exist and must be replaced
    m0, _, _ = pf(obs)
    v0 = vf(obs)
    assert util.rel_err(m0.cpu(), gold["fwd_mean"]) > 0.05 and util.rel_err(v0.cpu(), gold["fwd_value"]) > 0.05
    from vision4leg_amd.torchrl.policies import RolloutActor
    E = 8
    actor = RolloutActor(pf, vf, E)
    a0 = np.array(actor.eval_act(obs[:E]))          # per-step caller: keeps its own view of "parameters unchanged"
    # the viewer's two lines, critic first: with the shared encoder (ppo_locotransformer.py:94-96; ppo_state.py:104 `vf.base =
    # pf.base`) the policy file's encoder tensors are the ones that stay — in the reference as here
    vf.load_state_dict(torch.load(os.path.join(CKPT, name, "model_vf_2.pth"), map_location=device))
    missing = pf.load_state_dict(torch.load(os.path.join(CKPT, name, "model_pf_2.pth"), map_location=device))
    assert not missing.missing_keys and not missing.unexpected_keys
    mean, std, log_std = pf(obs)
    value = vf(obs)
    em, es = util.rel_err(mean.cpu(), gold["fwd_mean"]), util.rel_err(std.cpu(), gold["fwd_std"])
    util.record("ckpt/%s/%s/mean_vs_reference" % (name, mode), em)
    assert em <= TOL[mode] and es <= 1e-6, (em, es)
    # the critic file carries the encoder as the CRITIC step left it; the policy file's (loaded second) overwrote it, exactly as
    # in the reference, whose `fwd_value` golden was computed on the live shared objects = the policy file's encoder state
    ev = util.rel_err(value.cpu(), gold["fwd_value"])
    util.record("ckpt/%s/%s/value_vs_reference" % (name, mode), ev)
    assert ev <= TOL[mode], ev
    # the per-step actor notices the in-place load through the parameters' version counters (engine.ensure_bound(fast=True))
    a1 = np.array(actor.eval_act(obs[:E]))
    assert util.rel_err(a1, gold["fwd_mean"][:E]) <= TOL[mode] and util.rel_err(a0, gold["fwd_mean"][:E]) > 0.05


@pytest.mark.parametrize("name", ["loco_s84", "mlp_s93"])
def test_hip_snapshot_file_round_trip(name, device, tmp_path):
    """HIP-trained `PPO.snapshot` -> the reference's file names, the reference's keys in the reference's order, plain tensors;
    the oracle (CPU, fp32) fed the FILE reproduces what the HIP modules compute from their live parameters."""
    from vision4leg_amd.torchrl.algo import PPO
    case = util.CASES[name]
    gold = np.load(os.path.join(util.GOLDEN, "ckpt_%s.npz" % name))
    pf, vf = _build(case, "f32", device, seed=case["seed"])

    class Coll: epoch_frames = 1
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, entropy_coeff=0.005,
                collector=Coll(), device=device, batch_size=case["B"])
    agent.trainer.sync_target()
    for u in range(2):
        b = util.make_batch(case, update=u)
        agent.update({k: b[k] for k in ("obs", "acts", "advs", "estimate_returns", "values")})
    agent.snapshot(str(tmp_path), 2)
    files = sorted(os.listdir(tmp_path))
    assert files == ["model_pf_2.pth", "model_vf_2.pth"], files
    sd_pf = torch.load(os.path.join(tmp_path, "model_pf_2.pth"), map_location="cpu")
    sd_vf = torch.load(os.path.join(tmp_path, "model_vf_2.pth"), map_location="cpu")
    assert list(sd_pf.keys()) == list(gold["pf_keys"]) and list(sd_vf.keys()) == list(gold["vf_keys"])
    assert all(type(v) is torch.Tensor and v.dtype == torch.float32 for v in list(sd_pf.values()) + list(sd_vf.values()))
    # same two updates from the same seed: the HIP-trained file is the reference-trained file up to fp32 summation order
    ref_pf = torch.load(os.path.join(CKPT, name, "model_pf_2.pth"), map_location="cpu")
    worst = max((sd_pf[k] - ref_pf[k]).abs().max().item() for k in ref_pf)
    util.record("ckpt/%s/f32/hip_file_vs_reference_file_max_abs" % name, worst)
    assert worst <= case.get("param_tol_f32", 2e-5), worst
    obs = torch.tensor(util.make_batch(case, update=5)["obs"], dtype=torch.float32)
    with torch.no_grad():
        om = orc.FORWARDS[case["kind"]]({k: v for k, v in sd_pf.items() if k != "logstd"}, obs, case["S"], "f32")
    mean, _, _ = pf(obs.to(device))
    value = vf(obs.to(device))
    assert util.rel_err(mean.cpu(), om) <= 2e-5
    # leave the HIP-written files + what the HIP modules compute from them in gpurun_out/: committed as tests/golden/ckpt_hip/<name>/
    # they are what tests/test_overlay_cpu.py::test_hip_written_checkpoint_loads_into_reference_classes feeds the reference classes
    out = os.path.join(util.ROOT, "gpurun_out", "ckpt_hip", name)
    os.makedirs(out, exist_ok=True)
    for f in files:
        with open(os.path.join(tmp_path, f), "rb") as src, open(os.path.join(out, f), "wb") as dst:
            dst.write(src.read())
    np.savez_compressed(os.path.join(out, "hip_fwd.npz"), fwd_mean=mean.cpu().numpy(), fwd_value=value.cpu().numpy())
