"""Checkpoint cross-load fixtures: `.pth` files written by the UNMODIFIED reference's own
`RLAlgo.snapshot` (torchrl/algo/rl_algo.py:84-95: `torch.save(network.state_dict(), model_{pf,vf}_{epoch}.pth)`) after two
reference `PPO.update` calls on seeded nets — i.e. parameters no seeded construction of ours reproduces — plus what the
reference's classes compute from them on a seeded observation batch. Run: `python tests/golden/make_golden_ckpt.py`
(needs /root/reference; imports it exactly like make_golden.py).

  tests/golden/ckpt/<case>/model_pf_2.pth, model_vf_2.pth   the reference's files, byte for byte
  tests/golden/ckpt_<case>.npz                              keys (in the reference's state_dict order), forward outputs

The loaders this pins: starter/locotransformer_viewer.py:125-147 (`pf.load_state_dict(torch.load(PATH, map_location=...))`).
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (also puts the repo root and tests/ on sys.path)
import util  # noqa: E402

CKPT_CASES = ("loco_s84", "mlp_s93")


def main():
    networks, policies, RefPPO, RefBuffer, Box = mg.import_reference()
    torch.set_num_threads(8)
    for name in CKPT_CASES:
        case = util.CASES[name]
        pf, vf = mg.build_ref_nets(networks, policies, case)

        class Env: action_space = Box()
        class Coll: epoch_frames = 1
        class Log:
            def add_update_info(self, info): pass
        tmp = tempfile.mkdtemp()
        agent = RefPPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, shuffle=True,
                       entropy_coeff=0.005, env=Env(), replay_buffer=None, collector=Coll(), logger=Log(),
                       device=torch.device("cpu"), discount=0.99, num_epochs=1500, batch_size=case["B"],
                       save_interval=100, eval_interval=10, save_dir=tmp)
        for u in range(2):
            b = util.make_batch(case, update=u)
            agent.update({k: b[k] for k in ("obs", "acts", "advs", "estimate_returns", "values")})
        agent.snapshot(tmp, 2)  # the reference's own writer: model_pf_2.pth, model_vf_2.pth
        dst = os.path.join(HERE, "ckpt", name)
        os.makedirs(dst, exist_ok=True)
        for f in ("model_pf_2.pth", "model_vf_2.pth"):
            shutil.copyfile(os.path.join(tmp, f), os.path.join(dst, f))
        obs = torch.tensor(util.make_batch(case, update=5)["obs"], dtype=torch.float32)
        with torch.no_grad():
            mean, std, log_std = pf(obs)
            value = vf(obs)
        out = {"fwd_mean": mean.numpy(), "fwd_std": std.numpy(), "fwd_value": value.numpy(),
               "pf_keys": np.array(list(pf.state_dict().keys())), "vf_keys": np.array(list(vf.state_dict().keys()))}
        # the files really hold the trained parameters (not the seeded ones)
        torch.manual_seed(case["seed"])
        pf0, _ = util.build_nets(networks, policies, case)
        moved = sum(int(not torch.equal(a, b)) for a, b in zip(pf.state_dict().values(), pf0.state_dict().values()))
        assert moved >= len(pf.state_dict()) - 1, moved
        np.savez_compressed(os.path.join(HERE, "ckpt_%s.npz" % name), **out)
        print("==", name, "pf/vf checkpoints + forward goldens written;", moved, "of", len(pf.state_dict()), "pf tensors moved")


if __name__ == "__main__":
    main()
