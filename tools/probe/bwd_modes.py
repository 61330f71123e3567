"""Is the two-valued launch time of wps_layer_bwd_kernel (55 / 67 us policy pass, 65 / 76 us critic pass, fixed within a process,
different between processes) a property of where the DATA lies or of where the CODE lies? K trainers with their own nets,
workspaces and rollouts in ONE process (same code addresses, different data addresses), U graph-replayed updates each;
run under rocprofv3 --kernel-trace and feed the trace to tools/probe/bwd_modes_report.py."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from vision4leg_amd.torchrl.algo import PPO
import vision4leg_amd.torchrl.networks as networks
import vision4leg_amd.torchrl.policies as policies

K, U = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 40
dev = torch.device("cuda:0")
case = util.CASES["loco_b1024"]
keep = []
for k in range(K):
    torch.manual_seed(k)
    pf, vf = util.build_nets(networks, policies, case)
    pf, vf = pf.to(dev), vf.to(dev)
    class Coll: epoch_frames = 1
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, entropy_coeff=0.005,
                collector=Coll(), device=dev, batch_size=case["B"])
    agent.trainer.sync_target()
    b = util.make_batch(case, update=k)
    batch = {x: b[x] for x in ("obs", "acts", "advs", "estimate_returns", "values")}
    for u in range(U):
        agent.update(batch)
    torch.cuda.synchronize()
    keep.append((agent, torch.empty((k + 1) * 3 * 1024 * 1024 + 4096 * k, dtype=torch.uint8, device=dev)))  # shift later allocations
    print("agent", k, "done", flush=True)
