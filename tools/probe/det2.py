import os, sys
os.environ["V4L_PAR"] = sys.argv[1] if len(sys.argv) > 1 else "0"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, util
os.environ["V4L_COMPUTE"]="f32"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd.engine import HipTrainer
from vision4leg_amd.torchrl.algo import PPO
np.set_printoptions(linewidth=250, precision=7)
device=torch.device("cuda:0")
case = dict(util.CASES["loco_s84"], B=32)
T, E, B = 8, 8, 32
rs = np.random.RandomState(7)
obs = np.concatenate([np.clip(rs.randn(T * E, case["S"]), -10, 10), np.clip(rs.randn(T * E, 4 * 64 * 64), -2.5, 2.8)], 1)
acts, advs, rets = 0.1 * rs.randn(T * E, case["A"]), rs.randn(T * E), rs.randn(T * E)
rows = np.stack([rs.permutation(T * E)[:B] for _ in range(4)]).astype(np.int32)
res=[]
for it in range(3):
    torch.manual_seed(case["seed"]); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(device), vf.to(device)
    class Coll: epoch_frames = T * E
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, entropy_coeff=0.005, collector=Coll(), device=device, batch_size=B)
    agent.use_graph = False
    net = pf.hip; net.ensure_bound()
    state, image = net.alloc_rollout(T * E, device)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    net.ingest(t(obs), state, image)
    ro = HipTrainer.rollout(state, image, t(acts), t(advs), t(rets), t(rets))
    stats = torch.zeros(len(rows), 24, device=device)
    agent.trainer.sync_target()
    agent.run_updates(ro, torch.tensor(rows, device=device), stats)
    torch.cuda.synchronize()
    res.append(stats.cpu().numpy().copy())
    print("run", it); print(res[-1][:, :18])
for it in (1,2):
    print("max |run%d - run0| per update:"%it, np.abs(res[it][:, :18]-res[0][:, :18]).max(axis=1))
    d=np.abs(res[it][:, :18]-res[0][:, :18]); u=np.argmax(d.max(axis=1)>1e-6); print(" first deviating update", u, "cols", np.nonzero(d[u]>1e-6)[0])
