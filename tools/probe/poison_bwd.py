"""Poison the workspace (NaN / large values) before forward + backward: a kernel that reads bytes nobody wrote shows up as NaN
or as a changed gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
name = sys.argv[2] if len(sys.argv) > 2 else "loco_rag"
os.environ["V4L_COMPUTE"] = mode
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
dev = torch.device("cuda:0")
case = util.CASES[name]; n = case["B"]
torch.manual_seed(case["seed"]); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32, device=dev)
for tag, net, A in (("pf", pf, 6), ("vf", vf, 1)):
    w = torch.tensor(np.random.RandomState(5).randn(n, A), dtype=torch.float32, device=dev)
    hip = net.hip
    st, im, _ = hip.stage(obs)
    res = []
    for fill in (0.0, float("nan"), 1e3, -7.0):
        ws = hip.workspace(n)
        ws.fill_(fill)
        hip.forward(st, im, n, train=True)
        dout = torch.zeros(n, 16, device=dev); dout[:, :A] = w
        grads = torch.full((hip.total_params,), float("nan"), device=dev)
        hip.backward(st, im, n, dout, grads)
        torch.cuda.synchronize()
        res.append({k: hip.grad_view(grads, k).clone() for k in hip.param_names if k != "logstd"})
    for k in res[0]:
        d = [(res[i][k] - res[0][k]).abs().max().item() for i in range(1, 4)]
        nan = [torch.isnan(res[i][k]).any().item() for i in range(4)]
        if max(x if x == x else 1e9 for x in d) > 0 or any(nan):
            print("%s %-52s depends on workspace garbage: max|d| vs zero-fill %s nan %s (max|g| %.2e)"
                  % (tag, k, ["%.2e" % x for x in d], nan, res[0][k].abs().max().item()))
print("done")
