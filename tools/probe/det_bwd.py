"""Run the same forward + backward several times; report which gradient tensors / saved gradient tensors differ between runs."""
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
w = torch.tensor(np.random.RandomState(5).randn(n, 6), dtype=torch.float32, device=dev)
hip = pf.hip
st, im, _ = hip.stage(obs)
runs, taps = [], []
names = ["dz2_0", "df0", "dz1_0", "dqkv0", "dx0", "dz2_1", "df1", "dz1_1", "dqkv1", "dx1", "dx2"]
tfwd = ["x1", "x2", "qkv0", "qkv1", "ff0", "ff1", "mid0", "mid1"]
for it in range(6):
    hip.forward(st, im, n, train=True)
    dout = torch.zeros(n, 16, device=dev); dout[:, :6] = w
    grads = torch.full((hip.total_params,), float("nan"), device=dev)
    hip.backward(st, im, n, dout, grads)
    torch.cuda.synchronize()
    runs.append({k: hip.grad_view(grads, k).clone() for k in hip.param_names})
    d = {}
    for nm in names + tfwd:
        off = hip.ws_offset(n, nm)
        d[nm] = hip.workspace(n)[off:off + n * 17 * 64].clone()   # a prefix is enough to see a difference
    taps.append(d)
bad = False
for k in runs[0]:
    diffs = [(runs[i][k] - runs[0][k]).abs().max().item() for i in range(1, 6)]
    if max(diffs) > 0:
        bad = True
        print("GRAD differs between runs: %-50s max|d| %s  (max|g| %.3e)" % (k, ["%.2e" % x for x in diffs], runs[0][k].abs().max().item()))
for nm in names + tfwd:
    diffs = [(taps[i][nm] - taps[0][nm]).abs().max().item() for i in range(1, 6)]
    if max(diffs) > 0 or any(torch.isnan(taps[i][nm]).any().item() for i in range(6)):
        print("TAP  differs between runs: %-10s %s" % (nm, ["%.2e" % x for x in diffs]))
        i = int(np.argmax(diffs)) + 1
        idx = torch.nonzero((taps[i][nm] - taps[0][nm]).abs() > 0).flatten()
        print("     first differing flat indices:", idx[:12].tolist(), "count", idx.numel())
print("deterministic" if not bad else "NON-DETERMINISTIC")
