import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
os.environ["V4L_COMPUTE"] = "f32"
dev = torch.device("cuda:0")
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
case = util.CASES["loco_rag"]; n = case["B"]; R = n * 17
torch.manual_seed(case["seed"]); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32, device=dev)
hip = pf.hip
st, im, _ = hip.stage(obs)
hip.forward(st, im, n, train=True)
torch.cuda.synchronize()
ws = hip.workspace(n)
def tap(nm, cols):
    off = hip.ws_offset(n, nm); return ws[off:off + R * cols].view(R, cols).double()
sd = pf.state_dict()
for l in range(2):
    x1, f = tap("mid%d" % l, 64), tap("ff%d" % l, 256)
    W, b = sd["visual_append_layers.%d.linear1.weight" % l].double(), sd["visual_append_layers.%d.linear1.bias" % l].double()
    pre = x1 @ W.t() + b
    mism = (pre > 0) != (f > 0)
    idx = torch.nonzero(mism)
    print("layer %d: ReLU mask mismatches between HIP's saved f and a float64 recomputation from HIP's x1: %d of %d" % (l, idx.shape[0], pre.numel()))
    for r, c in idx[:10].tolist():
        print("   row %d col %d: fp64 pre-activation %.3e, HIP f %.3e" % (r, c, pre[r, c].item(), f[r, c].item()))
    print("   max |relu(pre) - f| = %.3e" % (torch.relu(pre) - f).abs().max().item())
# ---- against the float64 ORACLE forward (not HIP's own x1): which ReLU decisions differ, and by what margin
import math
import torch.nn.functional as F
from oracle import ppo_oracle as orc
p64 = {k: v.detach().cpu().double() for k, v in pf.state_dict().items() if k != "logstd"}
taps = {}
with torch.no_grad():
    orc.loco_forward(p64, obs.cpu().double(), case["S"], "f32", taps)
    x = taps["x0"]
    for l in range(2):
        pre_ = "visual_append_layers.%d" % l
        d = 64
        qkv = F.linear(x, p64[pre_ + ".self_attn.in_proj_weight"], p64[pre_ + ".self_attn.in_proj_bias"])
        q, k, v = qkv.split(d, dim=-1)
        ctx = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1) @ v
        a = F.linear(ctx, p64[pre_ + ".self_attn.out_proj.weight"], p64[pre_ + ".self_attn.out_proj.bias"])
        x1 = F.layer_norm(x + a, (d,), p64[pre_ + ".norm1.weight"], p64[pre_ + ".norm1.bias"], 1e-5)
        pre = F.linear(x1, p64[pre_ + ".linear1.weight"], p64[pre_ + ".linear1.bias"]).reshape(R, 256)
        f_hip = tap("ff%d" % l, 256).cpu()
        mism = (pre > 0) != (f_hip > 0)
        idx = torch.nonzero(mism)
        print("layer %d vs float64 oracle: %d ReLU decisions differ; |pre-activation| there: %s" %
              (l, idx.shape[0], ["%.2e" % pre[r, c].abs().item() for r, c in idx[:8].tolist()]))
        print("   max |x1_hip - x1_oracle64| = %.2e" % (tap("mid%d" % l, 64).cpu() - x1.reshape(R, 64)).abs().max().item())
        x = taps["x%d" % (l + 1)]
