"""Vision-only Transformer: every parameter gradient of the wave-per-sample path (17-row kernels, dummy row 0) against the
layer-by-layer path (V4L_NO_WPS_LAYERS=1) and against itself with the test taps on, per tensor.
usage: python tools/probe/vis_wps_grads.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, util
os.environ["V4L_COMPUTE"] = "bf16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
dev = torch.device("cuda:0")
n = 32
case = dict(util.CASES["loco_vis"], B=n)
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
A = case["A"]
w = torch.randn(n, A, generator=torch.Generator().manual_seed(11))
res = []
for variant in ("wps_taps", "wps", "wps", "general"):
    os.environ.pop("V4L_NO_WPS_LAYERS", None); os.environ.pop("V4L_LAYER_TAPS", None)
    if variant == "general": os.environ["V4L_NO_WPS_LAYERS"] = "1"
    if variant == "wps_taps": os.environ["V4L_LAYER_TAPS"] = "1"
    torch.manual_seed(0)
    pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
    hip = pf.hip
    st, im, _ = hip.stage(obs.to(dev))
    out = hip.forward(st, im, n, train=True)[:, :A].cpu().clone()
    dout = torch.zeros(n, 16, device=dev); dout[:, :A] = w.to(dev)
    grads = torch.full((hip.total_params,), float("nan"), device=dev)
    hip.backward(st, im, n, dout, grads)
    torch.cuda.synchronize()
    res.append({k: hip.grad_view(grads, k).cpu().clone() for k in pf.state_dict() if k != "logstd"})
for k in res[0]:
    a, b, c, g = res[0][k], res[1][k], res[2][k], res[3][k]
    print("%-55s nan %d/%d/%d  taps-vs-wps %.3e  wps-vs-wps %.3e  wps-vs-general %.3e (scale %.3e)" % (
        k, torch.isnan(a).sum(), torch.isnan(b).sum(), torch.isnan(g).sum(), (a - b).abs().max(), (b - c).abs().max(), (b - g).abs().max(), g.abs().max()))
