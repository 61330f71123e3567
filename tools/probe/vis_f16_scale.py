"""Vision-only Transformer, compute=f16: every parameter gradient of the four update paths (17-row wave-per-sample kernels with
test taps, layer-by-layer kernels, 17-row untapped, native 16-token) against the f16 oracle and the fp32 oracle, for a ladder of
loss-gradient scales (which path leaves half's range first, and at which scale).
usage: python tools/probe/vis_f16_scale.py [n]"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, util
from oracle import ppo_oracle as orc
os.environ["V4L_COMPUTE"] = "f16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
case = dict(util.CASES["loco_vis"], B=n)
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
A = case["A"]
w = torch.randn(n, A, generator=torch.Generator().manual_seed(11))
VARS = ("wps_taps", "general", "wps17", "wps")


def hip_grads(variant, scale):
    for k in ("V4L_NO_WPS_LAYERS", "V4L_LAYER_TAPS", "V4L_VIS17"):
        os.environ.pop(k, None)
    if variant == "general": os.environ["V4L_NO_WPS_LAYERS"] = "1"
    if variant == "wps_taps": os.environ["V4L_LAYER_TAPS"] = "1"
    if variant == "wps17": os.environ["V4L_VIS17"] = "1"
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
    hip = pf.hip
    st, im, _ = hip.stage(obs.to(dev))
    hip.forward(st, im, n, train=True)
    dout = torch.zeros(n, 16, device=dev); dout[:, :A] = w.to(dev)
    grads = torch.full((hip.total_params,), float("nan"), device=dev)
    hip.backward(st, im, n, dout, grads, scale=scale)
    torch.cuda.synchronize()
    return pf, {k: hip.grad_view(grads, k).cpu().clone() for k in pf.state_dict() if k != "logstd"}, hip.last_grad_scale


def oracle_grads(pf, mode, scale):
    op = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in pf.state_dict().items() if k != "logstd"}
    out = orc.FORWARDS[case["kind"]](op, obs, case["S"], mode)
    keys = list(op)
    g = torch.autograd.grad((out * w).sum() * scale, [op[k] for k in keys], allow_unused=True)
    return {k: (torch.zeros_like(op[k]) if x is None else x / scale) for k, x in zip(keys, g)}


flat = lambda d, keys: torch.cat([d[k].flatten().double() for k in keys])
for scale in (None, 1.0, 64.0, 512.0, 2048.0, 16384.0):
    res = {}
    for v in VARS:
        pf, res[v], used = hip_grads(v, scale)
    o16, o32 = oracle_grads(pf, "f16", used), oracle_grads(pf, "f32", 1.0)
    keys = list(o32)
    print("== scale %s (used %g): flat rel L2 vs f16 oracle / vs f32 oracle" % (scale, used))
    for v in VARS:
        a = flat(res[v], keys)
        print("   %-9s %.3e / %.3e   nan %d" % (v, float((a - flat(o16, keys)).norm() / flat(o16, keys).norm()),
                                                 float((a - flat(o32, keys)).norm() / flat(o32, keys).norm()), int(torch.isnan(a).sum())))
    print("   oracle16 vs oracle32 %.3e" % float((flat(o16, keys) - flat(o32, keys)).norm() / flat(o32, keys).norm()))
    worst = sorted(keys, key=lambda k: -util.rel_err(res["wps_taps"][k], res["general"][k]))[:4]
    for k in worst:
        print("   %-50s taps-vs-general %.3e  taps-vs-o16 %.3e  general-vs-o16 %.3e  native-vs-o16 %.3e" % (
            k, util.rel_err(res["wps_taps"][k], res["general"][k]), util.rel_err(res["wps_taps"][k], o16[k]),
            util.rel_err(res["general"][k], o16[k]), util.rel_err(res["wps"][k], o16[k])))
