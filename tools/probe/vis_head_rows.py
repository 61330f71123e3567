"""Vision-only Transformer: the heads' saved activations and data-grad rows of the 17-row wave-per-sample path against the
layer-by-layer path, row by row (which rows of hh0 / hh1 / dhh0 / dhh1 / pooled differ), in a fresh process per mode.
usage: python tools/probe/vis_head_rows.py [mode] [n]"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, util
mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
os.environ["V4L_COMPUTE"] = mode
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
dev = torch.device("cuda:0")
case = dict(util.CASES["loco_vis"], B=n)
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
A = case["A"]
w = torch.randn(n, A, generator=torch.Generator().manual_seed(11))
NAMES = [("hh0", 256), ("hh1", 256), ("dhh0", 256), ("dhh1", 256), ("dout", 16), ("out", 16), ("dpool", 128)]
res = {}
for variant in ("wps17", "general", "wps"):
    for k in ("V4L_NO_WPS_LAYERS", "V4L_LAYER_TAPS", "V4L_VIS17"):
        os.environ.pop(k, None)
    if variant == "general": os.environ["V4L_NO_WPS_LAYERS"] = "1"
    if variant == "wps17": os.environ["V4L_VIS17"] = "1"
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
    hip = pf.hip
    st, im, _ = hip.stage(obs.to(dev))
    hip.workspace(n).fill_(float("nan"))   # an element the path reads without having written it shows
    hip.forward(st, im, n, train=True)
    dout = torch.zeros(n, 16, device=dev); dout[:, :A] = w.to(dev)
    grads = torch.full((hip.total_params,), float("nan"), device=dev)
    hip.backward(st, im, n, dout, grads, scale=None if mode != "f16" else 2048.0)
    torch.cuda.synchronize()
    d = {}
    for nm, cols in NAMES:
        try:
            d[nm] = hip.ws_view(n, nm, n, cols).cpu().clone()
        except KeyError:
            pass
    d["g.b0"] = hip.grad_view(grads, "visual_seq_append_fcs.0.bias").cpu().clone()
    d["g.w0"] = hip.grad_view(grads, "visual_seq_append_fcs.0.weight").cpu().clone()
    res[variant] = d
for other in ("wps17", "wps"):
    print("== %s vs general (%s, n = %d)" % (other, mode, n))
    for nm in res["general"]:
        if nm not in res[other]:
            continue
        a, b = res[other][nm], res["general"][nm]
        if nm == "dhh0" or nm == "dhh1" or nm == "dout" or nm == "dpool":
            pass
        diff = (a - b).abs()
        nan_a, nan_b = int(torch.isnan(a).sum()), int(torch.isnan(b).sum())
        diff = torch.nan_to_num(diff, nan=float("inf"))
        rows = (diff.reshape(diff.shape[0], -1).max(1).values > 1e-6 * max(1e-30, float(torch.nan_to_num(b).abs().max()))).nonzero().flatten().tolist()
        print("   %-6s max|b| %.3e  max diff %.3e  nan %d/%d  rows differing: %s" % (
            nm, float(torch.nan_to_num(b).abs().max()), float(diff.max()), nan_a, nan_b, rows[:20]))
