"""Layer weight-grads of the fused backward vs dY^T X recomputed (fp64, torch on the GPU) from the tensors the kernels saved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
dev = torch.device("cuda:0")
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
def run(name, mode, reps=3):
    os.environ["V4L_COMPUTE"] = mode
    case = util.CASES[name]; n = case["B"]; R = n * 17
    torch.manual_seed(case["seed"]); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
    obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32, device=dev)
    tdt = torch.float32 if mode == "f32" else torch.bfloat16
    for tag, net, A in (("pf", pf, 6), ("vf", vf, 1)):
        w = torch.tensor(np.random.RandomState(5).randn(n, A), dtype=torch.float32, device=dev)
        hip = net.hip
        st, im, _ = hip.stage(obs)
        for rep in range(reps):
            hip.forward(st, im, n, train=True)
            dout = torch.zeros(n, 16, device=dev); dout[:, :A] = w
            grads = torch.full((hip.total_params,), float("nan"), device=dev)
            hip.backward(st, im, n, dout, grads)
            torch.cuda.synchronize()
            ws = hip.workspace(n)
            def tap(nm, cols):
                off = hip.ws_offset(n, nm)
                raw = ws[off:off + R * cols]
                if tdt == torch.bfloat16:
                    return raw.view(torch.bfloat16)[:R * cols].view(R, cols).double()
                return raw.view(R, cols).double()
            for l in range(2):
                pre = "visual_append_layers.%d." % l
                checks = [("linear1", tap("df%d" % l, 256), tap("mid%d" % l, 64)),
                          ("linear2", tap("dz2_%d" % l, 64), tap("ff%d" % l, 256)),
                          ("self_attn.out_proj", tap("dz1_%d" % l, 64), tap("ctx%d" % l, 64)),
                          ("self_attn.in_proj", tap("dqkv%d" % l, 192), tap("xin%d" % l, 64))]
                for nm, dy, x in checks:
                    wn = pre + nm + ("_weight" if "in_proj" in nm else ".weight")
                    bn = pre + nm + ("_bias" if "in_proj" in nm else ".bias")
                    gw, gb = hip.grad_view(grads, wn).double(), hip.grad_view(grads, bn).double()
                    rw, rb = dy.t() @ x, dy.sum(0)
                    ew = ((gw - rw).abs().max() / rw.abs().max()).item(); eb = ((gb - rb).abs().max() / rb.abs().max()).item()
                    flag = "  <<<<" if max(ew, eb) > 2e-5 else ""
                    if flag or rep == 0 and l == 0 and nm == "linear1":
                        print("%-10s %-5s %s rep%d %-28s dW rel %.2e db rel %.2e%s" % (name, mode, tag, rep, pre + nm, ew, eb, flag))
for name, mode in [("loco_s93", "f32"), ("loco_b1024", "bf16"), ("loco_b1024", "f32"), ("loco_rag", "bf16"), ("loco_rag", "f32"), ("loco_b1024", "f32")]:
    run(name, mode)
print("done")
