"""dW3 (conv3 weight gradient) of the fused conv backward for different block counts of its launch (V4L_CONV3_WGRAD_BLOCKS):
same sums in a different order — differences must be fp32 summation noise.  usage: python tools/probe/dw3_blocks.py [f32|bf16] [n]"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, util
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
os.environ["V4L_COMPUTE"] = mode
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
dev = torch.device("cuda:0")
case = dict(util.CASES["loco_s93"], B=n)
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
w = torch.randn(n, 1, generator=torch.Generator().manual_seed(3))
res = {}
for blocks in (64, 96, 128, 256):
    os.environ["V4L_CONV3_WGRAD_BLOCKS"] = str(blocks)
    torch.manual_seed(0)
    pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
    hip = vf.hip
    st, im, _ = hip.stage(obs.to(dev))
    hip.forward(st, im, n, train=True)
    dout = torch.zeros(n, 16, device=dev); dout[:, :1] = w.to(dev)
    grads = torch.full((hip.total_params,), float("nan"), device=dev)
    hip.backward(st, im, n, dout, grads)
    torch.cuda.synchronize()
    res[blocks] = {k: hip.grad_view(grads, k).cpu().clone() for k in vf.state_dict()}
ref = res[64]
for blocks in (96, 128, 256):
    worst = max(((res[blocks][k] - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-30), k) for k in ref)
    k3 = "encoder.depth_visual_base.layers.4.weight"
    d3 = (res[blocks][k3] - ref[k3]).abs().max().item() / ref[k3].abs().max().item()
    print("%s n=%d blocks=%d vs 64: worst rel diff %.3e (%s); dW3 %.3e; nan %d" % (mode, n, blocks, worst[0], worst[1], d3,
          sum(int(torch.isnan(v).sum()) for v in res[blocks].values())))
