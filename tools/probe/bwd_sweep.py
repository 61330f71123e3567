"""Per-kernel HIP-event times of one train forward + backward of the value net at several minibatch sizes (fixed vs
per-sample cost of every launch). Run on the GPU box: python tools/probe/bwd_sweep.py [n ...]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, util
os.environ.setdefault("V4L_COMPUTE", "bf16")
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib

dev = torch.device("cuda:0")
ns = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]
case = dict(util.CASES["loco_s93"], B=max(ns))
torch.manual_seed(0)
pf, vf = util.build_nets(networks, policies, case)
vf = vf.to(dev)
hip = vf.hip
L = _lib.lib()
res = {}
for n in ns:
    obs = torch.randn(n, 93 + 16384, device=dev)
    st_, im, _ = hip.stage(obs)
    dout = torch.randn(n, 16, device=dev)
    grads = torch.zeros(hip.total_params, device=dev)
    def step():
        hip.forward(st_, im, n, train=True)
        hip.backward(st_, im, n, dout, grads)
    for _ in range(3): step()
    torch.cuda.synchronize()
    L.v4l_prof_enable(1)
    for _ in range(5): step()
    buf = C.create_string_buffer(1 << 20)
    L.v4l_prof_collect(buf, len(buf))
    L.v4l_prof_enable(0)
    for line in buf.value.decode().splitlines():
        label, calls, us, flops = line.split("\t")
        k = label.split("|")[-1]
        c0, u0 = res.setdefault(k, {}).get(n, (0, 0.0))
        res[k][n] = (c0 + int(calls), u0 + float(us))
print("%-28s" % "kernel (avg us per launch)" + "".join("%10d" % n for n in ns))
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get(ns[-1], (1, 0))[1]):
    print("%-28s" % k + "".join("%10.1f" % (d[n][1] / d[n][0]) if n in d else "%10s" % "-" for n in ns))
