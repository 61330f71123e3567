"""Does the launch-time mode of wps_layer_bwd_kernel (DESIGN.md 4.4 d) depend on the HIP stream (hardware queue) inside ONE process?
Runs the same training forward + backward of the headline net on several freshly created streams and prints the kernel's average
time on each (the library's HIP-event profiler). usage: python tools/probe/bwd_mode_streams.py [n_streams]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import util  # noqa: E402

os.environ.setdefault("V4L_COMPUTE", "bf16")
import vision4leg_amd.torchrl.networks as networks  # noqa: E402
import vision4leg_amd.torchrl.policies as policies  # noqa: E402
from vision4leg_amd import _lib  # noqa: E402

case = dict(util.CASES["loco_b1024"])
n = case["B"]
dev = torch.device("cuda:0")
torch.manual_seed(case["seed"])
pf, vf = util.build_nets(networks, policies, case)
pf, vf = pf.to(dev), vf.to(dev)
hip = vf.hip
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32, device=dev)
st, im, _ = hip.stage(obs)
dout = torch.randn(n, 16, device=dev)
grads = torch.zeros(hip.total_params, device=dev)
L = _lib.lib()


def measure(reps=30):
    for _ in range(5):
        hip.forward(st, im, n, train=True); hip.backward(st, im, n, dout, grads)
    torch.cuda.synchronize()
    L.v4l_prof_enable(1)
    for _ in range(reps):
        hip.forward(st, im, n, train=True); hip.backward(st, im, n, dout, grads)
    buf = C.create_string_buffer(1 << 20)
    L.v4l_prof_collect(buf, len(buf))
    L.v4l_prof_enable(0)
    out = {}
    for line in buf.value.decode().splitlines():
        label, calls, us, _ = line.split("\t")
        k = label.split("|")[-1]
        if k in ("wps_layer_bwd_stack", "wps_layer_stack_head", "fused_conv_bwd", "fused_encoder"):
            out[k] = float(us) / int(calls)
    return out


print("default stream:", {k: round(v, 1) for k, v in measure().items()})
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        r = measure()
    torch.cuda.synchronize()
    print("stream %d:" % i, {k: round(v, 1) for k, v in r.items()})
print("default stream again:", {k: round(v, 1) for k, v in measure().items()})
