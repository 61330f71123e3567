"""Which kernels does one training forward + backward of a test case launch? usage: python tools/probe/which_kernels.py <case> [B]
(the library's HIP-event profiler: phase|op|kernel, calls, us). Used to confirm that a net geometry took the path it should."""
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

name = sys.argv[1]
case = dict(util.CASES[name])
if len(sys.argv) > 2:
    case["B"] = int(sys.argv[2])
n = case["B"]
dev = torch.device("cuda:0")
torch.manual_seed(case["seed"])
pf, vf = util.build_nets(networks, policies, case)
pf, vf = pf.to(dev), vf.to(dev)
hip = pf.hip
obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32, device=dev)
st, im, _ = hip.stage(obs)
dout = torch.randn(n, 16, device=dev)
grads = torch.zeros(hip.total_params, device=dev)


def run():
    hip.forward(st, im, n, train=True)
    hip.backward(st, im, n, dout, grads)


for _ in range(2):
    run()
torch.cuda.synchronize()
L = _lib.lib()
L.v4l_prof_enable(1)
run()
buf = C.create_string_buffer(1 << 20)
L.v4l_prof_collect(buf, len(buf))
L.v4l_prof_enable(0)
tot = 0.0
for line in buf.value.decode().splitlines():
    label, calls, us, _ = line.split("\t")
    tot += float(us)
    print("%-70s %4s %9.1f" % (label, calls, float(us)))
print("%s B=%d: %d launches, %.1f us of kernel time" % (name, n, len(buf.value.decode().splitlines()), tot))
