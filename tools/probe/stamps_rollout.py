"""Phase stamps (s_memtime, block (0,0), thread 0) of the two rollout-step kernels; needs the diagnostic build:
hipcc ... -DV4L_INFER_TIMING (tools/probe/build_variant.sh timing -> vision4leg_amd/libv4l_hip_timing.so)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("V4L_LIB", os.path.join(ROOT, "vision4leg_amd/libv4l_hip_timing.so"))  # tools/probe/build_variant.sh timing
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
os.environ["V4L_COMPUTE"] = "bf16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib
dev = torch.device("cuda:0")
case = dict(util.CASES["loco_s93"]); E = 32
torch.manual_seed(0); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
actor = policies.RolloutActor(pf, vf, E)
obs = torch.randn(64, E, 93 + 16384, device=dev)
for i in range(64): actor.step(obs[i])
L = _lib.lib(); L.v4l_debug_stamps.argtypes = [C.c_void_p]; L.v4l_debug_stamps.restype = C.c_int
buf = (C.c_longlong * 128)(); L.v4l_debug_stamps(buf)
st = np.array(buf[:], dtype=np.int64)
def show(title, i0, names):
    print(title)
    for k, n in enumerate(names):
        print("  %-28s %8d" % (n, st[i0 + k + 1] - st[i0 + k]))
    print("  total %d cycles" % (st[i0 + len(names)] - st[i0]))
show("rollout_stack_kernel block(0,0), layer 0:", 64,
     ["load x", "in_proj", "scores", "softmax", "PV", "out_proj+res", "LN1", "FF1", "FF2+res", "LN2"])
print("  FF1 detail: first tile MFMAs done +%d, both tiles stored +%d, next loads issued +%d, barrier +%d" % (st[75]-st[71], st[76]-st[75], st[77]-st[76], st[72]-st[77]))
print("  layer 1 (all phases)         %8d" % (st[80] - st[74]))
show("head:", 80, ["pool", "fc0", "fc1", "fc2", "sample+file"])
print("  kernel total %d cycles" % (st[85] - st[64]))
show("rollout_encoder2_kernel block 0:", 96, ["image load+cast+file", "conv1", "conv2", "conv3", "sum+relu", "up-conv"])
