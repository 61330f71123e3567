"""Stage stamps (block 0 = net 0 / tile 0, lane 0) of rollout_dense_kernel; needs the diagnostic build (tools/probe/build_variant.sh timing).
usage: python tools/probe/stamps_dense1.py [cnn_vis|cnn_s93] [E]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("V4L_LIB", os.path.join(ROOT, "vision4leg_amd/libv4l_hip_timing.so"))  # tools/probe/build_variant.sh timing
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
os.environ["V4L_COMPUTE"] = "bf16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cnn_vis"
case = dict(util.CASES[name]); E = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
actor = policies.RolloutActor(pf, vf, E)
obs = torch.randn(64, E, util.obs_dim(case), device=dev)
for i in range(64): actor.step(obs[i])
torch.cuda.synchronize()
L = _lib.lib(); L.v4l_debug_stamps.argtypes = [C.c_void_p]; L.v4l_debug_stamps.restype = C.c_int
buf = (C.c_longlong * 128)(); L.v4l_debug_stamps(buf)
st = np.array(buf[:], dtype=np.int64)
names = {100: "entry", 101: "weights requested", 102: "projector tile", 103: "signal 0", 104: "wait 0", 105: "fc0 tile",
         106: "signal 1", 107: "wait 1", 108: "fc1 tile", 109: "signal 3", 110: "wait 3", 111: "last linear + epilogue"}
prev = None
print("%s E=%d rollout_dense_kernel block 0" % (name, E))
for i in range(100, 112):
    if st[i] == 0: continue
    if prev is not None: print("  %-26s %8d" % (names[i], st[i] - st[prev]))
    prev = i
print("  total %d" % (st[111] - st[100]))
