"""Phase stamps (block (0,0), lane 0) of rollout_linear_kernel<32> on the vision-only NatureCNN step (E = 32, split launches);
needs the diagnostic build (tools/probe/build_variant.sh timing)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("V4L_LIB", os.path.join(ROOT, "vision4leg_amd/libv4l_hip_timing.so"))  # tools/probe/build_variant.sh timing
os.environ["V4L_ROLLOUT_DENSE_SPLIT"] = "1"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
os.environ["V4L_COMPUTE"] = "bf16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib
dev = torch.device("cuda:0")
case = dict(util.CASES["cnn_vis"]); E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0); pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)
actor = policies.RolloutActor(pf, vf, E)
obs = torch.randn(64, E, util.obs_dim(case), device=dev)
for i in range(64): actor.step(obs[i])
torch.cuda.synchronize()
L = _lib.lib(); L.v4l_debug_stamps.argtypes = [C.c_void_p]; L.v4l_debug_stamps.restype = C.c_int
buf = (C.c_longlong * 128)(); L.v4l_debug_stamps(buf)
st = np.array(buf[:], dtype=np.int64)
print("rollout_linear_kernel<32> block (0,0), E=%d (cycles of the 100 MHz s_memtime clock x 24 = core clocks?) raw deltas:" % E)
names = ["entry -> loads issued (W + x tile 0)", "loads arrived", "MFMA + store issued", "store retired"]
for mt in range((E + 15) // 16):
    base = 112 if mt == 0 else 116 + 4 * (mt - 1)
    idx = [112 if mt == 0 else 116 + 4 * (mt - 1), 113 + 4 * mt, 114 + 4 * mt, 115 + 4 * mt, 116 + 4 * mt]
    for k, n in enumerate(names):
        print("  tile %d %-40s %8d" % (mt, n, st[idx[k + 1]] - st[idx[k]]))
print("  total %d" % (st[116 + 4 * ((E + 15) // 16 - 1)] - st[112]))
