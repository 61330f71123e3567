import os, sys, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("V4L_LIB", os.path.join(ROOT, "vision4leg_amd/libv4l_hip_timing.so"))  # tools/probe/build_variant.sh timing
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np, torch, util
os.environ["V4L_COMPUTE"]="bf16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib
dev=torch.device("cuda:0")
case=dict(util.CASES["loco_s93"], B=1024); n=1024
torch.manual_seed(0); pf,vf=util.build_nets(networks,policies,case); pf,vf=pf.to(dev),vf.to(dev)
hip=vf.hip
obs=torch.randn(n, 93+16384, device=dev)
st_,im,_=hip.stage(obs)
L=_lib.lib(); L.v4l_debug_stamps.argtypes=[C.c_void_p]; L.v4l_debug_stamps.restype=C.c_int
for train in (True, False):
    for it in range(3): hip.forward(st_, im, n, train=train)
    torch.cuda.synchronize()
    buf=(C.c_longlong*48)(); L.v4l_debug_stamps(buf)
    st=np.array(buf[:9],dtype=np.int64)
    print("train" if train else "infer", np.diff(st), "in_proj: gemm +%d epi +%d | linear1: gemm +%d epi +%d" % (buf[9]-buf[1], buf[10]-buf[9], buf[11]-buf[5], buf[12]-buf[11]))
