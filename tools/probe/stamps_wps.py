import os, sys, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("V4L_LIB", os.path.join(ROOT, "vision4leg_amd", os.environ.get("TIMING_LIB", "libv4l_hip_timing.so")))  # tools/probe/build_variant.sh timing
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np, torch, util
os.environ.setdefault("V4L_COMPUTE","f16")
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib
dev=torch.device("cuda:0")
case=dict(util.CASES["loco_s93"], B=1024); n=1024
torch.manual_seed(0); pf,vf=util.build_nets(networks,policies,case); pf,vf=pf.to(dev),vf.to(dev)
hip=vf.hip
obs=torch.randn(n, 93+16384, device=dev)
st_,im,_=hip.stage(obs)
L=_lib.lib(); L.v4l_debug_stamps.argtypes=[C.c_void_p]; L.v4l_debug_stamps.restype=C.c_int
dout=torch.zeros(n,16,device=dev); dout[:,0]=1.0
grads=torch.zeros(hip.total_params, device=dev)
for it in range(3):
    hip.forward(st_, im, n, train=True)
    hip.backward(st_, im, n, dout, grads)
torch.cuda.synchronize()
buf=(C.c_longlong*128)(); L.v4l_debug_stamps(buf)
st=np.array(buf[:128],dtype=np.int64)
def show(title, idx, names):
    print(title, "total cycles", st[idx[-1]]-st[idx[0]])
    for a,b,nm in zip(idx[:-1], idx[1:], names): print("   %-26s %8d" % (nm, st[b]-st[a]))
show("FWD", [0,7,6,8,15,14,16,17], ["L0 stage+rows+wait","L0 layer","barrier","L1 stage+wait","L1 layer","(to heads)","heads"])
show("BWD", [32,33,34,35,36,37,38,41,42,43,44,45,46,50,51],
     ["heads","L1 stageW+rows","L1 recompute","barrier","L1 stageWt","L1 backward","LNred+barrier","L0 stageW+rows","L0 recompute","barrier","L0 stageWt","L0 backward","LNred+dx","tail"])
show("BWD heads", [32,60,61,62,63,33], ["dout/masks/rings issue + stage dt","gemm w2t + mask","gemm w1t + mask","gemm w0t","unpool -> barrier"])
show("BWD tail", [50,66,67,68,51], ["upconv' per wave (+c3, dc3)","token-0 rows -> LDS","proj' gemm + mask","fc2' gemm + mask"])
