import os, sys, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["V4L_LIB"] = os.path.join(ROOT, "tools/probe/libv4l_timing.so")
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
for it in range(3): hip.forward(st_, im, n, train=True)
torch.cuda.synchronize()
buf=(C.c_longlong*128)(); L.v4l_debug_stamps(buf)
st=np.array(buf[:18],dtype=np.int64)
names=["L0 stage","q,k","v","attn","outproj+ln1","ffn","ln2+saves","(gap)","L1 stage","q,k","v","attn","outproj+ln1","ffn","ln2+saves","(gap)","->heads","heads"]
print("total cycles", st[17]-st[0])
for i in range(1,18): print("%-14s %8d" % (names[i-1], st[i]-st[i-1]))
