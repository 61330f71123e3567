import os, sys, ctypes as C
os.environ.setdefault("V4L_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "vision4leg_amd/libv4l_hip_timing.so"))  # tools/probe/build_variant.sh timing
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, util
os.environ["V4L_COMPUTE"]="bf16"
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
from vision4leg_amd import _lib
dev=torch.device("cuda:0")
case=dict(util.CASES["loco_s93"]); E=32
torch.manual_seed(0); pf,vf=util.build_nets(networks,policies,case); pf,vf=pf.to(dev),vf.to(dev)
actor=policies.RolloutActor(pf,vf,E)
obs=torch.randn(E, 93+16384, device=dev)
for i in range(20): actor.step(obs)
L=_lib.lib(); L.v4l_debug_stamps.argtypes=[C.c_void_p]; L.v4l_debug_stamps.restype=C.c_int
buf=(C.c_longlong*32)(); L.v4l_debug_stamps(buf)
st=np.array(buf[:9],dtype=np.int64)
names=["load x","in_proj","attention","out_proj+res","LN1","FF1","FF2+res","LN2 out"]
d=np.diff(st)
print("layer kernel block(0,0) phase cycles (s_memtime):")
for n,c in zip(names,d): print("  %-14s %8d"%(n,c))
print("  total %d cycles"%(st[8]-st[0]))
