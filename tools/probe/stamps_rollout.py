import os, sys, ctypes as C
os.environ["V4L_LIB"] = "/root/repo/tools/probe/libv4l_timing.so"
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
for i in range(50): actor.step(obs)
torch.cuda.synchronize()
L=_lib.lib(); L.v4l_debug_stamps.argtypes=[C.c_void_p]; L.v4l_debug_stamps.restype=C.c_int
buf=(C.c_longlong*32)(); L.v4l_debug_stamps(buf)
st=np.array(buf[:],dtype=np.int64)
names={0:"load x",1:"L0 in_proj",2:"L0 attention",3:"L0 out_proj",4:"L0 LN1",5:"L0 FF1",6:"L0 FF2",7:"L0 LN2",8:"L0 end",
       10:"L1 in_proj",11:"L1 attention",12:"L1 out_proj",13:"L1 LN1",14:"L1 FF1",15:"L1 FF2",16:"L1 LN2",17:"L1 end",
       20:"pool",21:"head L0",22:"head L1",23:"head L2",24:"finish"}
idx=sorted(names)
print("rollout_layer_kernel block (0,0) phase cycles:")
for a,b in zip(idx[:-1], idx[1:]):
    print("  %-14s %8d" % (names[a], st[b]-st[a]))
print("  total %d" % (st[24]-st[0]))
