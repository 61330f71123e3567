"""A few training forward + backward passes of the value net at B = 1024 (for rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import torch, util
os.environ.setdefault("V4L_COMPUTE","bf16")
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
dev=torch.device("cuda:0")
n=1024
case=dict(util.CASES["loco_s93"], B=n)
torch.manual_seed(0); pf,vf=util.build_nets(networks,policies,case); pf,vf=pf.to(dev),vf.to(dev)
hip=vf.hip
obs=torch.randn(n, 93+16384, device=dev)
st_,im,_=hip.stage(obs)
dout=torch.zeros(n,16,device=dev); dout[:,0]=1.0
grads=torch.zeros(hip.total_params, device=dev)
for it in range(int(os.environ.get("ITERS","10"))):
    hip.forward(st_, im, n, train=True)
    hip.backward(st_, im, n, dout, grads)
torch.cuda.synchronize()
print("done")
