import os, sys, ctypes as C
os.environ.setdefault("V4L_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "vision4leg_amd/libv4l_hip_timing.so"))  # tools/probe/build_variant.sh timing
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
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
for it in range(3):
    hip.forward(st_, im, n, train=True)
    dout=torch.randn(n,16,device=dev); grads=torch.zeros(hip.total_params,device=dev)
    hip.backward(st_, im, n, dout, grads)
torch.cuda.synchronize()
L=_lib.lib(); L.v4l_debug_stamps.argtypes=[C.c_void_p]; L.v4l_debug_stamps.restype=C.c_int
buf=(C.c_longlong*48)(); L.v4l_debug_stamps(buf)
st=np.array(buf[16:23],dtype=np.int64)
names=["loads->LDS","dgrad3 (gather GEMM)","bias2+wgrad2","dgrad2 (gather GEMM)","bias1","wgrad1"]
print("bwd_conv_kernel block 0, 4th sample, phase cycles (clock64, 100 MHz s_memtime or shader clock):")
for nm,c in zip(names,np.diff(st)): print("  %-22s %8d"%(nm,c))
print("  total %d"%(st[6]-st[0]))
st=np.array(buf[:9],dtype=np.int64); print("layer kernel stamps diff", np.diff(st))
names=["stage x / wait","in_proj","attention","out_proj+res","norm1","linear1+relu","linear2+res","norm2"]
print("infer_layer_kernel (training forward, last layer's pass) block 0 phase cycles:")
for nm,c in zip(names,np.diff(st)): print("  %-22s %8d"%(nm,c))
print("  in_proj:  GEMM done +%d, epilogue (bias, qkv saves, q|k|v^T to LDS) +%d, barrier +%d" % (buf[9]-buf[1], buf[10]-buf[9], buf[2]-buf[10]))
print("  linear1:  GEMM done +%d, epilogue (bias, relu, f saves) +%d, barrier +%d" % (buf[11]-buf[5], buf[12]-buf[11], buf[6]-buf[12]))
st=np.array(buf[32:42],dtype=np.int64)
names=["load dy / heads","ln2 bwd","df gemm","dx1 gemm","qkv,P loads + ln1 bwd","dctx gemm","attention bwd","dx_in gemm","tail"]
print("bwd_layer_kernel (layer 0, TAIL variant) block 0 phase cycles:")
for nm,c in zip(names,np.diff(st)): print("  %-24s %8d"%(nm,c))
print("  total %d"%(st[9]-st[0]))
b=buf
print("  df:   GEMM done +%d, epilogue (mask, df saves, df to LDS) +%d, barrier +%d" % (b[32+10]-b[32+2], b[32+11]-b[32+10], b[32+3]-b[32+11]))
print("  dx1:  qkv / P / norm1 rows requested +%d, GEMM +%d, epilogue + barrier +%d" % (b[32+12]-b[32+3], b[32+13]-b[32+12], b[32+4]-b[32+13]))
print("  park: q|k|v^T to LDS +%d, norm1 backward +%d" % (b[32+14]-b[32+4], b[32+5]-b[32+14]))
