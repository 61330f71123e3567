"""Which contraction groups would have to leave half precision for the f16 mode to meet 1e-3 against the fp32 reference in EVERY case?
CPU oracle only (oracle/ppo_oracle.py with a per-group mode map), forward distances of the policy mean / the value per parity case.
usage: python tools/f16_precision_map.py > profiles/r6_f16_precision_map.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np, util
from oracle import ppo_oracle as orc
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
torch.set_num_threads(8)
allf16={g:"f16" for g in orc.GROUPS}
maps={"all f16":"f16",
      "heads f32":dict(allf16, heads="f32"),
      "heads+proprio+projector f32":dict(allf16, heads="f32", proprio="f32", projector="f32"),
      "heads+out_proj+upconv+proprio+projector+attn f32 (cheap groups)":dict(allf16, heads="f32", proprio="f32", projector="f32", out_proj="f32", upconv="f32", attn="f32")}
names=[n for n in util.CASES if util.CASES[n]["B"]<=128] + ["loco_b1024"]
worst={k:0.0 for k in maps}
for name in names:
    case=util.CASES[name]
    if case["kind"] not in orc.FORWARDS: continue
    torch.manual_seed(case["seed"])
    pf,vf=util.build_nets(networks,policies,case)
    opf={k:v.detach().clone() for k,v in pf.state_dict().items() if k!="logstd"}
    ovf=util.share_encoder({k:v.detach().clone() for k,v in pf.state_dict().items()}, {k:v.detach().clone() for k,v in vf.state_dict().items()}, case["kind"])
    obs=torch.tensor(util.make_batch(case)["obs"],dtype=torch.float32)
    with torch.no_grad():
        try:
            r_m=orc.FORWARDS[case["kind"]](opf,obs,case["S"],"f32"); r_v=orc.FORWARDS[case["kind"]](ovf,obs,case["S"],"f32")
        except Exception as e:
            print(name,"skip",e); continue
        row=[]
        for k,m in maps.items():
            try:
                a=orc.FORWARDS[case["kind"]](opf,obs,case["S"],m); b=orc.FORWARDS[case["kind"]](ovf,obs,case["S"],m)
            except Exception as e:
                row.append("n/a"); continue
            em,ev=util.rel_err(a,r_m),util.rel_err(b,r_v)
            worst[k]=max(worst[k],em,ev); row.append("%.1e/%.1e"%(em,ev))
    print("%-14s"%name, "  ".join(row), flush=True)
print({k:"%.2e"%v for k,v in worst.items()})
