"""Can the rollout step read its observation rows straight out of pinned host memory, and write the action straight into it?
(instead of H2D copy -> launch -> D2H copy). Run on the GPU box."""
import os, sys, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vision4leg_amd import recipes
from vision4leg_amd._lib import check
import vision4leg_amd.torchrl.networks as networks
import vision4leg_amd.torchrl.policies as policies
from vision4leg_amd.torchrl.policies import RolloutActor
dev = torch.device("cuda:0")
case = dict(kind="loco", S=93, A=6, seed=0, enc=[256, 256], head=[256, 256], layers=2, ff=256)
E, N = 32, 300
rs = np.random.RandomState(0)
rows = [recipes.obs_rows(rs, E, case) for _ in range(4)]
D = rows[0].shape[1]
pin = [torch.empty(E, D, dtype=torch.float32).pin_memory() for _ in range(2)]
dbuf = [torch.empty(E, D, dtype=torch.float32, device=dev) for _ in range(2)]
torch.manual_seed(0)
pf, vf = recipes.build_nets(networks, policies, case)
pf, vf = pf.to(dev), vf.to(dev)
actor = RolloutActor(pf, vf, E)
a = actor._actor
stream = torch.cuda.current_stream().cuda_stream
act_pin = torch.zeros(E, 6, dtype=torch.float32).pin_memory()
torch.set_num_threads(8)
def bench(name, fn, n=N):
    for _ in range(10): fn(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    print("%-64s %8.1f us" % (name, (time.perf_counter() - t) / n * 1e6), flush=True)
def classic(i):
    pin[i & 1].copy_(torch.from_numpy(rows[i & 3]))
    dbuf[i & 1].copy_(pin[i & 1], non_blocking=True)
    return actor.step(dbuf[i & 1])["action"].cpu().numpy()
def raw_step(obs_ptr, act_ptr=None):
    a.eps.normal_()
    args = list(a._args)
    args[0] = C.c_void_p(obs_ptr)
    if act_ptr is not None:
        args[7] = C.c_void_p(act_ptr)
    a.seek(0)
    check(a.L.v4l_actor_step(a.h, *args, stream), "step")
def zc_obs(i):
    pin[i & 1].copy_(torch.from_numpy(rows[i & 3]))
    raw_step(pin[i & 1].data_ptr())
    return a.action.cpu().numpy()
def zc_both(i):
    pin[i & 1].copy_(torch.from_numpy(rows[i & 3]))
    raw_step(pin[i & 1].data_ptr(), act_pin.data_ptr())
    torch.cuda.current_stream().synchronize()
    return act_pin.numpy()
def zc_act(i):
    pin[i & 1].copy_(torch.from_numpy(rows[i & 3]))
    dbuf[i & 1].copy_(pin[i & 1], non_blocking=True)
    raw_step(dbuf[i & 1].data_ptr(), act_pin.data_ptr())
    torch.cuda.current_stream().synchronize()
    return act_pin.numpy()
bench("classic: cast + H2D + step + action.cpu()", classic)
try:
    bench("obs read from pinned host by the kernel + action.cpu()", zc_obs)
    r1 = zc_obs(0).copy(); torch.manual_seed(1); 
except Exception as e:
    print("zc_obs failed", e)
try:
    bench("H2D copy, action written to pinned host + stream sync", zc_act)
    bench("obs from pinned host + action to pinned host + stream sync", zc_both)
except Exception as e:
    print("zc act failed", e)
# correctness of the zero-copy paths: same seed -> same action as the classic path
torch.manual_seed(5); x = classic(1).copy()
torch.manual_seed(5); y = zc_both(1).copy()
print("max |classic - zero-copy| action:", np.abs(x - y).max())
