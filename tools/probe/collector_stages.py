"""Where one env step of the fast collector goes (bench.py's fast_collector leg): host cast variants, pinned H2D, the rollout
launches, the action's D2H and the Python bookkeeping around them. Run on the GPU box: python tools/probe/collector_stages.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vision4leg_amd import recipes
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
pin64 = [torch.empty(E, D, dtype=torch.float64).pin_memory() for _ in range(2)]
dbuf = [torch.empty(E, D, dtype=torch.float32, device=dev) for _ in range(2)]
dbuf64 = [torch.empty(E, D, dtype=torch.float64, device=dev) for _ in range(2)]

def bench(name, fn, n=N, sync=True):
    for _ in range(10): fn(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): fn(i)
    if sync: torch.cuda.synchronize()
    print("%-52s %8.1f us" % (name, (time.perf_counter() - t) / n * 1e6), flush=True)

bench("numpy copyto f64->f32 (pinned dst)", lambda i: np.copyto(pin[i & 1].numpy(), rows[i & 3], casting="same_kind"))
for th in (1, 2, 4, 8, 16, 32):
    torch.set_num_threads(th)
    bench("torch copy_ f64->f32 pinned, %2d threads" % th, lambda i: pin[i & 1].copy_(torch.from_numpy(rows[i & 3])))
torch.set_num_threads(8)
bench("torch copy_ f64->f64 pinned (memcpy), 8 threads", lambda i: pin64[i & 1].copy_(torch.from_numpy(rows[i & 3])))
bench("H2D pinned fp32 2.1 MB (async + sync each)", lambda i: (dbuf[i & 1].copy_(pin[i & 1], non_blocking=True), torch.cuda.synchronize()), sync=False)
bench("H2D pinned fp64 4.2 MB (async + sync each)", lambda i: (dbuf64[i & 1].copy_(pin64[i & 1], non_blocking=True), torch.cuda.synchronize()), sync=False)
bench("H2D pageable f64 .to(dev) + sync", lambda i: (torch.from_numpy(rows[i & 3]).to(dev), torch.cuda.synchronize()), sync=False)
bench("device cast f64->f32 kernel", lambda i: dbuf[i & 1].copy_(dbuf64[i & 1]))
torch.manual_seed(0)
pf, vf = recipes.build_nets(networks, policies, case)
pf, vf = pf.to(dev), vf.to(dev)
actor = RolloutActor(pf, vf, E)
bench("actor.step (2 launches), no sync", lambda i: actor.step(dbuf[i & 1]))
bench("actor.step + action.cpu().numpy()", lambda i: actor.step(dbuf[i & 1])["action"].cpu().numpy(), sync=False)
def full(i):
    pin[i & 1].copy_(torch.from_numpy(rows[i & 3]))
    dbuf[i & 1].copy_(pin[i & 1], non_blocking=True)
    return actor.step(dbuf[i & 1])["action"].cpu().numpy()
bench("cast(8 thr) + H2D + step + D2H", full, sync=False)
a = np.zeros((E, 6))
bench("np.isfinite(acts).all()", lambda i: np.isfinite(a).all(), sync=False)
# ---- round 4: the split hand-over (fp32 proprio + bf16 depth rows read in place over PCIe)
S = case["S"]
prop = [torch.empty(E, S, dtype=torch.float32).pin_memory() for _ in range(2)]
img16 = [torch.empty(E, D - S, dtype=torch.bfloat16).pin_memory() for _ in range(2)]
src = [torch.from_numpy(r) for r in rows]
def cast_split(i):
    prop[i & 1].copy_(src[i & 3][:, :S]); img16[i & 1].copy_(src[i & 3][:, S:])
for th in (4, 8, 16, 32, 64):
    torch.set_num_threads(th)
    bench("split cast f64 -> f32 | bf16 pinned, %2d threads" % th, cast_split, sync=False)
actor2 = RolloutActor(pf, vf, E)
st, im = pf.hip.alloc_rollout(4 * E, dev)
actor2.attach((st, im, torch.zeros(4 * E, 6, device=dev), torch.zeros(4 * E, device=dev), torch.zeros(4 * E, device=dev)))
def host_split(i):
    actor2.seek(i & 3)
    return actor2.step_host_split(prop[i & 1], img16[i & 1])
def host_rows(i):
    actor2.seek(i & 3)
    return actor2.step_host(pin[i & 1])
def host_split_copy(i):
    actor2.seek(i & 3)
    return actor2._actor.step_host_split(prop[i & 1], img16[i & 1], via_copy=True)
def host_split_inplace(i):
    actor2.seek(i & 3)
    return actor2._actor.step_host_split(prop[i & 1], img16[i & 1], via_copy=False)
bench("step_host_split via 2 async H2D copies, incl. sync", host_split_copy, sync=False)
bench("step_host_split, rows read in place over PCIe, incl. sync", host_split_inplace, sync=False)
bench("step_host_split (default), incl. sync", host_split, sync=False)
bench("step_host (fp32 rows in place, incl. sync)", host_rows, sync=False)
for th in (8, 16, 32):
    torch.set_num_threads(th)
    bench("split cast (%2d thr) + step_host_split" % th, lambda i: (cast_split(i), host_split(i)), sync=False)
