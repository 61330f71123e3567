"""The fast collector's observation hand-over, cast included: kernels reading the pinned rows in place
(round 4's default) against the cast -> DMA pipeline (cast a row chunk, start its asynchronous copy to HBM, cast the next chunk
under it; the kernels then read HBM), for 1 / 2 / 4 / 8 chunks. One env step = float64 rows -> action in host memory.
Run on the GPU box: python tools/probe/collector_pipe.py > gpurun_out/collector_pipe.txt"""
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
E, N, S = int(os.environ.get("E", "32")), 400, 93
rs = np.random.RandomState(0)
rows = [recipes.obs_rows(rs, E, case) for _ in range(4)]
D = rows[0].shape[1]
torch.manual_seed(0)
pf, vf = recipes.build_nets(networks, policies, case)
pf, vf = pf.to(dev), vf.to(dev)
actor = RolloutActor(pf, vf, E)
st, im = pf.hip.alloc_rollout(4 * E, dev)
actor.attach((st, im, torch.zeros(4 * E, 6, device=dev), torch.zeros(4 * E, device=dev), torch.zeros(4 * E, device=dev)))
prop = [torch.empty(E, S, dtype=torch.float32).pin_memory() for _ in range(2)]
img = [torch.empty(E, D - S, dtype=torch.bfloat16).pin_memory() for _ in range(2)]
src = [torch.from_numpy(r) for r in rows]
dprop, dimg = actor.split_device_buffers()
torch.set_num_threads(int(os.environ.get("V4L_CAST_THREADS", "8")))


def bench(name, fn, n=N):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    print("%-78s %8.1f us per env step" % (name, (time.perf_counter() - t) / n * 1e6), flush=True)


def inplace(i):
    actor.seek(i & 3)
    p, g = prop[i & 1], img[i & 1]
    p.copy_(src[i & 3][:, :S]); g.copy_(src[i & 3][:, S:])
    return actor.step_host_split(p, g)


def piped(chunks):
    step = -(-E // chunks)
    def fn(i):
        actor.seek(i & 3)
        p, g, s = prop[i & 1], img[i & 1], src[i & 3]
        p.copy_(s[:, :S]); dprop.copy_(p, non_blocking=True)
        for a in range(0, E, step):
            b = min(E, a + step)
            g[a:b].copy_(s[a:b, S:]); dimg[a:b].copy_(g[a:b], non_blocking=True)
        return actor.step_host_split(dprop, dimg, on_device=True)
    return fn


def cast_only(i):
    prop[i & 1].copy_(src[i & 3][:, :S]); img[i & 1].copy_(src[i & 3][:, S:])


def step_resident(i):
    actor.seek(i & 3)
    return actor.step_host_split(dprop, dimg, on_device=True)


print("# E = %d envs, %d cast threads; one env step = float64 rows [E][%d] -> [E][6] action in host memory" % (E, torch.get_num_threads(), D))
bench("host cast alone (f64 -> f32 proprio | bf16 depth, pinned)", cast_only)
bench("step on rows already in HBM (2 launches + action D2H + synchronise)", step_resident)
bench("cast, then kernels read the pinned rows in place over PCIe (round 4)", inplace)
if actor._actor._poll is not None:
    actor._actor._poll = not actor._actor._poll
    bench("  the same with V4L_STEP_POLL=%d" % int(actor._actor._poll), inplace)
    bench("  step on rows already in HBM, V4L_STEP_POLL=%d" % int(actor._actor._poll), step_resident)
    actor._actor._poll = not actor._actor._poll
for c in (1, 2):
    bench("cast -> DMA pipeline, %d chunk%s, kernels read HBM" % (c, "" if c == 1 else "s"), piped(c))
