"""the collector's env step as one library call (v4l_actor_step_rows, csrc/host_step.h) against the round-5
arrangement (torch's copy kernel for the cast + RolloutActor.step_host_split from Python). (a) the cast alone, library pool vs
torch, by thread count; (b) the whole step on a bare RolloutActor (no collector bookkeeping); (c) bench.py's fast_collector leg
(VecOnPolicyCollector over the zero-cost env + update) per arrangement and thread count.
usage: python tools/probe/host_step.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ.setdefault("V4L_COMPUTE", "f16")
from vision4leg_amd import _lib, recipes
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies
import bench
L = _lib.lib()
wl = dict(bench.WORKLOADS["loco"]); E, S, img = wl["E"], wl["S"], 16384
rs = np.random.RandomState(0)
pool = [recipes.obs_rows(rs, E, dict(wl, seed=0)) for _ in range(8)]
prop = torch.empty(E, S).pin_memory(); out = torch.empty(E, img, dtype=torch.float16).pin_memory()
ONLY = os.environ.get("PROBE_ONLY", "abc")
print("AVX-512 cast path: %d; logical CPUs %d" % (L.v4l_host_cast_simd(), os.cpu_count()))
print("(a) cast alone, us per env step (E = %d rows of %d doubles)" % (E, S + img))
for th in (1, 2, 4, 8, 16, 32) if "a" in ONLY else ():
    f = lambda it: L.v4l_host_cast_rows(C.c_void_p(pool[it % 8].ctypes.data), S + img, E, S, img, C.c_void_p(prop.data_ptr()),
                                        C.c_void_p(out.data_ptr()), _lib.V4L_F16, th)
    for it in range(100): f(it)
    t0 = time.perf_counter()
    for it in range(1000): f(it)
    print("   library pool, %2d threads: %6.1f" % (th, (time.perf_counter() - t0) / 1000 * 1e6))
for th in (8, 16) if "a" in ONLY else ():
    torch.set_num_threads(th)
    def f(it):
        src = torch.from_numpy(pool[it % 8]); prop.copy_(src[:, :S]); out.copy_(src[:, S:])
    for it in range(50): f(it)
    t0 = time.perf_counter()
    for it in range(500): f(it)
    print("   torch copy kernel, %2d threads: %6.1f" % (th, (time.perf_counter() - t0) / 500 * 1e6))
dev = torch.device("cuda:0")
torch.manual_seed(0)
pf, vf = recipes.build_nets(networks, policies, dict(wl, seed=0)); pf, vf = pf.to(dev), vf.to(dev)
actor = policies.RolloutActor(pf, vf, E)
T = 512
st, im = pf.hip.alloc_rollout(T * E, dev)
actor.attach((st, im, torch.zeros(T * E, wl["A"], device=dev), torch.zeros(T * E, device=dev), torch.zeros(T * E, device=dev)))
print("(b) whole step on a bare RolloutActor, us per env step (median of 3 x %d steps)" % T)
def loop(fn):
    res = []
    for rep in range(4):
        actor.seek(0); actor.draw_noise(T); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(T): fn(t)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / T * 1e6)
    return sorted(res[1:])[1]
pp = [(torch.empty(E, S).pin_memory(), torch.empty(E, img, dtype=torch.float16).pin_memory()) for _ in range(2)]
def old(t):
    p, i16 = pp[t & 1]; src = torch.from_numpy(pool[t % 8]); p.copy_(src[:, :S]); i16.copy_(src[:, S:])
    actor._actor.step_host_split(p, i16)
torch.set_num_threads(8)
if "b" in ONLY: print("   torch cast (8 threads) + step_host_split: %6.1f" % loop(old))
for th in (4, 8, 16, 32) if "b" in ONLY else ():
    print("   step_host_rows, %2d threads:              %6.1f" % (th, loop(lambda t: actor._actor.step_host_rows(pool[t % 8], threads=th))))
dimg = torch.empty(E, img, dtype=torch.float16, device=dev); dprop = torch.empty(E, S, device=dev)
if "b" in ONLY: print("   step on rows already in HBM (step_host_split on_device): %6.1f" % loop(lambda t: actor._actor.step_host_split(dprop, dimg, on_device=True)))
print("(c) bench.py fast_collector leg (collector + update over the zero-cost env)")
for hs, th in (("0", "8"), ("1", "8"), ("1", "12"), ("1", "16"), ("1", "8"), ("0", "8")):
    os.environ["V4L_COLLECT_HOST_STEP"], os.environ["V4L_CAST_THREADS"] = hs, th
    r = bench.fast_collector(wl, "f16", dev)
    print("   host_step=%s threads=%s: collect %.1f us per env step, value_incl_transfers %.0f" % (hs, th, r["collect_us_per_env_step"], r["value"]))
