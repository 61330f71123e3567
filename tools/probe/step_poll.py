"""What does waiting for the VALUE (besides the action) cost a host step? (round 6: step_host returns when action AND value of every env
have arrived in pinned memory, so that no block still reads the caller's observation buffer.)
One env step = pinned rows already cast -> action in host memory; variants of the completion wait:
  action only (round 5) | action + value (round 6) | stream synchronise (V4L_STEP_POLL=0)
Run on the GPU box: python tools/probe/step_poll.py > gpurun_out/step_poll.txt"""
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
E, N, S = 32, 600, 93
rs = np.random.RandomState(0)
rows = recipes.obs_rows(rs, E, case)
torch.manual_seed(0)
pf, vf = recipes.build_nets(networks, policies, case)
pf, vf = pf.to(dev), vf.to(dev)
actor = RolloutActor(pf, vf, E)
st, im = pf.hip.alloc_rollout(4 * E, dev)
actor.attach((st, im, torch.zeros(4 * E, 6, device=dev), torch.zeros(4 * E, device=dev), torch.zeros(4 * E, device=dev)))
prop = torch.from_numpy(rows[:, :S].astype(np.float32)).pin_memory()
img = torch.from_numpy(rows[:, S:]).to(pf.hip.image_dtype()).pin_memory()
h = actor._actor


def bench(name, n=N):
    for i in range(30):
        actor.seek(i & 3); actor.step_host_split(prop, img)
    torch.cuda.synchronize()
    lat = []
    t = time.perf_counter()
    for i in range(n):
        actor.seek(i & 3)
        t0 = time.perf_counter()
        actor.step_host_split(prop, img)
        lat.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    lat = np.array(lat) * 1e6
    print("%-46s %7.1f us per env step (call only: median %.1f, p90 %.1f)" % (name, (time.perf_counter() - t) / n * 1e6, np.median(lat), np.percentile(lat, 90)), flush=True)


bench("action + value arrival (round 6 default)")
orig = type(h)._await_action


def action_only(self):
    a = self._act_np
    for _ in range(4000):
        if not np.isnan(a).any():
            return a
    torch.cuda.current_stream(self.device).synchronize()
    return a


type(h)._await_action = action_only
bench("action arrival only (round 5)")
type(h)._await_action = orig
h._poll = False
bench("stream synchronise (V4L_STEP_POLL=0)")
h._poll = True
bench("action + value arrival again")
