"""the process-dependent launch time of the wave-per-sample backward (now: of wps_layer_fb_kernel). Is a slow
process one where every block is uniformly slower (a clock / power state), or one where the blocks land differently on the XCDs /
CUs? Timing build (tools/probe/build_variant.sh timing): every block of the fused forward-loss-backward launch logs its XCC_ID /
HW_ID, the chip-wide 100 MHz wall clock and its XCD's shader clock at entry and exit. One process = one line block; run it several
times (tools/gpu_session.sh fbmodes) and compare fast and slow processes.
usage: python tools/probe/fb_blocks.py [updates]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("V4L_LIB", os.path.join(ROOT, "vision4leg_amd", "libv4l_hip_timing.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
os.environ.setdefault("V4L_COMPUTE", "f16")
from vision4leg_amd import _lib
from vision4leg_amd.engine import HipTrainer
from vision4leg_amd.torchrl.algo import PPO
import vision4leg_amd.torchrl.networks as networks, vision4leg_amd.torchrl.policies as policies

U = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
case = dict(util.CASES["loco_s93"], B=1024)
B, TE = 1024, 4096
torch.manual_seed(0)
pf, vf = util.build_nets(networks, policies, case); pf, vf = pf.to(dev), vf.to(dev)


class Coll: epoch_frames = TE
agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, entropy_coeff=0.005, collector=Coll(),
            device=dev, batch_size=B)
rs = np.random.RandomState(3)
net = pf.hip; net.ensure_bound()
state, image = net.alloc_rollout(TE, dev)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
net.ingest(t(util.obs_rows(rs, TE, case)), state, image)
ro = HipTrainer.rollout(state, image, t(0.1 * rs.randn(TE, case["A"])), t(rs.randn(TE)), t(rs.randn(TE)), t(rs.randn(TE)),
                        logp_old=t(-5.0 + 0.1 * rs.randn(TE)))
agent.trainer.sync_target()
rows = torch.tensor(np.stack([rs.permutation(TE)[:B] for _ in range(U)]).astype(np.int32), device=dev)
stats = torch.zeros(U, 24, device=dev)
for ep in range(3):  # update by update, capture of the run, replay
    agent.run_updates(ro, rows, stats)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(agent.trainer.stream if hasattr(agent.trainer, "stream") else None)
agent.run_updates(ro, rows, stats)
e1.record(agent.trainer.stream if hasattr(agent.trainer, "stream") else None)
torch.cuda.synchronize()
print("run of %d updates: %.1f us per update" % (U, 1000.0 * e0.elapsed_time(e1) / U))
L = _lib.lib()
if not hasattr(L, "v4l_debug_block_log"):  # not the timing build (PMC passes run the shipped kernels)
    sys.exit(0)
L.v4l_debug_block_log.argtypes = [C.c_void_p]; L.v4l_debug_block_log.restype = C.c_int
buf = (C.c_longlong * 5120)()
assert L.v4l_debug_block_log(buf) == 0
lg = np.array(buf[:5120], dtype=np.int64).reshape(1024, 5)[:256]   # the last fused launch of the run (the policy's pass): 256 blocks
wall0, clk0, ids, wall1, clk1 = lg.T
hw, xcc = ids & 0xffffffff, (ids >> 32) & 0xf
cu, se = (hw >> 8) & 0xf, (hw >> 13) & 0x7
dur_wall = (wall1 - wall0) / 100.0            # us (100 MHz)
dur_clk = (clk1 - clk0).astype(np.float64)    # shader cycles
mhz = dur_clk / dur_wall
start = (wall0 - wall0.min()) / 100.0
print("launch: first entry -> last exit %.1f us; block duration us: mean %.1f min %.1f max %.1f; entry spread %.1f us; "
      "shader clock during the block: mean %.0f MHz (min %.0f, max %.0f)"
      % ((wall1.max() - wall0.min()) / 100.0, dur_wall.mean(), dur_wall.min(), dur_wall.max(), start.max(), mhz.mean(), mhz.min(), mhz.max()))
print("per XCD: blocks | mean block us | mean MHz | mean entry us | distinct (se, cu)")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("   xcc %d: %3d | %6.1f | %5.0f | %5.1f | %d" % (x, int(m.sum()), dur_wall[m].mean(), mhz[m].mean(), start[m].mean(),
                                                        len(set(zip(se[m].tolist(), cu[m].tolist())))))
order = np.argsort(wall0)
print("blockIdx -> xcc of the first 16 blocks: %s" % " ".join("%d" % xcc[b] for b in range(16)))
print("slowest 8 blocks: " + ", ".join("b%d xcc%d se%d cu%d %.1fus %.0fMHz" % (b, xcc[b], se[b], cu[b], dur_wall[b], mhz[b])
                                        for b in np.argsort(-dur_wall)[:8]))
