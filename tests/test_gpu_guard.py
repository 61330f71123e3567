"""-m gpu: out-of-bounds WRITE check of the HIP path's device buffers (SURVEY.md §5 "sanitizers" row).

Device-side AddressSanitizer cannot run on this GPU pool (the instrumented library — 75 minutes of hipcc at
-O1, a compiler error at -O0 — was built and taken to the box: the agent is gfx950:xnack- and the image has no ASAN build of the
ROCm runtime, profiles/r4_gpu_asan.txt), so the suite checks what it can on the hardware: with V4L_GUARD=1 every buffer the library writes into (workspaces, control
blocks, packed weights, descriptor tables, gradient buckets, Adam moments, rollout arrays, actor outputs) sits between two
64 KB canary bands, the shapes that stress the hand-computed offsets run (ragged batches, E = 33 / 7 rollout steps, the
B = 1024 update with graph replays, both compute modes), and no band may have been touched."""
import os

import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def _nets(name, mode, device):
    os.environ["V4L_COMPUTE"] = mode
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = util.CASES[name]
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    return case, pf.to(device), vf.to(device)


@pytest.mark.timeout(900)
def test_no_kernel_writes_outside_its_buffers(device, monkeypatch):
    from vision4leg_amd import engine
    from vision4leg_amd.torchrl.algo import PPO
    from vision4leg_amd.torchrl.policies import RolloutActor
    monkeypatch.setenv("V4L_GUARD", "1")
    engine.check_guards(reset=True)
    ran = []
    for mode in ("bf16", "f16", "f32"):
        # PPO updates: small, ragged (300 = 256 + 44), B = 1024 with graph replays; every net kind once
        for name, n_upd in (("loco_s84", 2), ("loco_rag", 1), ("loco_b1024", 3), ("cnn_s93", 2), ("mlp_s93", 2), ("loco_vis", 1),
                            ("cnn_vis", 1), ("loco_gen", 1), ("loco_max", 1), ("loco_tn", 1), ("loco_pe", 1)):
            case, pf, vf = _nets(name, mode, device)

            class Coll: epoch_frames = 1
            agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=3, tau=0.95, entropy_coeff=0.005,
                        collector=Coll(), device=device, batch_size=case["B"],
                        clipped_value_loss=case.get("clipped_value_loss", False))
            agent.trainer.sync_target()
            if n_upd >= 3:  # the resident path: row-index minibatches, graph replays
                from vision4leg_amd.engine import HipTrainer
                b = util.make_batch(case)
                t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
                net = pf.hip
                net.ensure_bound()
                st, im = net.alloc_rollout(case["B"], device)
                net.ingest(t(b["obs"]), st, im)
                ro = HipTrainer.rollout(st, im, t(b["acts"]), t(b["advs"]).reshape(-1), t(b["estimate_returns"]).reshape(-1),
                                        t(b["values"]).reshape(-1))
                rows = torch.stack([torch.randperm(case["B"], device=device).int() for _ in range(n_upd)])
                stats = torch.zeros(n_upd, 24, device=device)
                agent.run_updates(ro, rows, stats)
            else:
                for u in range(n_upd):
                    b = util.make_batch(case, update=u)
                    agent.update({k: b[k] for k in ("obs", "acts", "advs", "estimate_returns", "values")})
            ran.append((mode, name))
        # ragged forward / backward through the module API (n = 1, 30)
        case, pf, vf = _nets("loco_s84", mode, device)
        for n in (1, 30):
            obs = torch.tensor(util.make_batch(dict(case, B=n))["obs"], dtype=torch.float32, device=device)
            hip = vf.hip
            st, im, _ = hip.stage(obs)
            hip.forward(st, im, n, train=True)
            dout = torch.zeros(n, 16, device=device)
            dout[:, 0] = 1.0
            hip.backward(st, im, n, dout, engine._buf(hip.total_params, torch.float32, device))
        # rollout steps at odd env counts (fused LocoTransformer step, dense NatureCNN step with a ragged row tile, state MLP)
        for name, E in (("loco_s84", 7), ("cnn_s93", 33), ("mlp_s93", 5), ("loco_vis", 3), ("cnn_vis", 16)):
            case, pf, vf = _nets(name, mode, device)
            actor = RolloutActor(pf, vf, E)
            T = 3
            st, im = pf.hip.alloc_rollout(T * E, device)
            acts = engine._buf(T * E * case["A"], torch.float32, device, zero=True).view(T * E, case["A"])
            vals, logp = engine._buf(T * E, torch.float32, device, zero=True), engine._buf(T * E, torch.float32, device, zero=True)
            actor.attach((st, im, acts, vals, logp))
            actor.seek(0)
            rs = np.random.RandomState(0)
            for t_ in range(T):
                actor.step(torch.tensor(util.obs_rows(rs, E, case), dtype=torch.float32, device=device))
            ran.append((mode, name, E))
    torch.cuda.synchronize()
    n_guarded = len(engine._guards)
    assert n_guarded > 100, n_guarded   # the buffers really were guarded
    bad = engine.check_guards(reset=True)
    print("\n[guard bands] %d guarded buffers over %d runs, corrupted bands: %s" % (n_guarded, len(ran), bad))
    util.record("guard_bands/buffers_checked", n_guarded)
    util.record("guard_bands/corrupted", len(bad))
    assert not bad, bad


def test_guard_bands_catch_a_stray_write(device, monkeypatch):
    """The checker itself: one float written one element past a guarded buffer is reported."""
    from vision4leg_amd import engine
    monkeypatch.setenv("V4L_GUARD", "1")
    engine.check_guards(reset=True)
    buf = engine._buf(1000, torch.float32, device, zero=True)
    assert engine.check_guards() == []
    torch.as_strided(buf, (1,), (1,), storage_offset=buf.storage_offset() + 1000).fill_(1.0)
    bad = engine.check_guards(reset=True)
    assert len(bad) == 1 and bad[0][1] == "tail", bad
