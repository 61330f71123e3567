"""-m gpu: the collector-side fast path (vision4leg_amd.torchrl.collector.VecOnPolicyCollector + RolloutActor +
DeviceOnPolicyReplayBuffer + pinned uploads) against the restated reference collection loop (oracle/collector_ref.py,
pinned to /root/reference/torchrl/collector/on_policy.py:84-155 by tests/test_overlay_cpu.py) driving the same HIP
policy / value modules through the reference's call protocol with the reference's float64 host buffer."""
import os

import numpy as np
import pytest
import torch

import util
from oracle.collector_ref import RefVecOnPolicyCollector

pytestmark = pytest.mark.gpu


class _Log:
    def __init__(self):
        self.infos = []

    def add_update_info(self, info):
        self.infos.append(dict(info))


def _run(which, case, mode, device, E, T, B, epochs, max_episode_frames):
    os.environ["V4L_COMPUTE"] = mode
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    from vision4leg_amd.torchrl.algo import PPO
    from vision4leg_amd.torchrl.collector import VecOnPolicyCollector
    from vision4leg_amd.torchrl.replay_buffers import DeviceOnPolicyReplayBuffer, OnPolicyReplayBuffer
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    img = 0 if case["kind"] == "mlp" else 4 * 64 * 64
    env = util.FakeVecEnv(E, case["S"], case["A"], img=img, seed=9, p_done=0.12, time_limit_key=True)
    if which == "fast":
        buf = DeviceOnPolicyReplayBuffer(max_replay_buffer_size=E * T, env_nums=E, time_limit_filter=True)
        coll = VecOnPolicyCollector(vf, env=env, eval_env=util.FakeVecEnv(E, case["S"], case["A"], img=img, seed=1), pf=pf,
                                    replay_buffer=buf, device=device, epoch_frames=E * T,
                                    max_episode_frames=max_episode_frames)
        assert coll.fast_path
    else:
        buf = OnPolicyReplayBuffer(max_replay_buffer_size=E * T, env_nums=E, time_limit_filter=True)
        pf.to(device); vf.to(device)
        coll = RefVecOnPolicyCollector(vf, pf, env, buf, E * T, device, discount=0.99, max_episode_frames=max_episode_frames)
        coll.epoch_frames = E * T
    log = _Log()
    agent = PPO(pf=pf, vf=vf, plr=1e-4, vlr=1e-4, clip_para=0.2, opt_epochs=2, tau=0.95, entropy_coeff=0.005, shuffle=True,
                collector=coll, replay_buffer=buf, logger=log, device=device, discount=0.99, num_epochs=100, batch_size=B)
    snaps = []
    for ep in range(epochs):
        agent.current_epoch = ep
        torch.manual_seed(50 + ep)    # exploration noise
        out = coll.train_one_epoch()
        np.random.seed(70 + ep)       # minibatch permutations
        agent.update_per_epoch()
        torch.cuda.synchronize()
        snaps.append(dict(acts=np.array(buf._acts, dtype=np.float64).reshape(T, E, -1).copy(),
                          values=np.array(buf._values, dtype=np.float64).reshape(T, E, 1).copy(),
                          rewards=buf._rewards.copy(), terminals=buf._terminals.copy(), time_limits=buf._time_limits.copy(),
                          advs=np.array(buf._advs).copy(), rets=np.array(buf._estimate_returns).copy(),
                          epoch_reward=out["train_epoch_reward"], train_rewards=list(out["train_rewards"])))
    params = {k: v.detach().cpu().clone() for k, v in pf.state_dict().items()}
    return snaps, log.infos, params, np.stack(env.log)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["loco_s84", "mlp_s93"])
def test_fast_collector_equals_reference_protocol(name, mode, device):
    case = util.CASES[name]
    E, T, B = 4, 8, 16
    fast = _run("fast", case, mode, device, E, T, B, 2, 3)
    ref = _run("ref", case, mode, device, E, T, B, 2, 3)
    tol = 2e-5 if mode == "f32" else 2e-2
    assert fast[3].shape == ref[3].shape == (2 * T, E, case["A"])
    d_act = np.abs(fast[3] - ref[3]).max()
    print("\n[collector %s %s] max |action fed to env: fast - reference protocol| = %.2e" % (name, mode, d_act))
    util.record("collector/%s/%s/max_abs_action_diff" % (name, mode), d_act)
    assert d_act <= tol
    for ep, (a, b) in enumerate(zip(fast[0], ref[0])):
        assert np.array_equal(a["terminals"], b["terminals"]) and np.array_equal(a["time_limits"], b["time_limits"])
        assert b["terminals"].sum() > 0
        for k in ("acts", "values", "rewards", "advs", "rets"):
            assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
            d = np.abs(a[k] - b[k]).max()
            assert d <= tol * (10 if k in ("advs", "rets", "rewards") else 1) * max(1.0, np.abs(b[k]).max()), (ep, k, d)
        assert abs(a["epoch_reward"] - b["epoch_reward"]) <= tol * 100
        assert len(a["train_rewards"]) == len(b["train_rewards"])
    # the truncation bootstrap really fired (max_episode_frames = 3): some reward differs from the env's raw reward
    assert len(fast[1]) == len(ref[1]) == 2 * 2 * (E * T // B)
    itol = 5e-4 if mode == "f32" else 3e-2
    for u, (x, y) in enumerate(zip(fast[1], ref[1])):
        for k in util.STAT_KEYS:
            assert abs(x[k] - y[k]) <= itol * max(1.0, abs(y[k])), (u, k, x[k], y[k])
    drift = sum((fast[2][k] - ref[2][k]).abs().sum().item() for k in ref[2]) / sum(v.numel() for v in ref[2].values())
    util.record("collector/%s/%s/mean_abs_param_diff_after_2_epochs" % (name, mode), drift)
    assert drift <= (5e-7 if mode == "f32" else 5e-5), drift


def test_fast_collector_eval_and_uploads(device):
    """eval_one_epoch on the policy mean, pinned double-buffered uploads carry the exact fp32 cast of the float64 rows."""
    os.environ["V4L_COMPUTE"] = "bf16"
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    from vision4leg_amd.torchrl.collector import VecOnPolicyCollector
    from vision4leg_amd.torchrl.replay_buffers import DeviceOnPolicyReplayBuffer
    case = util.CASES["loco_s84"]
    E, T = 4, 4
    torch.manual_seed(0)
    pf, vf = util.build_nets(networks, policies, case)
    env = util.FakeVecEnv(E, case["S"], case["A"], seed=2, p_done=0.3)
    eval_env = util.FakeVecEnv(E, case["S"], case["A"], seed=4, p_done=0.3)
    buf = DeviceOnPolicyReplayBuffer(max_replay_buffer_size=E * T, env_nums=E, time_limit_filter=False)
    coll = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=device, epoch_frames=E * T,
                                eval_episodes=2)
    rows = [np.random.RandomState(i).randn(E, util.obs_dim(case)) for i in range(5)]
    ups = [coll._upload(r).clone() for r in rows]
    for r, u in zip(rows, ups):
        assert np.array_equal(u.cpu().numpy(), r.astype(np.float32))
    out = coll.train_one_epoch()
    assert buf._top == 0 and buf._size == T and np.isfinite(out["train_epoch_reward"])
    # filed rows == what an ingest of the same observation rows produces
    st, im = pf.hip.alloc_rollout(E, device)
    ev = coll.eval_one_epoch()
    assert len(ev["eval_rewards"]) == 2 * E and ev["eval_traj_length"] >= 1 and not eval_env.training
    coll.terminate()
    assert env.closed and eval_env.closed
