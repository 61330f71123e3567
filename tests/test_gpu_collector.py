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


def _run(which, case, mode, device, E, T, B, epochs, max_episode_frames, gae=True):
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
                collector=coll, replay_buffer=buf, logger=log, device=device, discount=0.99, num_epochs=100, batch_size=B, gae=gae)
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
    if which == "fast" and not gae:  # the filed epoch's discounted rewards were computed where the values are (the host path never sets this)
        assert "_advs_dev64" in buf.__dict__
    return snaps, log.infos, params, np.stack(env.log)


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
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


def test_fast_collector_discounted_rewards_stay_on_device(device):
    """PPO(gae=False) (replay_buffers/on_policy.py:47-71) behind the fast collector: DeviceOnPolicyReplayBuffer.discount_reward of a
    filed epoch reads the fp32 values where the rollout kernels filed them (round-5 advisor finding: it fell back to the host path);
    advantages / returns as the reference protocol's, to the values' own agreement."""
    case = util.CASES["mlp_s93"]
    E, T, B = 4, 8, 16
    fast = _run("fast", case, "f32", device, E, T, B, 2, 3, gae=False)
    ref = _run("ref", case, "f32", device, E, T, B, 2, 3, gae=False)
    for a, b in zip(fast[0], ref[0]):
        for k in ("advs", "rets"):
            assert a[k].shape == b[k].shape and np.abs(a[k] - b[k]).max() <= 2e-4 * max(1.0, np.abs(b[k]).max()), k


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


@pytest.mark.parametrize("name", ["cnn_s93", "cnn_vis"])
def test_lost_handover_is_never_silent(name, device):
    """The NatureCNN nets' rollout step hands activations between the blocks of ONE launch through monotonic device-side
    counters with bounded spins (csrc/rollout_dense.h). Force the time-out path — rewind the counters behind the kernel's
    back, so that no wait of the next step can be satisfied — and require that (a) the GPU does not hang, (b) the step's
    actions are NaN (the collector's "NaN detected" check, collector/on_policy.py:102-107, then stops the epoch) instead of
    numbers computed from stale activations, (c) RolloutActor.check() reports it and clears the flag, (d) after the counters
    are put back the actor steps normally again."""
    os.environ["V4L_COMPUTE"] = "bf16"
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = util.CASES[name]
    E = 16
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    pf, vf = pf.to(device), vf.to(device)
    actor = policies.RolloutActor(pf, vf, E)
    rs = np.random.RandomState(1)
    obs = torch.tensor(util.obs_rows(rs, E, case), dtype=torch.float32, device=device)
    good = actor.step(obs, deterministic=True)["action"].clone()
    actor.step(obs, deterministic=True)
    torch.cuda.synchronize()
    assert torch.isfinite(good).all()
    actor.check()  # healthy: no exception
    ctl = actor._actor.ctl  # ActCtl { i64 t; u64 done; u32 seq, err; u32 stage[6] } (csrc/elem.h)
    words = ctl[:48].view(torch.int32)
    saved = words[6:12].clone()
    assert int(saved.max()) >= 32, saved  # 16 tiles x 2 launches went through the counters: the hand-over path is in use
    words[6:12] = 0
    t0 = __import__("time").time()
    bad = actor.step(obs, deterministic=True)["action"].clone()
    torch.cuda.synchronize()
    assert __import__("time").time() - t0 < 60.0
    assert torch.isnan(bad).all(), bad
    with pytest.raises(RuntimeError, match="hand-over"):
        actor.check()
    actor.check()  # cleared
    # the collector's own check on such an action (collector/on_policy.py take_actions)
    assert not np.isfinite(bad.cpu().numpy()).all()
    # put the counters where three launches leave them: the next step is healthy again and equals the first one
    words[6:12] = saved // 2 * 3
    again = actor.step(obs, deterministic=True)["action"].clone()
    torch.cuda.synchronize()
    actor.check()
    assert torch.equal(again, good)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("name", ["loco_s84", "cnn_s93", "mlp_s93", "loco_vis"])
def test_step_host_equals_step(name, mode, device):
    """HipActor.step_host — the fast collector's per-step call: the rollout kernels read the pinned host observation rows in
    place and write the action into pinned host memory — must give bit for bit what step() gives on the same rows in HBM:
    the action and the filed action / value / log pi_old, with per-step draws, with draw_noise() slices, and after attach()
    swapped the rollout arrays."""
    os.environ["V4L_COMPUTE"] = mode
    dt16 = torch.float16 if mode == "f16" else torch.bfloat16
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = util.CASES[name]
    E, T, A = 4, 6, case["A"]
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    pf, vf = pf.to(device), vf.to(device)
    rs = np.random.RandomState(3)
    rows = [util.obs_rows(rs, E, case).astype(np.float32) for _ in range(T)]

    def rollout_arrays():
        st, im = pf.hip.alloc_rollout(T * E, device)
        return st, im, torch.zeros(T * E, A, device=device), torch.zeros(T * E, device=device), torch.zeros(T * E, device=device)

    split = policies.RolloutActor(pf, vf, E).split_supported()
    assert split == (name != "mlp_s93")  # image nets in bf16 take the observation split (bf16 depth rows over PCIe)

    def run(host, arrays):
        actor = policies.RolloutActor(pf, vf, E)
        first = rollout_arrays()
        actor.attach(first)
        actor.seek(0)
        pinned = torch.zeros(E, util.obs_dim(case), dtype=torch.float32).pin_memory()
        acts = []
        for t in range(T):
            if t == 2:             # swap the rollout arrays mid-way: attach() rebuilds the argument tuples
                actor.attach(arrays)
                actor.seek(t)
            if t == 4:
                torch.manual_seed(77)
                actor.draw_noise(T - 4)  # the remaining steps consume slices of one bulk draw
            elif t < 4:
                torch.manual_seed(500 + t)
            if host == "split_pipe":  # the collector's cast -> DMA pipeline (_step_split_pipelined): chunked copies, kernels read HBM
                S = case["S"]
                dprop, dimg = actor.split_device_buffers()
                if S:
                    dprop.copy_(torch.from_numpy(rows[t][:, :S].copy()).pin_memory(), non_blocking=True)
                img16 = torch.from_numpy(rows[t][:, S:].copy()).to(dt16).pin_memory()
                for a0 in range(0, E, 3):  # ragged chunks on purpose (3 + 1 rows)
                    dimg[a0:a0 + 3].copy_(img16[a0:a0 + 3], non_blocking=True)
                acts.append(np.array(actor.step_host_split(dprop if S else None, dimg, on_device=True), copy=True))
            elif host == "rows":  # round 6: the one-call step on float64 rows (cast on the library's pool, launches, completion)
                acts.append(np.array(actor._actor.step_host_rows(rows[t].astype(np.float64), threads=3), copy=True))
            elif host in ("split", "split_copy"):
                S = case["S"]
                prop = torch.from_numpy(rows[t][:, :S].copy()).pin_memory() if S else None
                img16 = torch.from_numpy(rows[t][:, S:].copy()).to(dt16).pin_memory()
                acts.append(np.array(actor._actor.step_host_split(prop, img16, via_copy=(host == "split_copy")), copy=True))
            elif host:
                pinned.copy_(torch.from_numpy(rows[t]))
                acts.append(np.array(actor.step_host(pinned), copy=True))
            else:
                acts.append(actor.step(torch.from_numpy(rows[t]).to(device))["action"].cpu().numpy().copy())
        torch.cuda.synchronize()
        return np.stack(acts), [a.cpu().clone() if a is not None else None for a in arrays], [a.cpu().clone() if a is not None else None for a in first]
    a_dev, filed_dev, first_dev = run(False, rollout_arrays())
    a_host, filed_host, first_host = run(True, rollout_arrays())
    assert np.array_equal(a_dev, a_host)
    for x, y in list(zip(filed_dev, filed_host)) + list(zip(first_dev, first_host)):
        assert (x is None and y is None) or torch.equal(x, y)
    assert np.isfinite(a_host).all() and np.abs(a_host[2:]).max() > 0 and filed_host[4][2 * E:].abs().max() > 0
    if split:  # ... and the split hand-over (fp32 proprio + bf16 depth rows) gives the same bits again
        for how in ("split", "split_copy", "split_pipe", "rows"):  # rows read in place over PCIe / copied to HBM first / chunked DMA pipeline / one library call
            a_split, filed_split, first_split = run(how, rollout_arrays())
            assert np.array_equal(a_dev, a_split), how
            for x, y in list(zip(filed_dev, filed_split)) + list(zip(first_dev, first_split)):
                assert (x is None and y is None) or torch.equal(x, y), how


@pytest.mark.parametrize("name,mode", [("mlp_s93", "f32"), ("mlp_tanh", "bf16"), ("cnn_s93", "f32")])
def test_step_host_returns_only_when_the_observation_buffer_is_free(name, mode, device):
    """step_host reads the caller's pinned observation rows in place. The one-launch step kernels (rollout_mlp_kernel<T> /
    rollout_cnn_kernel, grid (E, 2)) run policy and value blocks side by side, and with more blocks than the chip holds at once
    (E = 1024: 2048 blocks of 1024 threads on 256 CUs) the value blocks are dispatched after the policy blocks: a host that only
    watched the ACTION arrive could overwrite rows the value blocks had not read yet (round-5 advisor finding). The call now
    returns when action and value of every env have arrived. Here ONE pinned buffer is reused and poisoned right after every
    call; the filed values must equal what step() files from the same rows in HBM."""
    os.environ["V4L_COMPUTE"] = mode
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = util.CASES[name]
    E, T, A = (1024 if case["kind"].startswith("mlp") else 384), 3, case["A"]
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    pf, vf = pf.to(device), vf.to(device)
    rs = np.random.RandomState(5)
    rows = [util.obs_rows(rs, E, case).astype(np.float32) for _ in range(T)]

    def arrays():
        st, im = pf.hip.alloc_rollout(T * E, device)
        return st, im, torch.zeros(T * E, A, device=device), torch.zeros(T * E, device=device), torch.zeros(T * E, device=device)

    def run(host):
        actor = policies.RolloutActor(pf, vf, E)
        arr = arrays()
        actor.attach(arr)
        actor.seek(0)
        pinned = torch.zeros(E, util.obs_dim(case), dtype=torch.float32).pin_memory()
        for t in range(T):
            if host:
                pinned.copy_(torch.from_numpy(rows[t]))
                actor.step_host(pinned, deterministic=True)
                pinned.fill_(float("nan"))  # the buffer is the caller's again the moment step_host returns
            else:
                actor.step(torch.from_numpy(rows[t]).to(device), deterministic=True)
        torch.cuda.synchronize()
        return [a.cpu().clone() if a is not None else None for a in arr]
    dev_arr, host_arr = run(False), run(True)
    assert torch.isfinite(host_arr[3]).all() and torch.isfinite(host_arr[2]).all()
    for x, y in zip(dev_arr, host_arr):
        assert (x is None and y is None) or torch.equal(x, y)
