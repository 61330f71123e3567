"""-m gpu: world_size 2 THROUGH THE PRODUCT CODE on one MI355X (reference semantics: N ranks with batch n ==
one process with batch N*n, torchrl/algo/on_policy/ppo.py:125-153).

Two spawned processes share cuda:0 and form a gloo process group (RCCL refuses two ranks on one device; gloo all-reduces
CUDA tensors through the host). Each constructs the real `algo.PPO` — world_size = 2 -> rank 0's parameters are broadcast
(the ranks are seeded differently on purpose), the library's communicator is declined by all ranks together ("ranks share a
GPU") and the update runs as `_update_phases`: critic_grads -> bucket_tail(pack) -> all_reduce -> bucket_tail(unpack) ->
critic_step -> actor_grads -> ... on each rank's own minibatch rows — three updates. The result must be what ONE trainer
produces on the concatenated 2n-row minibatches: parameters, advs/*, grad_norm/*, losses, and the shard-local statistics
combined over the two ranks.
"""
import os
import socket

import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu

ROWS, N, U = 64, 16, 3     # rollout rows per rank, minibatch rows per rank, updates
LR = 1e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(case, world):
    rs = np.random.RandomState(77)
    D = util.obs_dim(case)
    tot = world * ROWS
    obs = np.concatenate([np.clip(rs.randn(tot, case["S"]), -10, 10), np.clip(rs.randn(tot, D - case["S"]), -2.5, 2.8)], 1)
    acts, advs, rets = 0.1 * rs.randn(tot, case["A"]), rs.randn(tot), rs.randn(tot)
    rows = np.stack([[rs.permutation(ROWS)[:N] for _ in range(U)] for _ in range(world)]).astype(np.int32)  # [rank][u][n]
    return obs, acts, advs, rets, rows


def _agent(case, mode, device, seed, batch):
    os.environ["V4L_COMPUTE"] = mode
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    from vision4leg_amd.torchrl.algo import PPO
    torch.manual_seed(seed)
    pf, vf = util.build_nets(networks, policies, case)

    class Coll: epoch_frames = ROWS
    agent = PPO(pf=pf, vf=vf, plr=LR, vlr=LR, clip_para=0.2, opt_epochs=1, tau=0.95, entropy_coeff=0.005,
                collector=Coll(), device=device, batch_size=batch)
    return agent


def _run(agent, case, device, obs, acts, advs, rets, rows):
    from vision4leg_amd.engine import HipTrainer
    net = agent.pf.hip
    net.ensure_bound()
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    state, image = net.alloc_rollout(len(obs), device)
    net.ingest(t(obs), state, image)
    ro = HipTrainer.rollout(state, image, t(acts), t(advs), t(rets), t(rets))
    stats = torch.zeros(len(rows), 24, device=device)
    agent.trainer.sync_target()
    agent.run_updates(ro, torch.tensor(rows, device=device), stats)
    torch.cuda.synchronize()
    sd = {"pf." + k: v.detach().cpu().clone() for k, v in agent.pf.state_dict().items()}
    sd.update({"vf." + k: v.detach().cpu().clone() for k, v in agent.vf.state_dict().items()})
    return sd, stats.cpu().numpy()


def _worker(rank, world, port, out, mode, name):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("V4L_DP_COMM", None)          # the default: try the library's communicator, fall back together
    os.environ.pop("V4L_FORCE_DP_PHASES", None)
    import torch.distributed as dist
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = util.CASES[name]
    agent = _agent(case, mode, device, seed=100 + rank, batch=N)   # different parameters per rank until PPO broadcasts
    assert agent.world_size == world and agent.dp_phases and not agent.dp_in_library and not agent.trainer.has_comm
    assert "share a GPU" in agent.dp_comm_note, agent.dp_comm_note
    obs, acts, advs, rets, rows = _data(case, world)
    lo, hi = rank * ROWS, (rank + 1) * ROWS
    sd, stats = _run(agent, case, device, obs[lo:hi], acts[lo:hi], advs[lo:hi], rets[lo:hi], rows[rank])
    torch.save({"sd": sd, "stats": stats, "note": agent.dp_comm_note, "updates": agent.training_update_num},
               "%s.%d" % (out, rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["loco_s84"])
def test_two_processes_one_gpu_equal_one_big_batch(name, mode, device, tmp_path):
    import torch.multiprocessing as mp
    world, out = 2, str(tmp_path / "dp2")
    mp.start_processes(_worker, args=(world, _free_port(), out, mode, name), nprocs=world, join=True, start_method="spawn")
    res = [torch.load("%s.%d" % (out, r), weights_only=False) for r in range(world)]
    assert all(r["updates"] == U for r in res)
    # every rank applied the same all-reduced gradient: replicas stay bit-identical
    for k, v in res[0]["sd"].items():
        assert torch.equal(v, res[1]["sd"][k]), k
    # ONE process, world 1, on the concatenated minibatches (rank 0's seed: its parameters are the broadcast ones)
    case = util.CASES[name]
    for k in ("V4L_DP_COMM", "V4L_FORCE_DP_PHASES"):
        os.environ.pop(k, None)
    agent = _agent(case, mode, device, seed=100, batch=world * N)
    assert agent.world_size == 1 and not agent.dp_phases
    obs, acts, advs, rets, rows = _data(case, world)
    big_rows = np.concatenate([rows[r] + r * ROWS for r in range(world)], axis=1)   # [u][2n]
    sd, stats = _run(agent, case, device, obs, acts, advs, rets, big_rows)
    K = {k: i for i, k in enumerate(util.STAT_KEYS)}
    s0, s1 = res[0]["stats"], res[1]["stats"]
    rt = 1e-5 if mode == "f32" else 2e-3   # bf16: from the second update on the operands' rounding sees 1e-7 parameter differences
    for u in range(U):
        tol = rt * (1 if u == 0 or mode == "f32" else 10)
        # global quantities: both ranks hold the big batch's value
        for k in ("advs/mean", "advs/std", "Training/vf_loss", "grad_norm/vf", "Training/policy_loss", "grad_norm/pf"):
            for s in (s0, s1):
                assert abs(s[u, K[k]] - stats[u, K[k]]) <= 10 * tol * max(1.0, abs(stats[u, K[k]])), (u, k, s[u, K[k]], stats[u, K[k]])
        assert s0[u, K["advs/mean"]] == s1[u, K["advs/mean"]] and s0[u, K["grad_norm/pf"]] == s1[u, K["grad_norm/pf"]]
        # shard-local extrema / means combine to the big batch's
        for k, f in (("advs/max", max), ("advs/min", min), ("logprob/max", max), ("logprob/min", min), ("ratio/max", max),
                     ("ratio/min", min), ("logprob/mean", lambda a, b: 0.5 * (a + b))):
            got = f(float(s0[u, K[k]]), float(s1[u, K[k]]))
            assert abs(got - stats[u, K[k]]) <= 10 * tol * max(1.0, abs(stats[u, K[k]])), (u, k, got, stats[u, K[k]])
    worst, tot, cnt = 0.0, 0.0, 0
    for k, v in sd.items():
        d = (v - res[0]["sd"][k]).abs()
        worst = max(worst, d.max().item()); tot += d.sum().item(); cnt += d.numel()
    util.record("dp2/%s/%s/worst_abs_param_vs_big_batch" % (name, mode), worst)
    util.record("dp2/%s/%s/mean_abs_param_vs_big_batch" % (name, mode), tot / cnt)
    print("\n[dp2 %s %s] %s | params vs one big-batch trainer after %d updates: max %.2e mean %.2e"
          % (name, mode, res[0]["note"], U, worst, tot / cnt))
    # an Adam step moves an element by <= lr; a gradient element below the summation noise may pick the other sign (2 lr per update)
    assert worst <= 2.1 * LR * U and tot / cnt <= (2e-7 if mode == "f32" else 2e-5) * U
