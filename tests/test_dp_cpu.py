"""CPU, world_size 2, gloo: the data-parallel update rule algo.PPO uses on N GPUs equals ONE process
updating on the concatenated batch.

Scheme under test (what `PPO._update_phases` + the library's loss scaling implement):
  * every rank holds a shard of n rows; losses are scaled by 1/(n*world) instead of 1/n;
  * the critic gradient buffer carries (sum adv, sum adv^2, count) in its tail -> ONE all_reduce(sum) gives both the
    big-batch critic gradient and the global advantage mean / Bessel std;
  * a second all_reduce(sum) gives the big-batch actor gradient; clip + Adam then run identically on every rank.
The per-rank arithmetic here is the CPU oracle (the HIP kernels need a GPU); the collective pattern, the scaling and
the statistics exchange are exactly the ones of the product path.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from oracle import ppo_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(gs):
    return torch.cat([g.reshape(-1) for g in gs])


def _shard_grads(case, p_pf, p_vf, batch, world, adv_mean, adv_std):
    """critic / actor gradients of one shard with the 1/(n*world) loss scaling."""
    S, kind = case["S"], case["kind"]
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    obs, acts, advs, rets = t(batch["obs"]), t(batch["acts"]), t(batch["advs"]), t(batch["estimate_returns"])
    n = obs.shape[0]
    scale = 1.0 / (n * world)
    vk = list(p_vf)
    for k in vk:
        p_vf[k].requires_grad_(True)
    v = orc.FORWARDS[kind](p_vf, obs, S)
    g_vf = torch.autograd.grad(((v - rets) ** 2).sum() * scale, [p_vf[k] for k in vk])
    for k in vk:
        p_vf[k].requires_grad_(False)
    pk = list(p_pf)
    for k in pk:
        p_pf[k].requires_grad_(True)
    mean = orc.FORWARDS[kind]({k: w for k, w in p_pf.items() if k != "logstd"}, obs, S)
    mean, std, _ = orc.gaussian(mean, p_pf["logstd"])
    lp, ent = orc.log_prob_entropy(mean, std, acts)
    ratio = torch.exp(lp - lp.detach())
    an = (advs - adv_mean) / (adv_std + 1e-5)
    loss = -(torch.min(ratio * an, torch.clamp(ratio, 0.8, 1.2) * an)).sum() * scale - 0.005 * ent.sum() * scale
    g_pf = torch.autograd.grad(loss, [p_pf[k] for k in pk])
    for k in pk:
        p_pf[k].requires_grad_(False)
    return _flat(g_vf), _flat(g_pf)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = dict(util.CASES["mlp_s93"], B=32)
    torch.manual_seed(123 + rank)  # ranks construct DIFFERENT parameters ...
    pf, vf = util.build_nets(networks, policies, case)
    for p in list(pf.parameters()) + list(vf.parameters()):
        dist.broadcast(p.data, src=0)  # ... and PPO broadcasts rank 0's (ppo.py shell, world > 1)
    p_pf = {k: v.clone() for k, v in pf.state_dict().items()}
    p_vf = util.share_encoder(p_pf, {k: v.clone() for k, v in vf.state_dict().items()}, case["kind"])
    batch = util.make_batch(case, update=rank)  # each rank: its own env shard
    advs = torch.tensor(batch["advs"], dtype=torch.float32)
    # critic phase: grads + the shard's advantage moments in ONE buffer / ONE all-reduce — the tail layout of
    # bucket_tail_kernel (csrc/elem.h): [sum a, M2 = sum (a - mean_shard)^2, count, count * mean_shard^2, ...]
    g_vf, _ = _shard_grads(case, p_pf, p_vf, batch, world, 0.0, 1.0)
    m_loc = advs.double().mean()
    tail = torch.stack([advs.double().sum(), ((advs.double() - m_loc) ** 2).sum(), torch.tensor(float(advs.numel()), dtype=torch.float64),
                        advs.numel() * m_loc * m_loc]).float()
    bucket = torch.cat([g_vf, tail, torch.zeros(4)])
    dist.all_reduce(bucket)
    s, m2, c, nm2 = (bucket[-8 + i].item() for i in range(4))
    mean = s / c
    std = max(0.0, (m2 + (nm2 - c * mean * mean)) / (c - 1.0)) ** 0.5  # adv_stats_finalize_kernel
    _, g_pf = _shard_grads(case, p_pf, p_vf, batch, world, mean, std)
    dist.all_reduce(g_pf)
    if rank == 0:
        torch.save({"g_vf": bucket[:-8].clone(), "g_pf": g_pf, "mean": mean, "std": std}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_update_equals_big_batch(tmp_path):
    world, out = 2, str(tmp_path / "dp.pt")
    mp.start_processes(_worker, args=(world, _free_port(), out), nprocs=world, join=True, start_method="spawn")
    got = torch.load(out)
    # single process on the concatenated batch, reference normalisation (ppo.py:148: unbiased std of the big batch)
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = dict(util.CASES["mlp_s93"], B=32)
    torch.manual_seed(123)
    pf, vf = util.build_nets(networks, policies, case)
    p_pf = {k: v.clone() for k, v in pf.state_dict().items()}
    p_vf = util.share_encoder(p_pf, {k: v.clone() for k, v in vf.state_dict().items()}, case["kind"])
    shards = [util.make_batch(case, update=r) for r in range(world)]
    big = {k: np.concatenate([b[k] for b in shards]) for k in shards[0]}
    advs = torch.tensor(big["advs"], dtype=torch.float32)
    mean, std = advs.mean().item(), advs.std().item()
    assert abs(got["mean"] - mean) < 1e-6 and abs(got["std"] - std) < 1e-5
    g_vf, g_pf = _shard_grads(case, p_pf, p_vf, big, 1, mean, std)
    assert util.rel_err(got["g_vf"], g_vf) < 1e-5
    assert util.rel_err(got["g_pf"], g_pf) < 1e-4


# ---- bootstrap of the library's own communicator (V4L_DP_COMM=rccl) and bench.py's self-launcher -----------------------
def _handshake_worker(rank, world, port, out, broken_rank):
    """The PPO.__init__ data-parallel branch up to (not including) ncclCommInitRank: the product's `exchange_comm_id` over
    a gloo group. `available` / `unique_id` stand in for v4l_comm_available / v4l_comm_unique_id (RCCL needs a GPU)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vision4leg_amd.torchrl.algo.on_policy.ppo import exchange_comm_id
    calls = {"id": 0}

    def unique_id():
        calls["id"] += 1
        return bytes((7 * i + 3) % 256 for i in range(128))
    got = exchange_comm_id(dist, torch.device("cpu"), lambda: rank != broken_rank, unique_id)
    torch.save({"id": got, "id_calls": calls["id"]}, "%s.%d" % (out, rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("broken_rank", [-1, 1])
def test_comm_id_handshake_world2(tmp_path, broken_rank):
    """Every rank gets rank 0's 128-byte id (only rank 0 generates one) — or, when ANY rank cannot load RCCL, every rank
    gets None together, so none of them enters the communicator rendezvous alone (a one-sided dlopen failure
    would hang the others in ncclCommInitRank)."""
    world, out = 2, str(tmp_path / "hs")
    mp.start_processes(_handshake_worker, args=(world, _free_port(), out, broken_rank), nprocs=world, join=True,
                       start_method="spawn")
    res = [torch.load("%s.%d" % (out, r), weights_only=False) for r in range(world)]
    if broken_rank >= 0:
        assert all(r["id"] is None for r in res) and all(r["id_calls"] == 0 for r in res)
        return
    want = bytes((7 * i + 3) % 256 for i in range(128))
    assert all(r["id"] == want for r in res)
    assert [r["id_calls"] for r in res] == [1, 0]


def test_bench_launcher_argv():
    """`python bench.py --gpus N ...` re-executes itself as N ranks with the driver's own command line (one process per GPU,
    rendezvous on 127.0.0.1) and hands its flags through unchanged."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2", "--workload", "loco64"]
    cmd = bench.launcher_argv(8, argv, port=29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == argv
    a = bench.parse(cmd[i + 1:])
    assert a.gpus == 8 and a.steps == 5 and a.warmup == 2 and a.workload == "loco64"
    # without --workload the BASELINE configuration follows the GPU count: configs[2] / [3] (32 envs per GPU) up to 4 GPUs,
    # configs[4] (64 envs per GPU) on 8
    assert [bench.parse(["--gpus", str(g)]).workload for g in (1, 2, 4, 8)] == ["loco", "loco", "loco", "loco64"]
    assert bench.parse(["--gpus", "8", "--workload", "cnn"]).workload == "cnn"
    # without an explicit port a free one is picked (or MASTER_PORT is honoured)
    p = int(bench.launcher_argv(2, [])[bench.launcher_argv(2, []).index("--master-port") + 1])
    assert 1024 < p < 65536
