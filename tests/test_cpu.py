"""CPU suite (`-m "not gpu"`): oracle vs the reference's golden outputs, host logic, C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import util
from oracle import ppo_oracle as orc
from oracle.gae_c import gae_c


@pytest.mark.parametrize("name", list(util.CASES))
def test_oracle_matches_reference_golden(name):
    """seeded construction fingerprints + forward + one PPO.update of the oracle vs what the reference produced."""
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = util.CASES[name]
    gold = util.load_golden("ppo_" + name)
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    for tag, net in (("pf", pf), ("vf", vf)):
        sd = net.state_dict()
        assert sorted("init_%s/%s" % (tag, k) for k in sd) == sorted(f for f in gold.files if f.startswith("init_%s/" % tag))
        for k, v in sd.items():
            fp = gold["init_%s/%s" % (tag, k)]  # bit-exact where the fixture was minted; last-ulp LAPACK slack elsewhere
            assert abs(v.double().sum().item() - fp[0]) <= 1e-5 * max(1.0, fp[1]), (tag, k)
            assert abs(v.double().abs().sum().item() - fp[1]) <= 1e-5 * max(1.0, fp[1]), (tag, k)
    opf = {k: v.clone() for k, v in pf.state_dict().items()}
    ovf = util.share_encoder(opf, {k: v.clone() for k, v in vf.state_dict().items()}, case["kind"])
    b = util.make_batch(case)
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    with torch.no_grad():
        om = orc.FORWARDS[case["kind"]]({k: v for k, v in opf.items() if k != "logstd"}, t(b["obs"]), case["S"])
        ov = orc.FORWARDS[case["kind"]](ovf, t(b["obs"]), case["S"])
    assert util.rel_err(om, gold["fwd_mean"]) < 2e-5 and util.rel_err(ov, gold["fwd_value"]) < 2e-5
    oracle = orc.PPOOracle(case["kind"], opf, ovf, {k: v.clone() for k, v in opf.items()}, case["S"],
                           clipped_value_loss=case.get("clipped_value_loss", False))
    oracle.sync_target()
    info = oracle.update(t(b["obs"]), t(b["acts"]), t(b["advs"]), t(b["estimate_returns"]), t(b["values"]), 1e-4, 1e-4)
    for k, g in zip(util.STAT_KEYS, gold["u0/info"]):
        assert abs(info[k] - g) <= 2e-4 * max(1.0, abs(g)), (k, info[k], g)
    for tag, onet in (("pf", opf), ("vf", ovf)):
        for k, v in onet.items():
            key = "u0/%s/%s" % (tag, k)
            if key in gold.files:
                assert np.abs(v.numpy() - gold[key]).max() <= 2e-5, key


def test_bf16_oracle_is_close_to_fp32_oracle():
    """the rounded flavour stays within the expected bf16 distance of the fp32 one."""
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    case = util.CASES["loco_s84"]
    torch.manual_seed(case["seed"])
    pf, _ = util.build_nets(networks, policies, case)
    p = {k: v for k, v in pf.state_dict().items() if k != "logstd"}
    obs = torch.tensor(util.make_batch(case, B=8)["obs"], dtype=torch.float32)
    with torch.no_grad():
        a = orc.loco_forward(p, obs, case["S"], "f32")
        b = orc.loco_forward(p, obs, case["S"], "bf16")
    e = util.rel_err(b, a)
    assert 1e-4 < e < 3e-2, e


@pytest.mark.parametrize("name", list(util.GAE_CASES))
def test_gae_oracles_match_reference_golden(name):
    g = util.GAE_CASES[name]
    ro = util.make_gae_inputs(g)
    gold = util.load_golden("gae")
    oa, orr = orc.gae(ro["rewards"], ro["values"], ro["terminals"], ro["time_limits"], ro["last_value"], g["gamma"],
                      g["tau"], g["tl_filter"])
    assert np.array_equal(oa, gold[name + "/advs"]) and np.array_equal(orr, gold[name + "/rets"])
    T, E = g["T"], g["E"]
    tl = ro["time_limits"].reshape(T, -1)
    ca, cr = gae_c(ro["rewards"].reshape(T, E), ro["values"].reshape(T, E), ro["terminals"].reshape(T, E),
                   tl.reshape(T) if tl.shape[1] == 1 and E > 1 else tl, ro["last_value"], g["gamma"], g["tau"],
                   g["tl_filter"])
    assert np.array_equal(ca, oa.reshape(T, E)) and np.array_equal(cr, orr.reshape(T, E))


@pytest.mark.parametrize("name", list(util.GAE_CASES))
def test_discount_reward_oracles_match_reference_golden(name):
    """PPO(gae=False): the numpy and C restatements of discount_reward (reference replay_buffers/on_policy.py:47-71) against
    what the reference's own buffer produced (tests/golden/make_golden.py), bit for bit."""
    g = util.GAE_CASES[name]
    ro = util.make_gae_inputs(g)
    gold = util.load_golden("gae")
    oa, orr = orc.discount_reward(ro["rewards"], ro["values"], ro["terminals"], ro["time_limits"], ro["last_value"],
                                  g["gamma"], g["tl_filter"])
    assert np.array_equal(oa, gold[name + "/dr_advs"]) and np.array_equal(orr, gold[name + "/dr_rets"])
    T, E = g["T"], g["E"]
    tl = ro["time_limits"].reshape(T, -1)
    ca, cr = gae_c(ro["rewards"].reshape(T, E), ro["values"].reshape(T, E), ro["terminals"].reshape(T, E),
                   tl.reshape(T) if tl.shape[1] == 1 and E > 1 else tl, ro["last_value"], g["gamma"], None, g["tl_filter"])
    assert np.array_equal(ca, oa.reshape(T, E)) and np.array_equal(cr, orr.reshape(T, E))


@pytest.mark.parametrize("name", list(util.OBSNORM_CASES))
def test_obsnorm_oracle_matches_reference_golden(name):
    """oracle/obsnorm_ref.c reproduces the reference Normalizer / NormObs outputs (tests/golden/obsnorm.npz, minted by
    tests/golden/make_golden_obsnorm.py from /root/reference) bit for bit, statistics included."""
    from oracle.obsnorm_c import NormalizerOracle
    case = util.OBSNORM_CASES[name]
    raws, training = util.obsnorm_inputs(case)
    gold = util.load_golden("obsnorm")
    o = NormalizerOracle(case["S"])
    for k, raw in enumerate(raws):
        assert np.array_equal(o.observation(raw, training[k]), gold["%s/y%d" % (name, k)])
    assert np.array_equal(o.mean, gold[name + "/mean"]) and np.array_equal(o.var, gold[name + "/var"])
    assert o.count[0] == gold[name + "/count"][0]
    if case["S"] == 5:  # the wild case really exercises the clip
        assert gold["%s/y%d" % (name, len(raws) - 1)][0, 0] == 10.0 and gold["%s/y%d" % (name, len(raws) - 1)][1, 0] == -10.0


def test_library_exports_every_declared_symbol():
    """include/v4l_hip.h is the contract: every declared entry point must be exported by the built library and
    bound by the ctypes layer (no compute calls here: there is no GPU)."""
    from vision4leg_amd import _lib
    hdr = open(os.path.join(util.ROOT, "include", "v4l_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(v4l_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == _lib.exported_symbols()
    assert os.path.exists(_lib.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(dll, name), name
    assert _lib.lib().v4l_version() >= 104
    for which, mirror in enumerate((_lib.NetCfg, _lib.PPOHyper, _lib.Rollout)):  # (also enforced when the library is loaded)
        assert _lib.lib().v4l_abi_sizeof(which) == ctypes.sizeof(mirror)
    assert _lib.lib().v4l_abi_sizeof(3) == -1


def test_plan_matches_reference_state_dict_names():
    """host-only part of the C ABI: the plan's parameter table equals the reference state_dict keys/shapes."""
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    for name, case in util.CASES.items():
        torch.manual_seed(0)
        pf, vf = util.build_nets(networks, policies, case)
        for net in (pf, vf):
            hip = net.hip
            sd = net.state_dict()
            assert sorted(hip.param_names) == sorted(sd.keys()), name
            for k, shp in zip(hip.param_names, hip.param_shapes):
                assert tuple(sd[k].shape) == shp, (name, k)
            assert hip.total_params == sum(v.numel() for v in sd.values())
    pf, vf = util.build_nets(networks, policies, util.CASES["mlp_s93"])
    assert pf.hip.total_params == 222988  # state MLP pf, SURVEY.md §8a parameter inventory
    case = util.CASES["loco_s93"]
    pf, vf = util.build_nets(networks, policies, case)
    assert (pf.hip.total_params, vf.hip.total_params) == (388780, 387489)


def test_unsupported_configs_fail_loudly():
    import vision4leg_amd.torchrl.networks as networks
    import vision4leg_amd.torchrl.policies as policies
    enc = networks.LocoTransformerEncoder(in_channels=4, state_input_dim=84, hidden_shapes=[256, 256])
    with pytest.raises(NotImplementedError):
        networks.LocoTransformer(encoder=enc, output_shape=1, state_input_shape=84, visual_input_shape=(4, 64, 64),
                                 transformer_params=[[2, 256]], append_hidden_shapes=[256]).hip
    with pytest.raises(NotImplementedError):
        enc(torch.zeros(1, 4, 64, 64), torch.zeros(1, 84))
    net = networks.Net(input_shape=8, output_shape=2, base_type=networks.MLPBase, hidden_shapes=[16],
                       append_hidden_shapes=[16])
    with pytest.raises(RuntimeError):  # CPU tensors never silently fall back
        net(torch.zeros(3, 8))


@pytest.mark.parametrize("dt,eps_log2", [(torch.bfloat16, -8), (torch.float16, -11)])
def test_host_16bit_cast_is_the_two_step_rounding(dt, eps_log2):
    """The fast collector hands the depth stack to the 16-bit rollout kernels as bfloat16 / float16 rows cast on the host (half the
    PCIe bytes). That is only bit-identical to the reference protocol (torch.Tensor(ob): float64 -> float32, then the kernels'
    float32 -> operand type at ingest) if torch's float64 -> 16-bit copy rounds THROUGH float32 — it does (c10::BFloat16 / c10::Half
    are built from float); values where a direct rounding would differ pin it."""
    h = 2.0 ** eps_log2  # half an ulp of 1.0 in the 16-bit type: 1 + h + tiny rounds to 1 + 2h directly, to 1 via float32 (tie to even)
    tricky = np.array([1 + h + 2 ** -30, -(1 + h + 2 ** -40), 3.0 + 3 * h + 2 ** -33, 1 + h - 2 ** -30])
    rs = np.random.RandomState(0)
    x = np.concatenate([tricky, np.clip(rs.randn(100000), -2.5, 2.8)])
    one = torch.empty(len(x), dtype=dt)
    one.copy_(torch.from_numpy(x))
    two = torch.from_numpy(x).to(torch.float32).to(dt)
    assert torch.equal(one, two)
    assert one[0].item() == 1.0  # a direct float64 -> 16-bit rounding would give 1 + 2h
    cols = torch.empty(50, 300, dtype=dt)
    wide = rs.randn(50, 393)
    cols.copy_(torch.from_numpy(wide)[:, 93:])  # a strided column block, as the collector slices the observation rows
    assert torch.equal(cols, torch.from_numpy(wide[:, 93:].copy()).float().to(dt))


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("mode", ["bf16", "f16", "f32"])
def test_library_host_cast_equals_torch_two_step_rounding(mode, threads):
    """v4l_host_cast_rows (csrc/host_step.h; the cast inside v4l_actor_step_rows, the collector's one-call env step): float64 rows
    [E][S + C*H*W] -> fp32 proprio block + depth block in the operand type, on the library's thread pool (AVX-512 where the CPU
    has it, the scalar path otherwise — both run here when the CPU allows). Bit for bit torch.Tensor(ob) (float64 -> float32)
    followed by the float32 -> operand-type rounding of the kernels' ingest, including ties that a direct float64 -> 16-bit
    rounding would decide the other way, subnormals of both types, overflow to inf and NaN; ragged sizes exercise the tails."""
    import ctypes as C
    from vision4leg_amd import _lib
    L = _lib.lib()
    compute = {"f32": _lib.V4L_F32, "bf16": _lib.V4L_BF16, "f16": _lib.V4L_F16}[mode]
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[mode]
    rs = np.random.RandomState(4)
    for E, S, img in ((5, 93, 16384), (3, 0, 4099), (32, 7, 2048 * 3 + 5)):
        rows = np.clip(rs.randn(E, S + img), -2.5, 2.8)
        h = 2.0 ** (-8 if mode == "bf16" else -11)
        tricky = [1 + h + 2 ** -30, -(1 + h + 2 ** -40), 3.0 + 3 * h + 2 ** -33, 1 + h - 2 ** -30, 0.0, -0.0, 6e-8, -3e-8, 5.9e-8, 1e-40, -1e-41,
                  1e-46, 65504.0, 65519.9, 65520.0, -7e4, 3.3e38, 3.5e38, -1e300, np.inf, -np.inf, np.nan, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25]
        rows[0, S:S + len(tricky)] = tricky
        rows[E - 1, -len(tricky):] = tricky
        prop = torch.full((E, max(S, 1)), -7.0, dtype=torch.float32)
        out = torch.zeros(E, img, dtype=dt)
        rc = L.v4l_host_cast_rows(C.c_void_p(rows.ctypes.data), rows.shape[1], E, S, img, C.c_void_p(prop.data_ptr() if S else 0),
                                  C.c_void_p(out.data_ptr()), compute, threads)
        _lib.check(rc, "v4l_host_cast_rows")
        with np.errstate(over="ignore"):
            want32 = torch.from_numpy(rows).to(torch.float32)
        want = want32[:, S:].to(dt)
        bits = lambda t: t.view(torch.int16 if t.element_size() == 2 else torch.int32)
        nan = torch.isnan(want.float())
        assert torch.equal(torch.isnan(out.float()), nan)
        assert torch.equal(bits(out)[~nan], bits(want)[~nan]), (mode, E, S, img)
        if S:
            assert torch.equal(prop[:, :S], want32[:, :S])
    assert L.v4l_host_cast_simd() in (0, 1)


def test_cast_pool_is_clean_under_thread_sanitizer(tmp_path):
    """The host thread pool behind v4l_actor_step_rows (csrc/host_step.h: workers that spin, sleep and wake, claim flags, a job
    object on the caller's stack) under ThreadSanitizer: tools/host/cast_pool_check.cpp runs jobs with changing thread counts,
    pool resizes and pauses longer than the workers' spin window, and compares every element with the scalar two-step rounding."""
    import shutil, subprocess
    cxx = os.environ.get("CXX", "/opt/rocm/lib/llvm/bin/clang++")
    if not os.path.exists(cxx) and shutil.which(cxx) is None:
        pytest.skip("no clang++ for the sanitizer build")
    exe = str(tmp_path / "cast_check_tsan")
    src = os.path.join(os.path.dirname(__file__), "..", "tools", "host", "cast_pool_check.cpp")
    b = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-pthread", src, "-o", exe],
                       capture_output=True, text=True)
    if b.returncode != 0 and "sanitizer" in (b.stderr or "").lower():
        pytest.skip("this toolchain has no ThreadSanitizer runtime")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, "90"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout[-500:], r.stderr[-3000:])


def test_replay_buffer_iteration_matches_reference_semantics():
    """one_iteration: batch_size/E time rows per minibatch, all envs of a row together, np.random stream."""
    from vision4leg_amd.torchrl.replay_buffers import OnPolicyReplayBuffer
    T, E = 8, 4
    buf = OnPolicyReplayBuffer(max_replay_buffer_size=T * E, env_nums=E)
    for t in range(T):
        buf.add_sample({"obs": np.full((E, 3), t) + np.arange(E)[:, None] * 0.1, "acts": np.full((E, 2), t)})
    assert buf._obs.shape == (T, E, 3) and buf._obs.dtype == np.float64
    np.random.seed(0)
    perm = np.random.permutation(T)
    np.random.seed(0)
    batches = list(buf.one_iteration(8, ["obs", "acts"], True))
    assert len(batches) == 4 and batches[0]["obs"].shape == (8, 3)
    assert np.array_equal(batches[0]["acts"][:, 0], np.repeat(perm[:2], E))
    with pytest.raises(AssertionError):
        list(buf.one_iteration(6, ["obs"], True))
    assert np.array_equal(buf.last_sample(["acts"])["acts"], np.full((E, 2), T - 1))


def test_linear_schedule_and_stat_keys():
    from vision4leg_amd import _lib
    from vision4leg_amd.torchrl.algo import utils as atu
    class Opt: param_groups = [{"lr": 1.0}]
    atu.update_linear_schedule(Opt, 750, 1500, 1e-4)
    assert Opt.param_groups[0]["lr"] == 1e-4 - (1e-4 * (750 / 1500.0))
    assert _lib.STAT_KEYS == util.STAT_KEYS and len(_lib.STAT_KEYS) == 18


def test_gae_without_tau_raises_and_discount_reward_has_its_own_entry():
    """PPO's default is tau=None with gae=True; the reference fails on `gamma * None` (replay_buffers/on_policy.py:31). Here a
    forgotten tau must not silently train on discounted rewards: only discount_reward() reaches v4l_discount_reward."""
    import inspect
    from vision4leg_amd import engine
    from vision4leg_amd.torchrl.replay_buffers import OnPolicyReplayBuffer
    buf = OnPolicyReplayBuffer(max_replay_buffer_size=4, env_nums=2, time_limit_filter=False)
    with pytest.raises(TypeError, match="tau is None"):
        buf.generalized_advantage_estimation(np.zeros((2, 1)), 0.99, None)
    z = torch.zeros(2, 2, dtype=torch.float64)
    with pytest.raises(TypeError, match="tau is None"):
        engine.gae(z, z, z, None, torch.zeros(2, dtype=torch.float64), 0.99, None, False)
    assert "v4l_discount_reward" in inspect.getsource(engine.discount_reward)
    assert "v4l_discount_reward" not in inspect.getsource(engine.gae)


def test_cast_threads_env_is_validated(monkeypatch):
    from vision4leg_amd.torchrl.collector import on_policy as coll
    monkeypatch.setenv("V4L_CAST_THREADS", "many")
    with pytest.raises(ValueError, match="V4L_CAST_THREADS"):
        coll._cast_threads_from_env()
    monkeypatch.setenv("V4L_CAST_THREADS", "0")
    with pytest.raises(ValueError, match="V4L_CAST_THREADS"):
        coll._cast_threads_from_env()
    monkeypatch.setenv("V4L_CAST_THREADS", "3")
    assert coll._cast_threads_from_env() == 3


def test_device_identity_ignores_visibility_strings_when_the_gpu_is_identified(monkeypatch):
    """Two ranks that reach ONE physical GPU through different *_VISIBLE_DEVICES strings must get the same fingerprint (the
    'ranks share a GPU' guard in front of ncclCommInitRank); without uuid / PCI ids the index + environment are the fallback."""
    import types
    from vision4leg_amd.torchrl.algo.on_policy import ppo
    props = types.SimpleNamespace(uuid="GPU-12345678-abcd", pci_domain_id=0, pci_bus_id=5, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    a = ppo.device_identity(torch.device("cuda", 0))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert ppo.device_identity(torch.device("cuda", 0)) == a
    props.pci_bus_id = 6
    props.uuid = "GPU-87654321-abcd"
    assert ppo.device_identity(torch.device("cuda", 0)) != a
    # partitions of one GPU reporting the SAME uuid and bus id (CPX / SR-IOV guests): the physical index tells them apart ...
    same = types.SimpleNamespace(uuid="GPU-00000000-0000", pci_domain_id=0, pci_bus_id=5, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: same)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    p0, p1 = ppo.device_identity(torch.device("cuda", 0)), ppo.device_identity(torch.device("cuda", 1))
    assert p0 != p1
    # ... and it is the index the visibility strings RESOLVE to, through both layers: "1" reaches what "0,1"[1] reaches
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert ppo.device_identity(torch.device("cuda", 0)) == p1
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,3")
    assert ppo.physical_device_index(torch.device("cuda", 0)) == 3
    assert ppo.physical_device_index(torch.device("cuda", 0), {"HIP_VISIBLE_DEVICES": "GPU-abc,GPU-def"}) == "GPU-abc"
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    anon = types.SimpleNamespace()
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: anon)
    b0 = ppo.device_identity(torch.device("cuda", 0))
    assert b0 != ppo.device_identity(torch.device("cuda", 1))


def test_header_lists_every_library_switch():
    """include/v4l_hip.h documents the environment switches csrc/v4l_hip.hip reads — the two lists must be the same set."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ""
    for f in sorted(os.listdir(os.path.join(root, "vision4leg_amd", "csrc"))):
        src += open(os.path.join(root, "vision4leg_amd", "csrc", f)).read()
    read = set(re.findall(r'(?:sw_on|sw_int|getenv)\("(V4L_[A-Z0-9_]+)"', src))
    header = open(os.path.join(root, "include", "v4l_hip.h")).read()
    table = header[header.index("Environment switches read by the library"):header.index("Read by the Python shell")]
    listed = set(re.findall(r"V4L_[A-Z0-9_]+", table)) - {"V4L_F32", "V4L_BF16"}
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    m = re.search(r"Environment switches read by the library: (\d+)", header)
    assert int(m.group(1)) == len(read), (m.group(1), len(read))
