"""-m gpu: the configuration bench.py times, against the pinned oracle fed the REFERENCE protocol.

bench.py's step = RolloutActor.step at batch E (stored action / value / log pi_old) followed by hipGraph-replayed B = 1024
updates that read the stored log pi_old instead of evaluating the frozen target policy (SURVEY 8d's declared saving). The
oracle (oracle/bench_path.py) does what the reference does: pf.explore / vf per env step (collector/on_policy.py:90-100),
Normal(mean, std).log_prob (policies/continuous_policy.py:127-146) and a target_pf forward inside every minibatch update
(algo/on_policy/ppo.py:34,55-59).

Gates (tensor distances are relative to the reference tensor's max-abs — policy means are O(1e-2), nothing is floored at 1)
  f32 : rollout mean / value <= 2e-5, log pi_old <= 2e-4 (abs), the 18 infos of every update <= 5e-4, parameters: all but <= 1e-3
        of the elements within half an Adam step (5e-5), none further than 2 lr U, mean <= 5e-6 (see the comment at the gate).
  bf16: rollout mean / value within 3 x the bf16 oracle's own sensitivity (p95 of 8 forward passes with parameters nudged by
        1e-7, oracle/bench_path.py::_forward_envelope), update 0's infos within 1e-2 of the bf16 oracle's, distances recorded
        (profiles/parity_r4.json), and the TRAJECTORY GATE: per update, HIP-bf16 is no further from the bf16 oracle than 3 x the
        spread of 4 bf16 oracles started from parameters nudged by 1e-7 (oracle/bench_path.py::traj_seeds). (Round 3's rule —
        "no further from the fp32 reference trajectory than 2 x the bf16 oracle is" — leaned on the bf16-vs-fp32 distance; that
        distance is now recorded only.)
"""
import pytest

import util
from oracle import bench_path

pytestmark = pytest.mark.gpu

TRAJ_SEEDS, TRAJ_FACTOR, TRAJ_FLOOR = 4, 3.0, 2e-3


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [32, 64])
def test_rollout_then_stored_logp_graph_updates_vs_reference_protocol(E, mode, device, monkeypatch):
    case = dict(util.CASES["loco_b1024"])
    T, B, U = 64, 1024, 4
    r = bench_path.run(case, E, T, B, U, mode, device, threads=16, traj_seeds=TRAJ_SEEDS)
    params = r.pop("_params")
    tag = "bench_path/E%d/%s/" % (E, mode)
    for k, v in r.items():
        if isinstance(v, float):
            util.record(tag + k, v)
    print("\n[bench path E=%d %s] %s" % (E, mode, {k: v for k, v in r.items() if k != "path"}))
    assert r["finite"] and r["graph_replays"]
    assert r["rollout_logp_abs_vs_own_normal"] <= 2e-5
    if mode == "f32":
        assert r["rollout_mean_vs_f32"] <= 2e-5 and r["rollout_value_vs_f32"] <= 2e-5 and r["rollout_std_vs_f32"] <= 1e-6
        assert r["rollout_logp_abs_vs_f32"] <= 2e-4  # (a - mu)^2 / (2 sigma^2) with sigma = 0.125 amplifies d(mu) by ~ |a - mu| / sigma^2 = 8
        assert r["infos_vs_f32"] <= 5e-4, (r["infos_vs_f32_worst_key"], r["infos_vs_f32_per_update"])
        # Parameters after U Adam steps. Adam normalises every gradient element by its own magnitude: an element whose true
        # value is below the fp32 summation noise (partial sums that cancel to ~1e-9) gets a step of up to +-lr whose SIGN is
        # that noise — and every last-bit difference upstream (the shared encoder's weights after the critic's step, the order a
        # weight-grad kernel sums its blocks in: dW3 on 64 / 96 / 128 blocks gives 0 / 130 / 0 such elements here) re-rolls it.
        # So: the bulk within half a step (5e-5), the stragglers few (<= 1e-3 of the elements) and bounded by 2 lr U, the mean
        # tight. tools/probe/dw3_blocks.py shows the gradients themselves agree to 1e-6 across those block counts.
        assert r["param_frac_above_5e-5_vs_f32"] <= 1e-3 and r["param_max_vs_f32"] <= 2 * 1e-4 * U, r["param_worst_key_vs_f32"]
        assert r["param_mean_vs_f32"] <= 5e-6
        if E == 32:
            # yardstick, recorded: the SAME kernels with dW3's partial sums taken over a different number of blocks (a last-bit
            # change of one gradient tensor) — how many parameters of the product itself move by more than half a step
            monkeypatch.setenv("V4L_CONV3_WGRAD_BLOCKS", "64")
            other = bench_path.run(case, E, T, B, U, mode, device, threads=16).pop("_params")
            d = (params - other).abs()
            util.record(tag + "self_frac_above_5e-5_other_dW3_blocks", (d > 5e-5).double().mean().item())
            util.record(tag + "self_max_other_dW3_blocks", d.max().item())
            print("[bench path E=%d f32] product vs product with dW3 on 64 blocks: max %.2e, %.1e of the elements > 5e-5"
                  % (E, d.max().item(), (d > 5e-5).double().mean().item()))
        return
    # 16-bit modes: where the oracle with the same operand rounding sits ...
    m = mode
    assert r["rollout_mean_vs_" + m] <= max(3.0 * r["rollout_mean_envelope_p95"], 1e-4), (r["rollout_mean_vs_" + m], r["rollout_mean_envelope_p95"])
    assert r["rollout_value_vs_" + m] <= max(3.0 * r["rollout_value_envelope_p95"], 1e-4), (r["rollout_value_vs_" + m], r["rollout_value_envelope_p95"])
    assert r["infos_vs_%s_per_update" % m][0] <= 1e-2, r["infos_vs_%s_per_update" % m]
    if m == "f16":  # the literal gate on the path the bench times: rollout outputs and the first update's logged scalars within 1e-3
        # of the fp32 reference arithmetic (B = 1024: profiles/r6_f16_attribution.txt has the oracle at 7.9e-4 / 6.8e-4)
        assert r["rollout_mean_vs_f32"] <= 1e-3 and r["rollout_value_vs_f32"] <= 1e-3, (r["rollout_mean_vs_f32"], r["rollout_value_vs_f32"])
        assert r["infos_vs_f32_per_update"][0] <= 1e-3, r["infos_vs_f32_per_update"]
    # ... and the trajectory gate. From the second update on a bf16 trajectory is not unique: grad_norm/pf jumps by a few per
    # cent whenever ONE sample changes sides of the PPO clip, and which update that happens in differs between any two bf16
    # evaluations. The yardstick is therefore the bf16 ORACLE against itself: TRAJ_SEEDS oracles started from parameters nudged
    # by 1e-7 run the same updates, and per update the HIP infos may be no further from the un-nudged bf16 oracle's than
    # TRAJ_FACTOR x the largest distance among them (+ a floor for updates where no decision happened to flip). The distance to
    # the fp32 reference trajectory is recorded (profiles/parity_r4.json), not gated; tests/test_gpu_soak.py shows it does not
    # compound over 360 updates.
    hip, env = r["infos_vs_%s_per_update" % m], r["traj_envelope_per_update"]
    print("[bench path E=%d %s] infos vs the %s oracle per update %s, nudged-oracle envelope %s" % (E, m, m, hip, env))
    for u in range(U):
        util.record(tag + "u%d/infos_vs_%s" % (u, m), hip[u])
        util.record(tag + "u%d/traj_envelope" % u, env[u])
        assert hip[u] <= TRAJ_FACTOR * env[u] + TRAJ_FLOOR, (u, hip, env)
    pm_env = r["oracle_%s_vs_f32_param_mean" % m]
    util.record(tag + "param_mean_vs_f32_over_oracle_bf16_vs_f32", r["param_mean_vs_f32"] / max(pm_env, 1e-12))
    assert r["param_mean_vs_" + m] <= 2e-5 * U  # mean |param - bf16 oracle| after U Adam steps of 1e-4 (test_ppo_update: 2e-5 per update)
