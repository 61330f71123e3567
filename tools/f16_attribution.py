#!/usr/bin/env python3
"""How far are the 16-bit operand flavours from the fp32 reference arithmetic, and where should the f16 flavour's gradient scale
sit? (CPU only, no GPU minutes.)

For every parity case (tests/util.py CASES) the oracle (oracle/ppo_oracle.py, TEST INFRASTRUCTURE) runs the forward and ONE
PPO.update in the fp32, bf16 and f16 flavours on the same seeded minibatch and reports, against the fp32 run:
  fwd   max |a - b| / max |b| of the policy mean / the value (the distance every parity test uses),
  info  worst |a - b| / max(1, |b|) over the 18 logged scalars of the update,
  grad  relative L2 of the flat parameter gradient (critic, actor),
and for the f16 flavour, per gradient scale 2^k * pow2ceil(B): the same gradient distance, the largest backward operand the
rounding saw (half overflows at 65 504) and the share of backward operands below half's smallest normal (6.1e-5).

usage: python tools/f16_attribution.py [case ...] [--scales 0,2,4,6,8,10] [--out profiles/r6_f16_attribution.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from oracle import ppo_oracle as orc  # noqa: E402
import vision4leg_amd.torchrl.networks as networks  # noqa: E402  (seeded parameter construction only; no kernel runs)
import vision4leg_amd.torchrl.policies as policies  # noqa: E402


class Watch:
    """wraps a rounding function; records what the BACKWARD operands look like (forward operands are not scaled)"""

    def __init__(self, fn):
        self.fn, self.on = fn, False
        self.maxabs, self.n, self.sub, self.inf = 0.0, 0, 0, 0

    def __call__(self, x):
        y = self.fn(x)
        if self.on and x.numel():
            a = x.detach().abs()
            self.maxabs = max(self.maxabs, float(a.max()))
            nz = a > 0
            self.n += int(nz.sum())
            self.sub += int((nz & (a < 6.1035e-5)).sum())
            self.inf += int(torch.isinf(y).sum())
        return y


def flat(gd, keys):
    return torch.cat([gd[k].reshape(-1) for k in keys]).double()


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def one(name, scales):
    case = util.CASES[name]
    kind, S = case["kind"], case["S"]
    b = util.make_batch(case)
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    obs = t(b["obs"])

    def fresh(mode):
        torch.manual_seed(case["seed"])
        pf, vf = util.build_nets(networks, policies, case)
        opf = {k: v.detach().clone() for k, v in pf.state_dict().items()}
        ovf = util.share_encoder(opf, {k: v.detach().clone() for k, v in vf.state_dict().items()}, kind)
        o = orc.PPOOracle(kind, opf, ovf, {k: v.clone() for k, v in opf.items()}, S, mode,
                          clipped_value_loss=case.get("clipped_value_loss", False))
        o.sync_target()
        return o

    def run(mode):
        o = fresh(mode)
        with torch.no_grad():
            mean = o.fwd({k: v for k, v in o.pf.items() if k != "logstd"}, obs, S, mode)
            val = o.fwd(o.vf, obs, S, mode)
        info = o.update(obs, t(b["acts"]), t(b["advs"]), t(b["estimate_returns"]), t(b["values"]), 1e-4, 1e-4)
        return mean, val, info, flat(o.last_grads["pf"], o.pf_keys), flat(o.last_grads["vf"], o.vf_keys)

    ref = run("f32")
    res = {"B": case["B"]}
    for mode in ("bf16", "f16"):
        r = run(mode)
        res[mode] = {"fwd_pf": util.rel_err(r[0], ref[0]), "fwd_vf": util.rel_err(r[1], ref[1]),
                     "info": max(abs(r[2][k] - ref[2][k]) / max(1.0, abs(ref[2][k])) for k in util.STAT_KEYS),
                     "grad_pf": rel_l2(r[3], ref[3]), "grad_vf": rel_l2(r[4], ref[4])}
    sweep = {}
    keep = orc.F16_SCALE_LOG2, orc.ROUND["f16"]
    for k in scales:
        w = Watch(orc.rf16)
        orc.ROUND["f16"] = w
        orc.F16_SCALE_LOG2 = k
        o = fresh("f16")
        # (the watch only counts inside autograd.grad: PPOOracle.update's forwards run first)
        g0 = o._grads

        def grads(loss, pd, keys, n, g0=g0, w=w):
            w.on = True
            try:
                return g0(loss, pd, keys, n)
            finally:
                w.on = False
        o._grads = grads
        o.update(obs, t(b["acts"]), t(b["advs"]), t(b["estimate_returns"]), t(b["values"]), 1e-4, 1e-4)
        sweep[k] = {"grad_pf": rel_l2(flat(o.last_grads["pf"], o.pf_keys), ref[3]),
                    "grad_vf": rel_l2(flat(o.last_grads["vf"], o.vf_keys), ref[4]),
                    "max_operand": w.maxabs, "below_min_normal": w.sub / max(1, w.n), "inf": w.inf}
    orc.F16_SCALE_LOG2, orc.ROUND["f16"] = keep
    res["f16_scale_sweep"] = sweep
    print("== %-12s B = %-5d           %10s %10s %10s %10s %10s" % (name, case["B"], "fwd pf", "fwd vf", "infos", "grad pf", "grad vf"))
    for mode in ("bf16", "f16"):
        r = res[mode]
        print("   %-32s %10.2e %10.2e %10.2e %10.2e %10.2e" % (mode + " vs fp32", r["fwd_pf"], r["fwd_vf"], r["info"], r["grad_pf"], r["grad_vf"]))
    for k, r in sweep.items():
        print("   f16 grad scale 2^%-2d * pow2ceil(B)  grad pf %9.2e vf %9.2e   max operand %9.3g   below min-normal %5.1f %%   inf %d"
              % (k, r["grad_pf"], r["grad_vf"], r["max_operand"], 100 * r["below_min_normal"], r["inf"]))
    sys.stdout.flush()
    return res


def main():
    argv = sys.argv[1:]
    out, scales = None, [0, 2, 4, 6, 8, 10]
    if "--out" in argv:
        i = argv.index("--out"); out = argv[i + 1]; del argv[i:i + 2]
    if "--scales" in argv:
        i = argv.index("--scales"); scales = [int(x) for x in argv[i + 1].split(",")]; del argv[i:i + 2]
    torch.set_num_threads(8)
    res = {name: one(name, scales) for name in (argv or list(util.CASES))}
    if out:
        with open(out, "w") as f:
            json.dump({"what": "oracle flavours against the fp32 oracle (tools/f16_attribution.py): forward max|a-b|/max|b|, worst logged "
                               "scalar of one PPO.update, flat-gradient relative L2; f16 per gradient scale", "cases": res}, f, indent=1)


if __name__ == "__main__":
    main()
