#!/usr/bin/env python3
"""Which contractions carry the bf16 mode's distance to the fp32 reference? (CPU only, no GPU minutes.)

The oracle (oracle/ppo_oracle.py, TEST INFRASTRUCTURE) runs the forward + backward of one net with exactly ONE contraction
group's operands rounded to bf16 (fp32 accumulate — the rounding points of the HIP bf16 path), every other group in fp32,
and reports that run's distance to the all-fp32 run: the forward output (max |a - b| / max |b|, the distance every parity test
uses) and the flat parameter gradient of a fixed probe loss (relative L2, cosine). Also: all groups rounded (= the bf16 oracle),
each group left OUT of the all-bf16 run, and the candidate precision maps built from the costliest groups.

usage: python tools/bf16_attribution.py [case ...] [--out profiles/r5_bf16_attribution.json]   (default case: loco_b1024)
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

# forward MACs per sample of each group (SURVEY.md 8a/8d, S = 93, A = 6) -> its share of the net's FLOPs
MACS = {"conv": 3612672, "upconv": 65536, "proprio": 23808 + 65536 + 16384, "in_proj": 2 * 208896, "attn": 2 * 2 * 18496,
        "out_proj": 2 * 69632, "ffn": 2 * 2 * 278528, "heads": 99840}


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def run(kind, params, obs, S, w, mode):
    q = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = orc.FORWARDS[kind](q, obs, S, mode)
    keys = list(q)
    g = torch.autograd.grad((out * w).sum(), [q[k] for k in keys], allow_unused=True)
    flat = torch.cat([(torch.zeros_like(q[k]) if x is None else x).reshape(-1) for k, x in zip(keys, g)]).double()
    return out.detach(), flat


def attribute(name, threads):
    case = util.CASES[name]
    torch.manual_seed(case["seed"])
    pf, vf = util.build_nets(networks, policies, case)
    b = util.make_batch(case)
    obs = torch.tensor(b["obs"], dtype=torch.float32)
    res = {}
    for tag, net in (("pf", pf), ("vf", vf)):
        params = {k: v.detach().clone() for k, v in net.state_dict().items() if k != "logstd"}
        A = case["A"] if tag == "pf" else 1
        w = torch.tensor(np.random.RandomState(5).randn(case["B"], A), dtype=torch.float32)
        out0, g0 = run(case["kind"], params, obs, case["S"], w, "f32")
        groups = [g for g in orc.GROUPS if g in MACS]
        rows = {}

        def dist(mode):
            out, g = run(case["kind"], params, obs, case["S"], w, mode)
            return {"fwd": util.rel_err(out, out0), "grad_l2": rel_l2(g, g0),
                    "grad_cos": float((g @ g0) / (g.norm() * g0.norm()))}
        rows["all_bf16"] = dist("bf16")
        for g in groups:
            rows["only_" + g] = dist({g: "bf16"})
        for g in groups:
            rows["all_but_" + g] = dist({h: "bf16" for h in groups if h != g})
        # candidate maps: the cheap groups (< 5 % of the FLOPs each) in fp32, the rest bf16
        cheap = [g for g in groups if MACS[g] / sum(MACS.values()) < 0.05]
        rows["map_cheap_f32(%s)" % "+".join(cheap)] = dist({h: "bf16" for h in groups if h not in cheap})
        res[tag] = rows
        print("== %s %s (B = %d): distance to the all-fp32 oracle" % (name, tag, case["B"]))
        print("   %-44s %10s %10s %12s %8s" % ("contractions rounded to bf16", "forward", "grad L2", "1 - cos", "FLOP %"))
        tot = sum(MACS.values())
        for k, r in rows.items():
            share = 100.0 * MACS[k[5:]] / tot if k.startswith("only_") else float("nan")
            print("   %-44s %10.2e %10.2e %12.2e %8.1f" % (k, r["fwd"], r["grad_l2"], 1 - r["grad_cos"], share))
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = None
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
        args = [a for a in args if a != out]
    torch.set_num_threads(8)
    res = {name: attribute(name, 8) for name in (args or ["loco_b1024"])}
    if out:
        with open(out, "w") as f:
            json.dump({"what": "distance to the fp32 oracle with ONE contraction group's operands rounded to bf16 "
                               "(tools/bf16_attribution.py); fwd = max|a-b|/max|b|, grad = flat gradient of a probe loss",
                       "flop_share": {g: MACS[g] / sum(MACS.values()) for g in MACS}, "cases": res}, f, indent=1)


if __name__ == "__main__":
    main()
