"""f16: tapped vs untapped 17-row wave-per-sample forward of the vision-only Transformer at n = 1024 — how many outputs differ, by how much,
and is each variant deterministic run to run? (probe for tests/test_gpu_bench_shapes.py::test_vision_only_transformer_on_wave_per_sample_kernels)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import util
from test_gpu_parity import _build
mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = torch.device("cuda:0")
for n in (32, 1024):
    case = dict(util.CASES["loco_vis"], B=n)
    obs = torch.tensor(util.make_batch(case)["obs"], dtype=torch.float32)
    outs = {}
    for variant in ("wps_taps", "wps17", "wps_taps2", "wps17b"):
        if variant.startswith("wps_taps"):
            os.environ["V4L_LAYER_TAPS"] = "1"
        os.environ["V4L_VIS17"] = "1"
        try:
            pf, vf = _build(case, mode, dev)
            hip = pf.hip
            st, im, _ = hip.stage(obs.to(dev))
            outs[variant] = hip.forward(st, im, n, train=True)[:, :6].cpu().clone()
        finally:
            os.environ.pop("V4L_LAYER_TAPS", None)
            os.environ.pop("V4L_VIS17", None)
    for a, b in (("wps_taps", "wps17"), ("wps_taps", "wps_taps2"), ("wps17", "wps17b")):
        d = (outs[a] - outs[b]).abs()
        print(mode, n, a, "vs", b, "differing elements", int((d > 0).sum()), "of", d.numel(), "rows", int((d.sum(1) > 0).sum()),
              "max abs", float(d.max()), "rel to max", float(d.max() / outs[a].abs().max()))
