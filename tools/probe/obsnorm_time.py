"""Times v4l_obs_norm at the BASELINE geometries (HIP events, 200 launches) next to the host numpy wrapper path."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision4leg_amd.torchrl.env import NormObsWithImg
dev = torch.device("cuda:0")
for E in (16, 32, 64):
    S, IMG = 93, 4 * 64 * 64
    env = NormObsWithImg(S, IMG, E, device=dev)
    raw = torch.randn(E, S, dtype=torch.float64, device=dev)
    for name, img in (("f32 img", torch.randn(E, IMG, device=dev)), ("f64 img", torch.randn(E, IMG, dtype=torch.float64, device=dev))):
        for _ in range(20):
            env.observation(raw, img)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200):
            env.observation(raw, img)
        e1.record(); torch.cuda.synchronize()
        print("E=%d %s: %.2f us/step on device" % (E, name, e0.elapsed_time(e1) * 1e3 / 200))
    # host path of the reference wrapper: np.mean/np.var/merge/filt + hstack of the image row
    r, im = raw.cpu().numpy(), np.random.randn(E, IMG)
    mean, var, cnt = np.zeros(S), np.ones(S), 1e-4
    t0 = time.perf_counter()
    for _ in range(200):
        bm, bv = r.mean(0), r.var(0)
        d = bm - mean; tot = cnt + E
        mean = mean + d * E / tot; var = (var * cnt + bv * E + np.square(d) * cnt * E / tot) / tot; cnt = tot
        ob = np.hstack([np.clip((r - mean) / (np.sqrt(var) + 1e-4), -10, 10), im])
        t = torch.Tensor(ob)
    print("E=%d host numpy wrapper + torch.Tensor(ob): %.1f us/step" % (E, (time.perf_counter() - t0) * 1e6 / 200))
