"""Is the rollout leg of bench.py bound by the host loop or by the GPU? Times T env steps of RolloutActor.step twice: until the
Python loop returns (launch side) and until the device is idle.  usage: python tools/probe/rollout_host.py [workload]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cnn"
a = bench.parse(["--workload", wl, "--no-cpu-baseline", "--no-parity"])
dev = torch.device("cuda:0")
ep = bench.Epoch(bench.WORKLOADS[wl], a.compute, dev, 1)
ep.rollout()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    ep.rollout()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    T = ep.wl["T"]
    print("%s: T=%d launch side %.2f ms (%.1f us/step), device idle after %.2f ms (%.1f us/step)" %
          (wl, T, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / T, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / T))
