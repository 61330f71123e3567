"""Samples rocm-smi's sclk while (a) the rollout step loop, (b) the update loop keep the GPU busy."""
import os, sys, subprocess, threading, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import bench
wl = dict(bench.WORKLOADS["loco"])
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
ep = bench.Epoch(wl, "bf16", dev, 1)
ep.step(True)
def sample(tag, fn, secs=3.0):
    stop = [False]; out = []
    def smi():
        while not stop[0]:
            r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
            out.extend(l.strip() for l in r.splitlines() if "sclk" in l)
            time.sleep(0.3)
    th = threading.Thread(target=smi); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        fn(); torch.cuda.synchronize(); n += 1
    stop[0] = True; th.join()
    print(tag, "iterations", n, "ms/iter %.2f" % (1e3 * (time.time() - t0) / n)); print("  ", out[-4:])
sample("rollout", ep.rollout)
sample("update", ep.update)
