#!/usr/bin/env python
"""Timeline of ONE graph-replayed PPO minibatch update from a rocprofv3 kernel trace (CSV): for every kernel of the update
its start offset, duration and the gap since the latest end of any earlier kernel — where the wall time of an update goes
(critical-path kernels vs dispatch gaps vs overlap of the forked weight-grad branches). Also the same for one rollout step.

usage: python tools/update_timeline.py <dir with *kernel_trace.csv> [out.txt]
An update = from one begin_pack_kernel dispatch to the next one; the median update (by wall time) of the trace is printed."""
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void ", "").replace("v4l::", "")
    if name.startswith("_ZN3v4l"):
        rest = name[7:]
        n = ""
        while rest and rest[0].isdigit():
            n += rest[0]; rest = rest[1:]
        name = rest[:int(n)] if n else rest
    return name.split("<")[0].split("(")[0][:32]


def load(path):
    rows = []
    for p in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    return rows


def segments(rows, opener):
    idx = [i for i, r in enumerate(rows) if r[2].startswith(opener)]
    return [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]


def describe(seg, title, out):
    t0 = seg[0][0]
    out.append("# %s: %d kernels, wall %.1f us (first start -> last end), sum of kernel durations %.1f us"
               % (title, len(seg), (max(r[1] for r in seg) - t0) * 1e-3, sum(r[1] - r[0] for r in seg) * 1e-3))
    out.append("# %-34s %10s %10s %10s" % ("kernel", "start_us", "dur_us", "gap_us"))
    latest = t0
    gaps = 0.0
    for s, e, k in seg:
        gap = (s - latest) * 1e-3
        if gap > 0:
            gaps += gap
        out.append("%-36s %10.1f %10.1f %10.1f" % (k, (s - t0) * 1e-3, (e - s) * 1e-3, gap))
        latest = max(latest, e)
    out.append("# idle gaps inside the segment (no kernel running): %.1f us" % gaps)


def main():
    rows = load(sys.argv[1])
    out = []
    upd = [s for s in segments(rows, "begin_pack") if 15 <= len(s) <= 40]
    if upd:
        upd.sort(key=lambda s: s[-1][1] - s[0][0])
        med = upd[len(upd) // 2]
        nxt = [(s[-1][1] - s[0][0]) * 1e-3 for s in upd]
        out.append("# %d updates in the trace; wall per update (begin_pack -> last kernel end): median %.1f us, min %.1f, max %.1f"
                   % (len(upd), nxt[len(nxt) // 2], nxt[0], nxt[-1]))
        describe(med, "median update", out)
    steps = [s for s in segments(rows, "rollout_encoder2") if len(s) == 2]
    if steps:
        steps.sort(key=lambda s: s[-1][1] - s[0][0])
        describe(steps[len(steps) // 2], "median rollout step", out)
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
