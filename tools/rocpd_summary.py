#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, the same columns
`rocprofv3 --stats` prints. Usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["# rocprofv3 --kernel-trace --stats summary (ns); total kernel time %.3f ms over %d dispatches"
           % (total / 1e6, sum(r[1] for r in rows)),
           "%-150s %8s %14s %12s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for name, calls, tot, avg, mn, mx in rows:
        out.append("%-150s %8d %14d %12.1f %10d %10d %6.2f%%" % (name[:150], calls, tot, avg, mn, mx, 100.0 * tot / total))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
