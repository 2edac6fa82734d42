#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof/*/…_results.db) into the
small text/CSV files committed under profiles/.

    python tools/rocprof_summary.py <kernel-trace.db> [<pmc.db> ...] > profiles/rNN_<what>.md
"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) <= 70 else name[:67] + "..."


def main(paths):
    print("| kernel | calls | total_us | avg_us | pct |")
    print("|---|---|---|---|---|")
    con = sqlite3.connect(paths[0])
    for name, calls, total, avg, pct in con.execute("select * from top_kernels"):
        if pct < 0.01:
            continue
        print(f"| `{short(name)}` | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |")
    for p in paths[1:]:
        con = sqlite3.connect(p)
        print(f"\nPMC pass `{p}` (per-dispatch average of our kernels):\n")
        print("| kernel | counter | dispatches | avg | min | max |")
        print("|---|---|---|---|---|---|")
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
             "from counters_collection where (kernel_name like '%kb::%' or kernel_name like '%rocprim%') group by kernel_name, counter_name")
        for k, c, n, a, lo, hi in con.execute(q):
            print(f"| `{short(k)}` | {c} | {n} | {a:.6g} | {lo:.6g} | {hi:.6g} |")


if __name__ == "__main__":
    main(sys.argv[1:])
