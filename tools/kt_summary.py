#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite (kernel trace and/or PMC) as text: per-kernel call count / avg / min / max and counter averages.
--split-us T: launches shorter than T microseconds (the early-exit launches of a converged update: `if (hdr.stop) return`) are reported on their own, so
that the average of the EXECUTING launches is visible next to rocprofv3's all-launch average."""
import sqlite3
import sys


def main(db, tag="", split_us=None):
    c = sqlite3.connect(db)
    tabs = {r[0] for r in c.execute("select name from sqlite_master")}
    if "kernels" in tabs:
        print(f"# {tag} kernel trace: name, calls, avg_us, min_us, max_us" + (f" [>= {split_us} us: calls, avg_us | < {split_us} us: calls, avg_us]" if split_us else ""))
        for r in c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"):
            line = f"{r[0][:60]}, {r[1]}, {r[2] / 1e3:.2f}, {r[3] / 1e3:.2f}, {r[4] / 1e3:.2f}"
            if split_us:
                thr = split_us * 1e3
                a = c.execute("select count(*), avg(end-start) from kernels where name = ? and (end-start) >= ?", (r[0], thr)).fetchone()
                b = c.execute("select count(*), avg(end-start) from kernels where name = ? and (end-start) < ?", (r[0], thr)).fetchone()
                line += f" [{a[0]}, {(a[1] or 0) / 1e3:.2f} | {b[0]}, {(b[1] or 0) / 1e3:.2f}]"
            print(line)
    if "counters_collection" in tabs:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*), max(value) from counters_collection group by kernel_name, counter_name").fetchall()
        if rows:
            print(f"# {tag} counters: kernel, counter, avg per dispatch, dispatches, max per dispatch")
            for r in rows:
                print(f"{r[0][:60]}, {r[1]}, {r[2]:.1f}, {r[3]}, {r[4]:.1f}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    split = None
    if "--split-us" in args:
        i = args.index("--split-us"); split = float(args[i + 1]); del args[i:i + 2]
    main(args[0], args[1] if len(args) > 1 else "", split)
