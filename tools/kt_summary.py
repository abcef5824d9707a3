#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite (kernel trace and/or PMC) as text: per-kernel call count / avg / min / max ns and counter averages."""
import sqlite3
import sys


def main(db, tag=""):
    c = sqlite3.connect(db)
    tabs = {r[0] for r in c.execute("select name from sqlite_master")}
    if "kernels" in tabs:
        print(f"# {tag} kernel trace: name, calls, avg_us, min_us, max_us")
        for r in c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name order by sum(end-start) desc"):
            print(f"{r[0][:60]}, {r[1]}, {r[2] / 1e3:.2f}, {r[3] / 1e3:.2f}, {r[4] / 1e3:.2f}")
    if "counters_collection" in tabs:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        if rows:
            print(f"# {tag} counters: kernel, counter, avg per dispatch, dispatches")
            for r in rows:
                print(f"{r[0][:60]}, {r[1]}, {r[2]:.1f}, {r[3]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
