"""every dispatch of a rocprofv3 kernel trace whose name contains one of the given substrings, in time order, with its duration and the gap to the dispatch before:
python tools/kt_list.py <results.db> k_mt_ rocprim"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
prev = None
for n, s, e in rows:
    if any(k in n for k in sys.argv[2:]):
        print("%-44s %9.1f us   gap before %7.1f us" % (n[:44], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
