"""the last N dispatches (and memory copies, when traced) of a rocprofv3 trace in time order: offset from the first listed one, duration, queue, name.
python tools/kt_frame.py <results.db> [N=120]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = [(s, e, "q%s" % q, n) for n, s, e, q in c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id")]
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
if mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({mc[0]})")]
    if "size" in cols:
        rows += [(s, e, "copy", "memcpy %d B" % (sz or 0)) for s, e, sz in c.execute(f"select start, end, size from {mc[0]}")]
rows.sort()
rows = rows[-N:]
t0 = rows[0][0]
prev = None
for s, e, q, n in rows:
    print("%9.1f us  %8.1f us  gap %7.1f  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, q, n[:70]))
    prev = e
