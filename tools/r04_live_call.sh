#!/bin/bash
set -u
export TMPDIR=/tmp
python - <<'PY'
import os, tempfile
from scenarios import live_inputs as LI
d = os.path.join(tempfile.gettempdir(), "livo2_live_c4_v3")
if not os.path.exists(os.path.join(d, "chain_cfg.bin")):
    LI.write_live_dir(d, LI.make_live(**LI.SIZES["c4"]))
PY
rm -rf /tmp/kt; rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kt -o kt -- fast-livo2_amd/lib/live_chain /tmp/livo2_live_c4_v3 lean > /tmp/lc.out 2>&1; grep -a "^live_chain" /tmp/lc.out | cut -c1-200
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/kt/**/*results.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
res = [(s, e) for n, s, e in rows if 'k_lidar_residual' in n]
print("residual launches:", len(res))
prev_end = None
allk = [(s, e, n) for n, s, e in rows]
import bisect
starts = [s for s, e, n in allk]
for k, (s, e) in enumerate(res):
    i = bisect.bisect_left(starts, s)
    pe = allk[i - 1][1] if i > 0 else s
    pn = allk[i - 1][2][:30] if i > 0 else ''
    print("  residual %2d start %9.3f ms dur %7.1f us, gap after previous kernel (%s) %9.1f us" % (k, (s - t0) / 1e6, (e - s) / 1e3, pn, (s - pe) / 1e3))
mc = [t for t in tabs if 'memory_cop' in t or 'memcpy' in t.lower()]
print("memcpy tables:", mc)
for t in mc[:1]:
    cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
    print(cols)
    big = c.execute(f"select * from {t} order by (end-start) desc limit 8").fetchall()
    for b in big: print("   ", [x for x in b][:12])
PY
