#!/bin/bash
set -u
O=gpurun_out/r04p; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY'
import os, tempfile
from scenarios import live_inputs as LI
for size in ("avia", "c4"):
    d = os.path.join(tempfile.gettempdir(), f"livo2_live_{size}_v2")
    if not os.path.exists(os.path.join(d, "chain_cfg.bin")):
        LI.write_live_dir(d, LI.make_live(**LI.SIZES[size]))
PY
for size in avia c4; do
  LIVO2_SHIM_PROF=1 fast-livo2_amd/lib/live_chain /tmp/livo2_live_${size}_v2 2>&1 | grep -a "frame \|live_chain" | cut -c1-330
done
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- fast-livo2_amd/lib/live_chain /tmp/livo2_live_c4_v2 lean > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/kt/**/*results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
slow = [(n[:50], (s - t0) / 1e6, (e - s) / 1e3) for n, s, e in rows if (e - s) > 300e3]
print("kernels longer than 300 us (name, start ms, duration us):")
for r in slow[:60]: print("  %-50s %10.3f %10.1f" % r)
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows: agg[n[:50]][0] += 1; agg[n[:50]][1] += (e - s) / 1e3
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]: print("  %-50s calls %5d total %10.1f us" % (n, k, t))
PY
python -m pytest tests/test_host_shim_gpu.py tests/test_live_chain_gpu.py tests/test_sequence_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
