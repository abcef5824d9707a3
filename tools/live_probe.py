"""host/live_chain on the avia-sized live sequence with per-frame stage prints (LIVO2_SHIM_PROF): lean with the map update on the second stream (default) and
with LIVO2_LIVE_SYNC_MAP=1 (the round-5 order).  python tools/live_probe.py [size]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import live_inputs  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "avia"
d = os.path.join(tempfile.gettempdir(), f"livo2_live_{size}_v3")
if not os.path.exists(os.path.join(d, "chain_cfg.bin")):
    live_inputs.write_live_dir(d, live_inputs.make_live(**live_inputs.SIZES[size]))
exe = os.path.join(ROOT, "fast-livo2_amd", "lib", "live_chain")
for rep in range(2):
    for env_add in ({}, {"LIVO2_LIVE_SYNC_MAP": "1"})[:1 if os.environ.get("LIVE_PROBE_ASYNC_ONLY") else 2]:
        env = dict(os.environ, LIVO2_SHIM_PROF="1", **env_add)
        r = subprocess.run([exe, d, "lean"], capture_output=True, text=True, timeout=600, env=env)
        print("==", env_add or "async map update", "rc", r.returncode)
        print(r.stdout.strip()[:600])
        if rep == 1:
            print("\n".join(l for l in r.stderr.splitlines() if l.startswith("frame") or "StateEstimation" in l or "retrieveFrom" in l or "computeJacobian" in l)[-3800:])
