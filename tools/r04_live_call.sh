#!/bin/bash
set -u
O=gpurun_out/r04o; mkdir -p $O
python - <<'PY'
import os, tempfile
from scenarios import live_inputs as LI
for size in ("c4",):
    d = os.path.join(tempfile.gettempdir(), f"livo2_live_{size}_v2")
    if not os.path.exists(os.path.join(d, "chain_cfg.bin")):
        LI.write_live_dir(d, LI.make_live(**LI.SIZES[size]))
PY
LIVO2_SHIM_PROF=1 fast-livo2_amd/lib/live_chain /tmp/livo2_live_c4_v2 2>&1 | cut -c1-300
LIVO2_SHIM_PROF=1 fast-livo2_amd/lib/live_chain /tmp/livo2_live_c4_v2 lean 2>&1 | cut -c1-300
