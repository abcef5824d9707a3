"""livo2_lio_frame alone (the bench's extra.lio_frame leg, torch-free): median of 30 calls.  python tools/lio_probe.py [n_raw=24000]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import synth  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
H = importlib.import_module("fast-livo2_amd.configs")
n_raw = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
ctx = livo2.Context(0)
lf = synth.lio_frame_scenario(seed=61, n_raw=n_raw, n_steps=20)
lcfg = H.lidar_cfg(lf.sc)
lst = livo2.State.from_pose(lf.sc.R_prior, lf.sc.t_prior, lf.sc.P)
lst.inv_expo = lf.inv_expo; lst.vel[:] = lf.vel.tolist(); lst.bg[:] = lf.bg.tolist(); lst.ba[:] = lf.ba.tolist(); lst.grav[:] = lf.grav.tolist()
licfg = livo2.ImuCfg()
for k in ("cov_gyr", "cov_acc", "cov_bias_gyr", "cov_bias_acc"):
    getattr(licfg, k)[:] = lf.imu[k]
licfg.cov_inv_expo, licfg.G_m_s2, licfg.mean_acc_norm = lf.imu["cov_inv_expo"], lf.imu["G_m_s2"], lf.imu["mean_acc_norm"]
licfg.ba_bg_est_en = licfg.gravity_est_en = licfg.exposure_estimate_en = 1
ctx.upload_map(lf.sc.fmap)
t = []
for rep in range(35):
    t0 = time.perf_counter()
    lres, lnd, _, _ = ctx.lio_frame(lst, lf.steps, licfg, lf.first_pose, lf.sc.xyz, lf.curvature, synth.AVIA["filter_size_surf"], lcfg, want_poses=False)
    if rep >= 5:
        t.append((time.perf_counter() - t0) * 1e3)
print("lio_frame %d raw points -> %d: one call median %.3f ms (min %.3f), %d iterations" % (n_raw, lnd, np.median(t), min(t), lres.n_iters))
