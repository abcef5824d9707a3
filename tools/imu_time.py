"""k_imu_propagate by HIP events: on an otherwise idle chip (the clocks of a device that runs one block at a time sit low) and right behind 30 ms of full-grid LiDAR
launches (the clocks a live frame finds).  python tools/imu_time.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
livo2 = importlib.import_module("fast-livo2_amd")
H = importlib.import_module("fast-livo2_amd.configs")
from scenarios import imu_inputs as I, synth  # noqa: E402


def state(seed):
    rng = np.random.default_rng(seed)
    return livo2.State.from_pose(synth.so3_exp(rng.normal(0, 0.3, 3)), rng.normal(0, 1, 3), synth.prior_cov(rng))


ctx = livo2.Context(0)
sc = synth.lidar_scenario(seed=2, n_points=100000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays_factor=4, downsample=0.1)
cfg = H.lidar_cfg(sc)
ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
prior = livo2.State.from_pose(sc.R_prior, sc.t_prior, sc.P)
for n in (20, 200):
    steps = I.make_steps(0, n=n)
    st = state(0)
    c = livo2.ImuCfg()
    for k in ("cov_gyr", "cov_acc", "cov_bias_gyr", "cov_bias_acc"):
        getattr(c, k)[:] = I.CFG[k]
    c.cov_inv_expo, c.G_m_s2, c.mean_acc_norm = I.CFG["cov_inv_expo"], I.CFG["G_m_s2"], I.CFG["mean_acc_norm"]
    c.ba_bg_est_en = c.gravity_est_en = c.exposure_estimate_en = 1
    cold, warm = [], []
    for _ in range(12):
        ctx.synchronize(); ctx.imu_propagate(st, steps, c); cold.append(ctx.imu_last_kernel_us())
    for _ in range(12):
        ctx.lidar_iterations_async(prior, prior, cfg, 600)                 # ~30 ms of full grids
        ctx.imu_propagate(st, steps, c); warm.append(ctx.imu_last_kernel_us())
    print("k_imu_propagate %3d samples: idle chip median %.1f us (min %.1f) ; behind full-grid launches median %.1f us (min %.1f)" % (n, np.median(cold), min(cold), np.median(warm), min(warm)))
