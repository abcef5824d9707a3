"""A short LIO sequence through the C++ shim the way LIVMapper::handleLIO runs it (reference src/LIVMapper.cpp:357-426): per frame
StateEstimation on the propagated state, world points / covariances of the scan from the posterior, UpdateVoxelMap — against the same chain on
the oracle.  Everything composes: resident map re-flattened after every map update, plane fits on the device, posterior of frame k feeding
frame k + 1.  The trajectories must agree to 1e-7 (frame 0: ~1e-9; later frames inherit the ~1e-8 differences of the re-fitted planes, see tests/test_plane_fit_gpu.py)
and the final maps must have the same shape."""
import os
import subprocess

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H
from tests.test_map_update_gpu import _compare, _load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "fast-livo2_amd", "lib", "shim_demo")


def _glue(xyz, R, t, P, extR, extT, c):
    """LIVMapper.cpp:413-423"""
    pl = xyz.astype(np.float64)
    pi = pl @ extR.T + extT
    pw = (pi @ R.T + t).astype(np.float32).astype(np.float64)
    cb = synth.body_cov(pl, c["dept_err"], c["beam_err"])
    RE = R @ extR
    X = synth.skew(pi)
    return pw, RE @ cb @ RE.T + X @ P[0:3, 0:3] @ X.transpose(0, 2, 1) + P[3:6, 3:6]


@pytest.mark.parametrize("device_map,slide", [(False, False), (True, False), (True, True), (False, True)])
def test_lio_sequence_matches_oracle(tmp_path, orc, device_map, slide):
    """device_map: the shim keeps the octree on the GPU (VoxelMapManager::device_map_, livo2_map_tree_*): BuildVoxelMap from the host points once, then per frame
    StateEstimation on the resident tree and UpdateVoxelMapFromPosterior — no host octree, no snapshot upload"""
    d = str(tmp_path)
    if device_map:
        np.array([1], np.int32).tofile(os.path.join(d, "seq_device_map.bin"))
    SLIDE = (0.3, 14)                       # sliding_thresh [m], half_map_size [voxels]: a 7 m box that follows the sensor (mapSliding, LIVMapper.cpp:430-433)
    if slide:
        np.array(SLIDE, np.float64).tofile(os.path.join(d, "seq_slide.bin"))
    rng = np.random.default_rng(91)
    c = dict(synth.AVIA["lio"])
    extR, extT = synth.AVIA["extrinsic_R"], synth.AVIA["extrinsic_T"]
    scene = synth.make_room(rng, (20.0, 20.0, 6.0), 8)
    R0, t0 = scene.R_ws @ synth.rot_from_rpy(0.01, -0.015, 0.4), scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = synth.default_cov() * 1e-3
    K = 4
    # true poses along a short path, the commanded motion between them (what the IMU propagation would deliver) with a small error
    dR = [synth.rot_from_rpy(0.0, 0.0, 0.05 * (k % 2 * 2 - 1)) for k in range(K)]
    dt = [np.array([0.25, 0.05, 0.0]) for _ in range(K)]
    Rt, tt = [R0], [t0]
    for k in range(K):
        Rt.append(Rt[-1] @ dR[k]); tt.append(tt[-1] + dt[k])
    xyz_map = synth.lidar_scan(rng, scene, R0, t0, extR, extT, 60000, c["dept_err"], c["beam_err"], synth.AVIA["blind"], False)
    pw0, var0 = synth.world_points_and_var(xyz_map, R0, t0, extR, extT, P0, c["dept_err"], c["beam_err"])
    scans = [synth.voxel_grid_downsample(synth.lidar_scan(rng, scene, Rt[k + 1], tt[k + 1], extR, extT, 8000, c["dept_err"], c["beam_err"], synth.AVIA["blind"], False), 0.1)
             for k in range(K)]
    motion = np.array([np.concatenate([(dR[k] @ synth.so3_exp(rng.normal(0, np.deg2rad(0.2), 3))).ravel(), dt[k] + rng.normal(0, 0.01, 3)]) for k in range(K)])
    q = np.concatenate([np.full(3, 1e-5), np.full(3, 1e-4), np.zeros(13)])
    pw0.tofile(os.path.join(d, "seq_bld_pw.bin")); var0.reshape(-1, 9).tofile(os.path.join(d, "seq_bld_var.bin"))
    np.array([c["voxel_size"], c["max_layer"], c["max_points_num"], c["min_eigen_value"]] + list(c["layer_init_num"])[:5], np.float64).tofile(os.path.join(d, "seq_map_cfg.bin"))
    np.concatenate([[c["max_iterations"], c["max_layer"], c["sigma_num"], c["dept_err"], c["beam_err"], c["voxel_size"]], extR.ravel(), extT]).astype(np.float64).tofile(os.path.join(d, "seq_lidar_cfg.bin"))
    np.concatenate(scans).astype(np.float32).tofile(os.path.join(d, "seq_scans.bin")); np.array([len(s) for s in scans], np.int32).tofile(os.path.join(d, "seq_counts.bin"))
    motion.tofile(os.path.join(d, "seq_motion.bin")); q.tofile(os.path.join(d, "seq_q.bin"))
    st0 = orc.make_state(R0, t0, P0)
    np.frombuffer(bytes(st0), dtype=np.float64).tofile(os.path.join(d, "seq_state0.bin"))
    r = subprocess.run([DEMO, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr

    # the same chain on the oracle
    om = orc.OracleMap.build(pw0, var0.reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    ocfg = orc.lidar_cfg(c, extR, extT)
    post = orc.state_arrays(st0)
    traj = []
    for k in range(K):
        Rp, tp = post["R"] @ motion[k, :9].reshape(3, 3), post["t"] + motion[k, 9:]
        Pp = post["P"] + np.diag(q)
        prop = orc.make_state(Rp, tp, Pp, inv_expo=post["inv_expo"], vel=post["vel"], bg=post["bg"], ba=post["ba"], grav=post["grav"])
        ref = orc.lidar_state_estimation(om, ocfg, scans[k], prop, prop, want_points=False)
        post = orc.state_arrays(ref["state"])
        assert ref["n_iters"] >= 2
        pw, var = _glue(scans[k], post["R"], post["t"], post["P"], extR, extT, c)
        om.update(pw, var.reshape(-1, 9))
        if slide:
            removed = om.slide(post["t"], SLIDE[0], SLIDE[1])          # position_last_ = the posterior position (voxel_map.cpp:492)
            assert removed != 0 or k > 0
        traj.append(np.concatenate([post["R"].ravel(), post["t"]]))
        assert np.linalg.norm(post["t"] - tt[k + 1]) < 0.02            # the filter tracks the true path
    got = np.fromfile(os.path.join(d, "seq_out_traj.bin")).reshape(K, 12)
    err = np.abs(got - np.array(traj)).max(axis=1)
    assert err[0] < 1e-8 and err.max() < 1e-7, err
    n_planes = _compare(_load(d, "seq_out_"), om.export(c["voxel_size"], c["max_layer"]), loose=True)
    assert n_planes > (300 if slide else 1000)
    print(r.stdout)
