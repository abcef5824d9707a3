"""Pins the oracle's pre-stage restatement (oracle/orc_preprocess.hpp: UndistortPcl backward loop, src/IMU_Processing.cpp:494-539; pcl::VoxelGrid
as published) with independent numpy evaluations."""
import numpy as np

from oracle import orc
from scenarios import synth


def _exp(w):
    th = np.linalg.norm(w)
    if th < 1e-7:
        return np.eye(3)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _undistort_np(rs, i, h):
    P = rs.poses[h]
    dt = rs.curvature[i] / 1000.0 - P[0]
    Ri = P[13:22].reshape(3, 3) @ _exp(P[4:7] * dt)
    T = P[10:13] + P[7:10] * dt + 0.5 * P[1:4] * dt * dt - rs.pos_end
    return rs.extR.T @ (rs.rot_end.T @ (Ri @ (rs.extR @ rs.xyz[i].astype(float) + rs.extT) + T) - rs.extT)


def test_undistort_matches_numpy():
    rs = synth.raw_scan_scenario(seed=52, n_raw=3000, extR=synth.rot_from_rpy(0.1, -0.05, 0.2))
    u = orc.undistort(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.extR, rs.extT)
    assert np.array_equal(u[0], rs.xyz[0])                      # t = 0: not later than any IMU pose -> untouched
    for i in range(1, len(rs.xyz), 7):
        h = int(np.searchsorted(rs.poses[:-1, 0], rs.curvature[i] / 1000.0, side="left")) - 1
        assert h >= 0
        assert np.abs(_undistort_np(rs, i, h) - u[i]).max() < 2e-6
    # points later than the last IMU pose use the last head (poses[-2]); a scan without IMU poses is untouched
    assert np.array_equal(orc.undistort(rs.xyz, rs.curvature, rs.poses[:1], rs.rot_end, rs.pos_end, rs.extR, rs.extT), rs.xyz)


def test_first_point_is_recompensated_by_every_earlier_segment():
    """The backward walk only leaves its inner loop at the first point (`if (it_pcl == begin) break`, IMU_Processing.cpp:531), so a first
    point later than several IMU poses is compensated once per earlier head, each time from the already-written float coordinates."""
    rs = synth.raw_scan_scenario(seed=53, n_raw=400)
    rs.curvature = np.sort(np.maximum(rs.curvature, np.float32(12.3))).astype(np.float32)      # first point inside the third IMU segment
    u = orc.undistort(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.extR, rs.extT)
    p = rs.xyz.copy()
    t = rs.curvature[0] / 1000.0
    h0 = int(np.searchsorted(rs.poses[:-1, 0], t, side="left")) - 1
    assert h0 == 2
    for h in range(h0, -1, -1):
        tmp = synth.RawScanScenario(p, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.extR, rs.extT, rs.leaf, rs.cfg)
        p = p.copy(); p[0] = _undistort_np(tmp, 0, h).astype(np.float32)
    assert np.abs(p[0] - u[0]).max() < 5e-6
    assert np.abs(u[0] - _undistort_np(rs, 0, h0)).max() > 1e-4          # and that differs from a single compensation


def test_voxel_grid_matches_numpy():
    rng = np.random.default_rng(3)
    xyz = rng.uniform(-12, 12, (30000, 3)).astype(np.float32)
    a, b = orc.voxel_grid(xyz, 0.5), synth.voxel_grid_downsample(xyz, 0.5)
    assert a.shape == b.shape and np.abs(a - b).max() < 5e-6             # same leaves, same order; float vs double centroid sums
    one = orc.voxel_grid(xyz[:1], 0.1)
    assert np.array_equal(one, xyz[:1])
    assert len(orc.voxel_grid(xyz[:0], 0.1)) == 0
    try:
        orc.voxel_grid(np.array([[0, 0, 0], [4000, 4000, 4000]], np.float32), 0.001)
        assert False, "int32 overflow of the leaf grid must be refused"
    except OverflowError:
        pass
