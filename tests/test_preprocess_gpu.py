"""GPU parity of livo2_lidar_preprocess_scan (raw scan -> UndistortPcl -> pcl::VoxelGrid -> resident scan) against the oracle
(oracle/orc_preprocess.hpp; reference src/IMU_Processing.cpp:494-539, src/LIVMapper.cpp:351-352).  The undistorted coordinates are
float32 roundings of f64 expressions containing sin/cos: device and host libm may differ in the last f64 bit, so a float32 result
may differ by one ulp on rare points (counted and bounded); the voxel-grid stage is exact given identical inputs."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _cfg(livo2, rs):
    c = livo2.LidarCfg()
    c.max_iterations, c.max_layer = int(rs.cfg["max_iterations"]), int(rs.cfg["max_layer"])
    c.sigma_num, c.dept_err, c.beam_err, c.voxel_size, c.deg2rad = float(rs.cfg["sigma_num"]), float(rs.cfg["dept_err"]), float(rs.cfg["beam_err"]), float(rs.cfg["voxel_size"]), 0.017453293
    c.extR[:] = rs.extR.ravel().tolist(); c.extT[:] = rs.extT.tolist()
    return c


@pytest.mark.parametrize("seed,ext", [(61, None), (62, (0.1, -0.05, 0.2))])
def test_preprocess_matches_oracle(ctx, livo2, orc, seed, ext):
    rs = synth.raw_scan_scenario(seed=seed, n_raw=24000, extR=None if ext is None else synth.rot_from_rpy(*ext))
    if seed == 62:
        rs.curvature = np.sort(np.maximum(rs.curvature, np.float32(7.7))).astype(np.float32)     # exercises the first-point re-compensation
    cfg = _cfg(livo2, rs)
    n_down, und, down = ctx.preprocess_scan(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.leaf, cfg)
    ref_u = orc.undistort(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.extR, rs.extT)
    diff = und != ref_u
    assert diff.mean() < 1e-3, f"{diff.sum()} undistorted coordinates differ"
    assert np.abs(und - ref_u).max() <= 2.0 * np.spacing(np.abs(ref_u).max().astype(np.float32))
    # the voxel grid on the device's own undistorted cloud must be exactly the oracle's filter of that cloud
    ref_d = orc.voxel_grid(und, rs.leaf)
    assert n_down == len(ref_d)
    assert np.array_equal(down, ref_d)
    assert 0.3 * len(rs.xyz) < n_down < len(rs.xyz)


def test_preprocessed_scan_feeds_the_update(ctx, livo2, orc):
    """After the call the filtered cloud is the resident scan: an update gives byte-identical results to set_scan(feats_down_body)."""
    import ctypes as C
    sc = synth.lidar_scenario(seed=63, n_points=3000, downsample=0.1)
    rs = synth.raw_scan_scenario(seed=63, n_raw=6000)
    cfg = H.lidar_cfg_product(sc)
    cur, prior = H.states(sc, livo2.State)
    ctx.upload_map(sc.fmap)
    n_down, und, down = ctx.preprocess_scan(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.leaf, cfg)
    ra, _ = ctx.lidar_update(cur, prior, cfg)
    ctx.set_scan(down, cfg)
    rb, _ = ctx.lidar_update(cur, prior, cfg)
    assert ra.n_iters == rb.n_iters
    assert C.string_at(C.addressof(ra.state), C.sizeof(ra.state)) == C.string_at(C.addressof(rb.state), C.sizeof(rb.state))


def test_preprocess_edges(ctx, livo2, orc):
    rs = synth.raw_scan_scenario(seed=64, n_raw=500)
    cfg = _cfg(livo2, rs)
    n0, _, _ = ctx.preprocess_scan(rs.xyz[:0], rs.curvature[:0], rs.poses, rs.rot_end, rs.pos_end, rs.leaf, cfg)
    assert n0 == 0
    n1, und, down = ctx.preprocess_scan(rs.xyz, rs.curvature, rs.poses[:1], rs.rot_end, rs.pos_end, rs.leaf, cfg)     # no IMU segment: no undistortion
    assert np.array_equal(und, rs.xyz) and np.array_equal(down, orc.voxel_grid(rs.xyz, rs.leaf))
    with pytest.raises(Exception):
        ctx.preprocess_scan(np.array([[0, 0, 0], [4000, 4000, 4000]], np.float32), np.array([0, 1], np.float32), rs.poses[:1], rs.rot_end, rs.pos_end, 0.001, cfg)
    with pytest.raises(Exception):
        ctx.preprocess_scan(rs.xyz, rs.curvature, rs.poses[::-1], rs.rot_end, rs.pos_end, rs.leaf, cfg)                 # unordered IMU poses
