"""GPU parity of livo2_visual_map_upload + livo2_visual_select — the selection half of VIOManager::retrieveFromVisualSparseMap (reference
src/vio.cpp:352-486, 598-635) — against the oracle (oracle/orc_select.hpp).  Everything here is discrete or float32: the selected point
per grid cell, map_dist, the depth-continuity verdicts and the in-frame flags must be identical."""
import numpy as np
import pytest

from scenarios import synth

pytestmark = pytest.mark.gpu


def _compare(ctx, orc, ss, keys=True):
    ref = orc.visual_select(ss)
    ctx.visual_map_upload(ss.pos, ss.keys if keys else None, ss.active)
    out = ctx.visual_select(ss)
    assert np.array_equal(out["in_fov"].astype(np.int32), ref["in_fov"])
    assert np.array_equal(out["cell_point"], ref["cell_point"])
    assert np.array_equal(out["cell_dist"], ref["cell_dist"])
    assert np.array_equal(out["discont"].astype(np.int32), ref["discont"])
    return ref


def test_select_matches_oracle(ctx, orc):
    ss = synth.select_scenario(seed=71, n_pg=10000, n_vis=6000)
    ref = _compare(ctx, orc, ss)
    assert (ref["cell_point"] >= 0).sum() > 100 and ref["discont"].sum() > 5


def test_select_with_radtan_camera(ctx, orc):
    """vk::PinholeCamera::world2cam with the avia camera's distortion coefficients in the selection half (projection of scan and map points)."""
    ss = synth.select_scenario(seed=72, n_pg=8000, n_vis=5000)
    pin = orc.visual_select(ss)
    ss.cam = dict(ss.cam); ss.cam["d"] = synth.AVIA_RADTAN
    ref = _compare(ctx, orc, ss)
    assert (ref["cell_point"] >= 0).sum() > 100
    assert not np.array_equal(ref["cell_point"], pin["cell_point"]) or not np.array_equal(ref["cell_dist"], pin["cell_dist"])


def test_select_with_equidistant_camera(ctx, orc):
    """vk::EquidistantCamera (config/camera_fisheye_HILTI22.yaml) in the selection half"""
    ss = synth.select_scenario(seed=73, n_pg=8000, n_vis=5000)
    pin = orc.visual_select(ss)
    ss.cam = dict(ss.cam); ss.cam["k"] = synth.HILTI_EQUIDISTANT
    ref = _compare(ctx, orc, ss)
    assert (ref["cell_point"] >= 0).sum() > 100
    assert not np.array_equal(ref["cell_point"], pin["cell_point"]) or not np.array_equal(ref["cell_dist"], pin["cell_dist"])


def test_select_keys_computed_on_device(ctx, orc):
    """voxel_key = NULL: the device files every visual point with insertPointIntoVoxelMap's own formula (negative coordinates included)."""
    ss = synth.select_scenario(seed=74, n_pg=4000, n_vis=3000)
    ss.pos = ss.pos - np.array([30.0, 25.0, 3.0]); ss.pg = ss.pg - np.array([30.0, 25.0, 3.0])      # push the scene into negative coordinates
    ss.t_cur = ss.t_cur + ss.R_cur @ np.array([30.0, 25.0, 3.0])
    ss.keys = synth.feat_map_key_np(ss.pos)
    ref = _compare(ctx, orc, ss, keys=False)
    assert 5 < (ref["cell_point"] >= 0).sum() < 120          # few: with all-negative coordinates the scan looks one voxel low on every axis (the reference's key mismatch)


def test_select_edges(ctx, orc):
    ss = synth.select_scenario(seed=75, n_pg=500, n_vis=300)
    ctx.visual_map_upload(ss.pos, ss.keys, ss.active)
    import copy
    s0 = copy.copy(ss); s0.pg = ss.pg[:0]
    out = ctx.visual_select(s0)                                       # no scan points: nothing is selected
    assert (out["cell_point"] == -1).all() and (out["cell_dist"] == 10000.0).all() and not out["in_fov"].any()
    ctx.visual_map_upload(ss.pos[:0], None, None)                     # empty visual map
    out = ctx.visual_select(ss)
    assert (out["cell_point"] == -1).all()
    with pytest.raises(Exception):
        ctx.visual_map_upload(np.array([[1e9, 0, 0]]), None, None)    # voxel key outside 21 bits
