"""GPU parity of livo2_visual_retrieve_warp — the per-point tail of VIOManager::retrieveFromVisualSparseMap (reference src/vio.cpp:698-767:
warp matrix, search level, warpAffine, getImagePatch, photometric / NCC gates, survivors appended to visual_submap) — against the oracle
(oracle/orc_warp.hpp).  Everything the reference computes in float (warped patches, current patch, error) must be bit-identical, and so
must the discrete outputs (accepted, search level); the double quantities (A_cur_ref, NCC) use the same operation order."""
import ctypes as C

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _compare(ctx, orc, rs):
    ref = orc.warp_candidates(rs)
    out = ctx.retrieve_warp(rs)
    assert np.array_equal(out["search_level"], ref["search_level"])
    np.testing.assert_allclose(out["A"], ref["A"], rtol=1e-12, atol=1e-13)
    assert np.array_equal(out["patch_wrap"], ref["patch_wrap"]), "warped reference patches differ"
    assert np.array_equal(out["error"], ref["error"]), "float photometric errors differ"
    np.testing.assert_allclose(out["ncc"], ref["ncc"], rtol=1e-12, atol=1e-14)
    assert np.array_equal(out["accepted"], ref["accepted"])
    assert out["n_accepted"] == int(ref["accepted"].sum())
    return ref, out


def test_homography_warp_matches_oracle(ctx, orc):
    rs = synth.retrieve_scenario(seed=21, n_cand=1500, normal_en=True, ncc_en=False)
    ref, out = _compare(ctx, orc, rs)
    frac = ref["accepted"].mean()
    assert 0.4 < frac < 0.95                                  # both branches of the gate are exercised
    assert (ref["search_level"] > 0).sum() > 20
    assert (ref["patch_wrap"] == 0).mean() > 0.01             # out-of-image samples


def test_affine_warp_and_ncc_gate_match_oracle(ctx, orc):
    rs = synth.retrieve_scenario(seed=22, n_cand=700, normal_en=False, ncc_en=True, ncc_thre=0.9)
    ref, out = _compare(ctx, orc, rs)
    assert len(np.unique(ref["search_level"])) == 3
    assert ((ref["ncc"] < 0.9) & (ref["error"] <= 1000.0 * 64)).sum() > 0       # some candidates fall to the NCC gate alone


@pytest.mark.parametrize("normal_en", [True, False])
def test_radtan_camera_matches_oracle(ctx, orc, normal_en):
    """The avia camera's radial-tangential coefficients (config/camera_pinhole.yaml:9-12) through world2cam AND cam2world (OpenCV's undistortPoints
    iteration inside vk::PinholeCamera::cam2world): same bit-level bars as the pinhole camera."""
    rs = synth.retrieve_scenario(seed=25, n_cand=900, normal_en=normal_en, ncc_en=not normal_en, ncc_thre=0.85)
    pin = orc.warp_candidates(rs)
    rs.cam = dict(rs.cam); rs.cam["d"] = synth.AVIA_RADTAN
    ref, out = _compare(ctx, orc, rs)
    assert np.abs(ref["A"] - pin["A"]).max() > 1e-4           # the distortion is felt
    assert 0.3 < ref["accepted"].mean() < 0.98


def test_equidistant_camera_through_the_warp(ctx, orc):
    """vk::EquidistantCamera (config/camera_fisheye_HILTI22.yaml) through world2cam and cam2world of the affine warp (vio.cpp:247-290); atan / tan may differ in
    the last bit between host and device libm: accept / search-level decisions identical, patches to float rounding"""
    rs = synth.retrieve_scenario(seed=27, n_cand=900, normal_en=True)
    pin = orc.warp_candidates(rs)
    rs.cam = dict(rs.cam); rs.cam["k"] = synth.HILTI_EQUIDISTANT
    ref = orc.warp_candidates(rs)
    out = ctx.retrieve_warp(rs)
    assert np.abs(ref["A"] - pin["A"]).max() > 1e-4
    assert np.array_equal(out["accepted"], ref["accepted"]) and np.array_equal(out["search_level"], ref["search_level"])
    assert np.allclose(out["A"], ref["A"], rtol=1e-9, atol=1e-12)
    assert np.abs(out["patch_wrap"] - ref["patch_wrap"]).max() < 1e-2 and (out["patch_wrap"] == ref["patch_wrap"]).mean() > 0.999
    assert np.allclose(out["error"], ref["error"], rtol=1e-4)


def test_survivors_become_the_resident_frame(ctx, livo2, orc):
    """After the call the frame of the next visual update is the compacted survivor list: the update gives byte-identical results to
    livo2_visual_set_frame with the survivors' arrays taken from the oracle."""
    rs = synth.retrieve_scenario(seed=23, n_cand=600)
    vs = synth.visual_scenario(seed=3, n_patches=8)           # only for extrinsics / covariance / config of a visual update
    vs.img, vs.cam = rs.img, rs.cam
    ref = orc.warp_candidates(rs)
    keep = ref["accepted"] == 1
    pcfg = H.visual_cfg_product(vs)
    cur, prior = H.states(vs, livo2.State)
    out = ctx.retrieve_warp(rs, want_patches=False)
    assert out["n_accepted"] == keep.sum()
    ra, ea = ctx.visual_update(cur, prior, pcfg)
    ctx.set_frame(rs.img, rs.pos[keep], ref["patch_wrap"][keep], ref["search_level"][keep], rs.ref_inv_expo[keep])
    rb, eb = ctx.visual_update(cur, prior, pcfg)
    assert C.string_at(C.addressof(ra.state), C.sizeof(ra.state)) == C.string_at(C.addressof(rb.state), C.sizeof(rb.state))
    assert ra.n_steps == rb.n_steps and np.array_equal(ea, eb)


def test_retrieve_edge_cases(ctx, orc):
    rs = synth.retrieve_scenario(seed=24, n_cand=40)
    # no candidates
    import copy
    r0 = copy.copy(rs)
    for k in ("pos", "normal", "ref_img_idx", "ref_px", "ref_f", "ref_R", "ref_t", "ref_level", "ref_inv_expo"):
        setattr(r0, k, getattr(rs, k)[:0])
    out = ctx.retrieve_warp(r0)
    assert out["n_accepted"] == 0
    # a point projecting onto the image border: rejected with error = +inf instead of an out-of-bounds read
    r1 = copy.copy(rs)
    r1.pos = rs.pos.copy()
    p_c = np.array([(2.0 - rs.cam["cx"]) / rs.cam["fx"] * 5.0, 0.0, 5.0])
    r1.pos[0] = rs.R_cur.T @ (p_c - rs.t_cur)
    out = ctx.retrieve_warp(r1)
    assert out["accepted"][0] == 0 and np.isinf(out["error"][0])
    ref = orc.warp_candidates(rs)
    assert np.array_equal(out["accepted"][1:], ref["accepted"][1:])
    # bad arguments
    r2 = copy.copy(rs)
    r2.ref_img_idx = rs.ref_img_idx.copy(); r2.ref_img_idx[3] = 99
    with pytest.raises(Exception):
        ctx.retrieve_warp(r2)


def test_many_candidates(ctx, orc):
    """More than 16 384 candidates: the compaction scan takes its long-run path (more than 16 flags per thread)."""
    rs = synth.retrieve_scenario(seed=24, n_cand=17500, L=2)
    ref, out = _compare(ctx, orc, rs)
    assert 0.4 < ref["accepted"].mean() < 0.95
