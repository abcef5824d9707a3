"""GPU parity of livo2_visual_obs_upload + livo2_visual_retrieve_from_map — the whole VIOManager::retrieveFromVisualSparseMap (reference
src/vio.cpp:352-780, raycast_en = false) as one chain of launches: selection -> reference-patch choice (vio.cpp:644-696,
src/visual_point.cpp:57-95) -> warp / gate tail — against the chained oracle (oracle/orc.py visual_retrieve: orc_select.hpp, orc_choice.hpp,
orc_warp.hpp).  Discrete outputs (selected point and chosen observation per grid cell, remembered ref_patch, candidate order, search levels,
accept flags, the appended sub-map) and everything the reference computes in float must be identical; doubles agree to rounding."""
import ctypes as C

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(ctx, cs):
    ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
    ctx.visual_obs_upload(cs)
    return ctx.visual_retrieve_from_map(cs)


def _compare(ctx, orc, cs, run=None):
    ref = orc.visual_retrieve(cs)
    out = (run or _run)(ctx, cs)
    sel = ref["sel"]
    assert np.array_equal(out["cell_point"], sel["cell_point"])
    assert np.array_equal(out["cell_dist"], sel["cell_dist"])
    assert np.array_equal(out["discont"].astype(np.int32), sel["discont"])
    assert np.array_equal(out["cell_obs"], ref["cell_obs"]), "chosen observations differ"
    assert np.array_equal(out["ref_patch"], ref["ref_patch"]), "remembered reference patches differ"
    assert out["n_candidates"] == len(ref["cand_cell"]) and np.array_equal(out["cand_cell"], ref["cand_cell"])
    t, rt = out["tail"], ref["tail"]
    assert np.array_equal(t["search_level"], rt["search_level"])
    np.testing.assert_allclose(t["A"], rt["A"], rtol=1e-12, atol=1e-13)
    assert np.array_equal(t["patch_wrap"], rt["patch_wrap"]), "warped reference patches differ"
    assert np.array_equal(t["error"], rt["error"]), "float photometric errors differ"
    np.testing.assert_allclose(t["ncc"], rt["ncc"], rtol=1e-12, atol=1e-14)
    assert np.array_equal(t["accepted"], rt["accepted"])
    assert out["n_accepted"] == len(ref["sub_point"])
    assert np.array_equal(out["sub_point"], ref["sub_point"]) and np.array_equal(out["sub_obs"], ref["sub_obs"])
    return ref, out


@pytest.mark.parametrize("normal_en", [True, False])
def test_chain_matches_oracle(ctx, orc, normal_en):
    cs = synth.retrieve_chain_scenario(seed=81, n_pg=10000, n_vis=20000, grid_n_height=51, normal_en=normal_en, ncc_en=not normal_en, ncc_thre=0.8)
    ref, out = _compare(ctx, orc, cs)
    assert len(ref["cand_cell"]) > 300 and 0.3 < len(ref["sub_point"]) / len(ref["cand_cell"]) <= 1.0
    if normal_en:
        assert (ref["ref_patch"] != cs.ref_patch).sum() > 100          # choices were made and remembered
    else:
        ids = cs.obs_id[ref["cand_obs"]]
        assert len(np.unique(ids)) < len(ids) / 10                     # warp_map: a handful of frames lead all candidates


def test_chain_with_radtan_camera(ctx, orc):
    """The whole retrieveFromVisualSparseMap with the avia camera's distortion (world2cam in selection and warp, cam2world in the warp)."""
    cs = synth.retrieve_chain_scenario(seed=83, n_pg=8000, n_vis=12000, grid_n_height=51, normal_en=True)
    cs.sel.cam = dict(cs.sel.cam); cs.sel.cam["d"] = synth.AVIA_RADTAN
    ref, out = _compare(ctx, orc, cs)
    assert len(ref["cand_cell"]) > 200 and len(ref["sub_point"]) > 50


def test_remembered_choice_is_reused(ctx, orc):
    """A second call on the same map sees the ref_patch the first one remembered on the device (pt->ref_patch / has_ref_patch_)."""
    cs = synth.retrieve_chain_scenario(seed=82, n_pg=6000, n_vis=9000, normal_en=True)
    ref, out = _compare(ctx, orc, cs)
    again = ctx.visual_retrieve_from_map(cs)
    assert np.array_equal(again["cell_obs"], out["cell_obs"]) and np.array_equal(again["ref_patch"], out["ref_patch"])
    assert np.array_equal(again["tail"]["error"], out["tail"]["error"])
    # a point whose remembered patch is then changed by the host (re-upload) follows the new value
    cs.ref_patch = out["ref_patch"].copy()
    pts = out["cell_point"][out["cand_cell"]]
    multi = pts[(cs.obs_offset[pts + 1] - cs.obs_offset[pts]) >= 2]
    assert len(multi) > 5
    p = int(multi[0])
    cs.ref_patch[p] = cs.obs_offset[p] + (cs.ref_patch[p] - cs.obs_offset[p] + 1) % (cs.obs_offset[p + 1] - cs.obs_offset[p])
    _compare(ctx, orc, cs)


def test_chain_edges(ctx, orc):
    cs = synth.retrieve_chain_scenario(seed=83, n_pg=1500, n_vis=1200, normal_en=True)
    cs.normal_initialized[:] = 0                                       # nothing survives the normal gate
    ref, out = _compare(ctx, orc, cs)
    assert out["n_candidates"] == 0 and out["n_accepted"] == 0
    cs = synth.retrieve_chain_scenario(seed=84, n_pg=1500, n_vis=1200, normal_en=False, outlier_threshold=1e-3)
    ref, out = _compare(ctx, orc, cs)                                  # candidates, but the photometric gate rejects all of them
    assert out["n_candidates"] > 0 and out["n_accepted"] == 0
    cs.sel.pg = cs.sel.pg[:0]                                          # empty scan: no voxel is looked into
    ref, out = _compare(ctx, orc, cs)
    assert (out["cell_point"] == -1).all() and out["n_candidates"] == 0


def test_chain_feeds_the_visual_update(ctx, livo2, orc):
    """The survivors are the resident frame: the visual update that follows equals, byte for byte, the update after livo2_visual_set_frame
    with the oracle's sub-map arrays."""
    cs = synth.retrieve_chain_scenario(seed=86, n_pg=8000, n_vis=12000, grid_n_height=34, normal_en=True)
    ref = orc.visual_retrieve(cs)
    keep = ref["tail"]["accepted"] == 1
    vs = synth.visual_scenario(seed=3, n_patches=8)                    # only for extrinsics / covariance / config of a visual update
    vs.img, vs.cam = cs.img, cs.sel.cam
    pcfg = H.visual_cfg_product(vs)
    cur, prior = H.states(vs, livo2.State)
    out = _run(ctx, cs)
    assert out["n_accepted"] == keep.sum() > 50
    ra, _ = ctx.visual_update(cur, prior, pcfg)
    ctx.set_frame(cs.img, cs.sel.pos[ref["sub_point"]], ref["tail"]["patch_wrap"][keep], ref["tail"]["search_level"][keep], cs.obs_inv_expo[ref["sub_obs"]])
    rb, _ = ctx.visual_update(cur, prior, pcfg)
    assert C.string_at(C.addressof(ra.state), C.sizeof(ra.state)) == C.string_at(C.addressof(rb.state), C.sizeof(rb.state))


def test_retrieve_warp_reuses_warps_per_frame_id(ctx, orc):
    """livo2_visual_retrieve_warp with ref_id: the warp_map reuse of the !normal_en branch (src/vio.cpp:716-734)."""
    rs = synth.retrieve_scenario(seed=33, n_cand=900, normal_en=False)
    rs.ref_id = (np.arange(900) * 7919 % 13).astype(np.int32) + 1000
    ref = orc.warp_candidates(rs)
    out = ctx.retrieve_warp(rs)
    assert np.array_equal(out["search_level"], ref["search_level"])
    np.testing.assert_allclose(out["A"], ref["A"], rtol=1e-12, atol=1e-13)
    assert np.array_equal(out["patch_wrap"], ref["patch_wrap"]) and np.array_equal(out["error"], ref["error"])
    assert np.array_equal(out["accepted"], ref["accepted"])
    lead = {}
    for i, k in enumerate(rs.ref_id):
        lead.setdefault(int(k), i)
    assert all(np.array_equal(ref["A"][i], ref["A"][lead[int(k)]]) for i, k in enumerate(rs.ref_id))


def test_chain_argument_errors(livo2):
    """Error behaviour of the chained retrieval: order of the uploads, consistency of the observation table, value ranges."""
    c = livo2.Context(0)
    cs = synth.retrieve_chain_scenario(seed=87, n_pg=800, n_vis=600, normal_en=True)
    with pytest.raises(livo2.Livo2Error) as e:
        c.visual_obs_upload(cs)                                        # no visual map yet
    assert e.value.code == livo2.abi.ERR_NO_MAP
    c.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
    with pytest.raises(livo2.Livo2Error) as e:
        c.visual_retrieve_from_map(cs)                                 # points without their observations
    assert e.value.code == livo2.abi.ERR_NO_MAP
    keep = cs.ref_patch.copy()
    p = int(np.nonzero(cs.obs_offset[1:] - cs.obs_offset[:-1] >= 1)[0][0])
    cs.ref_patch[p] = cs.obs_offset[p + 1]                             # an observation of the NEXT point
    with pytest.raises(livo2.Livo2Error) as e:
        c.visual_obs_upload(cs)
    assert e.value.code == livo2.abi.ERR_INVALID
    cs.ref_patch = keep
    ids = cs.obs_id.copy()
    cs.obs_id[3] = 2**31 - 1                                           # reserved value of the warp_map table
    with pytest.raises(livo2.Livo2Error) as e:
        c.visual_obs_upload(cs)
    assert e.value.code == livo2.abi.ERR_RANGE
    cs.obs_id = ids
    c.visual_obs_upload(cs)
    out = c.visual_retrieve_from_map(cs)
    assert out["n_candidates"] > 0
    c.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)        # a new point set drops the old observation table
    with pytest.raises(livo2.Livo2Error) as e:
        c.visual_retrieve_from_map(cs)
    assert e.value.code == livo2.abi.ERR_NO_MAP
    c.close()
