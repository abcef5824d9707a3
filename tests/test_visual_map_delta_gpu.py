"""livo2_visual_map_apply — the incremental device mirror of the visual map (missing 3 / next-round item 3): what generateVisualMapPoints /
updateVisualMapPoints / updateReferencePatch change per frame (reference src/vio.cpp:227-246, 804-967, 969-1100; visual_point.cpp:35-55: addFrameRef pushes to the
FRONT of obs_, deleteFeatureRef clears ref_patch) is applied in O(changes).  After every scripted frame of maintenance (scenarios/visual_map_growth.py) the whole
retrieveFromVisualSparseMap runs three ways on the same map — the incrementally maintained mirror, a second context that gets the whole map re-uploaded
(livo2_visual_map_upload + livo2_visual_obs_upload), and the oracle — and every output must agree (observation indices through the global -> CSR translation)."""
import numpy as np
import pytest

from scenarios import synth
from scenarios.visual_map_growth import GrowingMap
from tests.test_retrieve_chain_gpu import _compare as _compare_chain

pytestmark = pytest.mark.gpu


def _same(out_i, out_f, g2c):
    tr = lambda a: np.where(a >= 0, g2c[np.maximum(a, 0)], -1)
    for k in ("cell_point", "cell_dist", "discont", "cand_cell", "sub_point"):
        assert np.array_equal(out_i[k], out_f[k]), k
    assert np.array_equal(tr(out_i["cell_obs"]), out_f["cell_obs"]) and np.array_equal(tr(out_i["sub_obs"]), out_f["sub_obs"])
    assert np.array_equal(tr(out_i["ref_patch"]), out_f["ref_patch"])
    assert out_i["n_candidates"] == out_f["n_candidates"] and out_i["n_accepted"] == out_f["n_accepted"]
    for k in ("accepted", "search_level", "error", "ncc", "A", "patch_wrap"):
        assert np.array_equal(out_i["tail"][k], out_f["tail"][k]), k


def _frame_pose(cs, rng, k):
    dR = synth.so3_exp(rng.normal(0, np.deg2rad(0.05), 3))
    R = dR @ cs.sel.R_cur
    c = -cs.sel.R_cur.T @ cs.sel.t_cur + rng.normal(0, 0.005, 3)
    return R, -R @ c


@pytest.mark.parametrize("normal_en", [True, False])
def test_incremental_mirror_equals_full_upload_and_oracle(ctx, livo2, orc, normal_en):
    cs = synth.retrieve_chain_scenario(seed=191, n_pg=8000, n_vis=9000, grid_n_height=34, normal_en=normal_en)
    gm = GrowingMap(cs)
    rng = np.random.default_rng(7)
    full = livo2.Context(0)
    try:
        ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active); ctx.visual_obs_upload(cs)
        grows0 = ctx.counter("visual_map_delta_grows")
        for k in range(5):
            flat, g2c = gm.flat()
            full.visual_map_upload(flat.sel.pos, flat.sel.keys, flat.sel.active); full.visual_obs_upload(flat)
            ref, out_f = None, None
            # the re-uploaded map against the oracle (every stage output), then the incrementally maintained mirror against the re-uploaded one
            orc_ref = orc.visual_retrieve(flat)
            out_f = full.visual_retrieve_from_map(flat)
            assert np.array_equal(out_f["cell_obs"], orc_ref["cell_obs"]) and np.array_equal(out_f["sub_point"], orc_ref["sub_point"]) and np.array_equal(out_f["ref_patch"], orc_ref["ref_patch"])
            assert np.array_equal(out_f["tail"]["error"], orc_ref["tail"]["error"]) and np.array_equal(out_f["tail"]["patch_wrap"], orc_ref["tail"]["patch_wrap"])
            out_i = ctx.visual_retrieve_from_map(flat)
            _same(out_i, out_f, g2c)
            assert out_i["n_candidates"] > 200 and out_i["n_accepted"] > 50
            c = ctx.visual_map_counts()
            assert c["points"] == gm.n_points and c["obs"] == gm.n_obs and c["ref_imgs"] == len(gm.ref_imgs)
            gm.set_ref_patch(out_i["ref_patch"])                                      # pt->ref_patch as the retrieval left it (the device remembers it, the host follows)
            R_fw, t_fw = _frame_pose(cs, rng, k)
            img_k = np.clip(cs.img.astype(np.int32) + rng.integers(-2, 3, cs.img.shape), 0, 255).astype(np.uint8)
            d = gm.step(rng, R_fw, t_fw, img_k, frame_id=300 + k, n_new=100, n_touch=150)
            ctx.visual_map_apply(**d)
        assert ctx.counter("visual_map_delta_calls") >= 5
        assert ctx.counter("visual_map_delta_grows") - grows0 <= 40                   # geometric growth: ~19 arrays grow once or twice over five frames, not per frame
    finally:
        full.close()


def test_a_map_built_from_nothing_by_deltas(ctx, livo2, orc):
    """an EMPTY map installed by the full upload (n_points = 0, n_obs = 0, no reference image), then everything arrives through livo2_visual_map_apply"""
    cs = synth.retrieve_chain_scenario(seed=192, n_pg=4000, n_vis=3000, grid_n_height=34, normal_en=True)
    import copy
    empty = copy.copy(cs); empty.sel = copy.copy(cs.sel)
    empty.sel.pos, empty.sel.keys, empty.sel.active = cs.sel.pos[:0], cs.sel.keys[:0], cs.sel.active[:0]
    empty.normal, empty.normal_initialized, empty.ref_patch, empty.obs_offset = cs.normal[:0], cs.normal_initialized[:0], cs.ref_patch[:0], np.zeros(1, np.int32)
    for k in ("id", "img_idx", "px", "f", "R", "t", "level", "inv_expo", "patch"):
        setattr(empty, "obs_" + k, getattr(cs, "obs_" + k)[:0])
    empty.ref_imgs = cs.ref_imgs[:1]                                                 # one image: names width / height / stride
    ctx.visual_map_upload(empty.sel.pos, empty.sel.keys, empty.sel.active); ctx.visual_obs_upload(empty)
    # the scenario's whole map as ONE delta: all points new, all observations new, every point touched with its list; the other five images one call each
    n = len(cs.sel.pos)
    lists = [list(range(int(cs.obs_offset[i]), int(cs.obs_offset[i + 1]))) for i in range(n)]
    for s in range(1, len(cs.ref_imgs)):
        ctx.visual_map_apply(img=cs.ref_imgs[s], img_slot=s)
    ctx.visual_map_apply(img=cs.ref_imgs[0], img_slot=0)                             # replacing a slot
    ctx.visual_map_apply(new_pos=cs.sel.pos, new_keys=cs.sel.keys, new_active=cs.sel.active,
                         obs={k: getattr(cs, "obs_" + k) for k in ("id", "img_idx", "px", "f", "R", "t", "level", "inv_expo", "patch")},
                         touched=dict(point=np.arange(n), lists=lists, normal=cs.normal, normal_initialized=cs.normal_initialized, ref_patch=cs.ref_patch, active=cs.sel.active))
    ref, out = _compare_chain(ctx, orc, cs, run=lambda c, s: c.visual_retrieve_from_map(s))      # (no re-upload: the mirror built by the deltas is what is under test)
    assert len(ref["cand_cell"]) > 100


def test_delta_argument_errors(livo2):
    c = livo2.Context(0)
    try:
        cs = synth.retrieve_chain_scenario(seed=193, n_pg=800, n_vis=600, normal_en=True)
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(img=cs.img, img_slot=0)                                # no map yet
        assert e.value.code == livo2.abi.ERR_NO_MAP
        c.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active); c.visual_obs_upload(cs)
        n, m, stride = len(cs.sel.pos), len(cs.obs_id), c.visual_map_counts()["stride"]
        assert stride >= 32
        base = dict(point=np.array([0]), normal=np.zeros((1, 3)), normal_initialized=np.ones(1, np.uint8))
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(touched=dict(base, lists=[list(range(stride + 1))], ref_patch=np.array([-1])))      # a list beyond the stride
        assert e.value.code == livo2.abi.ERR_RANGE
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(touched=dict(base, lists=[[0, 1]], ref_patch=np.array([5])))                         # ref_patch not in the list
        assert e.value.code == livo2.abi.ERR_INVALID
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(touched=dict(base, lists=[[m]], ref_patch=np.array([-1])))                           # an observation that does not exist
        assert e.value.code == livo2.abi.ERR_INVALID
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(touched=dict(base, point=np.array([n]), lists=[[0]], ref_patch=np.array([-1])))      # a point that does not exist
        assert e.value.code == livo2.abi.ERR_INVALID
        with pytest.raises(livo2.Livo2Error) as e:                                                                  # one point named by two rows (they would race on the device)
            c.visual_map_apply(touched=dict(point=np.array([0, 0]), normal=np.zeros((2, 3)), normal_initialized=np.ones(2, np.uint8), lists=[[0], [1]], ref_patch=np.array([-1, -1])))
        assert e.value.code == livo2.abi.ERR_INVALID
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(img=cs.img, img_slot=len(cs.ref_imgs) + 1)                                           # a hole in the image pool
        assert e.value.code == livo2.abi.ERR_INVALID
        with pytest.raises(livo2.Livo2Error) as e:
            c.visual_map_apply(new_pos=np.zeros((1, 3)), new_keys=np.array([[1 << 22, 0, 0]]))                       # key outside 21 bits
        assert e.value.code == livo2.abi.ERR_RANGE
        assert c.visual_map_counts()["points"] == n and c.visual_map_counts()["obs"] == m                           # nothing was applied by the failed calls
        c.visual_map_apply()                                                                                        # an empty delta is fine
        out = c.visual_retrieve_from_map(cs)
        assert out["n_candidates"] > 0
    finally:
        c.close()
