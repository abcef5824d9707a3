"""vio/raycast_en = true on the device (fast-livo2_amd/csrc/raycast_kernels.hpp) against the oracle's restatement of the RayCasting module (reference
src/vio.cpp:487-591 with the rays of initializeVIO, vio.cpp:80-118).  The LiDAR
VoxelMap the rays look into (plane_map) is the device-resident tree of the same context, built from the same points as the oracle's map.
Compared: which cells end TYPE_MAP, the point and float distance every cell keeps, in_fov, add_from_voxel_map (center_ exactly; normal_ up to the sign the eigen-solver
gives), and the whole retrieval (sub-map members, float errors, warped patches) downstream of that selection."""
import numpy as np
import pytest

from scenarios import synth

pytestmark = pytest.mark.gpu


def _maps(ctx, orc, cs, with_map):
    c = dict(synth.AVIA["lio"])
    var = np.tile((np.eye(3) * 1e-4).ravel(), (len(cs.sel.map_pw), 1))
    if not with_map:
        return None
    om = orc.OracleMap.build(cs.sel.map_pw, var, c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    ctx.map_tree_create(c, max_roots=20000)
    ctx.map_tree_update(cs.sel.map_pw, var, build=True)
    return om


@pytest.mark.parametrize("seed,camera,with_map", [(91, None, True), (92, None, True), (93, "radtan", True), (94, "equidistant", True), (95, None, False)])
def test_raycast_selection_and_retrieval_match_oracle(livo2, orc, seed, camera, with_map):
    cs = synth.retrieve_chain_scenario(seed=seed, n_pg=12000, n_vis=6000, L=2, grid_n_height=34, raycast=True)
    if camera == "radtan":
        cs.sel.cam = dict(cs.sel.cam); cs.sel.cam["d"] = synth.AVIA_RADTAN
    if camera == "equidistant":
        cs.sel.cam = dict(cs.sel.cam); cs.sel.cam["k"] = synth.HILTI_EQUIDISTANT
    ctx = livo2.Context(0)
    try:
        om = _maps(ctx, orc, cs, with_map)
        ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
        ctx.visual_obs_upload(cs)
        plain = orc.visual_select(cs.sel)
        ref = orc.visual_retrieve(cs, raycast=True, omap=om)
        sel = ref["sel"]
        assert int((sel["cell_point"] != plain["cell_point"]).sum()) >= 3                   # the rays changed the selection
        got = ctx.visual_select(cs.sel, raycast=True)
        on = sel["cell_type"] == 1
        assert np.array_equal(got["cell_point"] >= 0, on & (sel["cell_point"] >= 0))
        assert np.array_equal(got["cell_point"][on], sel["cell_point"][on]) and np.array_equal(got["cell_dist"][on], sel["cell_dist"][on])
        assert np.array_equal(got["discont"] != 0, sel["discont"] != 0)
        assert np.array_equal(got["in_fov"] != 0, sel["in_fov"] != 0)
        a, b = got["add_from_voxel_map"], sel["add_from_voxel_map"]
        assert len(a) == len(b) and (len(b) >= 3) == with_map
        if with_map:
            assert np.abs(a[:, :3] - b[:, :3]).max() < 1e-12                                   # center_ of the same planes in the same (grid cell) order
            assert np.abs(np.abs((a[:, 3:] * b[:, 3:]).sum(1)) - 1.0).max() < 1e-7
        # without the option nothing of it happens
        assert np.array_equal(ctx.visual_select(cs.sel)["cell_point"][plain["cell_type"] == 1], plain["cell_point"][plain["cell_type"] == 1])
        # the whole chain downstream of the ray-extended selection
        co = ctx.visual_retrieve_from_map(cs, raycast=True)
        assert np.array_equal(co["cell_obs"], ref["cell_obs"]) and np.array_equal(co["sub_point"], ref["sub_point"]) and len(ref["sub_point"]) > 50
        keep = ref["tail"]["accepted"] != 0
        assert np.array_equal(co["tail"]["error"][co["tail"]["accepted"] != 0], ref["tail"]["error"][keep])
        assert len(co["add_from_voxel_map"]) == len(b)
    finally:
        ctx.close()


def test_raycast_needs_the_device_tree_when_a_snapshot_map_is_resident(livo2, ctx):
    cs = synth.retrieve_chain_scenario(seed=91, n_pg=3000, n_vis=2000, L=2, grid_n_height=17, raycast=True)
    sc = synth.lidar_scenario(seed=11, n_points=2000, downsample=0.1)
    ctx.upload_map(sc.fmap)
    ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
    with pytest.raises(livo2.Livo2Error) as e:
        ctx.visual_select(cs.sel, raycast=True)
    assert e.value.code == livo2.abi.ERR_NO_MAP
    assert len(ctx.visual_select(cs.sel)["cell_point"]) > 0
