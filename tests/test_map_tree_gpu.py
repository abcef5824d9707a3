"""Device-resident VoxelMap (livo2_map_tree_*): BuildVoxelMap / UpdateVoxelMap with the octree itself on the GPU (reference src/voxel_map.cpp:137-290, 532-591,
609-641) against the oracle's serial restatement (oracle/orc_voxel_map.hpp): identical tree shape after the build and after every update of a sequence (which root
voxels exist, which nodes are planes, which children exist — i.e. every counter-driven decision of UpdateOctoTree was taken at the same point), plane
parameters within the plane-fit tolerances; then the LiDAR update reads the structure those calls maintain, with no snapshot upload in between."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H
from tests.test_map_update_gpu import _compare

pytestmark = pytest.mark.gpu


def _flat(d, c):
    return synth.FlatMap(c["voxel_size"], c["max_layer"], d["root_key"], d["root_node"], d["root_center"], d["root_quarter"], d["node_plane"], d["node_child"],
                         d["plane_normal"], d["plane_center"], d["plane_var"], d["plane_d"], d["plane_radius"])


def _scene(seed):
    rng = np.random.default_rng(seed)
    c = dict(synth.AVIA["lio"])
    scene = synth.make_room(rng, (20.0, 20.0, 6.0), 8)
    extR, extT = synth.AVIA["extrinsic_R"], synth.AVIA["extrinsic_T"]
    R0, t0 = scene.R_ws @ synth.rot_from_rpy(0.01, -0.015, 0.4), scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = synth.default_cov() * 1e-3
    def cloud(n, R, t):
        xyz = synth.lidar_scan(rng, scene, R, t, extR, extT, n, c["dept_err"], c["beam_err"], synth.AVIA["blind"], False)
        return xyz, synth.world_points_and_var(xyz, R, t, extR, extT, P0, c["dept_err"], c["beam_err"])
    return c, cloud, R0, t0, P0, extR, extT


@pytest.mark.parametrize("spread,wide", [(0, 1), (1, 1), (2, 1), (4, 1), (8, 0), (8, 1)])
def test_build_and_update_sequence_match_oracle(ctx, orc, spread, wide):
    """spread / wide: the lane layouts of k_mt_update (round 6) — eight, four, two or one root voxel per wave, the plane re-fit on 8 or on all 64 lanes; 0 = the choice
    the library makes itself from the previous update's touched roots.  Every one of them must take every counter-driven decision where the oracle takes it."""
    ctx.set_option("map_update_spread", spread); ctx.set_option("map_update_wide_fit", wide)
    try:
        _build_and_update_sequence(ctx, orc)
    finally:
        ctx.set_option("map_update_spread", 0); ctx.set_option("map_update_wide_fit", 1)


def _build_and_update_sequence(ctx, orc):
    c, cloud, R0, t0, P0, extR, extT = _scene(81)
    _, (pw0, var0) = cloud(40000, R0, t0)
    ctx.map_tree_create(c, max_roots=60000)
    ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
    om = orc.OracleMap.build(pw0, var0.reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    n_prev = _compare(_flat(ctx.map_tree_export(), c), om.export(c["voxel_size"], c["max_layer"]))
    assert n_prev > 1000
    # a sequence of scans: each overlaps the map (re-fits every 5 points, freezing at 50) and extends it (new roots, subdivision)
    for k in range(4):
        Rk, tk = R0 @ synth.rot_from_rpy(0.0, 0.0, 0.12 * (k + 1)), t0 + np.array([0.3 * (k + 1), 0.1 * k, 0.0])
        _, (pw, var) = cloud(12000, Rk, tk)
        ctx.map_tree_update(pw, var.reshape(-1, 9))
        om.update(pw, var.reshape(-1, 9))
        dev = ctx.map_tree_export()
        n = _compare(_flat(dev, c), om.export(c["voxel_size"], c["max_layer"]))
        st = ctx.map_tree_stats()
        assert n >= n_prev and st["error"] == 0 and st["touched"] > 500
        n_prev = n


@pytest.mark.parametrize("max_points_num", [100, 7])
def test_max_points_num_of_hilti22(ctx, orc, max_points_num):
    """lio/max_points_num = 100 (reference config/HILTI22.yaml:66; every other shipped config has 50): the point regions of the device tree are sized at creation
    (max_points_num + 2), so nodes keep collecting and re-fitting up to the 100th point and freeze there, like the oracle's (voxel_map.cpp:146-151, 240-245, 280-285).
    A small value (7) freezes almost every plane node during the build."""
    c, cloud, R0, t0, P0, extR, extT = _scene(87)
    c["max_points_num"] = max_points_num
    _, (pw0, var0) = cloud(60000, R0, t0)
    ctx.map_tree_create(c, max_roots=60000)
    ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
    om = orc.OracleMap.build(pw0, var0.reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    n_prev = _compare(_flat(ctx.map_tree_export(), c), om.export(c["voxel_size"], c["max_layer"]))
    assert n_prev > 500
    for k in range(3):
        Rk, tk = R0 @ synth.rot_from_rpy(0.0, 0.0, 0.05 * (k + 1)), t0 + np.array([0.1 * (k + 1), 0.05 * k, 0.0])
        _, (pw, var) = cloud(30000, Rk, tk)
        ctx.map_tree_update(pw, var.reshape(-1, 9))
        om.update(pw, var.reshape(-1, 9))
        n = _compare(_flat(ctx.map_tree_export(), c), om.export(c["voxel_size"], c["max_layer"]))
        assert n >= n_prev and ctx.map_tree_stats()["error"] == 0
        n_prev = n
    with pytest.raises(Exception):
        ctx.map_tree_create(dict(c, max_points_num=100000), max_roots=1000)              # beyond LIVO2_MAX_POINTS_NUM: LIVO2_ERR_INVALID


def test_map_sliding_matches_oracle_and_recycles(ctx, orc):
    """VoxelMapManager::mapSliding / clearMemOutOfMap (reference src/voxel_map.cpp:924-972) on the device tree: the same root voxels disappear as in the oracle,
    a call below sliding_thresh changes nothing, and the next updates take the released nodes / plane rows / point regions before fresh pool memory while the
    tree keeps matching the oracle's."""
    c, cloud, R0, t0, P0, extR, extT = _scene(83)
    _, (pw0, var0) = cloud(40000, R0, t0)
    ctx.map_tree_create(c, max_roots=60000)
    ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
    st = ctx.map_tree_stats()
    # pools sized so that the build leaves them more than half used: from then on the updates run the recycling kernels (the plain ones only bump)
    ctx.map_tree_create(c, max_roots=60000, max_nodes=int(1.6 * st["nodes"]), max_planes=int(1.6 * st["planes"]), max_points=int(1.6 * st["points"]))
    ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
    om = orc.OracleMap.build(pw0, var0.reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    st0 = ctx.map_tree_stats()
    with pytest.raises(Exception):
        ctx.map_tree_slide(np.zeros(3), 1.0, -1)                                           # half_map_size < 0: LIVO2_ERR_INVALID
    # below the threshold (last_slide_position starts at the origin): nothing happens on either side
    assert ctx.map_tree_slide(t0 * 0.0 + 0.5, 8.0, 8)[0] == -1 and om.slide(t0 * 0.0 + 0.5, 8.0, 8) == -1
    assert ctx.map_tree_stats() == st0
    # a slide that keeps a box of +-8 voxels (4 m) around the sensor, far from the origin of last_slide_position: most of the map goes
    pos = np.asarray(t0, float) + 40.0 * 0.0
    thr = 0.5 * float(np.linalg.norm(pos))
    assert thr > 0.1
    removed, free = ctx.map_tree_slide(pos, thr, 8)
    assert removed == om.slide(pos, thr, 8) and removed > 300
    st1 = ctx.map_tree_stats()
    assert st1["roots"] == st0["roots"] - removed and st1["nodes"] == st0["nodes"]            # the pools keep their high-water marks
    assert free["nodes"] >= removed and free["planes"] > 100 and free["slabs"] >= 0
    n_kept = _compare(_flat(ctx.map_tree_export(), c), om.export(c["voxel_size"], c["max_layer"]))
    assert 0 < n_kept < st0["planes"]
    # a second call from (almost) the same place: below the threshold again
    assert ctx.map_tree_slide(pos + 0.01, thr, 8)[0] == -1 and om.slide(pos + 0.01, thr, 8) == -1
    # new scans re-create voxels in the emptied space out of the released resources
    for k in range(3):
        Rk, tk = R0 @ synth.rot_from_rpy(0.0, 0.0, 0.15 * (k + 1)), t0 + np.array([0.2 * (k + 1), 0.1 * k, 0.0])
        _, (pw, var) = cloud(12000, Rk, tk)
        ctx.map_tree_update(pw, var.reshape(-1, 9))
        om.update(pw, var.reshape(-1, 9))
        _compare(_flat(ctx.map_tree_export(), c), om.export(c["voxel_size"], c["max_layer"]))
    st2 = ctx.map_tree_stats()
    _, free2 = ctx.map_tree_slide(pos + 0.02, thr, 8)                                          # (below the threshold: only reports the stacks)
    assert st2["error"] == 0 and st2["roots"] > st1["roots"]
    assert free2["nodes"] < free["nodes"] and free2["planes"] < free["planes"]                # recycled ...
    assert st2["nodes"] - st1["nodes"] < (free["nodes"] - free2["nodes"])                      # ... so the pools grew by less than what the scans needed


def test_tiny_inputs_and_thresholds(ctx, orc):
    """voxels below the init threshold, exactly at it, one point per call (the counters live across calls), an empty call"""
    c = dict(synth.AVIA["lio"])
    rng = np.random.default_rng(5)
    ctx.map_tree_create(c, max_roots=1000)
    om = None
    base = np.array([3.02, -1.98, 0.3])                      # x in [3.02, 3.47], y in [-1.98, -1.53], z ~ 0.4: one root voxel
    pts = base + np.stack([rng.uniform(0, 0.45, 80), rng.uniform(0, 0.45, 80), 0.1 + 0.002 * rng.normal(size=80)], 1)     # one voxel, a thin slab: becomes a plane, then freezes at 50
    pts = pts.astype(np.float32).astype(np.float64)
    A = rng.normal(size=(80, 3, 3)); var = 1e-4 * (A @ A.transpose(0, 2, 1) + 0.1 * np.eye(3))
    ctx.map_tree_update(np.zeros((0, 3)), np.zeros((0, 9)))
    for lo, hi in ((0, 3), (3, 5), (5, 6), (6, 7), (7, 30), (30, 49), (49, 50), (50, 51), (51, 80)):
        ctx.map_tree_update(pts[lo:hi], var[lo:hi].reshape(-1, 9))
        if om is None:
            om = orc.OracleMap.build(np.zeros((0, 3)), np.zeros((0, 9)), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
        om.update(pts[lo:hi], var[lo:hi].reshape(-1, 9))
        _compare(_flat(ctx.map_tree_export(), c), om.export(c["voxel_size"], c["max_layer"]))
    dev = ctx.map_tree_export()
    assert len(dev["root_node"]) == 1
    assert dev["node_plane"][dev["root_node"][0]] >= 0 and dev["node_temp"][dev["root_node"][0]] == 0          # a plane, frozen: temp_points_ released
    _, free = ctx.map_tree_slide(np.zeros(3), 1e9, 1)                                      # (below the threshold: only reports the free stacks)
    assert free["slabs"] == 1                                                             # ... and its 52-point region is back in the pool (k_mt_collect)


def test_lidar_update_reads_the_device_tree(ctx, livo2, orc):
    sc = synth.lidar_scenario(seed=31, n_points=12000, downsample=0.1)
    c = sc.cfg
    # the map: built on the device from the same world points the scenario's numpy BuildVoxelMap used is not available here, so build from a fresh dense sweep
    cs, cloud, R0, t0, P0, extR, extT = _scene(31)
    _, (pw0, var0) = cloud(60000, R0, t0)
    ctx.map_tree_create(cs, max_roots=60000)
    ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
    fm = _flat(ctx.map_tree_export(), cs)
    xyz, _ = cloud(9000, R0, t0)
    sc2 = synth.LidarScenario(fm, np.ascontiguousarray(xyz, np.float32), R0, t0, R0 @ synth.so3_exp(np.array([0.004, -0.003, 0.005])), t0 + np.array([0.02, -0.015, 0.01]),
                              synth.prior_cov(np.random.default_rng(1)), extR, extT, cs)
    pcfg = H.lidar_cfg_product(sc2)
    pcur, pprop = H.states(sc2, livo2.State)
    ctx.set_scan(sc2.xyz, pcfg)
    res_tree, pts_tree = ctx.lidar_update(pcur, pprop, pcfg, want=("match_plane", "dis_to_plane"))
    # the same planes as a snapshot (livo2_map_upload of the exported tree) and in the oracle: same matches, same residual bits
    om = orc.OracleMap.from_flat(fm)
    ocur, oprop = H.states(sc2, orc.StatePOD)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(cs, extR, extT), sc2.xyz, ocur, oprop)
    assert res_tree.n_iters == ref["n_iters"]
    assert np.array_equal(pts_tree["match_plane"], ref["match_plane"]) and np.array_equal(pts_tree["dis_to_plane"], ref["dis"])
    d = H.state_diff(res_tree.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-7, d
    # map maintenance from the posterior, formed on the device from the resident scan (LIVMapper.cpp:413-423), against the same points fed from the host
    Rp, tp, P = np.array(res_tree.state.rot).reshape(3, 3), np.array(res_tree.state.pos), np.array(res_tree.state.cov).reshape(19, 19)
    pl = sc2.xyz.astype(np.float64)
    pi = pl @ extR.T + extT
    pw = (pi @ Rp.T + tp).astype(np.float32).astype(np.float64)
    cb = synth.body_cov(pl, cs["dept_err"], cs["beam_err"])
    RE, X = Rp @ extR, synth.skew(pi)
    var = RE @ cb @ RE.T + X @ P[0:3, 0:3] @ X.transpose(0, 2, 1) + P[3:6, 3:6]
    om_seq = orc.OracleMap.build(pw0, var0.reshape(-1, 9), cs["voxel_size"], cs["max_layer"], cs["layer_init_num"], cs["max_points_num"], cs["min_eigen_value"])   # (from_flat carries no temp_points_)
    om_seq.update(pw, var.reshape(-1, 9))
    ctx.map_tree_update_from_scan(res_tree.state, pcfg)
    n = _compare(_flat(ctx.map_tree_export(), cs), om_seq.export(cs["voxel_size"], cs["max_layer"]), loose=True)
    assert n > 1000 and ctx.map_tree_last_kernel_us() > 0
    # and the next frame's update runs on the refreshed structure
    res2, _ = ctx.lidar_update(pcur, pprop, pcfg)
    assert res2.n_iters >= 1 and abs(res2.iter_sums[0].n_eff - res_tree.iter_sums[0].n_eff) < 0.05 * res_tree.iter_sums[0].n_eff
    ctx.upload_map(fm)                                        # leave a snapshot resident for whatever test comes next on this ctx


def test_map_update_on_the_second_stream_equals_the_synchronous_one(livo2, orc):
    """livo2_map_tree_update_from_scan_async (UpdateVoxelMap beside handleVIO, LIVMapper.cpp:413-424 / 281-334): over three chained frames the tree it leaves equals
    the synchronous call's (same voxels, same plane decisions, same counters), with a visual update enqueued between the fork and the join; state == NULL takes the posterior on the device; an entry point that
    needs the tree (the next LiDAR update, an export) joins by itself; the kernel time is reported after the join."""
    cs, cloud, R0, t0, P0, extR, extT = _scene(47)
    _, (pw0, var0) = cloud(50000, R0, t0)
    vs = synth.visual_scenario(seed=48, n_patches=600)
    vcur, vprop = H.states(vs, livo2.State)
    vcfg = H.visual_cfg_product(vs)
    poses = [(R0 @ synth.rot_from_rpy(0.0, 0.0, 0.05 * f), t0 + np.array([0.15 * f, 0.05 * f, 0.0])) for f in range(3)]
    scans = [cloud(8000, Rf, tf)[0] for Rf, tf in poses]                          # (the scene's generator is stateful: one set of scans for all three modes)
    exports = {}
    for mode in ("sync", "async", "async_device_state_implicit_join"):
        c = livo2.Context(0)
        c.map_tree_create(cs, max_roots=60000)
        c.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
        c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
        rng = np.random.default_rng(5)
        out = []
        for f in range(3):
            (Rf, tf), xyz = poses[f], scans[f]
            sc = synth.LidarScenario(None, np.ascontiguousarray(xyz, np.float32), Rf, tf, Rf @ synth.so3_exp(rng.normal(0, 0.003, 3)), tf + rng.normal(0, 0.01, 3),
                                     synth.prior_cov(np.random.default_rng(1)), extR, extT, cs)
            pcfg = H.lidar_cfg_product(sc)
            pcur, pprop = H.states(sc, livo2.State)
            c.set_scan(sc.xyz, pcfg)
            res, _ = c.lidar_update(pcur, pprop, pcfg)
            if mode == "sync":
                c.map_tree_update_from_scan(res.state, pcfg)
                vres, _ = c.visual_update(vcur, vprop, vcfg)
            else:
                c.map_tree_update_from_scan_async(None if mode.startswith("async_device") else res.state, pcfg)
                vres, _ = c.visual_update(vcur, vprop, vcfg)                      # runs on the context's stream while the octree update runs on the second one
                if mode == "async":
                    c.map_tree_update_join()
                    assert c.map_tree_last_kernel_us() > 10.0
            out.append((bytes(res.state), bytes(vres.state), res.n_iters))
        exports[mode] = (out, c.map_tree_export(), c.map_tree_stats())            # (export / stats join a pending update themselves)
        c.close()
    ref_out, ref_tree, ref_stats = exports["sync"]
    for mode in ("async", "async_device_state_implicit_join"):
        o, t, st = exports[mode]
        assert o == ref_out and st == ref_stats, mode
        assert _compare(_flat(t, cs), _flat(ref_tree, cs)) > 500                 # (node / plane rows are handed out by atomics: compared by structure, values to 1e-13)
    assert ref_stats["planes"] > 500 and ref_stats["error"] == 0


def test_soak_random_walk_with_sliding_grows_recycles_and_matches_oracle(ctx, orc):
    """200 frames of a random walk through a long corridor with mapSliding on (a +-6 m box follows the sensor), pools deliberately small at creation: they must grow on
    demand instead of killing the tree (LIVO2_ERR_RANGE), what sliding and freezing release must be reused (pool high-water marks stay bounded once the box is full),
    candidate ranges are re-packed, and after the last frame the tree has the oracle's shape (voxel_map.cpp:609-641, 924-972)."""
    rng = np.random.default_rng(201)
    c = dict(synth.AVIA["lio"])
    scene = synth.make_room(rng, (80.0, 8.0, 5.0), 30)
    extR, extT = synth.AVIA["extrinsic_R"], synth.AVIA["extrinsic_T"]
    R0 = scene.R_ws @ synth.rot_from_rpy(0.0, 0.0, 0.1)
    t0 = scene.R_ws @ np.array([-30.0, 0.0, 1.4]) + scene.t_ws
    P0 = synth.default_cov() * 1e-3

    def cloud(n, R, t):
        xyz = synth.lidar_scan(rng, scene, R, t, extR, extT, n, c["dept_err"], c["beam_err"], synth.AVIA["blind"], False)
        return synth.world_points_and_var(xyz, R, t, extR, extT, P0, c["dept_err"], c["beam_err"])
    pw0, var0 = cloud(20000, R0, t0)
    ctx.map_tree_create(c, max_roots=40000, max_nodes=9000, max_planes=4000, max_points=260000, max_cand=1500)     # fits the build, not the walk
    ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
    om = orc.OracleMap.build(pw0, var0.reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    thresh, half = 1.0, 12
    marks = []
    R, t = R0, t0
    for k in range(200):
        step = scene.R_ws @ np.array([0.3 if k < 160 else -0.3, rng.normal(0, 0.05), 0.0])
        R, t = R @ synth.rot_from_rpy(0.0, 0.0, rng.normal(0, 0.01)), t + step
        pw, var = cloud(2500, R, t)
        ctx.map_tree_update(pw, var.reshape(-1, 9))
        om.update(pw, var.reshape(-1, 9))
        ra, rb = ctx.map_tree_slide(t, thresh, half)[0], om.slide(t, thresh, half)
        assert ra == rb, k
        st = ctx.map_tree_stats()
        assert st["error"] == 0, (k, st)
        marks.append((st["nodes"], st["points"], st["planes"]))
        if k % 20 == 0:
            print("soak frame", k, st, "grow events", ctx.counter("map_tree_grow_events"))
    assert ctx.counter("map_tree_grow_events") > 0                       # the initial pools were too small
    dev = _flat(ctx.map_tree_export(), c)
    assert _compare(dev, om.export(c["voxel_size"], c["max_layer"])) > 100
    # no leak: while the box moves at constant speed through similar clutter the bump counters stop climbing (recycling covers the demand)
    late = np.array(marks[120:160], float)
    assert late[-1, 0] <= 1.25 * late[0, 0] and late[-1, 1] <= 1.25 * late[0, 1] and late[-1, 2] <= 1.25 * late[0, 2], (late[0], late[-1])


def test_a_frame_that_exhausts_a_pool_is_dropped_and_the_pools_grow_for_the_next_one(livo2, orc):
    """advisor (round 3): after a capacity error the allocators roll back, the pool stays full, and — because growth only ran on error-free frames — every later frame
    failed the same way.  Now the exhausted pools are doubled on the failing call itself: that frame is reported (LIVO2_ERR_RANGE, 'doubled'), the next ones run,
    and a LiDAR update reads the structure."""
    c, cloud, R0, t0, P0, extR, extT = _scene(23)
    _, (pw0, var0) = cloud(6000, R0, t0)
    ctx = livo2.Context(0)
    try:
        ctx.map_tree_create(c, max_roots=20000, max_nodes=6000, max_planes=4000, max_points=400000, max_cand=64)
        ctx.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
        before = ctx.counter("map_tree_grow_events")
        dropped = 0
        for k in range(1, 7):                                            # a walk into unseen space: every frame brings new root voxels
            _, (pw, var) = cloud(30000, R0, t0 + np.array([0.8 * k, 0.3 * k, 0.0]))
            for attempt in range(6):
                try:
                    ctx.map_tree_update(pw, var.reshape(-1, 9))
                    break
                except livo2.Livo2Error as exc:
                    assert exc.code == livo2.abi.ERR_RANGE and "doubled" in str(exc), str(exc)
                    dropped += 1
            else:
                raise AssertionError("the tree never recovered from the capacity error")
        st = ctx.map_tree_stats()
        assert st["error"] == 0 and st["roots"] > 1000
        assert ctx.counter("map_tree_grow_events") > before
        print("dropped frames:", dropped, st)
    finally:
        ctx.close()


def test_three_contexts_update_their_maps_on_second_streams_from_three_host_threads(livo2, orc):
    """Each context owns a second stream and a helper thread for livo2_map_tree_update_from_scan_async; three of them driven from three host threads at once (the ctypes
    calls release the interpreter lock) run the same five chained frames.  Every context must end with the tree, the counters and the posteriors of a context that ran
    the sequence alone with the synchronous call — nothing of the asynchronous path may be shared between contexts (the resident visual grid's admission is, under a lock)."""
    import threading
    cs, cloud, R0, t0, P0, extR, extT = _scene(53)
    _, (pw0, var0) = cloud(40000, R0, t0)
    vs = synth.visual_scenario(seed=54, n_patches=500)
    vcur, vprop = H.states(vs, livo2.State)
    vcfg = H.visual_cfg_product(vs)
    F = 5
    poses = [(R0 @ synth.rot_from_rpy(0.0, 0.0, 0.04 * f), t0 + np.array([0.12 * f, 0.04 * f, 0.0])) for f in range(F)]
    scans = [cloud(9000, Rf, tf)[0] for Rf, tf in poses]
    rng = np.random.default_rng(6)
    frames = []
    for f in range(F):
        (Rf, tf), xyz = poses[f], scans[f]
        sc = synth.LidarScenario(None, np.ascontiguousarray(xyz, np.float32), Rf, tf, Rf @ synth.so3_exp(rng.normal(0, 0.003, 3)), tf + rng.normal(0, 0.01, 3),
                                 synth.prior_cov(np.random.default_rng(1)), extR, extT, cs)
        frames.append((sc, H.lidar_cfg_product(sc), H.states(sc, livo2.State)))

    def run(mode, out, k):
        try:
            c = livo2.Context(0)
            c.map_tree_create(cs, max_roots=60000)
            c.map_tree_update(pw0, var0.reshape(-1, 9), build=True)
            c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
            states = []
            for sc, pcfg, (pcur, pprop) in frames:
                c.set_scan(sc.xyz, pcfg)
                res, _ = c.lidar_update(pcur, pprop, pcfg)
                if mode == "sync":
                    c.map_tree_update_from_scan(res.state, pcfg)
                    vres, _ = c.visual_update(vcur, vprop, vcfg)
                else:
                    c.map_tree_update_from_scan_async(None, pcfg)
                    vres, _ = c.visual_update(vcur, vprop, vcfg)
                states.append((bytes(res.state), bytes(vres.state)))
            out[k] = (states, c.map_tree_stats(), c.map_tree_export())
            c.close()
        except Exception as exc:                                                    # (a thread's exception would otherwise vanish)
            out[k] = exc

    out = {}
    run("sync", out, "ref")
    assert not isinstance(out["ref"], Exception), out["ref"]
    threads = [threading.Thread(target=run, args=("async", out, k)) for k in range(3)]
    for t in threads: t.start()
    for t in threads: t.join()
    ref_states, ref_stats, ref_tree = out["ref"]
    for k in range(3):
        assert not isinstance(out[k], Exception), out[k]
        states, stats, tree = out[k]
        assert states == ref_states, k
        assert stats == ref_stats, (k, stats, ref_stats)
        assert _compare(_flat(tree, cs), _flat(ref_tree, cs), loose=True) > 500
