"""VoxelMapManager::BuildVoxelMap / UpdateVoxelMap of the C++ shim — the reference's schedule (src/voxel_map.cpp:532-591, 609-641, 137-290)
with every init_plane evaluated on the device in batched rounds — against the oracle's serial restatement (oracle/orc_voxel_map.hpp):
identical tree shape (which voxels exist, which nodes are planes, which children exist) and plane parameters within the plane-fit
tolerances."""
import os
import subprocess

import numpy as np
import pytest

from scenarios import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "fast-livo2_amd", "lib", "shim_demo")
FIELDS = dict(root_key=np.int64, root_node=np.int32, root_center=np.float64, root_quarter=np.float32, node_plane=np.int32, node_child=np.int32,
              plane_normal=np.float64, plane_center=np.float64, plane_var=np.float64, plane_d=np.float32, plane_radius=np.float32)


def _load(d, prefix):
    a = {k: np.fromfile(os.path.join(d, prefix + k + ".bin"), dtype=t) for k, t in FIELDS.items()}
    return synth.FlatMap(0.5, 2, a["root_key"].reshape(-1, 3), a["root_node"], a["root_center"].reshape(-1, 3), a["root_quarter"], a["node_plane"],
                         a["node_child"].reshape(-1, 8), a["plane_normal"].reshape(-1, 3), a["plane_center"].reshape(-1, 3), a["plane_var"].reshape(-1, 36),
                         a["plane_d"], a["plane_radius"])


def _canon(fm):
    """root key -> nested (plane index or -1, [children...]) with Nones for absent children"""
    def node(n):
        return (int(fm.node_plane[n]), tuple(node(c) if c >= 0 else None for c in fm.node_child[n]))
    return {tuple(int(x) for x in k): (node(int(r)), fm.root_center[i], fm.root_quarter[i]) for i, (k, r) in enumerate(zip(fm.root_key, fm.root_node))}


def _compare(a, b, loose=False):
    """loose: the two maps were fed world points derived from two posteriors that agree to ~1e-9 (float32 world points can differ by an ulp)"""
    tol = 1e6 if loose else 1.0              # loose: one float32 ulp of a world coordinate at ~10 m (1e-6) in a voxel of >= 10 points
    ca, cb = _canon(a), _canon(b)
    assert set(ca) == set(cb), "different sets of root voxels"
    n_planes = 0
    def walk(x, y):
        nonlocal n_planes
        assert (x is None) == (y is None), "a child exists in one tree only"
        if x is None:
            return
        assert (x[0] >= 0) == (y[0] >= 0), "plane / non-plane decision differs"
        if x[0] >= 0:
            n_planes += 1
            i, j = x[0], y[0]
            np.testing.assert_allclose(a.plane_center[i], b.plane_center[j], rtol=1e-13 * tol, atol=1e-13 * tol)
            assert np.linalg.norm(a.plane_normal[i] - b.plane_normal[j]) < 1e-6 * (100 if loose else 1)
            assert abs(a.plane_radius[i] - b.plane_radius[j]) <= 1e-5 * (10 if loose else 1) * b.plane_radius[j] and abs(a.plane_d[i] - b.plane_d[j]) < 1e-4 * (10 if loose else 1)
            assert np.linalg.norm(a.plane_var[i] - b.plane_var[j]) < 1e-4 * (100 if loose else 1) * np.linalg.norm(b.plane_var[j]) + 1e-18
        for p, q in zip(x[1], y[1]):
            walk(p, q)
    for k in ca:
        np.testing.assert_allclose(ca[k][1], cb[k][1]); assert ca[k][2] == cb[k][2]
        walk(ca[k][0], cb[k][0])
    return n_planes


def test_build_and_update_match_oracle(tmp_path, orc):
    d = str(tmp_path)
    rng = np.random.default_rng(81)
    c = dict(synth.AVIA["lio"])
    scene = synth.make_room(rng, (20.0, 20.0, 6.0), 8)
    extR, extT = synth.AVIA["extrinsic_R"], synth.AVIA["extrinsic_T"]
    R0, t0 = scene.R_ws @ synth.rot_from_rpy(0.01, -0.015, 0.4), scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = synth.default_cov() * 1e-3
    def cloud(n, R, t):
        xyz = synth.lidar_scan(rng, scene, R, t, extR, extT, n, c["dept_err"], c["beam_err"], synth.AVIA["blind"], False)
        return synth.world_points_and_var(xyz, R, t, extR, extT, P0, c["dept_err"], c["beam_err"])
    pw0, var0 = cloud(40000, R0, t0)
    pw1, var1 = cloud(12000, R0 @ synth.rot_from_rpy(0.0, 0.0, 0.15), t0 + np.array([0.4, 0.1, 0.0]))      # next scan: overlaps the map and extends it
    pw0.tofile(os.path.join(d, "bld_pw.bin")); var0.reshape(-1, 9).tofile(os.path.join(d, "bld_var.bin"))
    pw1.tofile(os.path.join(d, "upd_pw.bin")); var1.reshape(-1, 9).tofile(os.path.join(d, "upd_var.bin"))
    np.array([c["voxel_size"], c["max_layer"], c["max_points_num"], c["min_eigen_value"]] + list(c["layer_init_num"])[:5], np.float64).tofile(os.path.join(d, "map_cfg.bin"))
    r = subprocess.run([DEMO, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    om = orc.OracleMap.build(pw0, var0.reshape(-1, 9), c["voxel_size"], c["max_layer"], c["layer_init_num"], c["max_points_num"], c["min_eigen_value"])
    n0 = _compare(_load(d, "bld_out_"), om.export(c["voxel_size"], c["max_layer"]))
    om.update(pw1, var1.reshape(-1, 9))
    n1 = _compare(_load(d, "upd_out_"), om.export(c["voxel_size"], c["max_layer"]))
    assert n0 > 1000 and n1 >= n0
    fits, rounds = np.fromfile(os.path.join(d, "bld_out_stats.bin"), dtype=np.int32)
    assert fits > n0 and rounds <= c["max_layer"] + 1          # BuildVoxelMap: one device batch per octree layer
    fits_u, rounds_u = np.fromfile(os.path.join(d, "upd_out_stats.bin"), dtype=np.int32)
    assert fits_u > 100 and rounds_u < 40
    print(r.stdout)
