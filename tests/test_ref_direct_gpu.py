"""The HIP path against the REFERENCE'S OWN translation units, without the oracle in between: oracle/_ref/libref.so (voxel_map.cpp / vio.cpp compiled unmodified,
oracle/ref_build) is prebuilt in the container and travels to the GPU box with the snapshot; /root/reference itself is not read here.  The same seeded frame goes
through `VoxelMapManager::StateEstimation` / `VIOManager::computeJacobianAndUpdateEKF` of the reference and through the C ABI; the bars are those of the
GPU-vs-oracle tests (float decisions identical, accumulated update to 1e-7).  Skipped where the library is absent."""
import os

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref(orc):
    path = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref.so was not built (it is built by __graft_entry__.build() where /root/reference exists)")
    import ctypes as C
    lib = orc.load(path)
    lib.ref_describe.restype = C.c_char_p
    assert b"compiled unmodified" in lib.ref_describe()
    return lib


@pytest.mark.parametrize("seed,n", [(31, 12000), (32, 30000)])
def test_lidar_update_against_the_reference_build(ctx, livo2, orc, ref, seed, n):
    sc = synth.lidar_scenario(seed=seed, n_points=n, downsample=0.1)
    rmap = orc.OracleMap.from_flat(sc.fmap, ref)
    cur, prop = H.states(sc, orc.StatePOD)
    r = orc.lidar_state_estimation(rmap, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), sc.xyz, cur, prop)           # the reference's StateEstimation
    cfg = H.lidar_cfg_product(sc)
    pcur, pprop = H.states(sc, livo2.State)
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
    res, pts = ctx.lidar_update(pcur, pprop, cfg, want=("match_plane", "dis_to_plane", "point_w"))
    assert res.n_iters == r["n_iters"]                                                                           # parsed from the reference's own "[ LIO ]" lines
    assert [res.iter_sums[i].n_eff for i in range(res.n_iters)] == [t.n_eff for t in r["trace"]]
    assert np.array_equal(pts["match_plane"], r["match_plane"]) and np.array_equal(pts["dis_to_plane"], r["dis"]) and np.array_equal(pts["point_w"], r["pw"])
    d = H.state_diff(res.state, r["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8, d


@pytest.mark.parametrize("seed,M,scen_kw,cfg_kw,exact", [(41, 1500, {}, {}, True), (42, 800, dict(distortion=synth.AVIA_RADTAN), {}, True),
                                                           (43, 600, {}, dict(equidistant=synth.HILTI_EQUIDISTANT), False)])
def test_visual_update_against_the_reference_build(ctx, livo2, orc, ref, seed, M, scen_kw, cfg_kw, exact):
    """pinhole / radtan: per-patch float errors bit-identical; equidistant: the model calls atan(), device and host libm may differ in the last bit of a projected
    pixel (tests/test_visual_gpu.py::test_equidistant_camera_of_the_hilti22_config) — errors to 1e-4 relative there, every decision identical"""
    vs = synth.visual_scenario(seed=seed, n_patches=M, **scen_kw)
    cur, prop = H.states(vs, orc.StatePOD)
    r = orc.visual_update(orc.visual_cfg(vs, num_threads=1, **cfg_kw), vs, cur, prop, lib=ref)                  # the reference's computeJacobianAndUpdateEKF (serial build)
    cfg = H.visual_cfg_product(vs, mp_proc_num=1, **cfg_kw)
    pcur, pprop = H.states(vs, livo2.State)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    res, errors = ctx.visual_update(pcur, pprop, cfg)
    if exact:
        assert np.array_equal(errors, r["errors"])                                                               # float patch_error of the last evaluated step, per patch
    else:
        assert np.allclose(errors, r["errors"], rtol=1e-4)
    d = H.state_diff(res.state, r["state"])
    assert d["R"] < 1e-8 and d["t"] < 1e-8 and d["P"] < 1e-7 and d["inv_expo"] < 1e-8, d
    assert H.relerr(np.array(res.G).reshape(19, 19), r["G"]) < 1e-6
