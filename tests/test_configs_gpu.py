"""BASELINE.json configs[1] and configs[2] at their full sizes inside `pytest -m gpu` (they only ran in the off-suite sweep).
C2: 100 000 synthetic LiDAR rays through the reference's 0.1 m voxel grid (-> 93 746 points), full StateEstimation against the oracle.
C3: the C2 LiDAR update + the full visual update of 2 000 patches (8x8) of the same frame, each against the oracle."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc_c2():
    return synth.lidar_scenario(seed=2, n_points=100000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays_factor=12, downsample=synth.AVIA["filter_size_surf"])


def _lidar_parity(ctx, livo2, orc, sc):
    om = orc.OracleMap.from_flat(sc.fmap)
    ocur, oprop = H.states(sc, orc.StatePOD)
    pcur, pprop = H.states(sc, livo2.State)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT, num_threads=4), sc.xyz, ocur, oprop)
    pcfg = H.lidar_cfg_product(sc)
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, pcfg)
    res, pts = ctx.lidar_update(pcur, pprop, pcfg, want=("match_plane", "dis_to_plane", "point_w"))
    assert res.n_iters == ref["n_iters"]
    assert np.array_equal(pts["match_plane"], ref["match_plane"]), int((pts["match_plane"] != ref["match_plane"]).sum())
    assert np.array_equal(pts["dis_to_plane"], ref["dis"]) and np.array_equal(pts["point_w"], ref["pw"])
    for it in range(res.n_iters):
        assert res.iter_sums[it].n_eff == ref["trace"][it].n_eff
        assert H.relerr(np.array(res.iter_sums[it].HtH), np.array(ref["trace"][it].HtH)) < 1e-9
    so, sp = orc.state_arrays(ref["state"]), orc.state_arrays(res.state)
    dx_ref = np.concatenate([so["t"] - sc.t_prior, (sc.R_prior.T @ so["R"] - np.eye(3)).ravel()])
    dx_gpu = np.concatenate([sp["t"] - sc.t_prior, (sc.R_prior.T @ sp["R"] - np.eye(3)).ravel()])
    assert H.relerr(dx_gpu, dx_ref) < 1e-7            # contract: 1e-5
    assert H.relerr(sp["P"], so["P"]) < 1e-8
    return res


def test_c2_lidar_update_vs_oracle(ctx, livo2, orc, sc_c2):
    assert 90000 < len(sc_c2.xyz) <= 100000
    _lidar_parity(ctx, livo2, orc, sc_c2)


def test_c3_lidar_plus_visual_vs_oracle(ctx, livo2, orc, sc_c2):
    _lidar_parity(ctx, livo2, orc, sc_c2)
    vs = synth.visual_scenario(seed=3, n_patches=2000)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_update(orc.visual_cfg(vs, num_threads=4), vs, ocur, oprop)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    res, errors = ctx.visual_update(pcur, pprop, H.visual_cfg_product(vs, mp_proc_num=4))
    assert [(res.steps[k].level, res.steps[k].iteration, res.steps[k].accepted, res.steps[k].n_meas, res.steps[k].error) for k in range(res.n_steps)] == \
           [(t.level, t.iteration, t.accepted, t.n_meas, t.error) for t in ref["trace"]]
    assert np.array_equal(errors, ref["errors"])
    for k in range(res.n_steps):
        if res.steps[k].accepted:
            assert H.relerr(np.array(res.steps[k].HtH), np.array(ref["trace"][k].HtH)) < 1e-8
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8 and d["inv_expo"] < 1e-9, d
