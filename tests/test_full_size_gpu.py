"""BASELINE.json's full sizes (C4: 200k LiDAR points, 4k visual patches): parity of the whole update against the oracle plus
size-independent properties of the device reduction (additivity over a split scan, invariance under a permutation of the input)."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc_c4():
    return synth.lidar_scenario(seed=4, n_points=200000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays_factor=8)   # random point order on purpose


def test_c4_lidar_update_vs_oracle(ctx, livo2, orc, sc_c4):
    sc = sc_c4
    assert len(sc.xyz) == 200000
    om = orc.OracleMap.from_flat(sc.fmap)
    ocur, oprop = H.states(sc, orc.StatePOD)
    pcur, pprop = H.states(sc, livo2.State)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT, num_threads=4), sc.xyz, ocur, oprop)
    pcfg = H.lidar_cfg_product(sc)
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, pcfg)
    res, pts = ctx.lidar_update(pcur, pprop, pcfg, want=("match_plane", "dis_to_plane"))
    assert res.n_iters == ref["n_iters"]
    assert np.array_equal(pts["match_plane"], ref["match_plane"]), int((pts["match_plane"] != ref["match_plane"]).sum())
    assert np.array_equal(pts["dis_to_plane"], ref["dis"])
    for it in range(res.n_iters):
        assert res.iter_sums[it].n_eff == ref["trace"][it].n_eff
        assert H.relerr(np.array(res.iter_sums[it].HtH), np.array(ref["trace"][it].HtH)) < 1e-9
    so, sp = orc.state_arrays(ref["state"]), orc.state_arrays(res.state)
    dx_ref = np.concatenate([so["t"] - sc.t_prior, (sc.R_prior.T @ so["R"] - np.eye(3)).ravel()])
    dx_gpu = np.concatenate([sp["t"] - sc.t_prior, (sc.R_prior.T @ sp["R"] - np.eye(3)).ravel()])
    assert H.relerr(dx_gpu, dx_ref) < 1e-7            # contract: 1e-5
    assert H.relerr(sp["P"], so["P"]) < 1e-8


def test_c4_reduction_properties(ctx, livo2, sc_c4):
    sc = sc_c4
    pcfg = H.lidar_cfg_product(sc)
    pcur, pprop = H.states(sc, livo2.State)
    ctx.upload_map(sc.fmap)

    def sums_of(xyz):
        ctx.set_scan(xyz, pcfg)
        s, p = ctx.lidar_iterate(pcur, pprop, pcfg, want=("match_plane",))
        return np.array(s.HtH), np.array(s.Htz), s.n_eff, s.total_residual, p["match_plane"]

    HtH, Htz, n_eff, tr, match = sums_of(sc.xyz)
    # additivity: the sums of two halves of the scan add up to the sums of the whole scan
    a = sums_of(sc.xyz[:123457]); b = sums_of(sc.xyz[123457:])
    assert a[2] + b[2] == n_eff
    assert H.relerr(a[0] + b[0], HtH) < 1e-12 and H.relerr(a[1] + b[1], Htz) < 1e-10 and abs(a[3] + b[3] - tr) < 1e-9 * tr
    assert np.array_equal(np.concatenate([a[4], b[4]]), match)
    # permutation invariance: the decisions are per point, the sums agree to rounding
    perm = np.random.default_rng(0).permutation(len(sc.xyz))
    c = sums_of(sc.xyz[perm])
    assert c[2] == n_eff and np.array_equal(c[4], match[perm])
    assert H.relerr(c[0], HtH) < 1e-12 and H.relerr(c[1], Htz) < 1e-10
    # fixed point: restarting the iterated filter at the converged pose (prior covariance, same prior) barely moves it
    ctx.set_scan(sc.xyz, pcfg)
    res, _ = ctx.lidar_update(pcur, pprop, pcfg)
    conv = res.state.copy()
    conv.cov[:] = list(pcur.cov)
    res2, _ = ctx.lidar_update(conv, pprop, pcfg)
    first = np.linalg.norm(np.array(res.iter_solution[0])[:6])
    assert np.linalg.norm(np.array(res2.iter_solution[0])[:6]) < 0.02 * first


def test_c4_visual_update_vs_oracle(ctx, livo2, orc):
    vs = synth.visual_scenario(seed=6, n_patches=4000)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_update(orc.visual_cfg(vs, num_threads=4), vs, ocur, oprop)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    res, errors = ctx.visual_update(pcur, pprop, H.visual_cfg_product(vs))
    steps_gpu = [(s.level, s.iteration, s.accepted) for s in res.steps[:res.n_steps]]
    steps_ref = [(t.level, t.iteration, t.accepted) for t in ref["trace"]]
    assert steps_gpu == steps_ref
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-8 and d["t"] < 1e-8 and d["P"] < 1e-7, d
    assert np.allclose(errors, ref["errors"], rtol=1e-5)
