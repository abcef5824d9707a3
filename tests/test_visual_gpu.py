"""GPU parity: visual photometric pass and the full computeJacobianAndUpdateEKF loop vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vs_small():
    return synth.visual_scenario(seed=3, n_patches=300)


@pytest.mark.parametrize("level", [3, 0])
@pytest.mark.parametrize("exposure", [True, False])
def test_iterate_matches_oracle(ctx, livo2, orc, vs_small, level, exposure):
    vs = vs_small
    ocfg = orc.visual_cfg(vs, exposure=exposure)
    pcfg = H.visual_cfg_product(vs, exposure=exposure)
    ocur, _ = H.states(vs, orc.StatePOD)
    pcur, _ = H.states(vs, livo2.State)
    ref = orc.visual_iterate(ocfg, vs, level, ocur)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    sums, errors, z, Hs = ctx.visual_iterate(level, pcur, pcfg, rows=True)
    assert sums.n_meas == ref["n_meas"] == 64 * len(vs.pos)
    # residuals: same float32 bilinear samples, same double expression -> identical
    assert np.array_equal(z, ref["z"]), f"max |dz| = {np.abs(z - ref['z']).max()}"
    assert H.relerr(Hs, ref["H"]) < 1e-14
    assert np.allclose(errors, ref["errors"], rtol=2e-6, atol=0)
    assert abs(sums.error - ref["error"]) <= 2e-6 * abs(ref["error"])
    # moment-factorised H^T H / H^T z vs the dense row-by-row sums of the oracle
    assert H.relerr(np.array(sums.HtH).reshape(7, 7), ref["HtH"]) < 1e-11
    assert H.relerr(np.array(sums.Htz), ref["Htz"]) < 1e-10
    # without rows requested the production kernel variant must give the same sums
    sums2, errors2, _, _ = ctx.visual_iterate(level, pcur, pcfg, rows=False)
    assert np.array_equal(np.array(sums2.HtH), np.array(sums.HtH)) and np.array_equal(errors2, errors)


def test_full_update_matches_oracle(ctx, livo2, orc, vs_small):
    vs = vs_small
    ocfg = orc.visual_cfg(vs)
    pcfg = H.visual_cfg_product(vs)
    ocur, oprop = H.states(vs, orc.StatePOD)
    pcur, pprop = H.states(vs, livo2.State)
    ref = orc.visual_update(ocfg, vs, ocur, oprop)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    res, errors = ctx.visual_update(pcur, pprop, pcfg)
    assert res.n_steps == len(ref["trace"]), (res.n_steps, len(ref["trace"]))
    for k in range(res.n_steps):
        a, b = res.steps[k], ref["trace"][k]
        assert (a.level, a.iteration, a.accepted, a.n_meas) == (b.level, b.iteration, b.accepted, b.n_meas), k
        assert abs(a.error - b.error) <= 4e-6 * abs(b.error)
        if a.accepted:
            assert H.relerr(np.array(a.HtH), np.array(b.HtH)) < 1e-8
            assert H.relerr(np.array(a.solution), np.array(b.solution)) < 1e-6
    d = H.state_diff(res.state, ref["state"])
    assert d["R"] < 1e-9 and d["t"] < 1e-9 and d["P"] < 1e-8 and d["inv_expo"] < 1e-9, d
    assert H.relerr(np.array(res.G).reshape(19, 19), ref["G"]) < 1e-7
    assert H.relerr(np.array(res.Rcw).reshape(3, 3), ref["Rcw"]) < 1e-12 and H.relerr(np.array(res.Pcw), ref["Pcw"]) < 1e-9
    assert np.allclose(errors, ref["errors"], rtol=1e-5)
