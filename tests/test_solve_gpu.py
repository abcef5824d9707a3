"""The device ESIKF solve (matrix-inversion-lemma form, k x k) against the reference's two-full-inversion form in numpy."""
import numpy as np
import pytest

from oracle import orc as _orc
from scenarios import synth
from tests import helpers as H
from tests import numpy_ref as NR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,sign,scale", [(6, +1, 1.0), (7, -1, 100.0)])
def test_solve_matches_double_inversion(ctx, livo2, k, sign, scale):
    rng = np.random.default_rng(k)
    for trial in range(5):
        A = rng.normal(size=(400, k)) * (30.0 if k == 6 else 3.0)
        HtH = A.T @ A
        Htz = A.T @ rng.normal(size=400) * 0.02
        P = synth.prior_cov(rng)
        Rc, Rp = NR.so3_exp(rng.normal(size=3) * 0.02), NR.so3_exp(rng.normal(size=3) * 0.02)
        tc, tp = rng.normal(size=3), rng.normal(size=3) * 0.01
        cur = _orc.make_state(Rc, tc, P, inv_expo=0.97, vel=rng.normal(size=3), cls=livo2.State)
        prop = _orc.make_state(Rp, tc + tp, P, inv_expo=1.0, vel=rng.normal(size=3), cls=livo2.State)
        out, sol, G = ctx.esikf_solve(HtH, Htz, k, scale, sign, cur, prop)
        vec = np.zeros(19)
        vec[:3], vec[3:6] = NR.boxminus(Rp, tc + tp, Rc, tc)
        vec[6] = 1.0 - 0.97
        vec[7:10] = np.array(prop.vel) - np.array(cur.vel)
        ref_sol, ref_G = NR.esikf_solution(HtH, Htz, k, P, vec, sign=sign, scale=scale)
        assert H.relerr(sol, ref_sol) < 1e-9, (trial, H.relerr(sol, ref_sol))
        assert H.relerr(G, ref_G) < 1e-9
        assert np.allclose(out.R, Rc @ NR.so3_exp(ref_sol[:3]), atol=1e-12)
        assert np.allclose(out.t, tc + ref_sol[3:6], atol=1e-12)
