"""Pins the oracle's init_plane (oracle/orc_voxel_map.hpp, restating src/voxel_map.cpp:55-135) against an independent numpy evaluation
(scenarios/synth._fit_planes: batched eigh instead of the oracle's Jacobi sweeps)."""
import numpy as np

from oracle import orc
from scenarios import synth
from tests import plane_groups as PG

THR = 0.0025          # min_eigen_value (config/avia.yaml:56)


def test_init_plane_matches_numpy_second_opinion():
    pw, var, off, kinds = PG.make_groups(seed=3, n_groups=120, big=(900,))
    G = len(off) - 1
    gid = np.repeat(np.arange(G), np.diff(off))
    cnt, ctr, ev, evec, pv = synth._fit_planes(pw, var.reshape(-1, 3, 3), gid, G)
    n_planes = 0
    for g in range(G):
        o = orc.init_plane(pw[off[g]:off[g + 1]], var[off[g]:off[g + 1]], THR)
        assert o.points_size == off[g + 1] - off[g]
        np.testing.assert_allclose(np.array(o.center), ctr[g], rtol=1e-13)
        if abs(ev[g, 0] - THR) < 1e-9:
            continue
        assert o.is_plane == int(ev[g, 0] < THR), (g, kinds[g], ev[g])
        if not o.is_plane:
            continue
        n_planes += 1
        # E[pp^T] - cc^T loses ~8 digits at |p| ~ 50 m: eigenvalues of the two summation orders agree to ~1e-8 of the largest one
        assert abs(o.min_eigen_value - ev[g, 0]) < 1e-6 * ev[g, 2] + 1e-7 * abs(ev[g, 0])
        assert abs(o.max_eigen_value - ev[g, 2]) < 1e-6 * ev[g, 2]
        n = np.array(o.normal)
        s = np.sign(n @ evec[g][:, 0])
        gap = ev[g, 1] - ev[g, 0]
        assert np.linalg.norm(s * n - evec[g][:, 0]) < 1e-7 * ev[g, 2] / gap + 1e-12
        assert abs(o.radius - np.sqrt(ev[g, 2])) < 1e-6
        assert abs(o.d + s * float(evec[g][:, 0] @ ctr[g])) < 1e-4
        D = np.diag([s, s, s, 1.0, 1.0, 1.0])                 # the normal/centre cross blocks carry the solver-dependent sign of the normal
        P, Q = np.array(o.plane_var).reshape(6, 6), D @ pv[g].reshape(6, 6) @ D
        assert np.linalg.norm(P - Q) < 1e-5 * np.linalg.norm(Q) * max(1.0, ev[g, 2] / gap), (g, kinds[g])
    assert n_planes > 40
