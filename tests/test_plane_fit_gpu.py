"""GPU parity of livo2_plane_fit_batch (device-side VoxelOctoTree::init_plane, reference src/voxel_map.cpp:55-135) against the oracle.

Tolerances: the device reduces each group's sums in a lane-strided order, the oracle serially; covariance_ = E[pp^T] - cc^T cancels
~8 digits at |p| of tens of metres, so eigenvalues agree to ~1e-8 of the largest one, the normal to that over the eigen-gap, and
plane_var_ (which contains 1/(lambda_min - lambda_m)) to ~1e-6 relative.  Same Jacobi sweeps on both sides, so no sign ambiguity."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H
from tests import plane_groups as PG

pytestmark = pytest.mark.gpu
THR = 0.0025


def _check_group(o, r, kind):
    assert o.points_size == r.points_size
    np.testing.assert_allclose(np.array(o.center), np.array(r.center), rtol=1e-14, atol=0)
    big = max(abs(np.array(r.covariance)).max(), 1e-30)
    assert abs(np.array(o.covariance) - np.array(r.covariance)).max() < 1e-7 * big + 1e-11, kind
    assert o.is_plane == r.is_plane, kind
    if not r.is_plane:
        assert not np.any(np.array(o.plane_var)) and o.radius == 0 and o.min_eigen_value == 1.0
        return 0
    gap = max(r.mid_eigen_value - r.min_eigen_value, 1e-12)
    cond = max(1.0, r.max_eigen_value / gap)
    assert abs(o.min_eigen_value - r.min_eigen_value) <= 1e-6 * r.max_eigen_value
    assert abs(o.max_eigen_value - r.max_eigen_value) <= 1e-6 * r.max_eigen_value
    assert abs(o.radius - r.radius) <= 1e-6 * r.radius
    for name in ("normal", "y_normal", "x_normal"):
        a, b = np.array(getattr(o, name)), np.array(getattr(r, name))
        assert np.linalg.norm(a - b) < 1e-7 * cond + 1e-12, (kind, name)
    assert abs(o.d - r.d) < 1e-4 * max(1.0, abs(r.d)) * 1e-2 + 1e-6 * cond
    P, Q = np.array(o.plane_var), np.array(r.plane_var)
    assert np.linalg.norm(P - Q) <= 1e-6 * cond * np.linalg.norm(Q) or kind in ("single", "identical"), kind
    return 1


def test_fit_batch_matches_oracle(ctx, orc):
    pw, var, off, kinds = PG.make_groups(seed=7, n_groups=400)
    # degenerate groups: empty, a single point, identical points
    off = np.concatenate([off, [off[-1], off[-1] + 1, off[-1] + 9]]).astype(np.int32)
    pw = np.concatenate([pw, [[1.0, 2.0, 3.0]], np.tile([[5.0, -6.0, 7.0]], (8, 1))])
    var = np.concatenate([var, np.tile((np.eye(3) * 1e-4).ravel(), (9, 1))])
    kinds = kinds + ["empty", "single", "identical"]
    out = ctx.plane_fit_batch(pw, var, off, THR)
    planes = 0
    for g in range(len(off) - 1):
        if off[g + 1] == off[g]:
            assert out[g].is_plane == 0 and out[g].points_size == 0
            continue
        r = orc.init_plane(pw[off[g]:off[g + 1]], var[off[g]:off[g + 1]], THR)
        if r.is_plane and abs(r.min_eigen_value - THR) < 1e-7:
            continue
        planes += _check_group(out[g], r, kinds[g])
    assert planes > 150


def test_fit_batch_matches_numpy_second_opinion(ctx):
    """The device fit against an evaluation that shares NO code with it or with the oracle (scenarios/synth._fit_planes: batched numpy eigh, Eigen's role in
    voxel_map.cpp:70) — the Jacobi sweeps of oracle and kernel are the same algorithm, so the oracle comparison alone would check an algorithm against its twin."""
    pw, var, off, kinds = PG.make_groups(seed=11, n_groups=300, big=(700,))
    G = len(off) - 1
    gid = np.repeat(np.arange(G), np.diff(off))
    cnt, ctr, ev, evec, pv = synth._fit_planes(pw, var.reshape(-1, 3, 3), gid, G)
    out = ctx.plane_fit_batch(pw, var, off.astype(np.int32), THR)
    n_planes = 0
    for g in range(G):
        o = out[g]
        assert o.points_size == off[g + 1] - off[g]
        np.testing.assert_allclose(np.array(o.center), ctr[g], rtol=1e-13)
        if abs(ev[g, 0] - THR) < 1e-9:
            continue                                           # the plane / no-plane decision at the threshold is a rounding matter
        assert o.is_plane == int(ev[g, 0] < THR), (g, kinds[g], ev[g])
        if not o.is_plane:
            continue
        n_planes += 1
        assert abs(o.min_eigen_value - ev[g, 0]) < 1e-6 * ev[g, 2] + 1e-7 * abs(ev[g, 0])
        assert abs(o.max_eigen_value - ev[g, 2]) < 1e-6 * ev[g, 2]
        n = np.array(o.normal)
        sgn = np.sign(n @ evec[g][:, 0])
        gap = ev[g, 1] - ev[g, 0]
        assert np.linalg.norm(sgn * n - evec[g][:, 0]) < 1e-7 * ev[g, 2] / gap + 1e-12
        assert abs(o.radius - np.sqrt(ev[g, 2])) < 1e-6
        assert abs(o.d + sgn * float(evec[g][:, 0] @ ctr[g])) < 1e-4
        D = np.diag([sgn, sgn, sgn, 1.0, 1.0, 1.0])            # the normal / centre cross blocks carry the solver-dependent sign of the normal
        P, Q = np.array(o.plane_var).reshape(6, 6), D @ pv[g].reshape(6, 6) @ D
        assert np.linalg.norm(P - Q) < 1e-5 * np.linalg.norm(Q) * max(1.0, ev[g, 2] / gap), (g, kinds[g])
    assert n_planes > 100


def test_fit_refreshes_resident_map(ctx, livo2, orc):
    """Fitted records written in place by the kernel == records sent through livo2_map_update_planes from the oracle's fit,
    as seen by the LiDAR update that reads them."""
    sc = synth.lidar_scenario(seed=9, n_points=6000, downsample=0.1)
    pcfg = H.lidar_cfg_product(sc)
    cur, prior = H.states(sc, livo2.State)
    rng = np.random.default_rng(2)
    # re-fit 200 planes of the map from fresh synthetic point groups lying near each plane
    idx = rng.permutation(sc.fmap.n_planes)[:200].astype(np.int32)
    pts, var, off = [], [], [0]
    for p in idx:
        n = int(rng.integers(8, 50))
        nrm, c = sc.fmap.plane_normal[p], sc.fmap.plane_center[p]
        Q, _ = np.linalg.qr(np.c_[nrm, rng.normal(size=(3, 2))])
        q = (rng.normal(size=(n, 3)) * np.array([0.004, 0.12, 0.1])) @ Q.T + c
        pts.append(q.astype(np.float32).astype(np.float64)); var.append(PG.random_spd(rng, n)); off.append(off[-1] + n)
    pts, var, off = np.concatenate(pts), np.concatenate(var).reshape(-1, 9), np.array(off, np.int32)
    ref = [orc.init_plane(pts[off[g]:off[g + 1]], var[off[g]:off[g + 1]], THR) for g in range(len(idx))]
    assert all(r.is_plane for r in ref)
    # A: host path — oracle fit -> livo2_map_update_planes
    ctx.upload_map(sc.fmap)
    ctx.update_planes(idx, np.array([list(r.normal) for r in ref]), np.array([list(r.center) for r in ref]),
                      np.array([list(r.plane_var) for r in ref]), np.array([r.d for r in ref], np.float32), np.array([r.radius for r in ref], np.float32))
    ctx.set_scan(sc.xyz, pcfg)
    sa, pa = ctx.lidar_iterate(cur, prior, pcfg, want=("match_plane", "dis_to_plane"))
    # B: device path — fit + in-place refresh in one call
    ctx.upload_map(sc.fmap)
    out = ctx.plane_fit_batch(pts, var, off, THR, plane_idx=idx)
    assert all(o.is_plane for o in out)
    sb, pb = ctx.lidar_iterate(cur, prior, pcfg, want=("match_plane", "dis_to_plane"))
    assert sa.n_eff > 1000
    same = pa["match_plane"] == pb["match_plane"]
    assert same.mean() > 0.9995                          # records agree to ~1e-8: a gate decision may flip on a knife edge
    assert np.abs(pa["dis_to_plane"][same] - pb["dis_to_plane"][same]).max() < 1e-5
    assert H.relerr(np.array(sb.HtH), np.array(sa.HtH)) < 1e-4
    # and the refresh really happened: the sums differ from those of the untouched map
    ctx.upload_map(sc.fmap)
    s0, _ = ctx.lidar_iterate(cur, prior, pcfg)
    assert H.relerr(np.array(s0.HtH), np.array(sa.HtH)) > 1e-6
