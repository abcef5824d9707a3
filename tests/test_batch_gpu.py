"""GPU parity of the batched-frames path (livo2_lidar_batch_*): several independent StateEstimation problems against one resident
map in one grid per ESIKF iteration.  Every frame must make the same discrete decisions as its own livo2_lidar_update call (iteration
count, n_eff per iteration) and agree with it to rounding (the batched grid uses 64-point blocks, the single-scan grid 256-point
blocks: same per-point arithmetic, different grouping of the partial sums), and be within the usual tolerance of the CPU oracle
(src/voxel_map.cpp:338-511 restated in oracle/).  Two batched runs are bit-identical to each other."""
import ctypes as C

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    return synth.lidar_scenario(seed=5, n_points=12000, downsample=0.1)


def _frames(sc, orc):
    """(scan, R_iterate, t_iterate) per frame: different sub-scans of the scenario and different priors around the true pose."""
    rng = np.random.default_rng(11)
    n = len(sc.xyz)
    scans = [sc.xyz, sc.xyz[: n // 2], sc.xyz[rng.permutation(n)[: n // 3]], sc.xyz[:0], sc.xyz[n // 4:], sc.xyz[:300]]
    poses = []
    for k in range(len(scans)):
        dth = rng.normal(0, np.deg2rad(0.3), 3)
        dp = rng.normal(0, 0.02, 3)
        poses.append((sc.R_prior @ _rodrigues(dth), sc.t_prior + dp))
    return scans, poses


def _rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def _raw(x):
    return C.string_at(C.addressof(x), C.sizeof(x))


def _bytes(r):
    """the defined part of a livo2_lidar_result (iterations >= n_iters are unspecified)"""
    return (_raw(r.state), r.n_iters, r.converged, [_raw(r.iter_sums[k]) for k in range(r.n_iters)],
            [_raw(r.iter_solution[k]) for k in range(r.n_iters)], _raw(r.position_last))


def test_batch_equals_single_and_oracle(ctx, livo2, orc, sc):
    pcfg = H.lidar_cfg_product(sc)
    ocfg = orc.lidar_cfg(sc.cfg, sc.extR, sc.extT)
    om = orc.OracleMap.from_flat(sc.fmap)
    scans, poses = _frames(sc, orc)
    ctx.upload_map(sc.fmap)
    # reference 1: each frame alone through the single-frame entry points
    single, pstates = [], []
    for xyz, (R, t) in zip(scans, poses):
        prior = orc.make_state(R, t, sc.P, cls=livo2.State)
        cur = orc.make_state(R, t, sc.P, cls=livo2.State)
        pstates.append((cur, prior))
        ctx.set_scan(xyz, pcfg)
        res, _ = ctx.lidar_update(cur, prior, pcfg)
        single.append(res)
    # the batch
    ctx.batch_set_scans(scans, pcfg)
    batch = ctx.batch_update([c for c, _ in pstates], [p for _, p in pstates], pcfg)
    assert len(batch) == len(scans)
    for f, (b, s_) in enumerate(zip(batch, single)):
        assert b.n_iters == s_.n_iters and b.converged == s_.converged, f"frame {f}"
        for it in range(b.n_iters):
            assert b.iter_sums[it].n_eff == s_.iter_sums[it].n_eff
            assert H.relerr(np.array(b.iter_sums[it].HtH), np.array(s_.iter_sums[it].HtH)) < 1e-12 or b.iter_sums[it].n_eff == 0
            assert np.abs(np.array(b.iter_solution[it]) - np.array(s_.iter_solution[it])).max() < 1e-12
        d = H.state_diff(b.state, s_.state)
        assert d["R"] < 1e-13 and d["t"] < 1e-13 and d["P"] < 1e-11, (f, d)
    # reference 2: the oracle
    for f, (xyz, (R, t)) in enumerate(zip(scans, poses)):
        if len(xyz) == 0:
            assert batch[f].n_iters >= 1 and batch[f].iter_sums[0].n_eff == 0
            continue
        ocur = orc.make_state(R, t, sc.P, cls=orc.StatePOD)
        oprior = orc.make_state(R, t, sc.P, cls=orc.StatePOD)
        ref = orc.lidar_state_estimation(om, ocfg, xyz, ocur, oprior)
        assert batch[f].n_iters == ref["n_iters"], f"frame {f}"
        for it in range(ref["n_iters"]):
            tr = ref["trace"][it]
            assert batch[f].iter_sums[it].n_eff == tr.n_eff
            assert H.relerr(np.array(batch[f].iter_solution[it]), np.array(tr.solution)) < 1e-7
        d = H.state_diff(batch[f].state, ref["state"])
        assert d["R"] < 1e-11 and d["t"] < 1e-11 and d["P"] < 1e-9, (f, d)


def test_batch_rerun_and_resize(ctx, livo2, orc, sc):
    """A second update on the same batch gives the same answer; a later, smaller batch does not see stale frames."""
    pcfg = H.lidar_cfg_product(sc)
    scans, poses = _frames(sc, orc)
    ctx.upload_map(sc.fmap)
    st = [orc.make_state(R, t, sc.P, cls=livo2.State) for R, t in poses]
    ctx.batch_set_scans(scans, pcfg)
    a = ctx.batch_update(st, st, pcfg)
    ctx.batch_update_async(st, st, pcfg)
    b = ctx.batch_update_fetch()
    assert all(_bytes(x) == _bytes(y) for x, y in zip(a, b))
    ctx.batch_set_scans(scans[4:], pcfg)
    c = ctx.batch_update(st[4:], st[4:], pcfg)
    assert len(c) == 2 and _bytes(c[0]) == _bytes(a[4]) and _bytes(c[1]) == _bytes(a[5])


def test_batch_errors(ctx, livo2, orc, sc):
    pcfg = H.lidar_cfg_product(sc)
    ctx.upload_map(sc.fmap)
    with pytest.raises(Exception):
        ctx.batch_set_scans([sc.xyz[:10]] * 65, pcfg)                 # > LIVO2_MAX_BATCH
    ctx.batch_set_scans([sc.xyz[:100], sc.xyz[:50]], pcfg)
    st = orc.make_state(sc.R_prior, sc.t_prior, sc.P, cls=livo2.State)
    ctx.batch_n = 3
    with pytest.raises(Exception):
        ctx.batch_update([st] * 3, [st] * 3, pcfg)                    # frame count differs from the uploaded batch
