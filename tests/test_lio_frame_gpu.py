"""livo2_lio_frame — IMU forward propagation -> undistortion + voxel grid -> StateEstimation(state_propagat) as one call (reference
src/LIVMapper.cpp:342-377, src/IMU_Processing.cpp:298-539, src/voxel_map.cpp:338-511).  The fused call must equal, byte for byte, the three
entry points called in sequence (same kernels, the scan-end pose read from the device instead of the host), and the whole chain must agree with
the oracle's chain within the tolerances of its stages."""
import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H
from tests import imu_inputs as IMU

pytestmark = pytest.mark.gpu


def _frame(orc, cls, seed, n_raw=20000, n_steps=20):
    lf = synth.lio_frame_scenario(seed=seed, n_raw=n_raw, n_steps=n_steps)
    sc = lf.sc
    st = orc.make_state(sc.R_prior, sc.t_prior, sc.P, inv_expo=lf.inv_expo, vel=lf.vel, bg=lf.bg, ba=lf.ba, grav=lf.grav, cls=cls)
    return sc, lf.curvature, st, lf.steps, lf.first_pose


@pytest.mark.parametrize("seed,n_steps", [(61, 20), (62, 7)])
def test_fused_frame_equals_the_sequence(ctx, livo2, orc, seed, n_steps):
    sc, curv, st, steps, first = _frame(orc, livo2.State, seed, n_steps=n_steps)
    cfg = H.lidar_cfg_product(sc)
    icfg = orc.imu_cfg(IMU.CFG, cls=livo2.ImuCfg)
    leaf = synth.AVIA["filter_size_surf"]
    ctx.upload_map(sc.fmap)
    # the three entry points in sequence
    prop, poses = ctx.imu_propagate(st, steps, icfg)
    pa = orc.state_arrays(prop)
    nd, und, down = ctx.preprocess_scan(sc.xyz, curv, np.vstack([first, poses]), pa["R"], pa["t"], leaf, cfg)
    ref, _ = ctx.lidar_update(prop, prop, cfg)
    # one call
    res, nd2, prop2, poses2 = ctx.lio_frame(st, steps, icfg, first, sc.xyz, curv, leaf, cfg)
    assert nd2 == nd and nd > 1000
    assert bytes(prop2) == bytes(prop) and np.array_equal(poses2, poses)
    assert res.n_iters == ref.n_iters and bytes(res.state) == bytes(ref.state)
    assert [res.iter_sums[k].n_eff for k in range(res.n_iters)] == [ref.iter_sums[k].n_eff for k in range(ref.n_iters)]
    assert res.iter_sums[0].n_eff > 0.5 * nd                      # the propagated pose is good enough for most points to match
    # the scan of the frame stays resident: a further update on it gives the same answer
    again, _ = ctx.lidar_update(prop, prop, cfg)
    assert bytes(again.state) == bytes(ref.state)


def test_fused_frame_against_the_oracle_chain(ctx, livo2, orc):
    sc, curv, st, steps, first = _frame(orc, livo2.State, 63)
    _, _, ost, _, _ = _frame(orc, orc.StatePOD, 63)
    cfg = H.lidar_cfg_product(sc)
    leaf = synth.AVIA["filter_size_surf"]
    ctx.upload_map(sc.fmap)
    res, nd, prop, poses = ctx.lio_frame(st, steps, orc.imu_cfg(IMU.CFG, cls=livo2.ImuCfg), first, sc.xyz, curv, leaf, cfg)
    oprop, oposes, _ = orc.imu_propagate(ost, steps, IMU.CFG)
    a, b = orc.state_arrays(prop), orc.state_arrays(oprop)
    assert np.abs(a["R"] - b["R"]).max() < 1e-13 and np.abs(a["t"] - b["t"]).max() < 1e-12 and np.abs(a["P"] - b["P"]).max() < 1e-13 * np.abs(b["P"]).max()
    und = orc.undistort(sc.xyz, curv, np.vstack([first, oposes]), b["R"], b["t"], sc.extR, sc.extT)
    down = orc.voxel_grid(und, leaf)
    assert abs(nd - len(down)) <= 2                               # a 1-ulp difference of an undistorted coordinate can move a point across a leaf boundary
    om = orc.OracleMap.from_flat(sc.fmap)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), down, oprop, oprop, want_points=False)
    d = H.state_diff(res.state, ref["state"])
    assert res.n_iters == ref["n_iters"] and d["R"] < 1e-7 and d["t"] < 1e-7 and d["P"] < 1e-6


def test_fused_frame_edges(ctx, livo2, orc):
    sc, curv, st, steps, first = _frame(orc, livo2.State, 64, n_raw=3000, n_steps=5)
    cfg = H.lidar_cfg_product(sc)
    icfg = orc.imu_cfg(IMU.CFG, cls=livo2.ImuCfg)
    c2 = livo2.Context(0)
    with pytest.raises(livo2.Livo2Error) as e:
        c2.lio_frame(st, steps, icfg, first, sc.xyz, curv, 0.1, cfg)            # no map
    assert e.value.code == livo2.abi.ERR_NO_MAP
    c2.upload_map(sc.fmap)
    res, nd, prop, poses = c2.lio_frame(st, steps[:0], icfg, first, sc.xyz, curv, 0.1, cfg)      # no IMU step: state_propagat = state_in, nothing is undistorted
    assert bytes(prop) == bytes(st) and len(poses) == 0 and nd > 100 and res.n_iters >= 1
    res, nd, prop, _ = c2.lio_frame(st, steps, icfg, first, sc.xyz[:0], curv[:0], 0.1, cfg)       # empty scan: the update leaves the propagated state
    assert nd == 0 and np.allclose(np.array(res.state.pos), np.array(prop.pos), atol=1e-12)
    with pytest.raises(livo2.Livo2Error) as e:
        c2.lio_frame(st, steps[::-1], icfg, first, sc.xyz, curv, 0.1, cfg)      # steps out of order
    assert e.value.code == livo2.abi.ERR_INVALID
    c2.close()
