"""GPU parity of livo2_imu_propagate (reference src/IMU_Processing.cpp:298-445) against the oracle: same k-ascending 19x19 dot products and 3x3
operation order, so the two differ only through sin/cos (last f64 bit) — tolerance 1e-13 relative on the covariance, 1e-14 on the pose."""
import numpy as np
import pytest

from tests import imu_inputs as I

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,flags,first", [(0, 20, (1, 1, 1), 0), (1, 3, (0, 0, 0), 0), (2, 200, (1, 0, 1), 0), (3, 20, (1, 1, 1), 1)])
def test_imu_propagate_matches_oracle(ctx, livo2, orc, seed, n, flags, first):
    c = dict(I.CFG); c["ba_bg_est_en"], c["gravity_est_en"], c["exposure_estimate_en"] = flags
    c["first_call"] = first                     # !imu_time_init: inv_expo_time leaves the first call as 1.0 (IMU_Processing.cpp:305-317, 444)
    steps = I.make_steps(seed, n=n)
    ref, rposes, _ = orc.imu_propagate(I.make_state(orc, orc.StatePOD, seed), steps, c)
    out, poses = ctx.imu_propagate(I.make_state(orc, livo2.State, seed), steps, orc.imu_cfg(c, cls=livo2.ImuCfg))
    a, b = orc.state_arrays(out), orc.state_arrays(ref)
    assert np.abs(a["R"] - b["R"]).max() < 1e-13 and np.abs(a["t"] - b["t"]).max() < 1e-12 and np.abs(a["vel"] - b["vel"]).max() < 1e-12
    assert np.abs(a["P"] - b["P"]).max() < 1e-13 * np.abs(b["P"]).max()
    assert np.abs(poses - rposes).max() < 1e-11
    assert a["inv_expo"] == b["inv_expo"] and np.array_equal(a["bg"], b["bg"]) and np.array_equal(a["grav"], b["grav"])
    assert (a["inv_expo"] == 1.0) if first else (a["inv_expo"] != 1.0)


def test_imu_edges(ctx, livo2, orc):
    st = I.make_state(orc, livo2.State, 5)
    out, poses = ctx.imu_propagate(st, I.make_steps(5, n=4)[:0], orc.imu_cfg(I.CFG, cls=livo2.ImuCfg))
    assert bytes(out) == bytes(st) and len(poses) == 0
    bad = dict(I.CFG); bad["mean_acc_norm"] = 0.0
    with pytest.raises(Exception):
        ctx.imu_propagate(st, I.make_steps(5, n=4), orc.imu_cfg(bad, cls=livo2.ImuCfg))
