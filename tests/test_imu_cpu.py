"""Pins the oracle's IMU forward propagation (oracle/orc_imu.hpp, reference src/IMU_Processing.cpp:298-445) with a plain numpy evaluation."""
import numpy as np

from oracle import orc
from tests import imu_inputs as I


def _exp(w, dt):
    n = np.linalg.norm(w)
    if n < 1e-7:
        return np.eye(3)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / n
    a = n * dt
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _numpy(st, steps, c):
    s = orc.state_arrays(st)
    R, p, v, P = s["R"].copy(), s["t"].copy(), s["vel"].copy(), s["P"].copy()
    poses = []
    for x in steps:
        w = x[:3] - s["bg"]; acc = x[3:6] * c["G_m_s2"] / c["mean_acc_norm"] - s["ba"]; dt = x[6]
        F, W = np.eye(19), np.zeros((19, 19))
        F[0:3, 0:3] = _exp(w, -dt)
        if c["ba_bg_est_en"]:
            F[0:3, 10:13] = -np.eye(3) * dt; F[7:10, 13:16] = -R * dt
        F[3:6, 7:10] = np.eye(3) * dt
        sk = np.array([[0, -acc[2], acc[1]], [acc[2], 0, -acc[0]], [-acc[1], acc[0], 0]])
        F[7:10, 0:3] = -R @ sk * dt
        if c["gravity_est_en"]:
            F[7:10, 16:19] = np.eye(3) * dt
        if c["exposure_estimate_en"]:
            W[6, 6] = c["cov_inv_expo"] * dt * dt
        W[0:3, 0:3] = np.diag(c["cov_gyr"]) * dt * dt
        W[7:10, 7:10] = R @ np.diag(c["cov_acc"]) @ R.T * dt * dt
        W[10:13, 10:13] = np.diag(c["cov_bias_gyr"]) * dt * dt; W[13:16, 13:16] = np.diag(c["cov_bias_acc"]) * dt * dt
        P = F @ P @ F.T + W
        R = R @ _exp(w, dt); ai = R @ acc + s["grav"]; p = p + v * dt + 0.5 * ai * dt * dt; v = v + ai * dt
        poses.append(np.concatenate([[x[7]], ai, w, v, p, R.ravel()]))
    return R, p, v, P, np.array(poses)


def test_imu_propagation_matches_numpy():
    for seed, flags in ((0, (1, 1, 1)), (1, (0, 0, 0)), (2, (1, 0, 1))):
        c = dict(I.CFG); c["ba_bg_est_en"], c["gravity_est_en"], c["exposure_estimate_en"] = flags
        st, steps = I.make_state(orc, orc.StatePOD, seed), I.make_steps(seed, n=25)
        out, poses, _ = orc.imu_propagate(st, steps, c)
        R, p, v, P, ps = _numpy(st, steps, c)
        o = orc.state_arrays(out)
        assert np.abs(o["R"] - R).max() < 1e-14 and np.abs(o["t"] - p).max() < 1e-13 and np.abs(o["vel"] - v).max() < 1e-13
        assert np.abs(o["P"] - P).max() < 1e-15 + 1e-13 * np.abs(P).max()
        assert np.abs(poses - ps).max() < 1e-12
        i = orc.state_arrays(st)
        assert o["inv_expo"] == i["inv_expo"] and np.array_equal(o["bg"], i["bg"]) and np.array_equal(o["ba"], i["ba"]) and np.array_equal(o["grav"], i["grav"])
    out0, poses0, _ = orc.imu_propagate(st, steps[:0], c)               # no IMU sample: the state is returned unchanged
    assert bytes(out0) == bytes(st)
