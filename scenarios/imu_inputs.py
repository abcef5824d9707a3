"""Synthetic IMU step lists for the forward-propagation tests (reference src/IMU_Processing.cpp:298-445)."""
import numpy as np

CFG = dict(cov_gyr=[0.1, 0.1, 0.1], cov_acc=[0.1, 0.12, 0.09], cov_bias_gyr=[1e-4, 1.2e-4, 0.9e-4], cov_bias_acc=[1e-4, 1e-4, 2e-4], cov_inv_expo=0.2,
           G_m_s2=9.81, mean_acc_norm=9.79, ba_bg_est_en=1, gravity_est_en=1, exposure_estimate_en=1)


def make_steps(seed=0, n=20, hz=200.0):
    rng = np.random.default_rng(seed)
    dt = 1.0 / hz
    steps = np.c_[rng.normal(0, 0.4, (n, 3)) + [0.1, -0.2, 0.3], rng.normal(0, 0.6, (n, 3)) + [0.2, -0.1, 9.8], np.full(n, dt), np.arange(1, n + 1) * dt]
    steps[0, 6] = 0.6 * dt                      # first step: from last_prop_end_time (IMU_Processing.cpp:355-360)
    steps[-1, 6] = 0.4 * dt                     # last step: up to prop_end_time (367-372)
    steps[-1, 7] = steps[-2, 7] + 0.4 * dt
    return steps


def make_state(orc, cls, seed=0):
    rng = np.random.default_rng(seed + 100)
    A = rng.normal(size=(19, 19))
    P = 1e-3 * (A @ A.T) / 19 + np.diag(np.full(19, 1e-4))
    w = rng.normal(0, 0.3, 3)
    th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    return orc.make_state(R, rng.normal(0, 2, 3), P, inv_expo=0.93, vel=rng.normal(0, 1, 3), bg=rng.normal(0, 0.01, 3), ba=rng.normal(0, 0.05, 3), grav=[0.05, -0.02, -9.81], cls=cls)
