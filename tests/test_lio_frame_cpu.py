"""CPU side of the one-call LiDAR-inertial frame (tests/test_lio_frame_gpu.py): the oracle's chain — IMU forward propagation -> undistortion ->
voxel grid -> StateEstimation(state_propagat) — on the seeded frame scenario behaves like a real frame (small propagation drift, most points
matched, the update pulls the pose back towards the truth), so that the GPU test compares something meaningful."""
import numpy as np

from oracle import orc
from scenarios import synth


def test_oracle_chain_on_the_frame_scenario():
    lf = synth.lio_frame_scenario(seed=63, n_raw=12000, n_steps=20)
    sc = lf.sc
    st = orc.make_state(sc.R_prior, sc.t_prior, sc.P, inv_expo=lf.inv_expo, vel=lf.vel, bg=lf.bg, ba=lf.ba, grav=lf.grav)
    prop, poses, _ = orc.imu_propagate(st, lf.steps, lf.imu)
    a, b = orc.state_arrays(st), orc.state_arrays(prop)
    assert np.abs(b["t"] - a["t"]).max() < 0.01 and np.abs(b["R"] - a["R"]).max() < 2e-3           # a sensor almost at rest for 100 ms
    assert np.all(np.diag(b["P"]) >= np.diag(a["P"]) - 1e-15)                                       # the prediction never shrinks a variance
    assert len(poses) == len(lf.steps) and np.allclose(poses[-1][10:13], b["t"]) and np.allclose(poses[-1][13:22].reshape(3, 3), b["R"])
    und = orc.undistort(sc.xyz, lf.curvature, np.vstack([lf.first_pose, poses]), b["R"], b["t"], sc.extR, sc.extT)
    assert 0 < np.abs(und - sc.xyz).max() < 0.01                                                   # mm-level compensation
    down = orc.voxel_grid(und, synth.AVIA["filter_size_surf"])
    assert 0.3 * len(und) < len(down) < len(und)
    om = orc.OracleMap.from_flat(sc.fmap)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), down, prop, prop, want_points=False)
    assert ref["n_iters"] >= 2 and ref["trace"][0].n_eff > 0.9 * len(down)
    post = orc.state_arrays(ref["state"])
    assert np.all(np.diag(post["P"])[:6] < np.diag(b["P"])[:6])                                     # the measurement update tightens the pose block


def test_first_pose_is_the_state_before_the_frame():
    lf = synth.lio_frame_scenario(seed=64, n_raw=500, n_steps=3)
    fp = lf.first_pose
    assert fp[0] == 0.0 and np.array_equal(fp[7:10], lf.vel) and np.array_equal(fp[10:13], lf.sc.t_prior) and np.array_equal(fp[13:22].reshape(3, 3), lf.sc.R_prior)
    assert np.all(np.diff(lf.steps[:, 7]) > 0) and lf.steps[0, 7] > fp[0]                           # offs_t ascending behind the first pose
    assert np.all(np.diff(lf.curvature) >= 0) and lf.curvature.dtype == np.float32
