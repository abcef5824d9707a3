// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the IMU forward propagation that produces state_propagat / its covariance and the IMUpose list:
//   ImuProcess::UndistortPcl, forward loop (LIO / VIO branch)        src/IMU_Processing.cpp:298-445
//   first IMUpose entry                                               src/IMU_Processing.cpp:281
//   Exp(ang_vel, dt)                                                  include/utils/so3_math.h:24-43
//   G_m_s2                                                            include/common_lib.h:29
// The time-stamp logic that turns the IMU message queue into (dt, offs_t) per step (IMU_Processing.cpp:332, 355-372) is the caller's; a
// step carries the averaged raw measurements 0.5 * (head + tail) (335-341), dt and offs_t.  The 19x19 products follow the expression
// F_x * cov * F_x^T + cov_w left to right with plain k-ascending dot products (Eigen's GEMM order / FMA use is build dependent: compare
// with a tolerance, 1e-12).
#pragma once
#include "orc_preprocess.hpp"
#include "orc_state.hpp"

namespace orc {

struct ImuStep { double gyr[3], acc[3], dt, offs_t; };
struct ImuCfg {
  double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3], cov_inv_expo;
  double G_m_s2, mean_acc_norm;                     // acc_avr * G_m_s2 / mean_acc.norm()
  int ba_bg_est_en, gravity_est_en, exposure_estimate_en;
  int first_call = 0;          // !imu_time_init (IMU_Processing.cpp:305-310): tau = 1.0
};

// state_inout: in = state at prop_beg_time, out = state_propagat.  poses[n_steps] receives the Pose6D pushed per step (the entry at offset 0,
// IMU_Processing.cpp:281, is the caller's: it is the previous frame's last pose).
inline void imu_propagate(StatesGroup &state_inout, const ImuStep *steps, int n_steps, const ImuCfg &cfg, Pose6D *poses) {
  V3 vel_imu = state_inout.vel_end, pos_imu = state_inout.pos_end;
  M3 R_imu = state_inout.rot_end;
  const double tau = cfg.first_call ? 1.0 : state_inout.inv_expo_time;      // IMU_Processing.cpp:305-317
  for (int i = 0; i < n_steps; i++) {
    V3 angvel_avr = vec3(steps[i].gyr[0], steps[i].gyr[1], steps[i].gyr[2]);
    V3 acc_avr = vec3(steps[i].acc[0], steps[i].acc[1], steps[i].acc[2]);
    const double dt = steps[i].dt;
    angvel_avr = angvel_avr - state_inout.bias_g;
    acc_avr = (acc_avr * cfg.G_m_s2) / cfg.mean_acc_norm - state_inout.bias_a;
    // covariance propagation
    const M3 Exp_f = ExpAngVel(angvel_avr, dt);
    const M3 acc_avr_skew = skew(acc_avr);
    MState F_x = MState::Identity(), cov_w = MState::Zero();
    const M3 Em = ExpAngVel(angvel_avr, -dt);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F_x(r, c) = Em(r, c);
    if (cfg.ba_bg_est_en) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F_x(r, 10 + c) = ((r == c) ? -1.0 : -0.0) * dt;      // -Eye3d * dt
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F_x(3 + r, 7 + c) = ((r == c) ? 1.0 : 0.0) * dt;
    const M3 nRa = ((R_imu * (-1.0)) * acc_avr_skew) * dt;                   // -R_imu * acc_avr_skew * dt
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F_x(7 + r, c) = nRa(r, c);
    if (cfg.ba_bg_est_en) { const M3 nR = (R_imu * (-1.0)) * dt; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F_x(7 + r, 13 + c) = nR(r, c); }
    if (cfg.gravity_est_en) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F_x(7 + r, 16 + c) = ((r == c) ? 1.0 : 0.0) * dt;
    if (cfg.exposure_estimate_en) cov_w(6, 6) = (cfg.cov_inv_expo * dt) * dt;
    for (int k = 0; k < 3; k++) cov_w(k, k) = (cfg.cov_gyr[k] * dt) * dt;
    {
      M3 D = M3::Zero(); for (int k = 0; k < 3; k++) D(k, k) = cfg.cov_acc[k];
      const M3 Q = (((R_imu * D) * R_imu.T()) * dt) * dt;                    // R_imu * cov_acc.asDiagonal() * R_imu^T * dt * dt
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cov_w(7 + r, 7 + c) = Q(r, c);
    }
    for (int k = 0; k < 3; k++) { cov_w(10 + k, 10 + k) = (cfg.cov_bias_gyr[k] * dt) * dt; cov_w(13 + k, 13 + k) = (cfg.cov_bias_acc[k] * dt) * dt; }
    state_inout.cov = (F_x * state_inout.cov) * F_x.T() + cov_w;
    // attitude, specific acceleration, position, velocity
    R_imu = R_imu * Exp_f;
    const V3 acc_imu = R_imu * acc_avr + state_inout.gravity;
    pos_imu = pos_imu + vel_imu * dt + ((acc_imu * 0.5) * dt) * dt;
    vel_imu = vel_imu + acc_imu * dt;
    Pose6D &p = poses[i];
    p.offset_time = steps[i].offs_t;
    for (int k = 0; k < 3; k++) { p.acc[k] = acc_imu[k]; p.gyr[k] = angvel_avr[k]; p.vel[k] = vel_imu[k]; p.pos[k] = pos_imu[k]; }
    std::memcpy(p.rot, R_imu.a, 72);
  }
  state_inout.vel_end = vel_imu; state_inout.rot_end = R_imu; state_inout.pos_end = pos_imu; state_inout.inv_expo_time = tau;
}

} // namespace orc
