// TEST INFRASTRUCTURE ONLY — the IMU part of oracle/_ref (see ref_api.cpp).  /root/reference/src/IMU_Processing.cpp is compiled textually unmodified by INCLUDING it
// (REF_IMU_CPP is set by the Makefile): its header defines a non-inline function (time_list, IMU_Processing.h:22), so it can be part of one translation unit only.
// Compiled with -fno-access-control: the members the real pipeline sets through Process2 / IMU_init (last_imu, last_prop_end_time, acc_s_last, angvel_last, mean_acc) are
// set directly.
#include REF_IMU_CPP
#include <cstring>
#include <sstream>
#include <fcntl.h>
#include <unistd.h>

namespace {
struct StatePOD { double rot[9], pos[3], inv_expo, vel[3], bg[3], ba[3], grav[3], cov[361]; };
template <class M> void rm_in(M &m, const double *a) { for (int i = 0; i < (int)m.rows(); i++) for (int j = 0; j < (int)m.cols(); j++) m(i, j) = a[i * m.cols() + j]; }
template <class M> void rm_out(const M &m, double *a) { for (int i = 0; i < (int)m.rows(); i++) for (int j = 0; j < (int)m.cols(); j++) a[i * m.cols() + j] = m(i, j); }
void from_pod(StatesGroup &s, const StatePOD &p) {
  rm_in(s.rot_end, p.rot); rm_in(s.pos_end, p.pos); s.inv_expo_time = p.inv_expo; rm_in(s.vel_end, p.vel); rm_in(s.bias_g, p.bg); rm_in(s.bias_a, p.ba);
  rm_in(s.gravity, p.grav); rm_in(s.cov, p.cov);
}
void to_pod(const StatesGroup &s, StatePOD &p) {
  rm_out(s.rot_end, p.rot); rm_out(s.pos_end, p.pos); p.inv_expo = s.inv_expo_time; rm_out(s.vel_end, p.vel); rm_out(s.bias_g, p.bg); rm_out(s.bias_a, p.ba);
  rm_out(s.gravity, p.grav); rm_out(s.cov, p.cov);
}
struct CoutCapture {
  std::ostringstream buf; std::streambuf *old; int saved_fd = -1;
  CoutCapture() : old(std::cout.rdbuf(buf.rdbuf())) { std::fflush(stdout); saved_fd = dup(1); int nul = open("/dev/null", O_WRONLY); if (nul >= 0) { dup2(nul, 1); close(nul); } }
  ~CoutCapture() { std::cout.rdbuf(old); std::fflush(stdout); if (saved_fd >= 0) { dup2(saved_fd, 1); close(saved_fd); } }
};
} // namespace

// ---- ImuProcess::UndistortPcl (IMU_Processing.cpp:237-541): forward propagation over the IMU samples + backward undistortion of the scan -------------------------
// Driven the way LIVMapper::handleFirstFrame / Process2 leave the object (this translation unit is compiled with -fno-access-control: the members the real
// pipeline sets through Process2 / IMU_init — last_imu, last_prop_end_time, acc_s_last, angvel_last, mean_acc — are set directly).
extern "C" {
struct ref_imu_cfg {
  double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3], cov_inv_expo, mean_acc_norm;
  int32_t ba_bg_est_en, gravity_est_en, exposure_estimate_en, first_call;
  double extR[9], extT[3];
  double last_prop_end_time, prop_end_time, acc_s_last[3], angvel_last[3];
};
// msgs: [n_msgs][7] = t, gyr3, acc3 with msgs[0] = last_imu; xyz/curvature: the scan (curvature = time offset in ms), undistorted in place; poses22: [n_msgs][22] out
int ref_imu_undistort(const ref_imu_cfg *c, const StatePOD *state_in, const double *msgs, int n_msgs, float *xyz, const float *curvature, int n_pts, StatePOD *state_out,
                      double *poses22, int *n_poses) {
  auto mk = [&](int i) { sensor_msgs::Imu::Ptr m(new sensor_msgs::Imu()); const double *p = msgs + (size_t)i * 7; m->header.stamp.t = p[0];
                         m->angular_velocity.x = p[1]; m->angular_velocity.y = p[2]; m->angular_velocity.z = p[3];
                         m->linear_acceleration.x = p[4]; m->linear_acceleration.y = p[5]; m->linear_acceleration.z = p[6]; return m; };
  auto setup = [&](ImuProcess &imu) {
  for (int k = 0; k < 3; k++) { imu.cov_gyr[k] = c->cov_gyr[k]; imu.cov_acc[k] = c->cov_acc[k]; imu.cov_bias_gyr[k] = c->cov_bias_gyr[k]; imu.cov_bias_acc[k] = c->cov_bias_acc[k]; }
  imu.cov_inv_expo = c->cov_inv_expo;
  if (!c->ba_bg_est_en) imu.disable_bias_est();
  if (!c->gravity_est_en) imu.disable_gravity_est();
  if (!c->exposure_estimate_en) imu.disable_exposure_est();
  imu.imu_time_init = !c->first_call;
  imu.mean_acc = V3D(0, 0, c->mean_acc_norm);
  { M3D R; V3D t; rm_in(R, c->extR); rm_in(t, c->extT); imu.set_extrinsic(t, R); }
  imu.last_prop_end_time = c->last_prop_end_time;
  for (int k = 0; k < 3; k++) { imu.acc_s_last[k] = c->acc_s_last[k]; imu.angvel_last[k] = c->angvel_last[k]; }
  imu.last_imu = mk(0);
  };
  ImuProcess imu; setup(imu);
  LidarMeasureGroup lm;
  lm.lio_vio_flg = LIO;
  MeasureGroup mg; mg.lio_time = c->prop_end_time; mg.vio_time = c->prop_end_time;
  for (int i = 1; i < n_msgs; i++) mg.imu.push_back(mk(i));
  lm.measures.push_back(mg);
  for (int i = 0; i < n_pts; i++) { PointType p; p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2]; p.curvature = curvature[i]; lm.pcl_proc_cur->points.push_back(p); }
  StatesGroup st; from_pod(st, *state_in);
  PointCloudXYZI out;
  // the poses are cleared at the end of UndistortPcl when the scan is not empty: take them from a run on a copy without points first
  {
    ImuProcess imu2; setup(imu2);
    LidarMeasureGroup lm2 = lm; lm2.pcl_proc_cur.reset(new PointCloudXYZI());
    StatesGroup st2 = st; PointCloudXYZI o2;
    CoutCapture cap;
    imu2.UndistortPcl(lm2, st2, o2);
    *n_poses = (int)imu2.IMUpose.size();
    for (int k = 0; k < *n_poses; k++) {
      const Pose6D &p = imu2.IMUpose[k]; double *o = poses22 + (size_t)k * 22;
      o[0] = p.offset_time;
      for (int j = 0; j < 3; j++) { o[1 + j] = p.acc[j]; o[4 + j] = p.gyr[j]; o[7 + j] = p.vel[j]; o[10 + j] = p.pos[j]; }
      for (int j = 0; j < 9; j++) o[13 + j] = p.rot[j];
    }
  }
  { CoutCapture cap; imu.UndistortPcl(lm, st, out); }
  to_pod(st, *state_out);
  if ((int)out.points.size() != n_pts && n_pts > 0) return -1;
  for (int i = 0; i < (int)out.points.size(); i++) { xyz[3 * i] = out.points[i].x; xyz[3 * i + 1] = out.points[i].y; xyz[3 * i + 2] = out.points[i].z; }
  return 0;
}
}
