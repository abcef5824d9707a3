// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// StatesGroup with boxplus / boxminus   (reference: include/common_lib.h:126-223, DIM_STATE = 19 at :30)
// Error-state layout: [0:3) dtheta, [3:6) dp, [6] inv_expo_time, [7:10) v, [10:13) b_g, [13:16) b_a, [16:19) g.
#pragma once
#include "orc_math.hpp"

namespace orc {

#define ORC_DIM_STATE 19
typedef Mat<ORC_DIM_STATE, ORC_DIM_STATE> MState;
typedef Mat<ORC_DIM_STATE, 1> VState;

// POD mirror used across the C boundary of the oracle library (386 doubles)
struct StatePOD {
  double rot[9];     // row-major rot_end
  double pos[3];
  double inv_expo;
  double vel[3], bg[3], ba[3], grav[3];
  double cov[ORC_DIM_STATE * ORC_DIM_STATE];   // row-major
};

struct StatesGroup {
  M3 rot_end; V3 pos_end, vel_end; double inv_expo_time; V3 bias_g, bias_a, gravity; MState cov;
  StatesGroup() {                      // common_lib.h:128-140
    rot_end = M3::Identity(); pos_end = V3::Zero(); vel_end = V3::Zero(); bias_g = V3::Zero(); bias_a = V3::Zero(); gravity = V3::Zero();
    inv_expo_time = 1.0;
    cov = MState::Identity() * 0.01;
    cov(6, 6) = 0.00001;
    for (int i = 10; i < 19; i++) cov(i, i) = 0.00001;
  }
  // operator+= (common_lib.h:182-192)
  StatesGroup &operator+=(const VState &d) {
    rot_end = rot_end * Exp(d[0], d[1], d[2]);
    for (int i = 0; i < 3; i++) pos_end[i] += d[3 + i];
    inv_expo_time += d[6];
    for (int i = 0; i < 3; i++) { vel_end[i] += d[7 + i]; bias_g[i] += d[10 + i]; bias_a[i] += d[13 + i]; gravity[i] += d[16 + i]; }
    return *this;
  }
  // operator- (common_lib.h:194-206):  this [-] b
  VState operator-(const StatesGroup &b) const {
    VState a;
    M3 rotd = b.rot_end.T() * rot_end;
    V3 l = Log(rotd);
    for (int i = 0; i < 3; i++) {
      a[i] = l[i]; a[3 + i] = pos_end[i] - b.pos_end[i]; a[7 + i] = vel_end[i] - b.vel_end[i];
      a[10 + i] = bias_g[i] - b.bias_g[i]; a[13 + i] = bias_a[i] - b.bias_a[i]; a[16 + i] = gravity[i] - b.gravity[i];
    }
    a[6] = inv_expo_time - b.inv_expo_time;
    return a;
  }
  void from_pod(const StatePOD &p) {
    std::memcpy(rot_end.a, p.rot, sizeof(p.rot)); std::memcpy(pos_end.a, p.pos, sizeof(p.pos)); inv_expo_time = p.inv_expo;
    std::memcpy(vel_end.a, p.vel, 24); std::memcpy(bias_g.a, p.bg, 24); std::memcpy(bias_a.a, p.ba, 24); std::memcpy(gravity.a, p.grav, 24);
    std::memcpy(cov.a, p.cov, sizeof(p.cov));
  }
  void to_pod(StatePOD &p) const {
    std::memcpy(p.rot, rot_end.a, sizeof(p.rot)); std::memcpy(p.pos, pos_end.a, sizeof(p.pos)); p.inv_expo = inv_expo_time;
    std::memcpy(p.vel, vel_end.a, 24); std::memcpy(p.bg, bias_g.a, 24); std::memcpy(p.ba, bias_a.a, 24); std::memcpy(p.grav, gravity.a, 24);
    std::memcpy(p.cov, cov.a, sizeof(p.cov));
  }
};

} // namespace orc
