// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the reference-patch choice in the middle of VIOManager::retrieveFromVisualSparseMap: for every
// grid cell whose selected visual point passed the depth-continuity test, the reference picks the observation (Feature) whose patch is
// warped into the current frame:
//   is_normal_initialized_ gate, normal_en branch (mutual photometric error of the stored patches)     src/vio.cpp:644-691
//   !normal_en branch: VisualPoint::getCloseViewObs (closest viewing direction, 60 degree gate)         src/vio.cpp:692-695, src/visual_point.cpp:57-95
//   Feature::pos() = T_f_w_.inverse().translation()                                                    include/feature.h:46
// Quirks restated as they are:
//   * observations are compared by id_, and id_ is the id of the FRAME a feature was created in (vio.cpp:882, 961), so two observations of one
//     point made in the same frame never count against each other (vio.cpp:673);
//   * a point whose observations all carry one id_ divides 0 by 0: the NaN score never beats FLT_MAX and the reference goes on with an
//     uninitialised ref_ftr (undefined behaviour).  Here such a point is SKIPPED (chosen = -1) and keeps has_ref_patch_ = false;
//   * getCloseViewObs starts from min_cos_angle = 0, keeps the FIRST observation of the best cosine (strict >), and rejects below 0.5.
// Third-party arithmetic not under /root/reference (Eigen normalize()/dot(), Sophus SE3 inverse — unpinned): v / sqrt((x*x + y*y) + z*z),
// (a0*b0 + a1*b1) + a2*b2, -(R^T t).  PARITY UNPINNED at that boundary.
#pragma once
#include "orc_math.hpp"
#include <limits>

namespace orc {

// One observation as the choice reads it.  patch = Feature::patch_ (patch_size_total floats, level 0).
struct ObsRef { int id; M3 R; V3 t; const float *patch; };

// src/vio.cpp:653-691 (normal_en).  obs[0..n): pt->obs_ in list order; has_ref_patch / ref_patch (index into obs) are read and updated like
// pt->has_ref_patch_ / pt->ref_patch.  Returns the index of ref_ftr, -1 if the point is skipped.
inline int choose_ref_by_patches(const ObsRef *obs, int n, int patch_size_total, int &has_ref_patch, int &ref_patch) {
  if (n < 1) return -1;
  int ref_ftr = -1;
  float phtometric_errors_min = std::numeric_limits<float>::max();
  if (n == 1) {
    ref_ftr = 0;
    ref_patch = ref_ftr; has_ref_patch = 1;
  } else if (!has_ref_patch) {
    for (int it = 0; it < n; it++) {
      const float *patch_temp = obs[it].patch;
      float phtometric_errors = 0.0;
      int count = 0;
      for (int itm = 0; itm < n; itm++) {
        if (obs[itm].id == obs[it].id) continue;
        const float *patch_cache = obs[itm].patch;
        for (int ind = 0; ind < patch_size_total; ind++) phtometric_errors += (patch_temp[ind] - patch_cache[ind]) * (patch_temp[ind] - patch_cache[ind]);
        count++;
      }
      phtometric_errors = phtometric_errors / count;
      if (phtometric_errors < phtometric_errors_min) { phtometric_errors_min = phtometric_errors; ref_ftr = it; }
    }
    if (ref_ftr < 0) return -1;                     // see the header: undefined in the reference
    ref_patch = ref_ftr; has_ref_patch = 1;
  } else ref_ftr = ref_patch;
  return ref_ftr;
}

// src/visual_point.cpp:57-95.  Returns the index of the chosen observation, -1 where the reference returns false.
inline int getCloseViewObs(const V3 &framepos, const V3 &pos, const ObsRef *obs, int n) {
  if (n <= 0) return -1;
  V3 obs_dir = framepos - pos;
  obs_dir = obs_dir / norm(obs_dir);
  int min_it = 0;
  double min_cos_angle = 0;
  for (int it = 0; it < n; it++) {
    V3 dir = (obs[it].R.T() * obs[it].t) * (-1.0) - pos;          // T_f_w_.inverse().translation() - pos_
    dir = dir / norm(dir);
    const double cos_angle = dot(obs_dir, dir);
    if (cos_angle > min_cos_angle) { min_cos_angle = cos_angle; min_it = it; }
  }
  if (min_cos_angle < 0.5) return -1;
  return min_it;
}

} // namespace orc
