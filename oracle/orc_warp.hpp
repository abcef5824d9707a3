// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the data-parallel tail of VIOManager::retrieveFromVisualSparseMap: what the reference does
// for every visual map point that survived the host-side grid / depth-continuity selection (src/vio.cpp:598-672):
//   getWarpMatrixAffineHomography   src/vio.cpp:247-272      (normal_en)
//   getWarpMatrixAffine             src/vio.cpp:274-290      (!normal_en)
//   getBestSearchLevel              src/vio.cpp:320-331
//   warpAffine                      src/vio.cpp:292-318      (patch_pyrimid_level levels of 8x8 bilinear samples of the reference image)
//   getImagePatch                   src/vio.cpp:203-225      (current image, level 0)
//   photometric error, NCC, gates   src/vio.cpp:742-760, calculateNCC 333-350
//   appended outputs                src/vio.cpp:762-767      (visual_submap->{voxel_points, search_levels, errors, warp_patch, inv_expo_list})
//   warp_map                        src/vio.cpp:369, 716-734 (!normal_en only: the warp is cached under ref_ftr->id_, and id_ is the id of the FRAME the
//                                   feature was created in (vio.cpp:882, 961) — every later point of this call whose ref_ftr comes from the same frame
//                                   reuses the FIRST such point's A_cur_ref and search_level instead of computing its own; restated in orc_api.cpp)
//
// Third-party arithmetic NOT under /root/reference (rpg_vikit, xuankuzcr fork, unpinned — README.md:76-84), restated from its
// published sources, PARITY UNPINNED:
//   vk::PinholeCamera::cam2world(px)  = normalize((u-cx)/fx, (v-cy)/fy, 1); with distortion: OpenCV's undistortPoints iteration first (see cam2world below)
//   vk::interpolateMat_8u(mat, u, v)  = float bilinear: w00=(1-sx)(1-sy), w01=(1-sx)sy, w10=sx(1-sy), w11=1-w00-w01-w10;
//                                       w00*p[0] + w01*p[stride] + w10*p[1] + w11*p[stride+1]
//   Sophus SE3 composition / inverse  = rigid-transform algebra
#pragma once
#include "orc_visual.hpp"

namespace orc {

struct WarpCfg {
  PinholeCam cam;
  M3 R_cur; V3 t_cur;               // new_frame_->T_f_w_  (p_f = R p_w + t)
  double inv_expo_cur;              // state->inv_expo_time
  int patch_pyrimid_level, normal_en, ncc_en;
  double ncc_thre, outlier_threshold;
};

struct WarpCand {                   // what the loop body reads of `pt` and `ref_ftr`
  V3 pos, normal;                   // pt->pos_, pt->normal_
  const uint8_t *img_ref;           // ref_ftr->img_ (same size as the current image)
  double px_ref[2];                 // ref_ftr->px_
  V3 f_ref;                         // ref_ftr->f_
  M3 R_ref; V3 t_ref;               // ref_ftr->T_f_w_
  int level_ref;                    // ref_ftr->level_
  double inv_expo_ref;              // ref_ftr->inv_expo_time_
};

// vk::PinholeCamera::cam2world.  Without distortion: normalize((u-cx)/fx, (v-cy)/fy, 1).  With distortion rpg_vikit calls
//   cv::undistortPoints(CV_32FC2 point, CV_32FC2 result, cvK_, cvD_)          (OpenCV >= 4.2, unpinned: README.md:57-61)
// whose published algorithm (cvUndistortPointsInternal, default TermCriteria(MAX_ITER, 5, 0.01): exactly five fixed-point iterations, no R / P, no tilt,
// k = (k1, k2, p1, p2, k3) = d[0..4]) is restated here: the pixel enters as float32, the iteration runs in double, the undistorted normalised point leaves
// as float32, and vikit normalises (x, y, 1) in double.  PARITY UNPINNED (third-party, from its published source).
inline V3 cam2world(const PinholeCam &c, double u, double v) {
  if (!c.distortion) {
    V3 xyz = vec3((u - c.cx) / c.fx, (v - c.cy) / c.fy, 1.0);
    return xyz / norm(xyz);
  }
  if (c.distortion == 2) {            // vk::EquidistantCamera: OpenCV-fisheye undistortPoints scheme (ten fixed-point iterations on theta); third-party, unpinned
    const double xd = (u - c.cx) / c.fx, yd = (v - c.cy) / c.fy;
    const double thetad = std::sqrt(xd * xd + yd * yd);
    double theta = thetad;
    for (int j = 0; j < 10; j++) {
      const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      theta = thetad / (1 + c.d[0] * t2 + c.d[1] * t4 + c.d[2] * t6 + c.d[3] * t8);
    }
    const double scaling = (thetad > 1e-8) ? std::tan(theta) / thetad : 1.0;
    V3 xyz = vec3(xd * scaling, yd * scaling, 1.0);
    return xyz / norm(xyz);
  }
  const double uf = (double)(float)u, vf = (double)(float)v;           // cv::Point2f uv(u, v)
  const double ifx = 1.0 / c.fx, ify = 1.0 / c.fy;
  const double x0 = (uf - c.cx) * ifx, y0 = (vf - c.cy) * ify;
  double x = x0, y = y0;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = 1.0 / (1.0 + ((c.d[4] * r2 + c.d[1]) * r2 + c.d[0]) * r2);
    if (icdist < 0) { x = x0; y = y0; break; }
    const double deltaX = 2.0 * c.d[2] * x * y + c.d[3] * (r2 + 2.0 * x * x);
    const double deltaY = c.d[2] * (r2 + 2.0 * y * y) + 2.0 * c.d[3] * x * y;
    x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
  }
  V3 xyz = vec3((double)(float)x, (double)(float)y, 1.0);               // dst point is CV_32FC2
  return xyz / norm(xyz);
}

inline float interpolateMat_8u(const uint8_t *img, int stride, float u, float v) {
  const int x = (int)std::floor(u), y = (int)std::floor(v);
  const float subpix_x = u - x, subpix_y = v - y;
  const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  const float w01 = (1.0f - subpix_x) * subpix_y;
  const float w10 = subpix_x * (1.0f - subpix_y);
  const float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t *ptr = img + y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

// src/vio.cpp:247-272
inline void getWarpMatrixAffineHomography(const PinholeCam &cam, const double px_ref[2], const V3 &xyz_ref, const V3 &normal_ref, const M3 &R_cur_ref,
                                          const V3 &t_cur_ref, int level_ref, double A[4]) {
  const V3 t = (R_cur_ref.T() * t_cur_ref) * (-1.0);                       // T_cur_ref.inverse().translation()
  const M3 H = R_cur_ref * (M3::Identity() * dot(normal_ref, xyz_ref) - t * normal_ref.T());
  const int kHalfPatchSize = 4;
  const V3 f_du_ref = cam2world(cam, px_ref[0] + (double)(kHalfPatchSize * (1 << level_ref)), px_ref[1]);
  const V3 f_dv_ref = cam2world(cam, px_ref[0], px_ref[1] + (double)(kHalfPatchSize * (1 << level_ref)));
  const V3 f_cur = H * xyz_ref, f_du_cur = H * f_du_ref, f_dv_cur = H * f_dv_ref;
  double px_cur[2], px_du[2], px_dv[2];
  cam.world2cam(f_cur, px_cur); cam.world2cam(f_du_cur, px_du); cam.world2cam(f_dv_cur, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / kHalfPatchSize; A[2] = (px_du[1] - px_cur[1]) / kHalfPatchSize;     // col 0
  A[1] = (px_dv[0] - px_cur[0]) / kHalfPatchSize; A[3] = (px_dv[1] - px_cur[1]) / kHalfPatchSize;     // col 1
}

// src/vio.cpp:274-290
inline void getWarpMatrixAffine(const PinholeCam &cam, const double px_ref[2], const V3 &f_ref, double depth_ref, const M3 &R_cur_ref, const V3 &t_cur_ref,
                                int level_ref, int pyramid_level, int halfpatch_size, double A[4]) {
  const V3 xyz_ref = f_ref * depth_ref;
  V3 xyz_du_ref = cam2world(cam, px_ref[0] + (double)(halfpatch_size * (1 << level_ref) * (1 << pyramid_level)), px_ref[1]);
  V3 xyz_dv_ref = cam2world(cam, px_ref[0], px_ref[1] + (double)(halfpatch_size * (1 << level_ref) * (1 << pyramid_level)));
  xyz_du_ref = xyz_du_ref * (xyz_ref[2] / xyz_du_ref[2]);
  xyz_dv_ref = xyz_dv_ref * (xyz_ref[2] / xyz_dv_ref[2]);
  double px_cur[2], px_du[2], px_dv[2];
  cam.world2cam(R_cur_ref * xyz_ref + t_cur_ref, px_cur);
  cam.world2cam(R_cur_ref * xyz_du_ref + t_cur_ref, px_du);
  cam.world2cam(R_cur_ref * xyz_dv_ref + t_cur_ref, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size; A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size; A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;
}

// src/vio.cpp:320-331
inline int getBestSearchLevel(const double A[4], int max_level) {
  int search_level = 0;
  double D = A[0] * A[3] - A[1] * A[2];
  while (D > 3.0 && search_level < max_level) { search_level += 1; D *= 0.25; }
  return search_level;
}

// src/vio.cpp:292-318 ; returns false when the warp is NaN (the reference prints a warning and leaves the patch untouched)
inline bool warpAffine(const double A_cur_ref[4], const uint8_t *img_ref, int width, int height, const double px_ref[2], int search_level, int pyramid_level,
                       int halfpatch_size, float *patch) {
  const int patch_size = halfpatch_size * 2, patch_size_total = patch_size * patch_size;
  // Eigen 2x2 inverse: adjugate times 1/det, then cast<float>
  const double invdet = 1.0 / (A_cur_ref[0] * A_cur_ref[3] - A_cur_ref[1] * A_cur_ref[2]);
  const float a00 = (float)(A_cur_ref[3] * invdet), a01 = (float)(-A_cur_ref[1] * invdet), a10 = (float)(-A_cur_ref[2] * invdet), a11 = (float)(A_cur_ref[0] * invdet);
  if (std::isnan(a00)) return false;
  const float pxr0 = (float)px_ref[0], pxr1 = (float)px_ref[1];
  for (int y = 0; y < patch_size; ++y)
    for (int x = 0; x < patch_size; ++x) {
      float p0 = (float)(x - halfpatch_size), p1 = (float)(y - halfpatch_size);
      p0 *= (float)(1 << search_level); p1 *= (float)(1 << search_level);
      p0 *= (float)(1 << pyramid_level); p1 *= (float)(1 << pyramid_level);
      const float px0 = (a00 * p0 + a01 * p1) + pxr0, px1 = (a10 * p0 + a11 * p1) + pxr1;
      float &dst = patch[patch_size_total * pyramid_level + y * patch_size + x];
      if (px0 < 0 || px1 < 0 || px0 >= width - 1 || px1 >= height - 1) dst = 0;
      else dst = interpolateMat_8u(img_ref, width, px0, px1);
    }
  return true;
}

// src/vio.cpp:203-225 (level 0 here, like the call at vio.cpp:740)
inline void getImagePatch(const uint8_t *img, int width, const double pc[2], float *patch_tmp, int level, int patch_size) {
  const int patch_size_half = patch_size / 2, patch_size_total = patch_size * patch_size;
  const float u_ref = (float)pc[0], v_ref = (float)pc[1];
  const int scale = (1 << level);
  const int u_ref_i = (int)(floorf((float)(pc[0] / scale)) * scale), v_ref_i = (int)(floorf((float)(pc[1] / scale)) * scale);
  const float subpix_u_ref = (u_ref - u_ref_i) / scale, subpix_v_ref = (v_ref - v_ref_i) / scale;
  const float w_ref_tl = (float)((1.0 - subpix_u_ref) * (1.0 - subpix_v_ref));
  const float w_ref_tr = (float)(subpix_u_ref * (1.0 - subpix_v_ref));
  const float w_ref_bl = (float)((1.0 - subpix_u_ref) * subpix_v_ref);
  const float w_ref_br = subpix_u_ref * subpix_v_ref;
  for (int x = 0; x < patch_size; x++) {
    const uint8_t *img_ptr = img + (v_ref_i - patch_size_half * scale + x * scale) * width + (u_ref_i - patch_size_half * scale);
    for (int y = 0; y < patch_size; y++, img_ptr += scale)
      patch_tmp[patch_size_total * level + x * patch_size + y] =
          w_ref_tl * img_ptr[0] + w_ref_tr * img_ptr[scale] + w_ref_bl * img_ptr[scale * width] + w_ref_br * img_ptr[scale * width + scale];
  }
}

// src/vio.cpp:333-350
inline double calculateNCC(const float *ref_patch, const float *cur_patch, int patch_size) {
  double sum_ref = 0.0; for (int i = 0; i < patch_size; i++) sum_ref += ref_patch[i];
  const double mean_ref = sum_ref / patch_size;
  double sum_cur = 0.0; for (int i = 0; i < patch_size; i++) sum_cur += cur_patch[i];
  const double mean_curr = sum_cur / patch_size;
  double numerator = 0, demoniator1 = 0, demoniator2 = 0;
  for (int i = 0; i < patch_size; i++) {
    const double n = (ref_patch[i] - mean_ref) * (cur_patch[i] - mean_curr);
    numerator += n;
    demoniator1 += (ref_patch[i] - mean_ref) * (ref_patch[i] - mean_ref);
    demoniator2 += (cur_patch[i] - mean_curr) * (cur_patch[i] - mean_curr);
  }
  return numerator / std::sqrt(demoniator1 * demoniator2 + 1e-10);
}

struct WarpOut { int accepted, search_level; float error; double ncc; double A[4]; };

// first half of the loop body, src/vio.cpp:698-735: the affine warp reference -> current and its search level
inline void warp_matrix(const WarpCfg &cfg, const WarpCand &c, WarpOut &o) {
  const int patch_size_half = 4;
  const M3 R_cur_ref = cfg.R_cur * c.R_ref.T();                               // new_frame_->T_f_w_ * ref_ftr->T_f_w_.inverse()
  const V3 t_cur_ref = cfg.t_cur - R_cur_ref * c.t_ref;
  if (cfg.normal_en) {
    V3 nv = c.R_ref * c.normal; nv = nv / norm(nv);                           // vio.cpp:701
    const V3 pf = c.R_ref * c.pos + c.t_ref;                                  // vio.cpp:703
    getWarpMatrixAffineHomography(cfg.cam, c.px_ref, pf, nv, R_cur_ref, t_cur_ref, 0, o.A);
  } else {
    const V3 ref_pos = (c.R_ref.T() * c.t_ref) * (-1.0);                      // Feature::pos()
    getWarpMatrixAffine(cfg.cam, c.px_ref, c.f_ref, norm(ref_pos - c.pos), R_cur_ref, t_cur_ref, c.level_ref, 0, patch_size_half, o.A);
  }
  o.search_level = getBestSearchLevel(o.A, 2);
}

// second half, src/vio.cpp:738-767, with o.A / o.search_level given; patch_wrap: [L*64]
inline void warp_finish(const WarpCfg &cfg, const uint8_t *img, const WarpCand &c, float *patch_wrap, WarpOut &o) {
  const int patch_size = 8, patch_size_half = 4, patch_size_total = 64;
  double pc[2];
  cfg.cam.world2cam(cfg.R_cur * c.pos + cfg.t_cur, pc);                       // new_frame_->w2c(pt->pos_), vio.cpp:606
  for (int k = 0; k < patch_size_total * cfg.patch_pyrimid_level; k++) patch_wrap[k] = 0.f;   // std::vector<float> patch_wrap(warp_len)
  for (int pyramid_level = 0; pyramid_level <= cfg.patch_pyrimid_level - 1; pyramid_level++)
    warpAffine(o.A, c.img_ref, cfg.cam.width, cfg.cam.height, c.px_ref, o.search_level, pyramid_level, patch_size_half, patch_wrap);
  float patch_buffer[64];
  getImagePatch(img, cfg.cam.width, pc, patch_buffer, 0, patch_size);
  float error = 0.0;
  for (int ind = 0; ind < patch_size_total; ind++)
    error += (c.inv_expo_ref * patch_wrap[ind] - cfg.inv_expo_cur * patch_buffer[ind]) * (c.inv_expo_ref * patch_wrap[ind] - cfg.inv_expo_cur * patch_buffer[ind]);
  o.error = error;
  o.ncc = calculateNCC(patch_wrap, patch_buffer, patch_size_total);
  o.accepted = 1;
  if (cfg.ncc_en && o.ncc < cfg.ncc_thre) o.accepted = 0;
  if (error > cfg.outlier_threshold * patch_size_total) o.accepted = 0;
}

// body of the per-point loop, src/vio.cpp:698-767, for a ref_ftr that meets no warp_map entry
inline void warp_candidate(const WarpCfg &cfg, const uint8_t *img, const WarpCand &c, float *patch_wrap, WarpOut &o) {
  warp_matrix(cfg, c, o);
  warp_finish(cfg, img, c, patch_wrap, o);
}

} // namespace orc
