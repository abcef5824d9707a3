// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the visual (direct photometric, 8x8 patch) ESIKF measurement update of FAST-LIVO2:
//   initializeVIO extrinsic constants        src/vio.cpp:27-38, 57-65
//   computeProjectionJacobian                src/vio.cpp:189-201
//   computeJacobianAndUpdateEKF              src/vio.cpp:784-802
//   updateState (forward compositional)      src/vio.cpp:1520-1688
//   precomputeReferencePatches               src/vio.cpp:1327-1396
//   updateStateInverse (inverse comp.)       src/vio.cpp:1398-1518
//   updateFrameState                         src/vio.cpp:1690-1697
// Float/double mix restated exactly (Q8/Q9 of SURVEY.md §8a): bilinear weights through double then narrowed to
// float, float pixel sums, float patch_error / error accumulators, P index uses `level`, sampling stride uses
// `level + search_level`.
//
// Third-party arithmetic: `cam->world2cam(pf)` (src/vio.cpp:1447,1574) is rpg_vikit (xuankuzcr fork, no version
// pinned, README.md:76-84) and is NOT under /root/reference.  Its published pinhole model is restated in
// `world2cam` below (projection to the z=1 plane, optional radial-tangential distortion d0..d3(+d4), then fx,fy,cx,cy).
// PARITY UNPINNED at this boundary; the benchmarks use zero distortion so it reduces to (fx*x/z+cx, fy*y/z+cy),
// consistent with computeProjectionJacobian.
#pragma once
#include "orc_state.hpp"
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

struct PinholeCam {                  // vk::PinholeCamera (values already multiplied by `scale`, vio.cpp:45-54)
  double fx, fy, cx, cy;
  double d[5];
  int distortion;                    // 0 pinhole, 1 radial-tangential, 2 equidistant
  int width, height;
  void world2cam(const V3 &xyz_c, double px[2]) const {
    double uv0 = xyz_c[0] / xyz_c[2], uv1 = xyz_c[1] / xyz_c[2];       // vk::project2d
    if (!distortion) { px[0] = fx * uv0 + cx; px[1] = fy * uv1 + cy; }
    else if (distortion == 2) {       // vk::EquidistantCamera (config/camera_fisheye_HILTI22.yaml): k1..k4 in d[0..3]; rpg_vikit's published Kannala-Brandt form
      const double r = std::sqrt(uv0 * uv0 + uv1 * uv1), theta = std::atan(r), t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      const double thetad = theta * (1 + d[0] * t2 + d[1] * t4 + d[2] * t6 + d[3] * t8);
      const double scaling = (r > 1e-8) ? thetad / r : 1.0;
      px[0] = fx * uv0 * scaling + cx; px[1] = fy * uv1 * scaling + cy;
    } else {
      double x = uv0, y = uv1, r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
      double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
      double cdist = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6;
      double xd = x * cdist + d[2] * a1 + d[3] * a2;
      double yd = y * cdist + d[2] * a3 + d[3] * a1;
      px[0] = xd * fx + cx; px[1] = yd * fy + cy;
    }
  }
};

struct VisualPoint { V3 pos_; };     // include/visual_point.h:23-46 (hot path reads pos_ only; inverse variant reads ref_patch, below)

struct RefPatch {                    // include/feature.h:19-54 subset read by precomputeReferencePatches
  const uint8_t *img_;               // reference image (same width as current)
  double px_[2];
  V3 f_;
  M3 R_ref_w;                        // T_f_w_.rotation_matrix()
  V3 pos_ref;                        // Feature::pos() = camera centre of the reference frame in world
};

struct SubSparseMap {                // include/vio.h:26-57
  std::vector<float> errors;
  std::vector<std::vector<float>> warp_patch;
  std::vector<int> search_levels;
  std::vector<VisualPoint *> voxel_points;
  std::vector<double> inv_expo_list;
  std::vector<RefPatch> ref_patches; // oracle-only carrier for the inverse-compositional variant
};

struct VisualIterTrace {
  int level, iteration, accepted, n_meas;
  float error;
  double HtH[49], Htz[7], solution[19];
};

class VIOManager {
public:
  PinholeCam cam;
  StatesGroup *state = nullptr, *state_propagat = nullptr;
  M3 Rli, Rci, Rcl, Rcw, Jdphi_dR, Jdp_dt, Jdp_dR;
  V3 Pli, Pci, Pcl, Pcw;
  bool inverse_composition_en = false, exposure_estimate_en = true, has_ref_patch_cache = false;
  int width = 0, height = 0;
  double fx, fy, cx, cy;
  int patch_pyrimid_level = 4, patch_size = 8, patch_size_total = 64, patch_size_half = 4;
  int max_iterations = 5, total_points = 0;
  double img_point_cov = 100;
  SubSparseMap *visual_submap = nullptr;
  MState G, H_T_H;
  std::vector<double> H_sub_inv;     // H_DIM x 6
  int num_threads_ = 1;
  // oracle-only dumps
  std::vector<double> dump_z_, dump_H_;     // last evaluated iteration: z (H_DIM), H_sub (H_DIM x 7 or x 6)
  std::vector<VisualIterTrace> trace_;

  void setImuToLidarExtrinsic(const V3 &transl, const M3 &rot) { Pli = -(rot.T()) * transl; Rli = rot.T(); }      // vio.cpp:27-31
  void setLidarToCameraExtrinsic(const M3 &R, const V3 &P) { Rcl = R; Pcl = P; }                                   // vio.cpp:33-37
  void initializeVIO() {                                                                                              // vio.cpp:41-65, 149-151
    fx = cam.fx; fy = cam.fy; cx = cam.cx; cy = cam.cy; width = cam.width; height = cam.height;
    Rci = Rcl * Rli;
    Pci = Rcl * Pli + Pcl;
    Jdphi_dR = Rci;
    V3 Pic = -(Rci.T()) * Pci;
    M3 tmp = skew(Pic);
    Jdp_dR = -Rci * tmp;
    patch_size_total = patch_size * patch_size;
    patch_size_half = patch_size / 2;
    G = MState::Zero(); H_T_H = MState::Zero();
  }

  void computeProjectionJacobian(const V3 &p, Mat<2, 3> &J) {          // vio.cpp:189-201
    const double x = p[0], y = p[1];
    const double z_inv = 1. / p[2];
    const double z_inv_2 = z_inv * z_inv;
    J(0, 0) = fx * z_inv; J(0, 1) = 0.0; J(0, 2) = -fx * x * z_inv_2;
    J(1, 0) = 0.0; J(1, 1) = fy * z_inv; J(1, 2) = -fy * y * z_inv_2;
  }

  // Evaluates residuals / Jacobian rows / errors for the CURRENT *state at `level` (vio.cpp:1538-1636); returns mean error.
  float eval_forward(const uint8_t *img, int level, std::vector<double> &z, std::vector<double> &H_sub /*H_DIM x 7*/, int &n_meas_out) {
    M3 Rwi = state->rot_end; V3 Pwi = state->pos_end;
    Rcw = Rci * Rwi.T();
    Pcw = -Rci * Rwi.T() * Pwi + Pci;
    Jdp_dt = Rci * Rwi.T();
    int n_meas = 0;
    // `#pragma omp parallel for reduction(+:error, n_meas)` with MP_PROC_NUM threads (vio.cpp:1551-1555): libgomp's default static schedule gives thread t
    // one contiguous block of patches, every thread adds its patches' float errors in index order into a private 0-initialised float, and the private sums
    // are added to `error` in the order the threads finish — the one part the reference leaves to chance (Q9).  The oracle pins that order to thread 0, 1, 2, ...
    // (reduce_error_static below), so the result does not depend on how many cores run the loop; num_threads_ = 1 is the serial loop.
#ifdef _OPENMP
    omp_set_num_threads(num_threads_);
#pragma omp parallel for reduction(+ : n_meas)
#endif
    for (int i = 0; i < total_points; i++) {
      Mat<1, 2> Jimg; Mat<2, 3> Jdpi; Mat<1, 3> Jdphi, Jdp, JdR, Jdt;
      float patch_error = 0.0;
      int search_level = visual_submap->search_levels[i];
      int pyramid_level = level + search_level;
      int scale = (1 << pyramid_level);
      float inv_scale = 1.0f / scale;
      VisualPoint *pt = visual_submap->voxel_points[i];
      if (pt == nullptr) continue;
      V3 pf = Rcw * pt->pos_ + Pcw;
      double pc[2]; cam.world2cam(pf, pc);
      computeProjectionJacobian(pf, Jdpi);
      M3 p_hat = skew(pf);
      float u_ref = pc[0];
      float v_ref = pc[1];
      int u_ref_i = floorf(pc[0] / scale) * scale;
      int v_ref_i = floorf(pc[1] / scale) * scale;
      float subpix_u_ref = (u_ref - u_ref_i) / scale;
      float subpix_v_ref = (v_ref - v_ref_i) / scale;
      float w_ref_tl = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
      float w_ref_tr = subpix_u_ref * (1.0 - subpix_v_ref);
      float w_ref_bl = (1.0 - subpix_u_ref) * subpix_v_ref;
      float w_ref_br = subpix_u_ref * subpix_v_ref;
      std::vector<float> P = visual_submap->warp_patch[i];            // deep copy every iteration, as the reference (vio.cpp:1591)
      double inv_ref_expo = visual_submap->inv_expo_list[i];
      for (int x = 0; x < patch_size; x++) {
        const uint8_t *img_ptr = img + (v_ref_i + x * scale - patch_size_half * scale) * width + u_ref_i - patch_size_half * scale;
        for (int y = 0; y < patch_size; ++y, img_ptr += scale) {
          float du = 0.5f * ((w_ref_tl * img_ptr[scale] + w_ref_tr * img_ptr[scale * 2] + w_ref_bl * img_ptr[scale * width + scale] +
                              w_ref_br * img_ptr[scale * width + scale * 2]) -
                             (w_ref_tl * img_ptr[-scale] + w_ref_tr * img_ptr[0] + w_ref_bl * img_ptr[scale * width - scale] + w_ref_br * img_ptr[scale * width]));
          float dv = 0.5f * ((w_ref_tl * img_ptr[scale * width] + w_ref_tr * img_ptr[scale + scale * width] + w_ref_bl * img_ptr[width * scale * 2] +
                              w_ref_br * img_ptr[width * scale * 2 + scale]) -
                             (w_ref_tl * img_ptr[-scale * width] + w_ref_tr * img_ptr[-scale * width + scale] + w_ref_bl * img_ptr[0] + w_ref_br * img_ptr[scale]));
          Jimg(0, 0) = du; Jimg(0, 1) = dv;
          Jimg = Jimg * state->inv_expo_time;
          Jimg = Jimg * (double)inv_scale;
          Jdphi = (Jimg * Jdpi) * p_hat;
          Jdp = (-Jimg) * Jdpi;
          JdR = Jdphi * Jdphi_dR + Jdp * Jdp_dR;
          Jdt = Jdp * Jdp_dt;
          double cur_value = w_ref_tl * img_ptr[0] + w_ref_tr * img_ptr[scale] + w_ref_bl * img_ptr[scale * width] + w_ref_br * img_ptr[scale * width + scale];
          double res = state->inv_expo_time * cur_value - inv_ref_expo * P[patch_size_total * level + x * patch_size + y];
          size_t row = (size_t)i * patch_size_total + x * patch_size + y;
          z[row] = res;
          patch_error += res * res;
          n_meas += 1;
          double *h = &H_sub[row * 7];
          h[0] = JdR(0, 0); h[1] = JdR(0, 1); h[2] = JdR(0, 2); h[3] = Jdt(0, 0); h[4] = Jdt(0, 1); h[5] = Jdt(0, 2);
          if (exposure_estimate_en) h[6] = cur_value;
        }
      }
      visual_submap->errors[i] = patch_error;
    }
    float error = reduce_error_static(visual_submap->errors.data(), total_points, num_threads_);
    error = error / n_meas;
    n_meas_out = n_meas;
    return error;
  }

  // float sum of the per-patch errors in the order of an OpenMP static partition over `threads` threads, partial sums joined in thread order
  static float reduce_error_static(const float *e, int n, int threads) {
    const int T = threads > 0 ? threads : 1;
    const int q = n / T, r = n % T;
    float error = 0.0f;
    for (int t = 0; t < T; t++) {
      const int begin = t < r ? t * (q + 1) : t * q + r, cnt = t < r ? q + 1 : q;
      float priv = 0.0f;
      for (int i = begin; i < begin + cnt; i++) priv += e[i];
      error += priv;
    }
    return error;
  }

  // src/vio.cpp:1520-1688
  void updateState(const uint8_t *img, int level) {
    if (total_points == 0) return;
    StatesGroup old_state = (*state);
    bool EKF_end = false;
    float last_error = std::numeric_limits<float>::max();
    const int H_DIM = total_points * patch_size_total;
    std::vector<double> z(H_DIM, 0.0), H_sub((size_t)H_DIM * 7, 0.0);
    for (int iteration = 0; iteration < max_iterations; iteration++) {
      VisualIterTrace tr; std::memset(&tr, 0, sizeof(tr)); tr.level = level; tr.iteration = iteration;
      int n_meas = 0;
      float error = eval_forward(img, level, z, H_sub, n_meas);
      tr.error = error; tr.n_meas = n_meas;
      if (error <= last_error) {
        old_state = (*state);
        last_error = error;
        H_T_H = MState::Zero();
        G = MState::Zero();
        double HtH7[49], HTz[7];
        for (int a = 0; a < 7; a++) {                       // H_sub^T z: row-major GEMV in Eigen (:1662); H_sub^T H_sub: GEMM (:1660); order of additions: orc_math.hpp
          HTz[a] = long_dot_gemv_rowmajor(&H_sub[a], 7, z.data(), 1, (size_t)H_DIM);
          for (int b = 0; b < 7; b++) { const double t = long_dot_gemm(&H_sub[a], 7, &H_sub[b], 7, (size_t)H_DIM); HtH7[a * 7 + b] = t; H_T_H(a, b) = t; }
        }
        MState Pinv, K_1;
        inverse_lu<ORC_DIM_STATE>(state->cov / img_point_cov, Pinv);
        inverse_lu<ORC_DIM_STATE>(H_T_H + Pinv, K_1);                                       // :1661
        VState vec = (*state_propagat) - (*state);
        for (int r = 0; r < ORC_DIM_STATE; r++)
          for (int c = 0; c < 7; c++) { double s = K_1(r, 0) * H_T_H(0, c); for (int k = 1; k < 7; k++) s = s + K_1(r, k) * H_T_H(k, c); G(r, c) = s; }
        VState solution;
        for (int r = 0; r < ORC_DIM_STATE; r++) {                                           // :1667
          double kz = (-K_1(r, 0)) * HTz[0]; for (int k = 1; k < 7; k++) kz = kz + (-K_1(r, k)) * HTz[k];
          double gv = G(r, 0) * vec[0]; for (int k = 1; k < 7; k++) gv = gv + G(r, k) * vec[k];
          solution[r] = kz + vec[r] - gv;
        }
        (*state) += solution;
        V3 rot_add = vec3(solution[0], solution[1], solution[2]);
        V3 t_add = vec3(solution[3], solution[4], solution[5]);
        if ((norm(rot_add) * 57.3f < 0.001f) && (norm(t_add) * 100.0f < 0.001f)) EKF_end = true;   // :1675
        tr.accepted = 1;
        std::memcpy(tr.HtH, HtH7, sizeof(HtH7)); std::memcpy(tr.Htz, HTz, sizeof(HTz)); std::memcpy(tr.solution, solution.a, sizeof(tr.solution));
      } else {
        (*state) = old_state;
        EKF_end = true;
        tr.accepted = 0;
      }
      trace_.push_back(tr);
      if (iteration == max_iterations || EKF_end) break;
    }
    dump_z_ = z; dump_H_ = H_sub;
  }

  // src/vio.cpp:1327-1396
  void precomputeReferencePatches(int level) {
    if (total_points == 0) return;
    Mat<1, 2> Jimg; Mat<2, 3> Jdpi; Mat<1, 3> JdR, Jdt;
    const int H_DIM = total_points * patch_size_total;
    H_sub_inv.assign((size_t)H_DIM * 6, 0.0);
    for (int i = 0; i < total_points; i++) {
      const int scale = (1 << level);
      VisualPoint *pt = visual_submap->voxel_points[i];
      const RefPatch &rp = visual_submap->ref_patches[i];
      const uint8_t *img = rp.img_;
      if (pt == nullptr) continue;
      double depth = norm(pt->pos_ - rp.pos_ref);
      V3 pf = rp.f_ * depth;
      M3 R_ref_w = rp.R_ref_w;
      computeProjectionJacobian(pf, Jdpi);
      M3 p_w_hat = skew(pt->pos_);
      const float u_ref = rp.px_[0];
      const float v_ref = rp.px_[1];
      const int u_ref_i = floorf(rp.px_[0] / scale) * scale;
      const int v_ref_i = floorf(rp.px_[1] / scale) * scale;
      const float subpix_u_ref = (u_ref - u_ref_i) / scale;
      const float subpix_v_ref = (v_ref - v_ref_i) / scale;
      const float w_ref_tl = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
      const float w_ref_tr = subpix_u_ref * (1.0 - subpix_v_ref);
      const float w_ref_bl = (1.0 - subpix_u_ref) * subpix_v_ref;
      const float w_ref_br = subpix_u_ref * subpix_v_ref;
      for (int x = 0; x < patch_size; x++) {
        const uint8_t *img_ptr = img + (v_ref_i + x * scale - patch_size_half * scale) * width + u_ref_i - patch_size_half * scale;
        for (int y = 0; y < patch_size; ++y, img_ptr += scale) {
          float du = 0.5f * ((w_ref_tl * img_ptr[scale] + w_ref_tr * img_ptr[scale * 2] + w_ref_bl * img_ptr[scale * width + scale] +
                              w_ref_br * img_ptr[scale * width + scale * 2]) -
                             (w_ref_tl * img_ptr[-scale] + w_ref_tr * img_ptr[0] + w_ref_bl * img_ptr[scale * width - scale] + w_ref_br * img_ptr[scale * width]));
          float dv = 0.5f * ((w_ref_tl * img_ptr[scale * width] + w_ref_tr * img_ptr[scale + scale * width] + w_ref_bl * img_ptr[width * scale * 2] +
                              w_ref_br * img_ptr[width * scale * 2 + scale]) -
                             (w_ref_tl * img_ptr[-scale * width] + w_ref_tr * img_ptr[-scale * width + scale] + w_ref_bl * img_ptr[0] + w_ref_br * img_ptr[scale]));
          Jimg(0, 0) = du; Jimg(0, 1) = dv;
          Jimg = Jimg * (1.0 / scale);
          JdR = ((Jimg * Jdpi) * R_ref_w) * p_w_hat;
          Jdt = ((-Jimg) * Jdpi) * R_ref_w;
          double *h = &H_sub_inv[((size_t)i * patch_size_total + x * patch_size + y) * 6];
          h[0] = JdR(0, 0); h[1] = JdR(0, 1); h[2] = JdR(0, 2); h[3] = Jdt(0, 0); h[4] = Jdt(0, 1); h[5] = Jdt(0, 2);
        }
      }
    }
    has_ref_patch_cache = true;
  }

  // src/vio.cpp:1398-1518
  void updateStateInverse(const uint8_t *img, int level) {
    if (total_points == 0) return;
    StatesGroup old_state = (*state);
    Mat<1, 3> JdR, Jdt;
    bool EKF_end = false;
    float last_error = std::numeric_limits<float>::max();
    const int H_DIM = total_points * patch_size_total;
    std::vector<double> z(H_DIM, 0.0), H_sub((size_t)H_DIM * 6, 0.0);
    for (int iteration = 0; iteration < max_iterations; iteration++) {
      VisualIterTrace tr; std::memset(&tr, 0, sizeof(tr)); tr.level = level; tr.iteration = iteration;
      if (has_ref_patch_cache == false) precomputeReferencePatches(level);
      int n_meas = 0;
      float error = 0.0;
      M3 Rwi = state->rot_end; V3 Pwi = state->pos_end;
      M3 P_wi_hat = skew(Pwi);
      Rcw = Rci * Rwi.T();
      Pcw = -Rci * Rwi.T() * Pwi + Pci;
      for (int i = 0; i < total_points; i++) {
        float patch_error = 0.0;
        const int scale = (1 << level);
        VisualPoint *pt = visual_submap->voxel_points[i];
        if (pt == nullptr) continue;
        V3 pf = Rcw * pt->pos_ + Pcw;
        double pc[2]; cam.world2cam(pf, pc);
        const float u_ref = pc[0];
        const float v_ref = pc[1];
        const int u_ref_i = floorf(pc[0] / scale) * scale;
        const int v_ref_i = floorf(pc[1] / scale) * scale;
        const float subpix_u_ref = (u_ref - u_ref_i) / scale;
        const float subpix_v_ref = (v_ref - v_ref_i) / scale;
        const float w_ref_tl = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
        const float w_ref_tr = subpix_u_ref * (1.0 - subpix_v_ref);
        const float w_ref_bl = (1.0 - subpix_u_ref) * subpix_v_ref;
        const float w_ref_br = subpix_u_ref * subpix_v_ref;
        std::vector<float> P = visual_submap->warp_patch[i];
        for (int x = 0; x < patch_size; x++) {
          const uint8_t *img_ptr = img + (v_ref_i + x * scale - patch_size_half * scale) * width + u_ref_i - patch_size_half * scale;
          for (int y = 0; y < patch_size; ++y, img_ptr += scale) {
            double res = w_ref_tl * img_ptr[0] + w_ref_tr * img_ptr[scale] + w_ref_bl * img_ptr[scale * width] + w_ref_br * img_ptr[scale * width + scale] -
                         P[patch_size_total * level + x * patch_size + y];
            size_t row = (size_t)i * patch_size_total + x * patch_size + y;
            z[row] = res;
            patch_error += res * res;
            Mat<1, 3> J_dR, J_dt;
            for (int k = 0; k < 3; k++) { J_dR(0, k) = H_sub_inv[row * 6 + k]; J_dt(0, k) = H_sub_inv[row * 6 + 3 + k]; }
            JdR = J_dR * Rwi + (J_dt * P_wi_hat) * Rwi;
            Jdt = J_dt * Rwi;
            double *h = &H_sub[row * 6];
            h[0] = JdR(0, 0); h[1] = JdR(0, 1); h[2] = JdR(0, 2); h[3] = Jdt(0, 0); h[4] = Jdt(0, 1); h[5] = Jdt(0, 2);
            n_meas++;
          }
        }
        visual_submap->errors[i] = patch_error;
        error += patch_error;
      }
      error = error / n_meas;
      tr.error = error; tr.n_meas = n_meas;
      if (error <= last_error) {
        old_state = (*state);
        last_error = error;
        H_T_H = MState::Zero();
        G = MState::Zero();
        double HtH6[36], HTz[6];
        for (int a = 0; a < 6; a++) {
          HTz[a] = long_dot_gemv_rowmajor(&H_sub[a], 6, z.data(), 1, (size_t)H_DIM);
          for (int b = 0; b < 6; b++) { const double t = long_dot_gemm(&H_sub[a], 6, &H_sub[b], 6, (size_t)H_DIM); HtH6[a * 6 + b] = t; H_T_H(a, b) = t; }
        }
        MState Pinv, K_1;
        inverse_lu<ORC_DIM_STATE>(state->cov / img_point_cov, Pinv);
        inverse_lu<ORC_DIM_STATE>(H_T_H + Pinv, K_1);
        VState vec = (*state_propagat) - (*state);
        for (int r = 0; r < ORC_DIM_STATE; r++)
          for (int c = 0; c < 6; c++) { double s = K_1(r, 0) * H_T_H(0, c); for (int k = 1; k < 6; k++) s = s + K_1(r, k) * H_T_H(k, c); G(r, c) = s; }
        VState solution;
        for (int r = 0; r < ORC_DIM_STATE; r++) {
          double kz = (-K_1(r, 0)) * HTz[0]; for (int k = 1; k < 6; k++) kz = kz + (-K_1(r, k)) * HTz[k];
          double gv = G(r, 0) * vec[0]; for (int k = 1; k < 6; k++) gv = gv + G(r, k) * vec[k];
          solution[r] = kz + vec[r] - gv;
        }
        (*state) += solution;
        V3 rot_add = vec3(solution[0], solution[1], solution[2]);
        V3 t_add = vec3(solution[3], solution[4], solution[5]);
        if ((norm(rot_add) * 57.3f < 0.001f) && (norm(t_add) * 100.0f < 0.001f)) { EKF_end = true; }
        tr.accepted = 1;
        for (int a = 0; a < 6; a++) { tr.Htz[a] = HTz[a]; for (int b = 0; b < 6; b++) tr.HtH[a * 7 + b] = HtH6[a * 6 + b]; }
        std::memcpy(tr.solution, solution.a, sizeof(tr.solution));
      } else {
        (*state) = old_state;
        EKF_end = true;
        tr.accepted = 0;
      }
      trace_.push_back(tr);
      if (iteration == max_iterations || EKF_end) break;
    }
    dump_z_ = z; dump_H_ = H_sub;
  }

  // src/vio.cpp:784-802  (updateFrameState :1690-1697 leaves Rcw/Pcw = T_f_w of the final state)
  void computeJacobianAndUpdateEKF(const uint8_t *img) {
    if (total_points == 0) return;
    trace_.clear();
    for (int level = patch_pyrimid_level - 1; level >= 0; level--) {
      if (inverse_composition_en) { has_ref_patch_cache = false; updateStateInverse(img, level); }
      else updateState(img, level);
    }
    state->cov = state->cov - G * state->cov;
    M3 Rwi = state->rot_end; V3 Pwi = state->pos_end;
    Rcw = Rci * Rwi.T();
    Pcw = -Rci * Rwi.T() * Pwi + Pci;
  }
};

} // namespace orc
