// Visual sub-map retrieval, per-point tail: for every visual map point that the host-side grid / depth-continuity
// selection of VIOManager::retrieveFromVisualSparseMap kept (reference src/vio.cpp:352-672), the reference
//   * builds the affine warp reference -> current  (getWarpMatrixAffineHomography 247-272 if normal_en, else getWarpMatrixAffine 274-290),
//   * picks the search level (getBestSearchLevel 320-331),
//   * warps patch_pyrimid_level 8x8 patches out of the reference image (warpAffine 292-318, vk::interpolateMat_8u),
//   * samples the 8x8 patch of the current image at the projected pixel (getImagePatch 203-225, level 0),
//   * gates on the exposure-compensated photometric error and (optionally) NCC (742-760, calculateNCC 333-350),
//   * appends the survivors to visual_submap (762-767).
// Here: one wave per candidate, one lane per patch pixel; the survivors are compacted in candidate order straight into the resident
// frame arrays (pos / warp_patch / search_levels / inv_expo_list) that k_visual_residual reads — the M*L*256-byte warp_patch upload
// of livo2_visual_set_frame disappears.  The patch-constant algebra is evaluated redundantly by all lanes with the oracle's
// operation order (oracle/orc_warp.hpp); the float error sum and the double NCC sums run serially on one lane in the reference's
// order so that the gate inputs are bit-identical.
#pragma once
#include "livo2_device.hpp"

#define WARP_WAVES 4

struct WarpKernelArgs {
  const uint8_t *img, *ref_imgs;
  int32_t width, height, stride, n, L, normal_en, ncc_en, pad;
  double fx, fy, cx, cy, inv_expo_cur, ncc_thre, outlier_threshold;
  double d[5];                      // vk::PinholeCamera radial-tangential coefficients, used when distortion != 0
  int32_t distortion, pad2;
  double R_cur[9], t_cur[3];
  const double *pos, *normal, *ref_px, *ref_f, *ref_R, *ref_t, *ref_inv_expo;
  const int32_t *ref_img_idx, *ref_level;
  const int32_t *n_dev;             // candidate count on the device (chained retrieval); NULL: n
  const int32_t *leader;            // !normal_en: the candidate whose warp this one reuses (warp_map, vio.cpp:716-734); NULL: itself
  float *patch_all;                 // [n][L][64]
  int32_t *accepted, *search_level;
  float *error;
  double *ncc, *A;
};

__device__ __forceinline__ void w_mat3_mul(const double *A, const double *B, double *C) {          // C = A B, ((a0 b0 + a1 b1) + a2 b2)
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) C[r * 3 + c] = (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c]) + A[r * 3 + 2] * B[6 + c];
}
__device__ __forceinline__ void w_mat3_vec(const double *A, const double *v, double *o) {
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = (A[r * 3] * v[0] + A[r * 3 + 1] * v[1]) + A[r * 3 + 2] * v[2];
}
__device__ __forceinline__ void w_mat3t_vec(const double *A, const double *v, double *o) {          // A^T v
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = (A[r] * v[0] + A[3 + r] * v[1]) + A[6 + r] * v[2];
}
// cam->world2cam / cam->cam2world: the camera models of livo2_device.hpp (cam_project / cam_unproject) — operation order of oracle/orc_visual.hpp (world2cam)
// and oracle/orc_warp.hpp (cam2world)
__device__ __forceinline__ void w_world2cam(const WarpKernelArgs &a, const double *p, double *px) {
  const double u = p[0] / p[2], v = p[1] / p[2];
  cam_project(a.distortion, a.d, a.fx, a.fy, a.cx, a.cy, u, v, px[0], px[1]);
}
__device__ __forceinline__ void w_cam2world(const WarpKernelArgs &a, double u, double v, double *o) {
  double x, y;
  cam_unproject(a.distortion, a.d, a.fx, a.fy, a.cx, a.cy, u, v, x, y);
  const double nrm = sqrt((x * x + y * y) + 1.0 * 1.0);
  o[0] = x / nrm; o[1] = y / nrm; o[2] = 1.0 / nrm;
}

__global__ void __launch_bounds__(WARP_WAVES *LIVO2_WAVE) k_warp_candidates(WarpKernelArgs a) {
  __shared__ float s_pair[WARP_WAVES][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * WARP_WAVES + wave);
  if (i >= (a.n_dev ? a.n_dev[0] : a.n)) return;
  // ---- candidate-constant algebra (every lane, same values) ----
  // the warp matrix comes from candidate j: i itself, or with !normal_en the first candidate whose ref_ftr was made in the same frame
  const int j = (a.leader && !a.normal_en) ? a.leader[i] : i;
  double pos[3], posj[3], Rr[9], tr[3], pxr[2], pxj[2];
#pragma unroll
  for (int k = 0; k < 3; k++) { pos[k] = a.pos[(size_t)i * 3 + k]; posj[k] = a.pos[(size_t)j * 3 + k]; tr[k] = a.ref_t[(size_t)j * 3 + k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) Rr[k] = a.ref_R[(size_t)j * 9 + k];
  pxr[0] = a.ref_px[(size_t)i * 2]; pxr[1] = a.ref_px[(size_t)i * 2 + 1];
  pxj[0] = a.ref_px[(size_t)j * 2]; pxj[1] = a.ref_px[(size_t)j * 2 + 1];
  double Rcr[9], tcr[3];                                            // T_cur_ref = T_cur * T_ref^-1
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) Rcr[r * 3 + c] = (a.R_cur[r * 3] * Rr[c * 3] + a.R_cur[r * 3 + 1] * Rr[c * 3 + 1]) + a.R_cur[r * 3 + 2] * Rr[c * 3 + 2];
  {
    double q[3]; w_mat3_vec(Rcr, tr, q);
#pragma unroll
    for (int k = 0; k < 3; k++) tcr[k] = a.t_cur[k] - q[k];
  }
  double pc[2];                                                     // new_frame_->w2c(pt->pos_)
  {
    double q[3], pf[3]; w_mat3_vec(a.R_cur, pos, q);
#pragma unroll
    for (int k = 0; k < 3; k++) pf[k] = q[k] + a.t_cur[k];
    w_world2cam(a, pf, pc);
  }
  double A[4];
  if (a.normal_en) {
    double nrm[3], nv[3], pf[3], q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) nrm[k] = a.normal[(size_t)i * 3 + k];
    w_mat3_vec(Rr, nrm, nv);
    const double nn = sqrt((nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) nv[k] = nv[k] / nn;
    w_mat3_vec(Rr, pos, q);
#pragma unroll
    for (int k = 0; k < 3; k++) pf[k] = q[k] + tr[k];
    // getWarpMatrixAffineHomography(cam, px_ref, pf, nv, T_cur_ref, 0, A)
    double t[3]; w_mat3t_vec(Rcr, tcr, t);
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = t[k] * (-1.0);
    const double s = (nv[0] * pf[0] + nv[1] * pf[1]) + nv[2] * pf[2];
    double K[9], H[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) K[r * 3 + c] = ((r == c) ? 1.0 : 0.0) * s - t[r] * nv[c];
    w_mat3_mul(Rcr, K, H);
    double fdu[3], fdv[3], c0[3], c1[3], c2[3], p0[2], p1[2], p2[2];
    w_cam2world(a, pxr[0] + 4.0, pxr[1], fdu);
    w_cam2world(a, pxr[0], pxr[1] + 4.0, fdv);
    w_mat3_vec(H, pf, c0); w_mat3_vec(H, fdu, c1); w_mat3_vec(H, fdv, c2);
    w_world2cam(a, c0, p0); w_world2cam(a, c1, p1); w_world2cam(a, c2, p2);
    A[0] = (p1[0] - p0[0]) / 4.0; A[2] = (p1[1] - p0[1]) / 4.0;
    A[1] = (p2[0] - p0[0]) / 4.0; A[3] = (p2[1] - p0[1]) / 4.0;
  } else {
    double f[3], rp[3], q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = a.ref_f[(size_t)j * 3 + k];
    w_mat3t_vec(Rr, tr, rp);
#pragma unroll
    for (int k = 0; k < 3; k++) { rp[k] = rp[k] * (-1.0); q[k] = rp[k] - posj[k]; }           // Feature::pos() - pt->pos_
    const double depth = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
    const int lr = a.ref_level[j];
    const double step = (double)(4 * (1 << lr) * (1 << 0));
    double xr[3], xdu[3], xdv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) xr[k] = f[k] * depth;
    w_cam2world(a, pxj[0] + step, pxj[1], xdu);
    w_cam2world(a, pxj[0], pxj[1] + step, xdv);
    const double su = xr[2] / xdu[2], sv = xr[2] / xdv[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { xdu[k] = xdu[k] * su; xdv[k] = xdv[k] * sv; }
    double c0[3], c1[3], c2[3], p0[2], p1[2], p2[2];
    w_mat3_vec(Rcr, xr, c0); w_mat3_vec(Rcr, xdu, c1); w_mat3_vec(Rcr, xdv, c2);
#pragma unroll
    for (int k = 0; k < 3; k++) { c0[k] = c0[k] + tcr[k]; c1[k] = c1[k] + tcr[k]; c2[k] = c2[k] + tcr[k]; }
    w_world2cam(a, c0, p0); w_world2cam(a, c1, p1); w_world2cam(a, c2, p2);
    A[0] = (p1[0] - p0[0]) / 4.0; A[2] = (p1[1] - p0[1]) / 4.0;
    A[1] = (p2[0] - p0[0]) / 4.0; A[3] = (p2[1] - p0[1]) / 4.0;
  }
  int search_level = 0;                                             // getBestSearchLevel(A, 2)
  {
    double D = A[0] * A[3] - A[1] * A[2];
    while (D > 3.0 && search_level < 2) { search_level += 1; D *= 0.25; }
  }
  // ---- warpAffine for every pyramid level: lane = (y, x) ----
  const double invdet = 1.0 / (A[0] * A[3] - A[1] * A[2]);
  const float a00 = (float)(A[3] * invdet), a01 = (float)(-A[1] * invdet), a10 = (float)(-A[2] * invdet), a11 = (float)(A[0] * invdet);
  const bool warp_ok = !(a00 != a00);                               // isnan(A_ref_cur(0,0)): the reference leaves the (zeroed) patch untouched
  const uint8_t *ref = a.ref_imgs + (size_t)a.ref_img_idx[i] * ((size_t)a.stride * a.height);
  const float pxr0 = (float)pxr[0], pxr1 = (float)pxr[1];
  const int x = lane & 7, y = lane >> 3;
  float *dst = a.patch_all + (size_t)i * a.L * 64;
  float wrap0 = 0.f;
  for (int lvl = 0; lvl < a.L; lvl++) {
    float v = 0.f;
    if (warp_ok) {
      float p0 = (float)(x - 4), p1 = (float)(y - 4);
      p0 *= (float)(1 << search_level); p1 *= (float)(1 << search_level);
      p0 *= (float)(1 << lvl); p1 *= (float)(1 << lvl);
      const float u = (a00 * p0 + a01 * p1) + pxr0, w = (a10 * p0 + a11 * p1) + pxr1;
      if (!(u < 0 || w < 0 || u >= (float)(a.width - 1) || w >= (float)(a.height - 1))) {
        const int xi = (int)floorf(u), yi = (int)floorf(w);
        const float sx = u - (float)xi, sy = w - (float)yi;
        const float w00 = (1.0f - sx) * (1.0f - sy), w01 = (1.0f - sx) * sy, w10 = sx * (1.0f - sy);
        const float w11 = 1.0f - w00 - w01 - w10;
        const uint8_t *p = ref + (size_t)yi * a.stride + xi;
        v = ((w00 * (float)p[0] + w01 * (float)p[a.stride]) + w10 * (float)p[1]) + w11 * (float)p[a.stride + 1];
      }
    }
    dst[lvl * 64 + lane] = v;
    if (lvl == 0) wrap0 = v;
  }
  // ---- getImagePatch(img, pc, patch_buffer, 0): lane = (row, col) ----
  const float u_ref = (float)pc[0], v_ref = (float)pc[1];
  const int u_i = (int)floorf((float)pc[0]), v_i = (int)floorf((float)pc[1]);          // scale = 1
  const float su = u_ref - (float)u_i, sv = v_ref - (float)v_i;
  const float w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv)), w_tr = (float)((double)su * (1.0 - (double)sv));
  const float w_bl = (float)((1.0 - (double)su) * (double)sv), w_br = su * sv;
  // the reference reads out of bounds when the 9x9 window leaves the image; here such a candidate is rejected (error = +inf)
  const bool inside = u_i - 4 >= 0 && v_i - 4 >= 0 && u_i + 4 < a.width && v_i + 4 < a.height && pc[0] == pc[0] && pc[1] == pc[1];
  float buf = 0.f;
  if (inside) {
    const uint8_t *p = a.img + (size_t)(v_i - 4 + y) * a.stride + (u_i - 4 + x);
    buf = ((w_tl * (float)p[0] + w_tr * (float)p[1]) + w_bl * (float)p[a.stride]) + w_br * (float)p[a.stride + 1];
  }
  s_pair[wave][0][lane] = wrap0; s_pair[wave][1][lane] = buf;
  __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the wave's own LDS writes have landed (no barrier: one wave)
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    const double ier = a.ref_inv_expo[i], iec = a.inv_expo_cur;
    const float *pw = s_pair[wave][0], *pb = s_pair[wave][1];
    float error = 0.0f;                                             // vio.cpp:742-749: float accumulator, double terms
    double sum_ref = 0.0, sum_cur = 0.0;
    for (int k = 0; k < 64; k++) {
      const double dlt = ier * (double)pw[k] - iec * (double)pb[k];
      error = (float)((double)error + dlt * dlt);
      sum_ref += (double)pw[k]; sum_cur += (double)pb[k];
    }
    const double mean_ref = sum_ref / 64, mean_cur = sum_cur / 64;
    double num = 0, d1 = 0, d2 = 0;
    for (int k = 0; k < 64; k++) {
      const double r = (double)pw[k] - mean_ref, c = (double)pb[k] - mean_cur;
      num += r * c; d1 += r * r; d2 += c * c;
    }
    const double ncc = num / sqrt(d1 * d2 + 1e-10);
    int ok = 1;
    if (a.ncc_en && ncc < a.ncc_thre) ok = 0;
    if ((double)error > a.outlier_threshold * 64) ok = 0;
    if (!inside) { ok = 0; error = __builtin_inff(); }
    a.accepted[i] = ok; a.search_level[i] = search_level; a.error[i] = error; a.ncc[i] = ncc;
    a.A[(size_t)i * 4] = A[0]; a.A[(size_t)i * 4 + 1] = A[1]; a.A[(size_t)i * 4 + 2] = A[2]; a.A[(size_t)i * 4 + 3] = A[3];
  }
}

// exclusive scan of the accept flags (one block, one pass): slot[i] = position among the survivors or -1 ; count[0] = number of survivors.
// Every thread owns a contiguous run of ceil(n / 1024) flags: it counts them, the block scans the 1024 counts, the thread then hands out the slots of
// its run (the first version walked the array 1024 flags at a time with three barriers per step: 14 us for the 13 056 cells of the avia grid, now ~4).
__global__ void __launch_bounds__(1024) k_warp_scan(const int32_t *__restrict__ accepted, int n, const int32_t *__restrict__ n_dev, int32_t *__restrict__ slot,
                                                    int32_t *__restrict__ count) {
  __shared__ int s_wave[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n_dev) n = n_dev[0];
  const int per = (n + 1023) / 1024;
  const int b = tid * per, e = (b + per < n) ? b + per : n;
  int sum = 0;
  unsigned own = 0;                                                 // the flags of a run of <= 16, so that its loads go out together and are read once
  if (per <= 16) {
    int f[16];
#pragma unroll
    for (int u = 0; u < 16; u++) f[u] = (b + u < e) ? accepted[b + u] : 0;
#pragma unroll
    for (int u = 0; u < 16; u++) if (f[u]) { own |= 1u << u; sum++; }
  } else {
    for (int i = b; i < e; i++) sum += accepted[i] ? 1 : 0;
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { const int t = s_wave[w]; if (w < wave) woff += t; tot += t; }
  int run = woff + incl - sum;                                      // survivors before this thread's run
  if (per <= 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) if (b + u < e) { const int f = (own >> u) & 1; slot[b + u] = f ? run : -1; run += f; }
  } else {
    for (int i = b; i < e; i++) { const int f = accepted[i] ? 1 : 0; slot[i] = f ? run : -1; run += f; }
  }
  if (tid == 0) count[0] = tot;
}

// survivors -> the resident frame arrays, in candidate order (visual_submap push_backs, vio.cpp:762-767)
// (cand_point / cand_obs -> sub_point / sub_obs: which visual point and observation each survivor is; NULL outside the chained retrieval)
__global__ void __launch_bounds__(WARP_WAVES *LIVO2_WAVE) k_warp_gather(const int32_t *__restrict__ slot, int n, const int32_t *__restrict__ n_dev, int L,
                                                                       const float *__restrict__ patch_all, const double *__restrict__ pos,
                                                                       const int32_t *__restrict__ search_level, const double *__restrict__ ref_inv_expo,
                                                                       float *__restrict__ warp, double *__restrict__ fpos, int32_t *__restrict__ fsearch,
                                                                       double *__restrict__ finvexpo, const int32_t *__restrict__ cand_point,
                                                                       const int32_t *__restrict__ cand_obs, int32_t *__restrict__ sub_point,
                                                                       int32_t *__restrict__ sub_obs) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * WARP_WAVES + (threadIdx.x >> 6);
  if (n_dev) n = n_dev[0];
  if (i >= n) return;
  const int s = slot[i];
  if (s < 0) return;
  for (int k = lane; k < L * 64; k += 64) warp[(size_t)s * L * 64 + k] = patch_all[(size_t)i * L * 64 + k];
  if (lane < 3) fpos[(size_t)s * 3 + lane] = pos[(size_t)i * 3 + lane];
  if (lane == 3) fsearch[s] = search_level[i];
  if (lane == 4) finvexpo[s] = ref_inv_expo[i];
  if (lane == 5 && sub_point) sub_point[s] = cand_point[i];
  if (lane == 6 && sub_obs) sub_obs[s] = cand_obs[i];
}
