// gfx950 kernels for the INVERSE-compositional visual update (vio/inverse_composition_en).
//   k_visual_ref_precompute   : reference src/vio.cpp:1327-1396 (precomputeReferencePatches, once per pyramid level)
//   k_visual_inverse_residual : reference src/vio.cpp:1418-1477 (residuals + Jacobian rows of updateStateInverse)
//   accept / revert / solve   : k_visual_solve (visual_kernels.hpp) — the 6x6 block of the reference is the 7x7 block with a zero
//                               exposure row/column, which yields the identical K_1[:,0:6], G and solution.
// Factorisation: the reference stores H_sub_inv rows [J_dR, J_dt] = g_ref * M_ref with the per-pixel REFERENCE-image gradient
// g_ref = [du,dv]/scale and a patch-constant 2x6 M_ref = [Jpi R_ref_w [p]x | -Jpi R_ref_w]; per iteration every row is multiplied by the
// state-dependent 6x6 T = [[Rwi, 0],[ [Pwi]x Rwi, Rwi ]] (vio.cpp:1470-1474).  So the precompute keeps g_ref (2 doubles / pixel), M_ref
// and sum(g g^T) per patch, and an iteration only needs the residuals and sum(g z): H^T H = N^T (sum g g^T) N, H^T z = N^T sum(g z),
// N = M_ref T — no 64M x 6 matrix is ever read back.
#pragma once
#include "visual_kernels.hpp"

struct VisualRefArgs {
  const uint8_t *ref_imgs;           // [n_ref][height][stride]
  const int32_t *ref_idx;            // [M] reference image of each point (ref_patch->img_)
  const double *ref_px;              // [M][2] ref_patch->px_
  const double *ref_f;               // [M][3] ref_patch->f_
  const double *ref_R;               // [M][9] ref_patch->T_f_w_.rotation_matrix()
  const double *ref_pos;             // [M][3] ref_patch->pos()
  double *gref;                      // [M][64][2]
  double *mref;                      // [M][16]: M_ref (12), sum g0g0, g0g1, g1g1, valid
  int32_t n_ref, pad;
};

// bilinear weights and integer anchor of the reference (vio.cpp:1359-1368 / 1449-1458), float/double mix reproduced
struct Anchor { int ui, vi; float w_tl, w_tr, w_bl, w_br; };
__device__ __forceinline__ Anchor make_anchor(double pcx, double pcy, int scale) {
  Anchor A;
  const float u_ref = (float)pcx, v_ref = (float)pcy;
  A.ui = (int)(floorf((float)(pcx / scale)) * (float)scale);
  A.vi = (int)(floorf((float)(pcy / scale)) * (float)scale);
  const float su = (u_ref - (float)A.ui) / (float)scale, sv = (v_ref - (float)A.vi) / (float)scale;
  A.w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv));
  A.w_tr = (float)((double)su * (1.0 - (double)sv));
  A.w_bl = (float)((1.0 - (double)su) * (double)sv);
  A.w_br = su * sv;
  return A;
}

__global__ void __launch_bounds__(VIS_BLOCK) k_visual_ref_precompute(VisualKernelArgs a, VisualRefArgs r) {
  __shared__ float Wf[VIS_WAVES][11 * 11 + 3];
  __shared__ float Bf[VIS_WAVES][10 * 10 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int patch = blockIdx.x * VIS_WAVES + wave;
  if (patch >= a.M) return;                                   // wave-uniform; no block barrier below
  const int scale = 1 << a.level;                            // no search level here (vio.cpp:1341)
  const double p[3] = {a.pos[(size_t)patch * 3], a.pos[(size_t)patch * 3 + 1], a.pos[(size_t)patch * 3 + 2]};
  const double *rp = r.ref_pos + (size_t)patch * 3, *rf = r.ref_f + (size_t)patch * 3, *RR = r.ref_R + (size_t)patch * 9;
  const double dx = p[0] - rp[0], dy = p[1] - rp[1], dz = p[2] - rp[2];
  const double depth = sqrt((dx * dx + dy * dy) + dz * dz);   // (pt->pos_ - ref_patch->pos()).norm()
  const double pf[3] = {rf[0] * depth, rf[1] * depth, rf[2] * depth};
  double Jpi[6];
  {
    const double z_inv = 1. / pf[2], z_inv_2 = z_inv * z_inv;
    Jpi[0] = a.fx * z_inv; Jpi[1] = 0.0; Jpi[2] = -a.fx * pf[0] * z_inv_2;
    Jpi[3] = 0.0; Jpi[4] = a.fy * z_inv; Jpi[5] = -a.fy * pf[1] * z_inv_2;
  }
  const Anchor A = make_anchor(r.ref_px[(size_t)patch * 2], r.ref_px[(size_t)patch * 2 + 1], scale);
  const int ridx = r.ref_idx[patch];
  const bool inside = (ridx >= 0) && (ridx < r.n_ref) && (A.ui - 5 * scale >= 0) && (A.ui + 5 * scale < a.width) && (A.vi - 5 * scale >= 0) && (A.vi + 5 * scale < a.height);
  double g0 = 0.0, g1 = 0.0;
  if (inside) {
    const uint8_t *img = r.ref_imgs + (size_t)ridx * a.height * a.stride;
    for (int e = lane; e < 121; e += LIVO2_WAVE) { const int wr = e / 11, wc = e - wr * 11; Wf[wave][e] = (float)img[(size_t)(A.vi + (wr - 5) * scale) * a.stride + (A.ui + (wc - 5) * scale)]; }
    vis_wave_sync();
    for (int e = lane; e < 100; e += LIVO2_WAVE) { const int br = e / 10, bc = e - br * 10; const float *w = &Wf[wave][br * 11 + bc]; Bf[wave][e] = ((A.w_tl * w[0] + A.w_tr * w[1]) + A.w_bl * w[11]) + A.w_br * w[12]; }
    vis_wave_sync();
    const int x = lane >> 3, y = lane & 7;
    const float *b = &Bf[wave][(x + 1) * 10 + (y + 1)];
    const float du = 0.5f * (b[1] - b[-1]), dv = 0.5f * (b[10] - b[-10]);
    const double isc = 1.0 / scale;                           // Jimg = Jimg * (1.0 / scale)  (vio.cpp:1385)
    g0 = (double)du * isc; g1 = (double)dv * isc;
  }
  r.gref[((size_t)patch * 64 + lane) * 2] = g0;
  r.gref[((size_t)patch * 64 + lane) * 2 + 1] = g1;
  const double S00 = wave_sum(g0 * g0), S01 = wave_sum(g0 * g1), S11 = wave_sum(g1 * g1);
  if (lane < 2) {                                             // M_ref row `lane`: JdR = ((g Jpi) R_ref_w) [p]x ; Jdt = ((-g) Jpi) R_ref_w   (vio.cpp:1387-1388)
    const double e0 = lane == 0 ? 1.0 : 0.0, e1 = lane == 1 ? 1.0 : 0.0;
    const double a3[3] = {e0 * Jpi[0] + e1 * Jpi[3], e0 * Jpi[1] + e1 * Jpi[4], e0 * Jpi[2] + e1 * Jpi[5]};
    double aR[3], naR[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { aR[j] = (a3[0] * RR[j] + a3[1] * RR[3 + j]) + a3[2] * RR[6 + j]; naR[j] = ((-a3[0]) * RR[j] + (-a3[1]) * RR[3 + j]) + (-a3[2]) * RR[6 + j]; }
    const double ph[9] = {0.0, -p[2], p[1], p[2], 0.0, -p[0], -p[1], p[0], 0.0};
    double *m = r.mref + (size_t)patch * 16 + lane * 6;
#pragma unroll
    for (int j = 0; j < 3; j++) { m[j] = (aR[0] * ph[j] + aR[1] * ph[3 + j]) + aR[2] * ph[6 + j]; m[3 + j] = naR[j]; }
  }
  if (lane == 2) { double *m = r.mref + (size_t)patch * 16; m[12] = S00; m[13] = S01; m[14] = S11; m[15] = inside ? 1.0 : 0.0; }
}

template <bool DEBUG_ROWS>
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_inverse_residual(VisualKernelArgs a, VisualRefArgs r, const DevCtl *__restrict__ ctl,
                                                                       double *__restrict__ partials, int check_stop) {
  if (check_stop && ctl->hdr.stop) return;
  __shared__ float Wf[VIS_WAVES][9 * 9 + 3];
  __shared__ double Rr[VIS_WAVES][LIVO2_WAVE];
  __shared__ double red[VIS_WAVES][VIS_PSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int patch = blockIdx.x * VIS_WAVES + wave;
  double out_val = 0.0;
  if (patch < a.M) {
    const double *Rwi = ctl->cur.rot, *Pwi = ctl->cur.pos;
    double Rcw[9], Pcw[3];
    mat3_mul_Bt(a.Rci, Rwi, Rcw);
#pragma unroll
    for (int j = 0; j < 3; j++) Pcw[j] = a.Pci[j] - ((Rcw[j * 3] * Pwi[0] + Rcw[j * 3 + 1] * Pwi[1]) + Rcw[j * 3 + 2] * Pwi[2]);
    const double p0 = a.pos[(size_t)patch * 3], p1 = a.pos[(size_t)patch * 3 + 1], p2 = a.pos[(size_t)patch * 3 + 2];
    double pf[3];
#pragma unroll
    for (int j = 0; j < 3; j++) pf[j] = ((Rcw[j * 3] * p0 + Rcw[j * 3 + 1] * p1) + Rcw[j * 3 + 2] * p2) + Pcw[j];
    double pcx, pcy;
    {
      const double u0 = pf[0] / pf[2], u1 = pf[1] / pf[2];
      cam_project(a.distortion, a.d, a.fx, a.fy, a.cx, a.cy, u0, u1, pcx, pcy);
    }
    const int scale = 1 << a.level;                          // vio.cpp:1437 (no search level)
    const Anchor A = make_anchor(pcx, pcy, scale);
    const double *mr = r.mref + (size_t)patch * 16;
    // the reference reads rows/cols -4..+4 of the strided grid unchecked (vio.cpp:1463-1468); a window leaving the image is skipped
    const bool inside = (mr[15] != 0.0) && (A.ui - 4 * scale >= 0) && (A.ui + 4 * scale < a.width) && (A.vi - 4 * scale >= 0) && (A.vi + 4 * scale < a.height);
    if (inside) {
      for (int e = lane; e < 81; e += LIVO2_WAVE) { const int wr = e / 9, wc = e - wr * 9; Wf[wave][e] = (float)a.img[(size_t)(A.vi + (wr - 4) * scale) * a.stride + (A.ui + (wc - 4) * scale)]; }
      vis_wave_sync();
      const int x = lane >> 3, y = lane & 7;
      const float *w = &Wf[wave][x * 9 + y];
      const float Pref = a.warp[((size_t)patch * a.L + a.level) * 64 + lane];
      const double res = (double)((((A.w_tl * w[0] + A.w_tr * w[1]) + A.w_bl * w[9]) + A.w_br * w[10]) - Pref);   // all-float expression (vio.cpp:1466-1467)
      const double g0 = r.gref[((size_t)patch * 64 + lane) * 2], g1 = r.gref[((size_t)patch * 64 + lane) * 2 + 1];
      // N = M_ref * T :  JdR = J_dR Rwi + (J_dt [Pwi]x) Rwi ; Jdt = J_dt Rwi    (vio.cpp:1472-1473)
      const double Ph[9] = {0.0, -Pwi[2], Pwi[1], Pwi[2], 0.0, -Pwi[0], -Pwi[1], Pwi[0], 0.0};
      double N0[6], N1[6];
#pragma unroll
      for (int row = 0; row < 2; row++) {
        const double *m = mr + row * 6;
        double tP[3];
#pragma unroll
        for (int j = 0; j < 3; j++) tP[j] = (m[3] * Ph[j] + m[4] * Ph[3 + j]) + m[5] * Ph[6 + j];
        double *N = row == 0 ? N0 : N1;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double t1 = (m[0] * Rwi[j] + m[1] * Rwi[3 + j]) + m[2] * Rwi[6 + j];
          const double t2 = (tP[0] * Rwi[j] + tP[1] * Rwi[3 + j]) + tP[2] * Rwi[6 + j];
          N[j] = t1 + t2;
          N[3 + j] = (m[3] * Rwi[j] + m[4] * Rwi[3 + j]) + m[5] * Rwi[6 + j];
        }
      }
      if (DEBUG_ROWS) {
        if (a.z) a.z[(size_t)patch * 64 + lane] = res;
        if (a.H_sub) {
          double *h = a.H_sub + ((size_t)patch * 64 + lane) * 7;
#pragma unroll
          for (int k = 0; k < 6; k++) h[k] = g0 * N0[k] + g1 * N1[k];
          h[6] = 0.0;
        }
      }
      const double Sgz0 = wave_sum(g0 * res), Sgz1 = wave_sum(g1 * res);
      const double S00 = mr[12], S01 = mr[13], S11 = mr[14];
      // float patch_error += res * res in pixel order (vio.cpp:1469), every lane runs the chain from broadcast LDS reads
      Rr[wave][lane] = res * res;
      vis_wave_sync();
      float patch_error = 0.0f;
#pragma unroll
      for (int i = 0; i < 64; i++) patch_error = (float)((double)patch_error + Rr[wave][i]);
      if (a.errors && lane == 0) a.errors[patch] = patch_error;
      if (lane < 28) {
        int rr = 0, q = lane;
        while (q >= 7 - rr) { q -= 7 - rr; rr++; }
        const int cc = rr + q;
        double n0r = 0, n1r = 0, n0c = 0, n1c = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { if (k == rr) { n0r = N0[k]; n1r = N1[k]; } if (k == cc) { n0c = N0[k]; n1c = N1[k]; } }
        out_val = (cc < 6) ? ((n0r * n0c) * S00 + (n0r * n1c + n1r * n0c) * S01 + (n1r * n1c) * S11) : 0.0;
      } else if (lane < 35) {
        const int rr = lane - 28;
        double n0r = 0, n1r = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k == rr) { n0r = N0[k]; n1r = N1[k]; }
        out_val = (rr < 6) ? (n0r * Sgz0 + n1r * Sgz1) : 0.0;
      } else if (lane == 35) out_val = (double)patch_error;
      else if (lane == 36) out_val = 64.0;
    } else if (a.errors && lane == 0) a.errors[patch] = 0.f;
  }
  if (lane < VIS_PSTRIDE) red[wave][lane] = out_val;
  __syncthreads();
  if (tid < VIS_PSTRIDE) {
    double v = red[0][tid];
#pragma unroll
    for (int w = 1; w < VIS_WAVES; w++) v = v + red[w][tid];
    partials[(size_t)blockIdx.x * VIS_PSTRIDE + tid] = v;
  }
}
