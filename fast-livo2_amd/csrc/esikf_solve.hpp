// Single-wavefront ESIKF solve on gfx950.
//   LiDAR : reference src/voxel_map.cpp:468-474 (K_1, G, vec, solution, state_ += solution)
//   visual: reference src/vio.cpp:1657-1669
//   StatesGroup operator+= / operator-  : reference include/common_lib.h:182-206
//
// The reference forms K_1 = (H_T_H + P^-1)^-1 with two general 19x19 inversions per iteration, although H_T_H is non-zero
// only in its leading k x k block (k = 6 LiDAR, 7 visual) and only the first k columns of K_1 are ever used
// (voxel_map.cpp:469,472; vio.cpp:1665,1667).  With P' = P / meas_cov_scale the matrix-inversion lemma gives exactly
//        K_1[:, 0:k] = P'[:, 0:k] * (I_k + H_k * P'[0:k, 0:k])^-1
// so one k x k Gauss-Jordan replaces both 19x19 LU inversions.  Algebraically identical; measured agreement with the
// double-inversion form on the test scenes is ~1e-15 relative (tests/test_solve_gpu.py), far inside the 1e-5 contract.
// The whole 19-dim algebra stays on the device so the <=5-iteration loop never returns to the host.
// Launch geometry: ONE wave (64 threads); LDS hand-offs are ordered by __syncthreads() on a 1-wave block.
#pragma once
#include "livo2_device.hpp"

#define KMAX 7

struct SolveLds {
  double P[DS * DS];           // P' = cov / scale
  double aug[KMAX * 2 * KMAX]; // [S | I] during Gauss-Jordan, row stride 2k
  double Kc[DS * KMAX];        // K_1[:, 0:k], row stride KMAX
  double G[DS * KMAX];         // G[:, 0:k],   row stride KMAX
  double hth[49];              // H_k, row stride k
  double htz[8];
  double vec[DS + 1];
  double sol[DS + 1];
};

// One Kalman update of ctl->cur given the reduced sums (s.hth: k x k row-major with stride k, s.htz).
// sign=+1: LiDAR form (K1*HTz + vec - G*vec) ; sign=-1: visual form (-K1*HTz + vec - G*vec).
// Leaves the solution in s.sol, G[:,0:k] in s.G (and the zero-padded 19x19 G in ctl->G).
// Part 1 (independent of the measurement sums, so it can overlap the partial-sum loads): P' = cov / scale and
// vec = state_propagat [-] state (common_lib.h:194-206) into LDS.  Call from wave 0 only; no barrier inside.
__device__ inline void esikf_prefetch_wave(const DevCtl *ctl, SolveLds &s, const double meas_cov_scale, const int lane) {
  double c[6];
#pragma unroll
  for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; c[q] = (e < DS * DS) ? ctl->cur.cov[e] : 0.0; }
  if (lane == 0) {
    double rotd[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      rotd[i * 3 + j] = (ctl->cur.rot[i] * ctl->prop.rot[j] + ctl->cur.rot[3 + i] * ctl->prop.rot[3 + j]) + ctl->cur.rot[6 + i] * ctl->prop.rot[6 + j];   // cur^T * prop
    double l[3]; so3_log(rotd, l);
    for (int i = 0; i < 3; i++) {
      s.vec[i] = l[i];
      s.vec[3 + i] = ctl->prop.pos[i] - ctl->cur.pos[i];
      s.vec[7 + i] = ctl->prop.vel[i] - ctl->cur.vel[i];
      s.vec[10 + i] = ctl->prop.bg[i] - ctl->cur.bg[i];
      s.vec[13 + i] = ctl->prop.ba[i] - ctl->cur.ba[i];
      s.vec[16 + i] = ctl->prop.grav[i] - ctl->cur.grav[i];
    }
    s.vec[6] = ctl->prop.inv_expo - ctl->cur.inv_expo;
  }
#pragma unroll
  for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; if (e < DS * DS) s.P[e] = c[q] / meas_cov_scale; }
}

// Part 2: needs s.P / s.vec (esikf_prefetch_wave + a barrier) and the sums in s.hth / s.htz.
template <int k>
__device__ inline void esikf_update_wave(DevCtl *ctl, SolveLds &s, const int sign, const int lane) {
  constexpr int k2 = 2 * k;
  // aug = [ I + H_k P'_kk | I ]
  for (int e = lane; e < k * k2; e += LIVO2_WAVE) {
    const int i = e / k2, j = e % k2;
    double v;
    if (j < k) {
      v = (i == j) ? 1.0 : 0.0;
      for (int m = 0; m < k; m++) v = fma(s.hth[i * k + m], s.P[m * DS + j], v);
    } else v = (j - k == i) ? 1.0 : 0.0;
    s.aug[e] = v;
  }
  __syncthreads();
  // Gauss-Jordan with partial pivoting; every lane scans the pivot column itself (LDS broadcast reads), so no extra hand-off
  for (int c = 0; c < k; c++) {
    int piv = c;
    double best = fabs(s.aug[c * k2 + c]);
    for (int i = c + 1; i < k; i++) { double v = fabs(s.aug[i * k2 + c]); if (v > best) { best = v; piv = i; } }
    if (piv != c) {                                          // wave-uniform
      if (lane < k2) { double t = s.aug[c * k2 + lane]; s.aug[c * k2 + lane] = s.aug[piv * k2 + lane]; s.aug[piv * k2 + lane] = t; }
      __syncthreads();
    }
    const double pinv = 1.0 / s.aug[c * k2 + c];
    double nv[2]; int ne[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int e = lane + q * LIVO2_WAVE;
      ne[q] = e;
      if (e < k * k2) {
        const int i = e / k2, j = e % k2;
        const double prow = s.aug[c * k2 + j] * pinv;
        nv[q] = (i == c) ? prow : fma(-s.aug[i * k2 + c], prow, s.aug[e]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++) if (ne[q] < k * k2) s.aug[ne[q]] = nv[q];
    __syncthreads();
  }
  // Kc = P'[:, 0:k] * S^-1
  for (int e = lane; e < DS * k; e += LIVO2_WAVE) {
    const int r = e / k, c = e % k;
    double v = 0.0;
    for (int m = 0; m < k; m++) v = fma(s.P[r * DS + m], s.aug[m * k2 + k + c], v);
    s.Kc[r * KMAX + c] = v;
  }
  __syncthreads();
  // G[:, 0:k] = K_1[:, 0:k] * H_k
  for (int e = lane; e < DS * DS; e += LIVO2_WAVE) {
    const int r = e / DS, c = e % DS;
    double g = 0.0;
    if (c < k) {
      for (int m = 0; m < k; m++) g = fma(s.Kc[r * KMAX + m], s.hth[m * k + c], g);
      s.G[r * KMAX + c] = g;
    }
    ctl->G[e] = g;
  }
  __syncthreads();
  if (lane < DS) {
    const int r = lane;
    double kz = 0.0, gv = 0.0;
    for (int m = 0; m < k; m++) { kz = fma(s.Kc[r * KMAX + m], s.htz[m], kz); gv = fma(s.G[r * KMAX + m], s.vec[m], gv); }
    s.sol[r] = ((sign > 0) ? kz : -kz) + s.vec[r] - gv;
  }
  __syncthreads();
  if (lane == 0) {                                           // state += solution   (common_lib.h:182-192)
    double E[9], Rn[9];
    so3_exp(s.sol[0], s.sol[1], s.sol[2], E);
    mat3_mul(ctl->cur.rot, E, Rn);
    for (int i = 0; i < 9; i++) ctl->cur.rot[i] = Rn[i];
    for (int i = 0; i < 3; i++) {
      ctl->cur.pos[i] += s.sol[3 + i]; ctl->cur.vel[i] += s.sol[7 + i]; ctl->cur.bg[i] += s.sol[10 + i];
      ctl->cur.ba[i] += s.sol[13 + i]; ctl->cur.grav[i] += s.sol[16 + i];
    }
    ctl->cur.inv_expo += s.sol[6];
  }
  __syncthreads();
}
