// Single-wavefront ESIKF solve on gfx950: 19x19 partial-pivot LU inverse, Kalman gain blocks, boxminus / boxplus.
//   LiDAR : reference src/voxel_map.cpp:468-474 (K_1, G, vec, solution, state_ += solution)
//   visual: reference src/vio.cpp:1657-1669
//   StatesGroup operator+= / operator-  : reference include/common_lib.h:182-206
// The whole 19-dim algebra stays on the device so that the <=5-iteration loop never returns to the host
// (a dependent kernel boundary costs ~1.5 us on MI355X, a host round trip >10 us).
// Launch geometry: ONE wave (64 threads).  All LDS hand-offs are wave-synchronous (__syncthreads on a 1-wave block).
#pragma once
#include "livo2_device.hpp"

struct SolveLds {
  double A[DS * DS];       // matrix being factorised (LU in place)
  double K1[DS * DS];      // inverse result
  double G[DS * DS];
  double cov[DS * DS];
  double hth[49];
  double htz[8];
  double vec[DS + 1];
  double sol[DS + 1];
  int perm[DS + 1];
};

// inv = A^-1 by partial-pivot LU + solve against the identity.  A is destroyed.  Same operation order as the
// CPU restatement (division for the multipliers, a(i,j) -= l*a(k,j), first-maximum pivot) so both agree to rounding.
__device__ inline void inverse19_wave(double *A, double *inv, int *perm, int lane) {
  if (lane < DS) perm[lane] = lane;
  __syncthreads();
  for (int k = 0; k < DS; k++) {
    int piv = k;
    double best = fabs(A[k * DS + k]);
    for (int i = k + 1; i < DS; i++) { double v = fabs(A[i * DS + k]); if (v > best) { best = v; piv = i; } }
    if (piv != k) {                                       // wave-uniform
      if (lane < DS) { double t = A[k * DS + lane]; A[k * DS + lane] = A[piv * DS + lane]; A[piv * DS + lane] = t; }
      if (lane == 0) { int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
    }
    __syncthreads();
    const int m = DS - 1 - k, w = m + 1, tot = m * w;     // rows k+1.., cols k.. (col k receives the multiplier)
    const double akk = A[k * DS + k];
    double newv[6];
#pragma unroll
    for (int e = 0; e < 6; e++) {
      int idx = lane + e * LIVO2_WAVE;
      if (idx < tot) {
        int i = k + 1 + idx / w, j = k + idx % w;
        double l = A[i * DS + k] / akk;
        newv[e] = (j == k) ? l : (A[i * DS + j] - l * A[k * DS + j]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 6; e++) {
      int idx = lane + e * LIVO2_WAVE;
      if (idx < tot) { int i = k + 1 + idx / w, j = k + idx % w; A[i * DS + j] = newv[e]; }
    }
    __syncthreads();
  }
  if (lane < DS) {
    const int c = lane;
    double y[DS];
#pragma unroll
    for (int i = 0; i < DS; i++) {                        // L y = P e_c
      double s = (perm[i] == c) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < i; j++) s = s - A[i * DS + j] * y[j];
      y[i] = s;
    }
#pragma unroll
    for (int i = DS - 1; i >= 0; i--) {                   // U x = y
      double s = y[i];
#pragma unroll
      for (int j = i + 1; j < DS; j++) s = s - A[i * DS + j] * y[j];
      y[i] = s / A[i * DS + i];
    }
#pragma unroll
    for (int i = 0; i < DS; i++) inv[i * DS + c] = y[i];
  }
  __syncthreads();
}

// One Kalman update of ctl->cur given the reduced sums (hth: k x k row-major in s.hth with stride k, htz in s.htz).
// sign=+1: LiDAR form (K1*HTz + vec - G*vec) ; sign=-1: visual form (-K1*HTz + vec - G*vec).
// Leaves solution in s.sol, K-gain blocks in s.G (also written to ctl->G).
__device__ inline void esikf_update_wave(DevCtl *ctl, SolveLds &s, int k, double meas_cov_scale, int sign, int lane) {
  // (P / scale)^-1 is iteration-invariant inside one update: cov is only rewritten when the update finishes.
  if (!ctl->hdr.pinv_valid) {
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) s.A[e] = ctl->cur.cov[e] / meas_cov_scale;
    __syncthreads();
    inverse19_wave(s.A, s.K1, s.perm, lane);
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) ctl->Pinv[e] = s.K1[e];
    __syncthreads();
    if (lane == 0) ctl->hdr.pinv_valid = 1;
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) { int r = e / DS, c = e % DS; s.A[e] = ((r < k && c < k) ? s.hth[r * k + c] : 0.0) + s.K1[e]; }
  } else {
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) { int r = e / DS, c = e % DS; s.A[e] = ((r < k && c < k) ? s.hth[r * k + c] : 0.0) + ctl->Pinv[e]; }
  }
  __syncthreads();
  inverse19_wave(s.A, s.K1, s.perm, lane);                 // K_1 = (H_T_H + P^-1)^-1
  for (int e = lane; e < DS * DS; e += LIVO2_WAVE) {       // G[:, :k] = K_1[:, :k] * H_T_H[:k,:k]
    int r = e / DS, c = e % DS;
    double g = 0.0;
    if (c < k) { g = s.K1[r * DS] * s.hth[c]; for (int j = 1; j < k; j++) g = g + s.K1[r * DS + j] * s.hth[j * k + c]; }
    s.G[e] = g; ctl->G[e] = g;
  }
  if (lane == 0) {                                         // vec = state_propagat [-] state   (common_lib.h:194-206)
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
  __syncthreads();
  if (lane < DS) {
    const int r = lane;
    double kz = s.K1[r * DS] * s.htz[0];
    for (int j = 1; j < k; j++) kz = kz + s.K1[r * DS + j] * s.htz[j];
    double gv = s.G[r * DS] * s.vec[0];
    for (int j = 1; j < k; j++) gv = gv + s.G[r * DS + j] * s.vec[j];
    s.sol[r] = ((sign > 0) ? kz : -kz) + s.vec[r] - gv;
  }
  __syncthreads();
  if (lane == 0) {                                         // state += solution   (common_lib.h:182-192)
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

// Deterministic reduction of per-block partial sums: partials[nblocks][32] -> out32 (LDS).  64 lanes = 2 slices x 32 values;
// slice s adds blocks s, s+2, ... in order, then slice0 + slice1.
__device__ inline void reduce_partials_wave(const double *partials, int nblocks, double *out32 /*LDS, 64 doubles scratch*/, int lane) {
  const int kidx = lane & 31, slice = lane >> 5;
  double acc = 0.0;
  for (int b = slice; b < nblocks; b += 2) acc += partials[(size_t)b * 32 + kidx];
  out32[lane] = acc;
  __syncthreads();
  if (lane < 32) out32[lane] = out32[lane] + out32[lane + 32];
  __syncthreads();
}
