// IMU forward propagation: ImuProcess::UndistortPcl's forward loop, reference src/IMU_Processing.cpp:298-445 — per IMU
// sample: F_x and cov_w, cov <- F_x cov F_x^T + cov_w (19x19), attitude / position / velocity integration, one Pose6D per sample.
// It exists so that state_propagat, its covariance and the IMUpose list are produced where the next stages (undistortion, update) consume them.
// Round 5 (the first form ran 4.5 us per sample): only TWO things are sequential over the samples; the kernel runs those as chains and
// everything else across the samples of a chunk in parallel:
//   A  per sample (one lane each): bias-corrected rates, Exp(w, dt), Exp(w, -dt) — the gyro / accelerometer biases do not change during the propagation, so
//      the sin / cos of every sample are independent of the state;
//   B  the attitude / velocity / position recursion: ONE lane, state in registers (R <- R Exp_f, a = R acc + g, p, v), the Pose6D rows;
//   C  per sample (one lane each): the sample's F_x blocks (-R [acc]x dt, -R dt) and cov_w (R cov_acc R^T dt^2) from the attitude BEFORE the sample;
//   D  cov <- F_x cov F_x^T + cov_w: F_x is the identity plus four 3x3 blocks and a few diagonals (at most 8 non-zeros per row), so each of the 361 entries is an
//      8-term dot product instead of a 19-term one — same k-ascending order as the oracle's dense product with the exact zeros left out.
// Same operations on the same operands as before for everything that reaches the state and the poses; the covariance differs from the dense evaluation by the
// sign of exact zeros at most.  Measured (tools/imu_time.py, profiles/r05_imu_propagate.txt): 90.8 -> 32.6 us for 20 samples, 1.2 us per further sample, at the
// full 2.4 GHz (tools/imu_clock_probe.py) — ~1 500 cycles of B (a hundred dependent f64 operations and ~50 LDS instructions on one lane) + ~1 200 of D per sample.
// Two pipelined forms (B on a wave of its own in D's shadow; one covariance row per half-wave with a single barrier per sample) measured 31-40 us: B itself is the
// longer chain, so hiding D behind it buys nothing.  One host core needs 20.9 us for the same 20 samples: the row stays for residency, not for speed.
#pragma once
#include "livo2_device.hpp"

#define IMU_THREADS 384

struct ImuKernelArgs {
  const double *steps;          // [n][8] gyr3 acc3 dt offs_t
  int32_t n, ba_bg_est_en, gravity_est_en, exposure_estimate_en, first_call;
  double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3], cov_inv_expo, G_m_s2, mean_acc_norm;
  const livo2_state *in;
  livo2_state *out;
  double *poses;                // [n][22]
};

__device__ __forceinline__ void imu_exp(const double *w, double dt, double *E) {        // Exp(ang_vel, dt), so3_math.h:24-43
#pragma unroll
  for (int k = 0; k < 9; k++) E[k] = (k % 4 == 0) ? 1.0 : 0.0;
  const double nrm = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
  if (nrm > 0.0000001) {
    const double r[3] = {w[0] / nrm, w[1] / nrm, w[2] / nrm};
    const double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
    const double ang = nrm * dt, s = sin(ang), c1 = 1.0 - cos(ang);
    double Kc[9], KK[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Kc[k] = K[k] * c1;
    mat3_mul(Kc, K, KK);
#pragma unroll
    for (int k = 0; k < 9; k++) E[k] = (E[k] + K[k] * s) + KK[k];
  }
}

#define IMU_CHUNK 64             // samples per pass of A / B / C / D
#define IMU_NZ 8                 // non-zeros per row of F_x, padded

__global__ void __launch_bounds__(IMU_THREADS) k_imu_propagate(ImuKernelArgs a) {
  __shared__ double P[DS * DS], T[DS * DS];
  __shared__ double R[9], pos[3], vel[3], bg[3], ba[3], grav[3];
  __shared__ double sEf[IMU_CHUNK][9], sEm[IMU_CHUNK][9], sW[IMU_CHUNK][3], sA[IMU_CHUNK][3], sDt[IMU_CHUNK], sOff[IMU_CHUNK], sRpre[IMU_CHUNK][9], sQ[IMU_CHUNK][9];
  __shared__ double sVal[IMU_CHUNK][10][IMU_NZ];     // values of the non-zeros of rows 0..9 of F_x (rows 10..18 are rows of the identity)
  __shared__ int sIdx[10][IMU_NZ];                   // their columns, ascending; padding: column 0 with value 0 behind the real entries
  const int tid = threadIdx.x;
  const int r = tid / DS, c = tid % DS;
  const bool cell = tid < DS * DS;
  if (cell) P[tid] = a.in->cov[tid];
  if (tid < 9) R[tid] = a.in->rot[tid];
  if (tid < 3) { pos[tid] = a.in->pos[tid]; vel[tid] = a.in->vel[tid]; bg[tid] = a.in->bg[tid]; ba[tid] = a.in->ba[tid]; grav[tid] = a.in->grav[tid]; }
  if (tid < 10) {                                    // sparsity pattern of F_x (IMU_Processing.cpp:383-399): depends on the flags only
    int k = 0;
    int *ix = sIdx[tid];
    if (tid < 3) { ix[k++] = 0; ix[k++] = 1; ix[k++] = 2; if (a.ba_bg_est_en) ix[k++] = 10 + tid; }                                       // Exp(w, -dt) | -dt I
    else if (tid < 6) { ix[k++] = tid; ix[k++] = 4 + tid; }                                                                                // I | dt I (columns 7..9)
    else if (tid == 6) ix[k++] = 6;
    else { ix[k++] = 0; ix[k++] = 1; ix[k++] = 2; ix[k++] = tid; if (a.ba_bg_est_en) { ix[k++] = 13; ix[k++] = 14; ix[k++] = 15; } if (a.gravity_est_en) ix[k++] = 9 + tid; }   // -R [a]x dt | I | -R dt | dt I
    for (; k < IMU_NZ; k++) ix[k] = 0;
  }
  __syncthreads();
  int ixr[IMU_NZ], ixc[IMU_NZ];                      // non-zero columns of row r / row c of F_x: in registers for the whole kernel
#pragma unroll
  for (int k = 0; k < IMU_NZ; k++) { ixr[k] = (cell && r < 10) ? sIdx[r][k] : 0; ixc[k] = (cell && c < 10) ? sIdx[c][k] : 0; }
  for (int base = 0; base < a.n; base += IMU_CHUNK) {
    const int m = min(IMU_CHUNK, a.n - base);
    // ---- A: per sample, in parallel
    if (tid < m) {
      const double *s = a.steps + (size_t)(base + tid) * 8;
      const double dt = s[6];
      double w[3], acc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { w[k] = s[k] - bg[k]; acc[k] = (s[3 + k] * a.G_m_s2) / a.mean_acc_norm - ba[k]; }
      double Ef[9], Em[9];
      imu_exp(w, dt, Ef); imu_exp(w, -dt, Em);
#pragma unroll
      for (int k = 0; k < 9; k++) { sEf[tid][k] = Ef[k]; sEm[tid][k] = Em[k]; }
#pragma unroll
      for (int k = 0; k < 3; k++) { sW[tid][k] = w[k]; sA[tid][k] = acc[k]; }
      sDt[tid] = dt; sOff[tid] = s[7];
    }
    __syncthreads();
    // ---- B: the recursion over the samples, one lane (IMU_Processing.cpp:411-421, 433-441)
    if (tid == 0) {
      double Rc[9], pc[3], vc[3];
#pragma unroll
      for (int k = 0; k < 9; k++) Rc[k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; k++) { pc[k] = pos[k]; vc[k] = vel[k]; }
      const double g0 = grav[0], g1 = grav[1], g2 = grav[2];
      for (int i = 0; i < m; i++) {
#pragma unroll
        for (int k = 0; k < 9; k++) sRpre[i][k] = Rc[k];
        const double dt = sDt[i];
        double Rn[9], ai[3];
        mat3_mul(Rc, sEf[i], Rn);
        const double acc0 = sA[i][0], acc1 = sA[i][1], acc2 = sA[i][2];
        ai[0] = ((Rn[0] * acc0 + Rn[1] * acc1) + Rn[2] * acc2) + g0;
        ai[1] = ((Rn[3] * acc0 + Rn[4] * acc1) + Rn[5] * acc2) + g1;
        ai[2] = ((Rn[6] * acc0 + Rn[7] * acc1) + Rn[8] * acc2) + g2;
        double *po = a.poses + (size_t)(base + i) * 22;
        po[0] = sOff[i];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double pn = (pc[k] + vc[k] * dt) + ((ai[k] * 0.5) * dt) * dt, vn = vc[k] + ai[k] * dt;
          pc[k] = pn; vc[k] = vn;
          po[1 + k] = ai[k]; po[4 + k] = sW[i][k]; po[7 + k] = vn; po[10 + k] = pn;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) { Rc[k] = Rn[k]; po[13 + k] = Rn[k]; }
      }
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = Rc[k];
#pragma unroll
      for (int k = 0; k < 3; k++) { pos[k] = pc[k]; vel[k] = vc[k]; }
    }
    __syncthreads();
    // ---- C: per sample, in parallel: the values of F_x's non-zeros and the 3x3 block of cov_w, from the attitude BEFORE the sample
    if (tid < m) {
      const double dt = sDt[tid];
      const double *Rp = sRpre[tid], *acc = sA[tid];
      const double sk[9] = {0.0, -acc[2], acc[1], acc[2], 0.0, -acc[0], -acc[1], acc[0], 0.0};
      double nR[9], nRa[9];
#pragma unroll
      for (int k = 0; k < 9; k++) nR[k] = Rp[k] * (-1.0);
      mat3_mul(nR, sk, nRa);
      double (*v)[IMU_NZ] = sVal[tid];
#pragma unroll
      for (int p = 0; p < 3; p++) {
        int k = 0;
        v[p][k++] = sEm[tid][p * 3]; v[p][k++] = sEm[tid][p * 3 + 1]; v[p][k++] = sEm[tid][p * 3 + 2];
        if (a.ba_bg_est_en) v[p][k++] = -1.0 * dt;
        for (; k < IMU_NZ; k++) v[p][k] = 0.0;
        k = 0;
        v[3 + p][k++] = 1.0; v[3 + p][k++] = 1.0 * dt;
        for (; k < IMU_NZ; k++) v[3 + p][k] = 0.0;
        k = 0;
        v[7 + p][k++] = nRa[p * 3] * dt; v[7 + p][k++] = nRa[p * 3 + 1] * dt; v[7 + p][k++] = nRa[p * 3 + 2] * dt; v[7 + p][k++] = 1.0;
        if (a.ba_bg_est_en) { v[7 + p][k++] = nR[p * 3] * dt; v[7 + p][k++] = nR[p * 3 + 1] * dt; v[7 + p][k++] = nR[p * 3 + 2] * dt; }
        if (a.gravity_est_en) v[7 + p][k++] = 1.0 * dt;
        for (; k < IMU_NZ; k++) v[7 + p][k] = 0.0;
      }
      v[6][0] = 1.0;
      for (int k = 1; k < IMU_NZ; k++) v[6][k] = 0.0;
      double RD[9], Q[9];
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) RD[p * 3 + q] = (Rp[p * 3] * ((q == 0) ? a.cov_acc[0] : 0.0) + Rp[p * 3 + 1] * ((q == 1) ? a.cov_acc[1] : 0.0)) + Rp[p * 3 + 2] * ((q == 2) ? a.cov_acc[2] : 0.0);
      mat3_mul_Bt(RD, Rp, Q);
#pragma unroll
      for (int k = 0; k < 9; k++) sQ[tid][k] = (Q[k] * dt) * dt;
    }
    __syncthreads();
    // ---- D: the covariance recursion, 361 entries in parallel per sample
    for (int i = 0; i < m; i++) {
      if (cell) {                                         // T = F P
        double sacc;
        if (r >= 10) sacc = P[tid];
        else {
          const double *v = sVal[i][r];
          sacc = v[0] * P[ixr[0] * DS + c];
#pragma unroll
          for (int k = 1; k < IMU_NZ; k++) sacc = sacc + v[k] * P[ixr[k] * DS + c];
        }
        T[tid] = sacc;
      }
      __syncthreads();
      if (cell) {                                         // P = T F^T + W
        double sacc;
        if (c >= 10) sacc = T[tid];
        else {
          const double *v = sVal[i][c];
          sacc = T[r * DS + ixc[0]] * v[0];
#pragma unroll
          for (int k = 1; k < IMU_NZ; k++) sacc = sacc + T[r * DS + ixc[k]] * v[k];
        }
        const double dt = sDt[i];
        double wv = 0.0;                                  // cov_w (IMU_Processing.cpp:400-407)
        if (r == c) {
          if (r < 3) wv = (a.cov_gyr[r] * dt) * dt;
          else if (r == 6) wv = a.exposure_estimate_en ? (a.cov_inv_expo * dt) * dt : 0.0;
          else if (r >= 10 && r < 13) wv = (a.cov_bias_gyr[r - 10] * dt) * dt;
          else if (r >= 13 && r < 16) wv = (a.cov_bias_acc[r - 13] * dt) * dt;
        }
        if (r >= 7 && r < 10 && c >= 7 && c < 10) wv = sQ[i][(r - 7) * 3 + (c - 7)];
        P[tid] = sacc + wv;
      }
      __syncthreads();
    }
  }
  if (cell) a.out->cov[tid] = P[tid];
  if (tid < 9) a.out->rot[tid] = R[tid];
  if (tid < 3) { a.out->pos[tid] = pos[tid]; a.out->vel[tid] = vel[tid]; a.out->bg[tid] = bg[tid]; a.out->ba[tid] = ba[tid]; a.out->grav[tid] = grav[tid]; }
  if (tid == 0) a.out->inv_expo = a.first_call ? 1.0 : a.in->inv_expo;      // tau (IMU_Processing.cpp:305-317, 444)
}
