// IMU forward propagation: ImuProcess::UndistortPcl's forward loop, reference src/IMU_Processing.cpp:298-445 — per IMU
// sample: F_x and cov_w, cov <- F_x cov F_x^T + cov_w (19x19), attitude / position / velocity integration, one Pose6D per sample.
// The samples are sequential (each uses the previous attitude), so this is ONE block: thread (r, c) of the 19x19 products keeps the
// oracle's k-ascending dot products, thread 0 does the 3x3 algebra of a step.  It exists so that state_propagat, its covariance and the
// IMUpose list can be produced where the next stages (undistortion, update) consume them; as a stand-alone call it is latency-bound
// (~3 us per sample) and no faster than a host core.
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

__global__ void __launch_bounds__(IMU_THREADS) k_imu_propagate(ImuKernelArgs a) {
  __shared__ double P[DS * DS], F[DS * DS], T[DS * DS], W[DS * DS];
  __shared__ double R[9], pos[3], vel[3], bg[3], ba[3], grav[3];
  const int tid = threadIdx.x;
  const int r = tid / DS, c = tid % DS;
  const bool cell = tid < DS * DS;
  if (cell) P[tid] = a.in->cov[tid];
  if (tid < 9) R[tid] = a.in->rot[tid];
  if (tid < 3) { pos[tid] = a.in->pos[tid]; vel[tid] = a.in->vel[tid]; bg[tid] = a.in->bg[tid]; ba[tid] = a.in->ba[tid]; grav[tid] = a.in->grav[tid]; }
  __syncthreads();
  for (int i = 0; i < a.n; i++) {
    if (cell) { F[tid] = (r == c) ? 1.0 : 0.0; W[tid] = 0.0; }
    __syncthreads();
    if (tid == 0) {
      const double *s = a.steps + (size_t)i * 8;
      const double dt = s[6];
      double w[3], acc[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { w[k] = s[k] - bg[k]; acc[k] = (s[3 + k] * a.G_m_s2) / a.mean_acc_norm - ba[k]; }
      double Ef[9], Em[9];
      imu_exp(w, dt, Ef); imu_exp(w, -dt, Em);
      const double sk[9] = {0.0, -acc[2], acc[1], acc[2], 0.0, -acc[0], -acc[1], acc[0], 0.0};
      double nR[9], nRa[9];
#pragma unroll
      for (int k = 0; k < 9; k++) nR[k] = R[k] * (-1.0);
      mat3_mul(nR, sk, nRa);
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
          F[p * DS + q] = Em[p * 3 + q];
          if (a.ba_bg_est_en) F[p * DS + 10 + q] = ((p == q) ? -1.0 : -0.0) * dt;
          F[(3 + p) * DS + 7 + q] = ((p == q) ? 1.0 : 0.0) * dt;
          F[(7 + p) * DS + q] = nRa[p * 3 + q] * dt;
          if (a.ba_bg_est_en) F[(7 + p) * DS + 13 + q] = nR[p * 3 + q] * dt;
          if (a.gravity_est_en) F[(7 + p) * DS + 16 + q] = ((p == q) ? 1.0 : 0.0) * dt;
        }
      if (a.exposure_estimate_en) W[6 * DS + 6] = (a.cov_inv_expo * dt) * dt;
      double RD[9], Q[9];
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) RD[p * 3 + q] = (R[p * 3] * ((q == 0) ? a.cov_acc[0] : 0.0) + R[p * 3 + 1] * ((q == 1) ? a.cov_acc[1] : 0.0)) + R[p * 3 + 2] * ((q == 2) ? a.cov_acc[2] : 0.0);
      mat3_mul_Bt(RD, R, Q);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        W[k * DS + k] = (a.cov_gyr[k] * dt) * dt;
        W[(10 + k) * DS + 10 + k] = (a.cov_bias_gyr[k] * dt) * dt;
        W[(13 + k) * DS + 13 + k] = (a.cov_bias_acc[k] * dt) * dt;
#pragma unroll
        for (int q = 0; q < 3; q++) W[(7 + k) * DS + 7 + q] = (Q[k * 3 + q] * dt) * dt;
      }
      // attitude, specific acceleration, position, velocity (IMU_Processing.cpp:411-421)
      double Rn[9], ai[3];
      mat3_mul(R, Ef, Rn);
#pragma unroll
      for (int k = 0; k < 3; k++) ai[k] = ((Rn[k * 3] * acc[0] + Rn[k * 3 + 1] * acc[1]) + Rn[k * 3 + 2] * acc[2]) + grav[k];
      double *po = a.poses + (size_t)i * 22;
      po[0] = s[7];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double pn = (pos[k] + vel[k] * dt) + ((ai[k] * 0.5) * dt) * dt, vn = vel[k] + ai[k] * dt;
        pos[k] = pn; vel[k] = vn;
        po[1 + k] = ai[k]; po[4 + k] = w[k]; po[7 + k] = vn; po[10 + k] = pn;
      }
#pragma unroll
      for (int k = 0; k < 9; k++) { R[k] = Rn[k]; po[13 + k] = Rn[k]; }
    }
    __syncthreads();
    if (cell) {                                         // T = F P
      double sacc = F[r * DS] * P[c];
#pragma unroll
      for (int k = 1; k < DS; k++) sacc = sacc + F[r * DS + k] * P[k * DS + c];
      T[tid] = sacc;
    }
    __syncthreads();
    if (cell) {                                         // P = T F^T + W
      double sacc = T[r * DS] * F[c * DS];
#pragma unroll
      for (int k = 1; k < DS; k++) sacc = sacc + T[r * DS + k] * F[c * DS + k];
      P[tid] = sacc + W[tid];
    }
    __syncthreads();
  }
  if (cell) a.out->cov[tid] = P[tid];
  if (tid < 9) a.out->rot[tid] = R[tid];
  if (tid < 3) { a.out->pos[tid] = pos[tid]; a.out->vel[tid] = vel[tid]; a.out->bg[tid] = bg[tid]; a.out->ba[tid] = ba[tid]; a.out->grav[tid] = grav[tid]; }
  if (tid == 0) a.out->inv_expo = a.first_call ? 1.0 : a.in->inv_expo;      // tau (IMU_Processing.cpp:305-317, 444)
}
