// gfx950 kernels for the visual (direct photometric, 8x8 patch) ESIKF update, forward-compositional form.
//   k_visual_residual : per (level, iteration) fused pass     reference src/vio.cpp:1540-1636 (+189-201)
//   k_visual_solve    : reduction + accept/revert + 19x19 solve reference src/vio.cpp:1636-1685
//   k_visual_finish   : cov -= G*cov, T_f_w                    reference src/vio.cpp:800-801, 1690-1697
//
// Mapping: ONE wavefront per patch, one lane per pixel (lane = 8*x + y, x = patch row, y = patch column — the reference's
// loop order).  The (8+3)^2 strided u8 window is staged once in LDS as float, the 10x10 grid of bilinear samples B is
// built from it with the reference's float expression, and every lane then reads its 5 samples (centre, +-u, +-v).
// All rows of one patch share the patch-constant 2x6 matrix M with  row = [g*M, cur]  (g = image gradient scaled by
// inv_expo/scale), so H^T H and H^T z are accumulated as 10 per-patch moment sums (wave butterflies) and expanded with M
// once per patch:  sum J^T J = M^T (sum g g^T) M  etc. instead of a 64M x 7 dense product.
#pragma once
#include "esikf_solve.hpp"
#include <float.h>

#define VIS_BLOCK 512
#define VIS_WAVES (VIS_BLOCK / LIVO2_WAVE)
#define VIS_NSUM 37          // 28 (sym 7x7) + 7 + err_sum + n_meas
#define VIS_PSTRIDE 40

struct VisualKernelArgs {
  const uint8_t *img; int32_t width, height, stride;
  const double *pos; const float *warp; const int32_t *search_levels; const double *inv_expo;
  int32_t M, L, level, exposure_en;
  double fx, fy, cx, cy, d[5]; int32_t distortion; int32_t pad;
  double Rci[9], Pci[3], Jdp_dR[9];            // initializeVIO constants (vio.cpp:57-65), Jdphi_dR == Rci
  float *errors; double *z; double *H_sub;     // optional outputs (device pointers or null)
};

// Jacobian row chain of the reference for an image-gradient row g (vio.cpp:1611-1617)
__device__ __forceinline__ void jac_row(double g0, double g1, const double Jpi[6], const double pf[3], const double *Rci, const double *JdpdR,
                                        const double *Rcw, double out[6]) {
  // a = Jimg * Jdpi (1x3)
  double a0 = g0 * Jpi[0] + g1 * Jpi[3], a1 = g0 * Jpi[1] + g1 * Jpi[4], a2 = g0 * Jpi[2] + g1 * Jpi[5];
  // Jdphi = a * p_hat, p_hat = skew(pf)
  double ph[9] = {0.0, -pf[2], pf[1], pf[2], 0.0, -pf[0], -pf[1], pf[0], 0.0};
  double f[3], na[3] = {-a0, -a1, -a2};
#pragma unroll
  for (int j = 0; j < 3; j++) f[j] = (a0 * ph[j] + a1 * ph[3 + j]) + a2 * ph[6 + j];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double t1 = (f[0] * Rci[j] + f[1] * Rci[3 + j]) + f[2] * Rci[6 + j];            // Jdphi * Jdphi_dR
    double t2 = (na[0] * JdpdR[j] + na[1] * JdpdR[3 + j]) + na[2] * JdpdR[6 + j];   // Jdp * Jdp_dR
    out[j] = t1 + t2;
    out[3 + j] = (na[0] * Rcw[j] + na[1] * Rcw[3 + j]) + na[2] * Rcw[6 + j];        // Jdp * Jdp_dt
  }
}

template <bool DEBUG_ROWS>
__global__ void __launch_bounds__(VIS_BLOCK) k_visual_residual(VisualKernelArgs a, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                               int check_stop) {
  if (check_stop && ctl->hdr.stop) return;
  __shared__ float Wf[VIS_WAVES][11 * 11 + 3];
  __shared__ float Bf[VIS_WAVES][10 * 10 + 4];
  __shared__ double red[VIS_WAVES][VIS_PSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int patch = blockIdx.x * VIS_WAVES + wave;          // wave-uniform
  double out_val = 0.0;                                      // lane q < VIS_NSUM holds value q of this patch

  if (patch < a.M) {
    const double *Rwi = ctl->cur.rot, *Pwi = ctl->cur.pos;
    const double tau = ctl->cur.inv_expo;
    double Rcw[9], Pcw[3];
    mat3_mul_Bt(a.Rci, Rwi, Rcw);                            // Rcw = Rci * Rwi^T
#pragma unroll
    for (int j = 0; j < 3; j++) Pcw[j] = a.Pci[j] - ((Rcw[j * 3] * Pwi[0] + Rcw[j * 3 + 1] * Pwi[1]) + Rcw[j * 3 + 2] * Pwi[2]);
    const double p0 = a.pos[(size_t)patch * 3], p1 = a.pos[(size_t)patch * 3 + 1], p2 = a.pos[(size_t)patch * 3 + 2];
    double pf[3];
#pragma unroll
    for (int j = 0; j < 3; j++) pf[j] = ((Rcw[j * 3] * p0 + Rcw[j * 3 + 1] * p1) + Rcw[j * 3 + 2] * p2) + Pcw[j];
    // pc = cam->world2cam(pf)
    double pcx, pcy;
    {
      double u0 = pf[0] / pf[2], u1 = pf[1] / pf[2];
      if (!a.distortion) { pcx = a.fx * u0 + a.cx; pcy = a.fy * u1 + a.cy; }
      else {
        double x = u0, y = u1, r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + a.d[0] * r2 + a.d[1] * r4 + a.d[4] * r6;
        double xd = x * cdist + a.d[2] * a1 + a.d[3] * a2, yd = y * cdist + a.d[2] * a3 + a.d[3] * a1;
        pcx = xd * a.fx + a.cx; pcy = yd * a.fy + a.cy;
      }
    }
    // computeProjectionJacobian (vio.cpp:189-201)
    double Jpi[6];
    {
      const double z_inv = 1. / pf[2], z_inv_2 = z_inv * z_inv;
      Jpi[0] = a.fx * z_inv; Jpi[1] = 0.0; Jpi[2] = -a.fx * pf[0] * z_inv_2;
      Jpi[3] = 0.0; Jpi[4] = a.fy * z_inv; Jpi[5] = -a.fy * pf[1] * z_inv_2;
    }
    const int search_level = a.search_levels[patch];
    const int scale = 1 << (a.level + search_level);
    const float inv_scale = 1.0f / (float)scale;
    const float u_ref = (float)pcx, v_ref = (float)pcy;
    const int u_ref_i = (int)(floorf((float)(pcx / scale)) * (float)scale);
    const int v_ref_i = (int)(floorf((float)(pcy / scale)) * (float)scale);
    const float subpix_u = (u_ref - (float)u_ref_i) / (float)scale;
    const float subpix_v = (v_ref - (float)v_ref_i) / (float)scale;
    const float w_tl = (float)((1.0 - (double)subpix_u) * (1.0 - (double)subpix_v));
    const float w_tr = (float)((double)subpix_u * (1.0 - (double)subpix_v));
    const float w_bl = (float)((1.0 - (double)subpix_u) * (double)subpix_v);
    const float w_br = subpix_u * subpix_v;
    // the reference reads this window unchecked (vio.cpp:1595-1609); a window leaving the image is skipped here
    const bool inside = (u_ref_i - 5 * scale >= 0) && (u_ref_i + 5 * scale < a.width) && (v_ref_i - 5 * scale >= 0) && (v_ref_i + 5 * scale < a.height);
    if (inside) {
      // stage the 11x11 strided window as float
      for (int e = lane; e < 121; e += LIVO2_WAVE) {
        int wr = e / 11, wc = e - wr * 11;
        Wf[wave][e] = (float)a.img[(size_t)(v_ref_i + (wr - 5) * scale) * a.stride + (u_ref_i + (wc - 5) * scale)];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      for (int e = lane; e < 100; e += LIVO2_WAVE) {
        int br = e / 10, bc = e - br * 10;
        const float *w = &Wf[wave][br * 11 + bc];
        Bf[wave][e] = ((w_tl * w[0] + w_tr * w[1]) + w_bl * w[11]) + w_br * w[12];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int x = lane >> 3, y = lane & 7;
      const float *b = &Bf[wave][(x + 1) * 10 + (y + 1)];
      const float du = 0.5f * (b[1] - b[-1]);
      const float dv = 0.5f * (b[10] - b[-10]);
      const double cur = (double)b[0];
      const double g0 = ((double)du * tau) * (double)inv_scale;
      const double g1 = ((double)dv * tau) * (double)inv_scale;
      const double Pref = (double)a.warp[((size_t)patch * a.L + a.level) * 64 + lane];
      const double res = tau * cur - a.inv_expo[patch] * Pref;
      const double cexp = a.exposure_en ? cur : 0.0;
      if (DEBUG_ROWS) {
        double row[6];
        jac_row(g0, g1, Jpi, pf, a.Rci, a.Jdp_dR, Rcw, row);
        if (a.z) a.z[(size_t)patch * 64 + lane] = res;
        if (a.H_sub) {
          double *h = a.H_sub + ((size_t)patch * 64 + lane) * 7;
#pragma unroll
          for (int k = 0; k < 6; k++) h[k] = row[k];
          h[6] = cexp;
        }
      }
      // 10 moment sums over the 64 pixels
      const double Sg00 = wave_sum(g0 * g0), Sg01 = wave_sum(g0 * g1), Sg11 = wave_sum(g1 * g1);
      const double Sgc0 = wave_sum(g0 * cexp), Sgc1 = wave_sum(g1 * cexp), Scc = wave_sum(cexp * cexp);
      const double Sgr0 = wave_sum(g0 * res), Sgr1 = wave_sum(g1 * res), Scr = wave_sum(cexp * res), Srr = wave_sum(res * res);
      const float patch_error = (float)Srr;
      if (a.errors && lane == 0) a.errors[patch] = patch_error;
      // patch-constant M (2x6)
      double M0[6], M1[6];
      jac_row(1.0, 0.0, Jpi, pf, a.Rci, a.Jdp_dR, Rcw, M0);
      jac_row(0.0, 1.0, Jpi, pf, a.Rci, a.Jdp_dR, Rcw, M1);
      // lane q: value q of the 37-vector.  q<28: upper-tri (r<=c) of the 7x7 ; 28..34: Htz ; 35: err ; 36: n_meas
      if (lane < 28) {
        int r = 0, q = lane;
        while (q >= 7 - r) { q -= 7 - r; r++; }
        int c = r + q;
        double m0r = 0, m1r = 0, m0c = 0, m1c = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { if (k == r) { m0r = M0[k]; m1r = M1[k]; } if (k == c) { m0c = M0[k]; m1c = M1[k]; } }
        if (c < 6) out_val = (m0r * m0c) * Sg00 + (m0r * m1c + m1r * m0c) * Sg01 + (m1r * m1c) * Sg11;
        else if (r < 6) out_val = m0r * Sgc0 + m1r * Sgc1;
        else out_val = Scc;
      } else if (lane < 35) {
        int r = lane - 28;
        double m0r = 0, m1r = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k == r) { m0r = M0[k]; m1r = M1[k]; }
        out_val = (r < 6) ? (m0r * Sgr0 + m1r * Sgr1) : Scr;
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

// mode 0: bare evaluation -> ctl->sums_v ; mode 1: full ESIKF step (accept / revert / solve) ; mode 2: benchmark (always accept, never stop)
#define VIS_SOLVE_THREADS 480         // 12 slices x 40 values (<= 512 threads: the register-resident solve needs > 128 VGPRs)
__global__ void __launch_bounds__(512) k_visual_solve(DevCtl *__restrict__ ctl, const double *__restrict__ partials, int nblocks, int mode, int level,
                                                             int iter, double img_point_cov) {
  __shared__ SolveLds s;
  __shared__ double sums[64];
  __shared__ double scratch[12 * 41];
  // every global read of this kernel is issued here, in one batch: loop-control words, covariance + states, the partial rows
  const int hdr_stop = ctl->hdr.stop, hdr_steps = ctl->hdr.n_steps;
  const float hdr_last_error = ctl->hdr.last_error;
  double craw[6];
  if (mode != 0 && threadIdx.x < LIVO2_WAVE) esikf_prefetch_wave(ctl, s, img_point_cov, threadIdx.x, craw);
  if (mode != 0 && threadIdx.x == LIVO2_WAVE) esikf_log_lane(ctl, s);                   // second wave: overlaps the partial rows
  {
    // partial rows come from other CUs: every thread issues all its loads before the first add (slice s = rows s, s+12, ...)
    const int t = threadIdx.x, kidx = t % VIS_PSTRIDE, slice = t / VIS_PSTRIDE;     // 12 slices
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; u++) { const int b = slice + 12 * u; v[u] = (b < nblocks) ? partials[(size_t)b * VIS_PSTRIDE + kidx] : 0.0; }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 32; u++) acc += v[u];
    for (int b = slice + 384; b < nblocks; b += 12) acc += partials[(size_t)b * VIS_PSTRIDE + kidx];
    scratch[slice * 41 + kidx] = acc;
    __syncthreads();
    if (t < VIS_PSTRIDE) {
      double r = scratch[t];
#pragma unroll
      for (int sl = 1; sl < 12; sl++) r += scratch[sl * 41 + t];
      sums[t] = r;
    }
    __syncthreads();
  }
  if (mode == 1 && iter > 0 && hdr_stop) return;
  if (threadIdx.x >= LIVO2_WAVE) return;            // the 19-dim algebra is one wave; s_barrier only counts live waves
  const int lane = threadIdx.x;
  if (lane < 49) {
    int r = lane / 7, c = lane % 7;
    int u = r < c ? r : c, v = r < c ? c : r;
    int idx = u * 7 - (u * (u - 1)) / 2 + (v - u);
    s.hth[lane] = sums[idx];
  }
  if (lane < 7) s.htz[lane] = sums[28 + lane];
  __syncthreads();
  const double err_sum = sums[35];
  const int n_meas = (int)sums[36];
  float error = (float)err_sum;
  error = error / n_meas;                                   // float / int (vio.cpp:1636); NaN when n_meas == 0
  if (mode == 0) {
    if (lane < 49) ctl->sums_v.HtH[lane] = s.hth[lane];
    if (lane < 7) ctl->sums_v.Htz[lane] = s.htz[lane];
    if (lane == 0) { ctl->sums_v.err_sum = err_sum; ctl->sums_v.error = error; ctl->sums_v.n_meas = n_meas; }
    return;
  }
  const int nsl = sizeof(livo2_state) / sizeof(double);
  // level entry (iter == 0): old_state = *state, last_error = FLT_MAX (vio.cpp:1523,1528) — the first evaluation is always accepted
  // unless the error is NaN, in which case reverting to old_state == *state is a no-op.
  const float last_error = (iter == 0) ? FLT_MAX : hdr_last_error;
  const bool accepted = (mode == 2) ? true : (error <= last_error);
  const int step = hdr_steps;
  livo2_visual_step *st = (step < LIVO2_MAX_LEVELS * LIVO2_MAX_ITERS) ? &ctl->visual.steps[step] : nullptr;
  int stop = 0;
  if (accepted) {
    {                                                       // old_state = *state (vio.cpp:1650), from the prefetched copy
      double *dst = reinterpret_cast<double *>(&ctl->old);
      if (lane < 25) dst[lane] = s.cur[lane];
#pragma unroll
      for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; if (e < DS * DS) dst[25 + e] = craw[q]; }
    }
    esikf_update_wave<7>(ctl, s, -1, lane);
    const double rn = sqrt((s.sol[0] * s.sol[0] + s.sol[1] * s.sol[1]) + s.sol[2] * s.sol[2]);
    const double tn = sqrt((s.sol[3] * s.sol[3] + s.sol[4] * s.sol[4]) + s.sol[5] * s.sol[5]);
    if ((rn * (double)57.3f < (double)0.001f) && (tn * (double)100.0f < (double)0.001f)) stop = 1;   // vio.cpp:1675
    if (st) {
      if (lane < 49) st->HtH[lane] = s.hth[lane];
      if (lane < 7) st->Htz[lane] = s.htz[lane];
      if (lane < DS) st->solution[lane] = s.sol[lane];
    }
  } else {                                                  // (*state) = old_state ; EKF_end (vio.cpp:1679-1680)
    if (iter > 0) {
      const double *src = reinterpret_cast<const double *>(&ctl->old); double *dst = reinterpret_cast<double *>(&ctl->cur);
      for (int e = lane; e < nsl; e += LIVO2_WAVE) dst[e] = src[e];
    }
    stop = 1;
    if (st) {
      if (lane < 49) st->HtH[lane] = s.hth[lane];
      if (lane < 7) st->Htz[lane] = s.htz[lane];
      if (lane < DS) st->solution[lane] = 0.0;
    }
  }
  if (lane == 0) {
    if (accepted) ctl->hdr.last_error = error;
    if (mode == 1) ctl->hdr.stop = stop;
    if (st) { st->level = level; st->iteration = iter; st->accepted = accepted ? 1 : 0; st->n_meas = n_meas; st->error = error; st->pad = 0; }
    ctl->hdr.n_steps = step + 1;
  }
}

__global__ void __launch_bounds__(LIVO2_WAVE) k_visual_finish(DevCtl *__restrict__ ctl, VisualKernelArgs a, int update_cov) {
  __shared__ double cov[DS * DS];
  const int lane = threadIdx.x;
  if (update_cov) {
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) cov[e] = ctl->cur.cov[e];
    __syncthreads();
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) {      // state->cov -= G * state->cov (vio.cpp:800)
      int r = e / DS, c = e % DS;
      double g = ctl->G[r * DS] * cov[c];
      for (int k = 1; k < DS; k++) g = g + ctl->G[r * DS + k] * cov[k * DS + c];
      ctl->cur.cov[e] = cov[e] - g;
    }
    __syncthreads();
  }
  if (lane == 0) {                                          // updateFrameState (vio.cpp:1690-1697)
    double Rcw[9];
    mat3_mul_Bt(a.Rci, ctl->cur.rot, Rcw);
    for (int j = 0; j < 9; j++) ctl->visual.Rcw[j] = Rcw[j];
    for (int j = 0; j < 3; j++)
      ctl->visual.Pcw[j] = a.Pci[j] - ((Rcw[j * 3] * ctl->cur.pos[0] + Rcw[j * 3 + 1] * ctl->cur.pos[1]) + Rcw[j * 3 + 2] * ctl->cur.pos[2]);
    ctl->visual.n_steps = ctl->hdr.n_steps;
  }
  __syncthreads();
  const double *src = reinterpret_cast<const double *>(&ctl->cur);
  double *dst = reinterpret_cast<double *>(&ctl->visual.state);
  for (int e = lane; e < (int)(sizeof(livo2_state) / sizeof(double)); e += LIVO2_WAVE) dst[e] = src[e];
  for (int e = lane; e < DS * DS; e += LIVO2_WAVE) ctl->visual.G[e] = ctl->G[e];
}
