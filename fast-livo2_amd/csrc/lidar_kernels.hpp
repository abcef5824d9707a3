// gfx950 kernels for the LiDAR point-to-plane ESIKF update.
//   k_body_cov        : once-per-scan precompute            reference src/voxel_map.cpp:15-34, 349-360
//   k_lidar_residual  : per-iteration fused pass            reference src/voxel_map.cpp:374-466, 513-530, 643-786
//                       (world transform -> world covariance -> voxel hash probe -> octree descent with 3-sigma gate and
//                        max-probability plane -> H / R^-1 / z row -> per-block partial sums of H^T R^-1 H and H^T R^-1 z)
//   k_lidar_solve     : partial-sum reduction + 19x19 solve + state update + convergence / rematch / covariance update
//                                                           reference src/voxel_map.cpp:464-499
// Layout: points SoA float x[],y[],z[] (coalesced 4-B/lane loads); body covariance SoA 6 x double[n];
// plane records 256-B aligned AoS gathered per lane with 16-B loads; hash slots 16 B.
#pragma once
#include "esikf_solve.hpp"

#define LIDAR_BLOCK 256
#define LIDAR_NSUM 29       // 21 (sym HtH) + 6 (Htz) + n_eff + sum|r|

struct LidarKernelArgs {
  const float *x, *y, *z;          // [n]
  const double *cb;                // [6][n] body covariance, symmetric (xx,xy,xz,yy,yz,zz)
  int32_t n;
  int32_t max_layer;
  DevMap map;
  double voxel_size, sigma_num;
  double ER[9], Et[3];
  // optional per-point outputs (device pointers or null)
  int32_t *match_plane; float *dis; float *pw; int32_t *normal_plane; double *var; double *r_inv; double *h_row;
};

// ---- once per scan: calcBodyCov -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_body_cov(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, int n,
                                                  float range_inc, float degree_inc, double deg2rad, double *__restrict__ cb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p0 = x[i], p1 = y[i], p2 = z[i];
  if (p2 == 0) p2 = 0.001;                                  // voxel_map.cpp:352 (calcBodyCov's own 0.0001 patch can then never fire)
  float range = (float)sqrt((p0 * p0 + p1 * p1) + p2 * p2); // float range (voxel_map.cpp:18)
  float range_var = range_inc * range_inc;
  double s = sin((double)degree_inc * deg2rad);
  double dv = s * s;
  double nrm = sqrt((p0 * p0 + p1 * p1) + p2 * p2);
  double d0 = p0 / nrm, d1 = p1 / nrm, d2 = p2 / nrm;
  double b10 = 1.0, b11 = 1.0, b12 = -(d0 + d1) / d2;
  double n1 = sqrt((b10 * b10 + b11 * b11) + b12 * b12);
  b10 /= n1; b11 /= n1; b12 /= n1;
  double b20 = b11 * d2 - b12 * d1, b21 = b12 * d0 - b10 * d2, b22 = b10 * d1 - b11 * d0;   // base_vector1 x direction
  double n2 = sqrt((b20 * b20 + b21 * b21) + b22 * b22);
  b20 /= n2; b21 /= n2; b22 /= n2;
  double r = (double)range;
  // direction_hat scaled by range, times N = [b1 b2]
  double h[9] = {r * 0.0, r * -d2, r * d1, r * d2, r * 0.0, r * -d0, r * -d1, r * d0, r * 0.0};
  double A[6];                                               // 3x2
#pragma unroll
  for (int row = 0; row < 3; row++) {
    A[row * 2 + 0] = (h[row * 3] * b10 + h[row * 3 + 1] * b11) + h[row * 3 + 2] * b12;
    A[row * 2 + 1] = (h[row * 3] * b20 + h[row * 3 + 1] * b21) + h[row * 3 + 2] * b22;
  }
  double dd[3] = {d0, d1, d2};
  double rv = (double)range_var;
  const int idx[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
#pragma unroll
  for (int e = 0; e < 6; e++) {
    int a = idx[e][0], b = idx[e][1];
    double t1 = (dd[a] * rv) * dd[b];
    double t2 = (A[a * 2] * dv) * A[b * 2] + (A[a * 2 + 1] * dv) * A[b * 2 + 1];
    cb[(size_t)e * n + i] = t1 + t2;
  }
}

// ---- per-candidate plane evaluation (build_single_residual's is_plane_ branch, voxel_map.cpp:721-768) -------------------
struct Candidate { double prob; int32_t plane; float r; bool success; };

__device__ __forceinline__ void eval_plane(const double *__restrict__ planes, int32_t pidx, double sigma_num, const double pw[3], const double Sw[6],
                                           Candidate &best) {
  const double *P = planes + (size_t)pidx * PLANE_REC_DOUBLES;
  const double2 *P2 = reinterpret_cast<const double2 *>(P);
  double2 v0 = P2[0], v1 = P2[1], v2 = P2[2];                 // n0 n1 | n2 c0 | c1 c2
  double n0 = v0.x, n1 = v0.y, n2 = v1.x, c0 = v1.y, c1 = v2.x, c2 = v2.y;
  float2 dr = *reinterpret_cast<const float2 *>(P + 27);      // {d_, radius_}
  double sd = ((n0 * pw[0] + n1 * pw[1]) + n2 * pw[2]) + (double)dr.x;
  float dis_to_plane = (float)fabs(sd);
  double e0 = c0 - pw[0], e1 = c1 - pw[1], e2 = c2 - pw[2];
  float dis_to_center = (float)((e0 * e0 + e1 * e1) + e2 * e2);
  float range_dis = sqrtf(dis_to_center - dis_to_plane * dis_to_plane);   // float ops; NaN (negative radicand) fails the gate below
  if (!((double)range_dis <= 3.0 * (double)dr.y)) return;
  // sigma_l = J_nq * plane_var * J_nq^T + n^T Sigma_w n,  J_nq = [p_w - c, -n]
  double J[6] = {-e0, -e1, -e2, -n0, -n1, -n2};
  double S[21];
  const double2 *S2 = reinterpret_cast<const double2 *>(P + 6);
#pragma unroll
  for (int q = 0; q < 10; q++) { double2 t = S2[q]; S[2 * q] = t.x; S[2 * q + 1] = t.y; }
  S[20] = P[26];
  double sigma_l = 0.0;
  {
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
      double rowacc = 0.0;
#pragma unroll
      for (int b = a; b < 6; b++) { double wgt = (a == b) ? 1.0 : 2.0; rowacc = fma(wgt * S[q], J[b], rowacc); q++; }
      sigma_l = fma(J[a], rowacc, sigma_l);
    }
  }
  double nSn = n0 * (Sw[0] * n0 + 2.0 * (Sw[1] * n1 + Sw[2] * n2)) + n1 * (Sw[3] * n1 + 2.0 * Sw[4] * n2) + n2 * Sw[5] * n2;
  sigma_l += nSn;
  double sq = sqrt(sigma_l);
  if ((double)dis_to_plane < sigma_num * sq) {
    best.success = true;
    double this_prob = 1.0 / sq * exp(-0.5 * (double)dis_to_plane * (double)dis_to_plane / sigma_l);
    if (this_prob > best.prob) { best.prob = this_prob; best.plane = pidx; best.r = (float)sd; }
  }
}

// recursive all-8-children descent of a NON-plane node, iteratively (voxel_map.cpp:769-785); depth <= LIVO2_MAX_LAYER
__device__ __forceinline__ void descend(const DevMap &map, int32_t node, int max_layer, double sigma_num, const double pw[3], const double Sw[6],
                                        Candidate &best) {
  // `node` is known to be a non-plane node at layer 0
  if (0 >= max_layer) return;
  int32_t stack_node[LIVO2_MAX_LAYER + 1];
  int32_t stack_next[LIVO2_MAX_LAYER + 1];
  int depth = 0;
  stack_node[0] = node; stack_next[0] = 0;
  while (depth >= 0) {
    int k = stack_next[depth];
    if (k >= 8) { depth--; continue; }
    stack_next[depth] = k + 1;
    int32_t child = map.node_child[(size_t)stack_node[depth] * 8 + k];
    if (child < 0) continue;
    int child_layer = depth + 1;
    int32_t pl = map.node_plane[child];
    if (pl >= 0) eval_plane(map.planes, pl, sigma_num, pw, Sw, best);
    else if (child_layer < max_layer) { depth++; stack_node[depth] = child; stack_next[depth] = 0; }
  }
}

// looks a voxel key up; returns slot index or -1
__device__ __forceinline__ int32_t hash_find(const DevMap &map, int32_t kx, int32_t ky, int32_t kz, int32_t &val) {
  uint32_t h = voxel_hash(kx, ky, kz) & map.hash_mask;
  for (uint32_t probe = 0; probe <= map.hash_mask; probe++) {
    HashSlot s = map.hash[h];
    if (s.val == -1) return -1;
    if (s.kx == kx && s.ky == ky && s.kz == kz) { val = s.val; return (int32_t)h; }
    h = (h + 1) & map.hash_mask;
  }
  return -1;
}

__device__ __forceinline__ void visit_root(const DevMap &map, int32_t val, int max_layer, double sigma_num, const double pw[3], const double Sw[6],
                                           Candidate &best) {
  if (val >= 0) eval_plane(map.planes, val, sigma_num, pw, Sw, best);
  else descend(map, -(val + 2), max_layer, sigma_num, pw, Sw, best);
}

// ---- fused per-iteration pass -----------------------------------------------------------------------------------------
// One thread per point.  blockIdx is remapped so that consecutive point chunks land on the same XCD (dispatcher places
// block b on XCD b % 8): neighbouring points share voxel planes, so each XCD's private 4-MiB L2 keeps one spatial slab.
__global__ void __launch_bounds__(LIDAR_BLOCK) k_lidar_residual(LidarKernelArgs a, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                                int check_stop) {
  if (check_stop && ctl->hdr.stop) return;
  const int per_xcd = gridDim.x >> 3;                            // host launches a multiple of 8 blocks
  const int vb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); // chunk index; chunks past the scan are empty
  const int tid = threadIdx.x;
  const int i = vb * LIDAR_BLOCK + tid;

  double acc[LIDAR_NSUM];
#pragma unroll
  for (int q = 0; q < LIDAR_NSUM; q++) acc[q] = 0.0;

  if (i < a.n) {
    // wave-uniform state (scalar loads)
    const double *R = ctl->cur.rot, *t = ctl->cur.pos, *Rp = ctl->prop.rot, *tp = ctl->prop.pos, *cov = ctl->cur.cov;
    const double plx = a.x[i], ply = a.y[i], plz = a.z[i];
    // p_i = extR * p_l + extT  (un-patched, voxel_map.cpp:522 / 418)
    double pi[3];
#pragma unroll
    for (int j = 0; j < 3; j++) pi[j] = ((a.ER[j * 3] * plx + a.ER[j * 3 + 1] * ply) + a.ER[j * 3 + 2] * plz) + a.Et[j];
    // p_w = float32( R * p_i + t )   (voxel_map.cpp:522-526)
    float pwf[3]; double pw[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { pwf[j] = (float)(((R[j * 3] * pi[0] + R[j * 3 + 1] * pi[1]) + R[j * 3 + 2] * pi[2]) + t[j]); pw[j] = (double)pwf[j]; }
    // cross matrix uses the z-patched point (voxel_map.cpp:352-358)
    double pc[3] = {pi[0], pi[1], pi[2]};
    if (plz == 0) {
      const double pz = 0.001;
#pragma unroll
      for (int j = 0; j < 3; j++) pc[j] = ((a.ER[j * 3] * plx + a.ER[j * 3 + 1] * ply) + a.ER[j * 3 + 2] * pz) + a.Et[j];
    }
    // body covariance (symmetric)
    double Cb[6];
#pragma unroll
    for (int e = 0; e < 6; e++) Cb[e] = a.cb[(size_t)e * a.n + i];
    const double Cbf[9] = {Cb[0], Cb[1], Cb[2], Cb[1], Cb[3], Cb[4], Cb[2], Cb[4], Cb[5]};
    // Sigma_w = R Cb R^T + X Prr X^T + Ptt   ((-X) Prr (-X)^T == X Prr X^T exactly)   (voxel_map.cpp:387)
    double Sw[6];
    {
      double T[9], RC[9];
      mat3_mul(R, Cbf, T);
      mat3_mul_Bt(T, R, RC);
      const double X[9] = {0.0, -pc[2], pc[1], pc[2], 0.0, -pc[0], -pc[1], pc[0], 0.0};
      const double Prr[9] = {cov[0], cov[1], cov[2], cov[DS], cov[DS + 1], cov[DS + 2], cov[2 * DS], cov[2 * DS + 1], cov[2 * DS + 2]};
      double XP[9], XPX[9];
      mat3_mul(X, Prr, XP);
      mat3_mul_Bt(XP, X, XPX);
      const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int e = 0; e < 6; e++) Sw[e] = RC[ii[e] * 3 + jj[e]] + XPX[ii[e] * 3 + jj[e]] + cov[(3 + ii[e]) * DS + 3 + jj[e]];
    }
    if (a.var) {
      const int map9[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
#pragma unroll
      for (int e = 0; e < 9; e++) a.var[(size_t)i * 9 + e] = Sw[map9[e]];
    }
    if (a.pw) { a.pw[(size_t)i * 3] = pwf[0]; a.pw[(size_t)i * 3 + 1] = pwf[1]; a.pw[(size_t)i * 3 + 2] = pwf[2]; }

    // voxel key (voxel_map.cpp:665-671): double divide, narrow to float, -1 for negatives, truncate
    float loc[3]; int32_t key[3]; bool in_range = true;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float l = (float)(pw[j] / a.voxel_size);
      if (l < 0) l = (float)((double)l - 1.0);
      loc[j] = l;
      in_range = in_range && (l > -2147483000.f) && (l < 2147483000.f);
      key[j] = (int32_t)l;
    }
    Candidate best; best.prob = 0.0; best.plane = -1; best.r = 0.f; best.success = false;
    if (in_range) {
      int32_t val = 0;
      int32_t slot = hash_find(a.map, key[0], key[1], key[2], val);
      if (slot >= 0) {
        visit_root(a.map, val, a.max_layer, a.sigma_num, pw, Sw, best);
        if (!best.success) {
          // neighbour rule (voxel_map.cpp:682-690): voxel-index units compared with metres, reproduced as is
          RootAux ra = a.map.root_aux[slot];
          int32_t nk[3] = {key[0], key[1], key[2]};
#pragma unroll
          for (int j = 0; j < 3; j++) {
            if ((double)loc[j] > (ra.center[j] + (double)ra.quarter)) nk[j] = nk[j] + 1;
            else if ((double)loc[j] < (ra.center[j] - (double)ra.quarter)) nk[j] = nk[j] - 1;
          }
          int32_t nval = 0;
          int32_t nslot = hash_find(a.map, nk[0], nk[1], nk[2], nval);
          if (nslot >= 0) visit_root(a.map, nval, a.max_layer, a.sigma_num, pw, Sw, best);
        }
      }
    }
    if (a.match_plane) a.match_plane[i] = best.success ? best.plane : -1;
    if (a.dis) a.dis[i] = best.success ? best.r : 0.f;
    double w_out = 0.0, h_out[6] = {0, 0, 0, 0, 0, 0};
    if (best.success) {
      if (a.normal_plane) a.normal_plane[i] = best.plane;
      // H / R^-1 / z row of the matched point (voxel_map.cpp:414-458)
      const double *P = a.map.planes + (size_t)best.plane * PLANE_REC_DOUBLES;
      const double n[3] = {P[0], P[1], P[2]}, c[3] = {P[3], P[4], P[5]};
      double q[3];                                              // PRIOR pose, un-rounded (voxel_map.cpp:425)
#pragma unroll
      for (int j = 0; j < 3; j++) q[j] = ((Rp[j * 3] * pi[0] + Rp[j * 3 + 1] * pi[1]) + Rp[j * 3 + 2] * pi[2]) + tp[j];
      const double J[6] = {q[0] - c[0], q[1] - c[1], q[2] - c[2], -n[0], -n[1], -n[2]};
      double sigma_l = 0.0;
      {
        int s = 0;
#pragma unroll
        for (int u = 0; u < 6; u++) {
          double rowacc = 0.0;
#pragma unroll
          for (int v = u; v < 6; v++) { double wgt = (u == v) ? 1.0 : 2.0; rowacc = fma(wgt * P[6 + s], J[v], rowacc); s++; }
          sigma_l = fma(J[u], rowacc, sigma_l);
        }
      }
      // var = (Rp*extR) Cb (Rp*extR)^T ; n^T var n = m^T Cb m with m = (Rp*extR)^T n     (voxel_map.cpp:445,449)
      double RE[9]; mat3_mul(Rp, a.ER, RE);
      double m[3]; mat3t_vec(RE, n, m);
      double nVn = m[0] * (Cb[0] * m[0] + 2.0 * (Cb[1] * m[1] + Cb[2] * m[2])) + m[1] * (Cb[3] * m[1] + 2.0 * Cb[4] * m[2]) + m[2] * Cb[5] * m[2];
      double w = 1.0 / (0.001 + sigma_l + nVn);
      // A = [p_i]x * R^T * n   (CURRENT rotation; un-patched p_i)   (voxel_map.cpp:453)
      double Rtn[3]; mat3t_vec(R, n, Rtn);
      double A0 = pi[1] * Rtn[2] - pi[2] * Rtn[1];              // skew(p_i) * v = p_i x v
      double A1 = pi[2] * Rtn[0] - pi[0] * Rtn[2];
      double A2 = pi[0] * Rtn[1] - pi[1] * Rtn[0];
      const double h[6] = {A0, A1, A2, n[0], n[1], n[2]};
      const double zz = -(double)best.r;                        // meas_vec(i) = -dis_to_plane_ (float32 residual, voxel_map.cpp:457)
      int s = 0;
#pragma unroll
      for (int u = 0; u < 6; u++) {
        double hw = h[u] * w;
#pragma unroll
        for (int v = u; v < 6; v++) { acc[s] = hw * h[v]; s++; }
        acc[21 + u] = hw * zz;
      }
      acc[27] = 1.0;
      acc[28] = fabs((double)best.r);
      w_out = w;
#pragma unroll
      for (int u = 0; u < 6; u++) h_out[u] = h[u];
    }
    if (a.r_inv) a.r_inv[i] = w_out;
    if (a.h_row) {
#pragma unroll
      for (int u = 0; u < 6; u++) a.h_row[(size_t)i * 6 + u] = h_out[u];
    }
  }

  // block reduction: wave butterflies, then 4 waves through LDS in fixed order (deterministic)
  __shared__ double red[LIDAR_BLOCK / LIVO2_WAVE][32];
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < LIDAR_NSUM; q++) {
    double v = wave_sum(acc[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid < 32) {
    double v = 0.0;
    if (tid < LIDAR_NSUM) v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    partials[(size_t)blockIdx.x * 32 + tid] = v;
  }
}

// ---- reduction + solve + loop control ----------------------------------------------------------------------------------
// mode 0: bare iterate (only reduce and publish sums_l) ; mode 1: full ESIKF iteration `iter` of `max_iter`;
// mode 2: like 1 but never stops (benchmark: fixed iteration count).
__global__ void __launch_bounds__(LIVO2_WAVE) k_lidar_solve(DevCtl *__restrict__ ctl, const double *__restrict__ partials, int nblocks, int mode, int iter,
                                                            int max_iter) {
  if (mode == 1 && ctl->hdr.stop) return;
  __shared__ SolveLds s;
  __shared__ double sums[64];
  const int lane = threadIdx.x;
  reduce_partials_wave(partials, nblocks, sums, lane);
  // expand symmetric 21 -> 6x6
  if (lane < 36) {
    int r = lane / 6, c = lane % 6;
    int u = r < c ? r : c, v = r < c ? c : r;
    int idx = u * 6 - (u * (u - 1)) / 2 + (v - u);
    s.hth[lane] = sums[idx];
  }
  if (lane < 6) s.htz[lane] = sums[21 + lane];
  __syncthreads();
  livo2_lidar_sums *out = (mode == 0) ? &ctl->sums_l : &ctl->lidar.iter_sums[iter];
  if (lane < 36) out->HtH[lane] = s.hth[lane];
  if (lane < 6) out->Htz[lane] = s.htz[lane];
  if (lane == 0) { out->total_residual = sums[28]; out->n_eff = (int32_t)sums[27]; out->pad = 0; }
  if (mode == 0) return;

  esikf_update_wave(ctl, s, 6, 1.0, +1, lane);
  if (lane < DS) ctl->lidar.iter_solution[iter][lane] = s.sol[lane];

  // convergence / rematch / covariance update (voxel_map.cpp:475-499)
  const double rn = sqrt((s.sol[0] * s.sol[0] + s.sol[1] * s.sol[1]) + s.sol[2] * s.sol[2]);
  const double tn = sqrt((s.sol[3] * s.sol[3] + s.sol[4] * s.sol[4]) + s.sol[5] * s.sol[5]);
  const bool conv = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
  int rematch = ctl->hdr.rematch_num;
  if (conv || ((rematch == 0) && (iter == (max_iter - 2)))) rematch++;
  const bool stop_now = (rematch >= 2 || (iter == max_iter - 1));
  __syncthreads();
  if (stop_now && mode == 1) {
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) s.cov[e] = ctl->cur.cov[e];
    __syncthreads();
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) {       // cov = (I - G) * cov
      int r = e / DS, c = e % DS;
      double v = (((r == 0) ? 1.0 : 0.0) - s.G[r * DS]) * s.cov[c];
      for (int k = 1; k < DS; k++) v = v + (((r == k) ? 1.0 : 0.0) - s.G[r * DS + k]) * s.cov[k * DS + c];
      ctl->cur.cov[e] = v;
    }
    if (lane < 3) ctl->lidar.position_last[lane] = ctl->cur.pos[lane];
  }
  if (lane == 0) {
    ctl->hdr.rematch_num = rematch;
    ctl->lidar.n_iters = iter + 1;
    ctl->lidar.converged = conv ? 1 : 0;
    if (stop_now && mode == 1) ctl->hdr.stop = 1;
  }
}

// copies the posterior into the result block after the loop (always runs)
__global__ void __launch_bounds__(LIVO2_WAVE) k_lidar_finish(DevCtl *__restrict__ ctl) {
  const int lane = threadIdx.x;
  const double *src = reinterpret_cast<const double *>(&ctl->cur);
  double *dst = reinterpret_cast<double *>(&ctl->lidar.state);
  for (int e = lane; e < (int)(sizeof(livo2_state) / sizeof(double)); e += LIVO2_WAVE) dst[e] = src[e];
}
