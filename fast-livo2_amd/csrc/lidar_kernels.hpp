// gfx950 kernels for the LiDAR point-to-plane ESIKF update.
//   k_body_cov        : once-per-scan precompute            reference src/voxel_map.cpp:15-34, 349-360
//   k_lidar_residual  : per-iteration fused pass            reference src/voxel_map.cpp:374-466, 513-530, 643-786
//                       (world transform -> voxel key -> cuckoo root lookup -> plane / candidate-list visit with radius gate,
//                        3-sigma gate and max-probability choice -> neighbour-voxel retry -> H / R^-1 / z row ->
//                        per-block partial sums of H^T R^-1 H and H^T R^-1 z)
//   k_lidar_solve     : partial-sum reduction + k x k solve + state update + convergence / rematch / covariance update + (stopping iteration) result block
//                                                           reference src/voxel_map.cpp:464-499
// Layout: points SoA float x[],y[],z[] (coalesced 4-B/lane loads); body covariance SoA 6 x double[n];
// root slots 64 B (two fetched per lookup), plane HOT records 128 B + a 16-B side word per lane (the 256-B master records are read by the map kernels only).
//
// The pass is LATENCY bound (17 000 points on an otherwise empty chip take 14 us, 200 000 take 24 us; rocprof shows waves parked in s_waitcnt),
// so it is organised as the shortest possible chain of dependent round trips:
//   T1 xyz + body covariance  ->  T2 both cuckoo slots  ->  T3 plane record  ||  both neighbour slots (speculative)
//   ->  T4 neighbour plane (only lanes whose first visit failed)  ->  reduction.
#pragma once
#include "esikf_solve.hpp"

#ifndef LIDAR_BLOCK
#define LIDAR_BLOCK 256          // threads (= points) per block of a single-scan launch (overridable: tools/block_probe.py)
#endif
#ifndef LIDAR_WPE
#define LIDAR_WPE 2             // waves per SIMD the single-scan kernel is compiled for (experiments: 3 = 168 VGPRs)
#endif
#define LIDAR_BLOCK_BATCH 64    // ... of a batched launch: single-wave blocks (17 KB LDS each) keep eight of them in flight per CU and make every barrier of
                                // the cooperative visit wave-local: 126 us (256) -> 109 us (128) -> 102 us (64) per 16 frames of 91k points
#define LIDAR_NSUM 29       // 21 (sym HtH) + 6 (Htz) + n_eff + sum|r|
#define LIDAR_LDS_BYTES_OF(B) (((B) / LIVO2_WAVE) * 32 * 65 * 8)
#define LIDAR_LDS_BYTES LIDAR_LDS_BYTES_OF(LIDAR_BLOCK)
#define LIDAR_LDS_DUMP 512          // behind the tiles: [0,256) landing area of the software-prefetch loads (touch_line), [256,352) sym(P_rr), sym(P_tt)
#define LIDAR_LDS_SP_OFF 256

struct LidarKernelArgs {
  const float *x, *y, *z;          // [n]
  const double *cb;                // [6][n] body covariance, symmetric (xx,xy,xz,yy,yz,zz)
  const int32_t *perm;             // [n] original index of the point stored at position i (scan is Morton-sorted at upload)
  int32_t n;
  int32_t max_layer;
  DevMap map;
  double voxel_size, sigma_num;
  double ER[9], Et[3];
  // optional per-point outputs (device pointers or null)
  int32_t *match_plane; float *dis; float *pw; int32_t *normal_plane; double *var; double *r_inv; double *h_row;
  // round 6 — fewer issued instructions per point (the pass is bound by them, profiles/r04_l2_retention_probe.txt):
  double inv_voxel_size;           // 1 / voxel_size when that is exact (voxel_size a power of two: every shipped LiDAR config but HILTI22's 0.4), else 0: the key needs a true division
  double sn2_lo, sn2_hi;           // sigma_num^2 (1 -+ 1e-14): the 3-sigma gate without its square root outside a 2e-14 band (sigma_gate_and_row)
#ifdef LIVO2_PHASE_PROF
  unsigned long long *prof;        // [waves][8] s_memtime stamps (profiling build only)
#endif
};

#ifdef LIVO2_PHASE_PROF
// drain every outstanding memory op, then stamp: phase k of this wave ends here
#define PHASE(k)                                                                                                   \
  do {                                                                                                             \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);           \
    if (a.prof && (threadIdx.x & 63) == 0) {                                                                       \
      a.prof[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter();              \
      if ((k) == 0 || (k) == 6) /* chip-wide 100 MHz clock: launch ramp and drain */                                \
        a.prof[((size_t)gridDim.x * 4 + 2) * 8 + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  } while (0)
#else
#define PHASE(k) do { } while (0)
#endif
#ifdef LIVO2_PHASE_PROF
// stamps inside the first round of the first cooperative visit (per wave): region behind the chip-wide clock stamps
#define COOP_PROF_PARAM , unsigned long long *cprof
#define COOP_PROF_ARG1 , a.prof
#define COOP_PROF_ARG2 , nullptr
#define CSTAMP(k)                                                                                                  \
  do {                                                                                                             \
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);           \
    if (cprof && base <= BLOCK && (threadIdx.x & 63) == 0)                                                    \
      cprof[((size_t)gridDim.x * 4 + 2) * 8 + (size_t)gridDim.x * 8 + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (base ? 6 : 0) + (k)] = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  } while (0)
#else
#define COOP_PROF_PARAM
#define COOP_PROF_ARG1
#define COOP_PROF_ARG2
#define CSTAMP(k) do { } while (0)
#endif

// ---- once per scan: spatial ordering --------------------------------------------------------------------------------------
// The residual pass gathers one plane record (a 128-B hot record + a 16-B side word) per point; with points in arbitrary order every lane of a wave touches
// different L2 lines (measured: 12 of 24 us).  Points are therefore re-ordered once per scan along a 30-bit Morton curve of their
// body-frame cell (cell = voxel_size): a rigid transform keeps neighbours neighbours, so a wave's lanes share planes and each
// XCD (consecutive chunks, see k_lidar_residual) keeps one spatial slab in its private L2.  Only the summation order changes.
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v &= 0x3ffu; v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__device__ __forceinline__ uint32_t morton_key_of(const float *__restrict__ p, float inv_cell) {
  uint32_t c[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float f = floorf(p[j] * inv_cell) + 512.f;
    f = fminf(fmaxf(f, 0.f), 1023.f);                          // NaN -> 0
    c[j] = (uint32_t)f;
  }
  return spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
}
__global__ void __launch_bounds__(256) k_morton_keys(const float *__restrict__ aos, int n, float inv_cell, uint32_t *__restrict__ keys, int32_t *__restrict__ idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = morton_key_of(aos + (size_t)i * 3, inv_cell);
  idx[i] = i;
}
__global__ void __launch_bounds__(256) k_gather_xyz(const float *__restrict__ aos, const int32_t *__restrict__ perm, int n, float *__restrict__ x,
                                                    float *__restrict__ y, float *__restrict__ z) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = perm[i];
  x[i] = aos[(size_t)o * 3]; y[i] = aos[(size_t)o * 3 + 1]; z[i] = aos[(size_t)o * 3 + 2];
}

// ---- once per scan: calcBodyCov -------------------------------------------------------------------------------------
// body covariance of one sensor-frame point, symmetric part {00, 01, 02, 11, 12, 22} (shared by k_body_cov and the single-launch scan preparation of frame_kernels.hpp)
__device__ __forceinline__ void body_cov_point(float fx, float fy, float fz, float range_inc, float degree_inc, double deg2rad, double (&out)[6]) {
  double p0 = fx, p1 = fy, p2 = fz;
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
    out[e] = t1 + t2;
  }
}
__global__ void __launch_bounds__(256) k_body_cov(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, int n,
                                                  float range_inc, float degree_inc, double deg2rad, double *__restrict__ cb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double c6[6];
  body_cov_point(x[i], y[i], z[i], range_inc, degree_inc, deg2rad, c6);
#pragma unroll
  for (int e = 0; e < 6; e++) cb[(size_t)e * n + i] = c6[e];
}

// ---- plane record in registers ------------------------------------------------------------------------------------------
struct PlaneRec {
  double n[3], c[3], See[6], v[3], k;
  float d, radius;
};

__device__ __forceinline__ void load_plane(const DevMap &map, int32_t pidx, PlaneRec &r) {
  const double2 *P2 = reinterpret_cast<const double2 *>(map.planes + (size_t)pidx * PLANE_HOT_DOUBLES);
  double2 v[8];
#pragma unroll
  for (int q = 0; q < 8; q++) v[q] = P2[q];                     // one batch of 16-B loads (one 128-B line) + the side word: a single round trip
  const float2 dr = *reinterpret_cast<const float2 *>(map.plane_aux + pidx);
  r.n[0] = v[0].x; r.n[1] = v[0].y; r.n[2] = v[1].x; r.c[0] = v[1].y; r.c[1] = v[2].x; r.c[2] = v[2].y;
#pragma unroll
  for (int q = 0; q < 3; q++) { r.See[2 * q] = v[3 + q].x; r.See[2 * q + 1] = v[3 + q].y; }
  r.v[0] = v[6].x; r.v[1] = v[6].y; r.v[2] = v[7].x; r.k = v[7].y;
  r.d = dr.x; r.radius = dr.y;
}

// J plane_var_ J^T for J = [a, -n]: a^T See a - 2 a^T v + k (livo2_device.hpp, PLANE_HOT_DOUBLES)
__device__ __forceinline__ double quad_plane(const double *See, const double *v, double k, const double *a) {
  const double av = fma(a[2], v[2], fma(a[1], v[1], a[0] * v[0]));
  return fma(-2.0, av, quad3_sym(See, a)) + k;
}

// Per-point invariants of the candidate search and of the Jacobian row — kept small on purpose: the kernel's occupancy is set by its VGPR count, so whatever can be
// re-derived in a few operations (the widened world point, the z-patched point, the prior-pose point q) is not held in registers across the visits.
struct PointCtx {
  float pwf[3];                // float32-rounded world point (voxel_map.cpp:524-526)
  double pi[3];                // un-patched IMU-frame point
  double Cb[6];                // body covariance, symmetric
  float plx, ply;              // sensor-frame x, y: only to rebuild the z-patched point when plz == 0 (voxel_map.cpp:352-358)
  bool zpatch;
};
// z-patched IMU-frame point (cross matrix of the matching covariance): p_i itself unless the sensor-frame z was exactly 0
__device__ __forceinline__ void point_pc(const PointCtx &pt, const double *ER, const double *Et, double pc[3]) {
  pc[0] = pt.pi[0]; pc[1] = pt.pi[1]; pc[2] = pt.pi[2];
  if (pt.zpatch) {
    const double plx = pt.plx, ply = pt.ply, pz = 0.001;
#pragma unroll
    for (int j = 0; j < 3; j++) pc[j] = ((ER[j * 3] * plx + ER[j * 3 + 1] * ply) + ER[j * 3 + 2] * pz) + Et[j];
  }
}

// State of the running max-probability search of one point (build_single_residual's in/out arguments).
// * The probability is evaluated LAZILY: the first accepted plane always wins against prob = 0 (exp(-0.5 d^2/sigma)/sqrt(sigma) > 0
//   because d < sigma_num*sqrt(sigma)), so exp/sqrt are only needed when a second plane is accepted for the same point.
// * The H / R^-1 row of an accepted plane is computed on the spot, while its record is still in registers, so the search keeps
//   8 doubles per point instead of a 29-double record copy (or a reload round trip at the end).
struct Best {
  double h[6];                 // Hsub row [A, n] of the best plane (voxel_map.cpp:453-454)
  double w;                    // R_inv (voxel_map.cpp:449)
  double dis2, sigma;          // of the best plane, for the lazy probability
  double prob;                 // valid iff prob_valid
  int32_t plane;
  float r;
  bool success, prob_valid;
};

// Radius gate of build_single_residual (voxel_map.cpp:724-730): needs only normal_, center_, d_, radius_ of the plane.
// All three float32 quantities use the reference's operand types and operation order (Q5).
struct GateOut { double sd; double e[3]; float dis_to_plane; };
__device__ __forceinline__ bool radius_gate(const double *n, const double *c, float d, float radius, const double *pw, GateOut &g) {
  g.sd = ((n[0] * pw[0] + n[1] * pw[1]) + n[2] * pw[2]) + (double)d;
  g.dis_to_plane = (float)fabs(g.sd);
  g.e[0] = c[0] - pw[0]; g.e[1] = c[1] - pw[1]; g.e[2] = c[2] - pw[2];
  const float dis_to_center = (float)((g.e[0] * g.e[0] + g.e[1] * g.e[1]) + g.e[2] * g.e[2]);
  const float range_dis = sqrtf(dis_to_center - g.dis_to_plane * g.dis_to_plane);   // float ops; NaN (negative radicand) fails the gate
  return (double)range_dis <= 3.0 * (double)radius;
}

// 3-sigma gate, max-probability choice (voxel_map.cpp:732-755) and — when the plane becomes the point's current best — its
// measurement row (voxel_map.cpp:425-457), for a plane that passed the radius gate.
// StateRefs: the wave-uniform operands of a plane evaluation.  R / RE / Rp / tp point into the control block (scalar loads), sP at the block's LDS copy of
// sym(P[0:3,0:3]) (6) and sym(P[3:6,3:6]) (6) — twelve broadcast LDS reads per evaluation instead of 24 VGPRs held for the whole kernel.
struct StateRefs {               // ONE base pointer: every operand is a scalar load at a constant offset from it (five separate pointers were five SGPR pairs held — and spilled — for the whole kernel)
  const DevCtl *ctl; const double *sP;
  __device__ __forceinline__ const double *R() const { return ctl->cur.rot; }
  __device__ __forceinline__ const double *RE() const { return ctl->hdr.RE; }
  __device__ __forceinline__ const double *Rp() const { return ctl->prop.rot; }
  __device__ __forceinline__ const double *tp() const { return ctl->prop.pos; }
};
__device__ __forceinline__ void sigma_gate_and_row(const double *n, const double *c, const double *See, const double *pv, double pk, const GateOut &g, int32_t pidx,
                                                   double sigma_num, double sn2_lo, double sn2_hi, const double *pc, const double *pi, const double *Cb, const StateRefs &st, Best &best) {
  // sigma_l = J_nq plane_var J_nq^T + n^T Sigma_w n ,  J_nq = [p_w - c, -n]
  const double aw[3] = {-g.e[0], -g.e[1], -g.e[2]};
  double sigma_l = quad_plane(See, pv, pk, aw);
  // n^T Sigma_w n = m^T Cb m + q^T Prr q + n^T Ptt n  with m = R^T n, q = n x p_i  (Sigma_w = R Cb R^T + X Prr X^T + Ptt, X = [p_i]x)
  double m[3]; mat3t_vec_fma(st.R(), n, m);
  const double qx[3] = {n[1] * pc[2] - n[2] * pc[1], n[2] * pc[0] - n[0] * pc[2], n[0] * pc[1] - n[1] * pc[0]};
  {
    double sPrr[6], sPtt[6];
#pragma unroll
    for (int e = 0; e < 6; e++) { sPrr[e] = st.sP[e]; sPtt[e] = st.sP[6 + e]; }
    sigma_l += (quad3_sym(Cb, m) + quad3_sym(sPrr, qx)) + quad3_sym(sPtt, n);
  }
  // 3-sigma gate  dis < sigma_num * sqrt(sigma_l)  (voxel_map.cpp:737).  The square root (a 16-cycle v_rsq_f64 + ~10 dependent f64 operations) only decides
  // when dis^2 lies within 1e-14 (relative) of sigma_num^2 sigma_l: both roundings of the reference's right-hand side are 2^-53 each and the three of the squared
  // form another 3 x 2^-53, so outside the band the two forms agree by construction; inside it (and for NaN / negative sigma_l, which fail both fast tests) the
  // reference's expression is evaluated as written.
  const double dd = (double)g.dis_to_plane, dis2 = dd * dd;
  bool pass = dis2 < sn2_lo * sigma_l;
  if (!pass && !(dis2 > sn2_hi * sigma_l)) pass = dd < sigma_num * sqrt(sigma_l);
  if (pass) {
    bool take = true;
    if (best.success) {
      if (!best.prob_valid) { best.prob = 1.0 / sqrt(best.sigma) * exp(-0.5 * best.dis2 / best.sigma); best.prob_valid = true; }
      const double this_prob = 1.0 / sqrt(sigma_l) * exp(-0.5 * dis2 / sigma_l);
      take = this_prob > best.prob;
      if (take) best.prob = this_prob;
    }
    best.success = true;
    if (take) {
      best.plane = pidx; best.r = (float)g.sd; best.dis2 = dis2; best.sigma = sigma_l;
      // H / R^-1 row (voxel_map.cpp:414-458): sigma_l' at the PRIOR-pose point, var with the PRIOR rotation, A with the CURRENT one
      // (R^-1 depends on the prior pose only; keeping it per point from one iteration to the next — 12 B per point read in T1, written on change — was built and
      //  measured: +1.0 us per iteration, profiles/r06_lidar_instruction_ab.txt.  Recomputed.)
      {
        double q[3];                                          // PRIOR-pose world point R^ p_i + t^, un-rounded (voxel_map.cpp:425)
#pragma unroll
        for (int j = 0; j < 3; j++) q[j] = ((st.Rp()[j * 3] * pi[0] + st.Rp()[j * 3 + 1] * pi[1]) + st.Rp()[j * 3 + 2] * pi[2]) + st.tp()[j];
        const double aq[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
        const double sig_q = quad_plane(See, pv, pk, aq);
        double mp[3]; mat3t_vec_fma(st.RE(), n, mp);            // (R^ extR)^T n ; n^T var n = mp^T Cb mp   (voxel_map.cpp:445,449)
        best.w = 1.0 / (0.001 + sig_q + quad3_sym(Cb, mp));
      }
      best.h[0] = pi[1] * m[2] - pi[2] * m[1];             // A = [p_i]x R^T n = p_i x (R^T n)   (voxel_map.cpp:453)
      best.h[1] = pi[2] * m[0] - pi[0] * m[2];
      best.h[2] = pi[0] * m[1] - pi[1] * m[0];
      best.h[3] = n[0]; best.h[4] = n[1]; best.h[5] = n[2];
    }
  }
}

__device__ __forceinline__ bool slot_match(const RootSlot &s, const int32_t key[3]) { return s.val != -1 && s.kx == key[0] && s.ky == key[1] && s.kz == key[2]; }

__device__ __forceinline__ RootSlot load_slot(const RootSlot *__restrict__ slots, uint32_t h) {
  const int4 *p = reinterpret_cast<const int4 *>(slots + h);
  int4 a = p[0], b = p[1], c = p[2], d = p[3];
  RootSlot s;
  s.kx = a.x; s.ky = a.y; s.kz = a.z; s.val = a.w;
  s.center[0] = __builtin_bit_cast(double, make_int2(b.x, b.y)); s.center[1] = __builtin_bit_cast(double, make_int2(b.z, b.w));
  s.center[2] = __builtin_bit_cast(double, make_int2(c.x, c.y)); s.quarter = __builtin_bit_cast(float, c.z); s.cand_begin = c.w;
  s.cand_count = d.x; s.pad = d.y;
  return s;
}

// the part of a root slot a NEIGHBOUR lookup needs (its centre / quarter are never read): 24 B instead of the whole 64-B slot in registers
struct SlotHead { int32_t kx, ky, kz, val, cand_begin, cand_count; };
__device__ __forceinline__ SlotHead load_slot_head(const RootSlot *__restrict__ slots, uint32_t h) {
  const int4 *p = reinterpret_cast<const int4 *>(slots + h);
  const int4 a = p[0];
  const int32_t *w = reinterpret_cast<const int32_t *>(slots + h);
  return {a.x, a.y, a.z, a.w, w[11], w[12]};                 // cand_begin (word 11), cand_count (word 12)
}
__device__ __forceinline__ bool head_match(const SlotHead &s, const int32_t key[3]) { return s.val != -1 && s.kx == key[0] && s.ky == key[1] && s.kz == key[2]; }

// Visit one root voxel (build_single_residual from layer 0).
//  * plane root: the whole hot record + side word in one batch, radius gate, 3-sigma gate (visit_plane_root).
//  * non-plane root: the depth-first list of descendant planes (layers <= max_layer) was flattened at upload into contiguous copies
//    of their hot records (side word: d_, radius_, plane index | layer); the block evaluates all such (point, candidate) pairs cooperatively
//    (coop_plan / coop_run below), in depth-first order so that ties keep the first.
struct RootRef { int32_t val, cand_begin, cand_count; };   // what a visit needs from a RootSlot

struct LidarKernelArgs;
struct GateConsts {                // the gate's three launch constants, read where they are used (scalar loads from the argument block: no SGPRs held across the body)
  const double *p;                 // -> {sigma_num, ..., sn2_lo, sn2_hi} inside LidarKernelArgs
  int off_lo, off_hi;
  __device__ __forceinline__ double sigma_num() const { return p[0]; }
  __device__ __forceinline__ double sn2_lo() const { return p[off_lo]; }
  __device__ __forceinline__ double sn2_hi() const { return p[off_hi]; }
};
__device__ __forceinline__ void visit_plane_root(const PlaneRec &p, int32_t pidx, const GateConsts &gc, const PointCtx &pt, const double *ER, const double *Et,
                                                 const StateRefs &st, Best &best) {
  GateOut g;
  const double pw[3] = {(double)pt.pwf[0], (double)pt.pwf[1], (double)pt.pwf[2]};
  if (radius_gate(p.n, p.c, p.d, p.radius, pw, g)) {
    double pc[3]; point_pc(pt, ER, Et, pc);
    sigma_gate_and_row(p.n, p.c, p.See, p.v, p.k, g, pidx, gc.sigma_num(), gc.sn2_lo(), gc.sn2_hi(), pc, pt.pi, pt.Cb, st, best);
  }
}

// Software prefetch (gfx950 has no prefetch instruction): a one-dword LDS-direct load (no destination VGPR) into a dump area
// behind the block's tiles.  Nothing reads the dump; the point is that the line is now on its way to this CU's L2/L1.
template <int BLOCK> __device__ __forceinline__ void touch_line(const void *p) {
  uint32_t keep_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep_m0) : "v"(p), "s"((uint32_t)LIDAR_LDS_BYTES_OF(BLOCK)) : "memory");
}

// ---- block-cooperative visit of non-plane roots ----------------------------------------------------------------------------
// A non-plane root owns a flattened depth-first list of descendant planes.  Walking it inside the owning lane serialises up to 73
// plane evaluations while the other lanes idle, and the kernel ends with its slowest wave (measured tail: 14 us against a 2.6 us
// median; the Morton order concentrates cluttered voxels in the same wave).  Instead every (point, candidate) pair of the BLOCK
// becomes one work item: owners publish their point context and their pair range in LDS, each of the 256 threads evaluates one
// pair per round (hot record + side word from the candidate-ordered arrays: one round trip -> radius gate -> 3-sigma gate ->
// probability and measurement row), and the owner folds the results of its pairs IN LIST ORDER with the reference's strict '>'
// (ties keep the first), which reproduces the serial recursion of build_single_residual exactly.
template <int BLOCK> struct __attribute__((aligned(16))) CoopLds {
  double ctx[BLOCK][15];       // owner context: pw pc pi (3 each) Cb (6)
  double res[BLOCK][10];       // result rows of accepted pairs: prob, w, h[6], {float r, int32 plane}
  unsigned long long omax[BLOCK];                // per owner: largest accepted probability of the round (bit pattern)
  int32_t omin[BLOCK];                           // per owner: first slot holding it
  int32_t pair_cand[2 * BLOCK];                  // per slot of the round: candidate record
  uint8_t pair_owner[2 * BLOCK];                 //                        owning thread
  uint8_t slot_row[2 * BLOCK];                   //                        result row of an accepted pair
  uint8_t oacc[BLOCK];                           // per owner: some pair of the round was accepted
  int32_t wave_tot[2][BLOCK / LIVO2_WAVE];       // one set per visit (first / neighbour): no barrier separates their plans
  int32_t res_count, overflow;                         // double rounds: rows handed out, more than BLOCK accepted pairs
};
static_assert(sizeof(CoopLds<256>) <= LIDAR_LDS_BYTES_OF(256) && sizeof(CoopLds<128>) <= LIDAR_LDS_BYTES_OF(128) && sizeof(CoopLds<64>) <= LIDAR_LDS_BYTES_OF(64),
              "CoopLds must fit in the block's reduction tiles (pair_owner / slot_row are bytes: BLOCK <= 256)");

// Planning half (every thread of the block calls it; cnt = 0 for threads without a pending candidate list): the pair counts are
// prefix-summed over the block.  One barrier, LDS only — global loads issued before it (the plane
// record of a plane-root lane) stay in flight across it, which is why the plan is made BEFORE those records are consumed.
struct CoopPlan { int cnt, cand_begin, excl, W; };
template <int BLOCK> __device__ __forceinline__ CoopPlan coop_plan(CoopLds<BLOCK> &L, int which, int cnt, int cand_begin) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int incl = cnt;                                            // inclusive prefix sum of the pair counts: wave scan + wave totals
#pragma unroll
  for (int off = 1; off < LIVO2_WAVE; off <<= 1) { const int t = __shfl_up(incl, off, LIVO2_WAVE); if (lane >= off) incl += t; }
  if (lane == LIVO2_WAVE - 1) L.wave_tot[which][wave] = incl;
  __syncthreads();
  int woff = 0, W = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / LIVO2_WAVE; w++) { const int t = L.wave_tot[which][w]; if (w < wave) woff += t; W += t; }
  return {cnt, cand_begin, woff + incl - cnt, W};
}

// Every thread parks its point context in LDS before the first plan: evaluators read the owners' rows, and the thread itself
// re-reads its own row after the evaluation instead of holding 36 VGPRs across it (the evaluation is the register peak).
template <int BLOCK> __device__ __forceinline__ void coop_park_ctx(CoopLds<BLOCK> &L, const PointCtx &pt, const double *ER, const double *Et) {
  double *c = L.ctx[threadIdx.x];
  double pc[3]; point_pc(pt, ER, Et, pc);
#pragma unroll
  for (int k = 0; k < 3; k++) { c[k] = (double)pt.pwf[k]; c[3 + k] = pc[k]; c[6 + k] = pt.pi[k]; }
#pragma unroll
  for (int k = 0; k < 6; k++) c[9 + k] = pt.Cb[k];
}
template <int BLOCK> __device__ __forceinline__ void coop_unpark_ctx(const CoopLds<BLOCK> &L, PointCtx &pt) {
  const double *c = L.ctx[threadIdx.x];
#pragma unroll
  for (int k = 0; k < 3; k++) { pt.pwf[k] = (float)c[k]; pt.pi[k] = c[6 + k]; }      // (float)(double)f == f
#pragma unroll
  for (int k = 0; k < 6; k++) pt.Cb[k] = c[9 + k];
}

// One round of the evaluation half over the slots [base, base + PAIRS * 256) of the block's pair list; every thread evaluates PAIRS pairs.
//  PAIRS == 1: hot record + side word in one batch of loads (one round trip), as everywhere else in this kernel.
//  PAIRS == 2 (blocks with more than 256 pairs left — the slowest blocks of a launch): gate first.  Both pairs' gate words
//    {normal_, center_} (the first 48 B of the hot record's line, so the whole record is on its way to this CU) and {d_, radius_, meta} (the side word) are
//    fetched together, the radius gate runs on both, and the covariance part is read — now an L1/L2 hit — only for a pair
//    that passed: one cold round trip and one set of barriers for up to 512 pairs instead of two of each.  Accepted pairs take
//    result rows from an LDS counter; should more than 256 of them turn up the round reports failure and the caller repeats the
//    slots with PAIRS == 1.
// Fold = the reference's serial "first plane with the strictly largest probability" over each owner's list: the evaluators elect
// the winner — LDS max over the probability bit patterns (accepted probabilities are non-negative, so the patterns order like the
// values), then LDS min over the slots holding that maximum (ascending slot = list order -> the first one) — and the owner copies
// one row (walking its accepted rows cost a dependent LDS round trip per accepted plane, ~1.8 us in cluttered blocks).
template <int PAIRS, int BLOCK> __device__ __forceinline__ bool coop_round(CoopLds<BLOCK> &L, const DevMap &map, const CoopPlan &pl, int base, int max_layer, const GateConsts &gc,
                                                              const StateRefs &st, Best &best COOP_PROF_PARAM) {
  const int tid = threadIdx.x;
  const int cnt = pl.cnt, cand_begin = pl.cand_begin, excl = pl.excl, W = pl.W;
  constexpr int SPAN = PAIRS * BLOCK;
  CSTAMP(0);
  // owners publish the pairs that fall into this round
  const int k_lo = max(0, base - excl), k_hi = min(cnt, base + SPAN - excl);
  for (int k = k_lo; k < k_hi; k++) { const int j = excl + k - base; L.pair_owner[j] = (uint8_t)tid; L.pair_cand[j] = cand_begin + k; }
  if (k_lo < k_hi) { L.omax[tid] = 0ull; L.omin[tid] = 0x7fffffff; L.oacc[tid] = 0; }
  if (PAIRS > 1 && tid == 0) { L.res_count = 0; L.overflow = 0; }
  __syncthreads();
  CSTAMP(1);
  bool act[PAIRS], acc[PAIRS];
  int owner[PAIRS];
  unsigned long long pbits[PAIRS];
  double n[PAIRS][3], c[PAIRS][3];
  float d[PAIRS], radius[PAIRS];
  int meta[PAIRS];
  const double2 *P2[PAIRS];
  const PlaneAux *PA[PAIRS];
#pragma unroll
  for (int q = 0; q < PAIRS; q++) {
    const int s = tid + q * BLOCK;
    act[q] = base + s < W; acc[q] = false; owner[q] = 0; pbits[q] = 0ull;
    const size_t ci = (size_t)(act[q] ? L.pair_cand[s] : 0);
    P2[q] = reinterpret_cast<const double2 *>(map.cand_rec + ci * PLANE_HOT_DOUBLES);
    PA[q] = map.cand_aux + ci;
    if (act[q]) owner[q] = L.pair_owner[s];
  }
  double2 g0[PAIRS], g1[PAIRS], g2[PAIRS];
  int4 ga[PAIRS];
  double2 sv[5];                                                    // covariance words 6..15 of ONE record at a time
  if (PAIRS == 1) {
    if (act[0]) {
      g0[0] = P2[0][0]; g1[0] = P2[0][1]; g2[0] = P2[0][2];
#pragma unroll
      for (int w = 0; w < 5; w++) sv[w] = P2[0][3 + w];
      ga[0] = *reinterpret_cast<const int4 *>(PA[0]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < PAIRS; q++)
      if (act[q]) { g0[q] = P2[q][0]; g1[q] = P2[q][1]; g2[q] = P2[q][2]; ga[q] = *reinterpret_cast<const int4 *>(PA[q]); }
  }
  CSTAMP(2);
  GateOut g[PAIRS];
  bool pass[PAIRS];
#pragma unroll
  for (int q = 0; q < PAIRS; q++) {
    pass[q] = false;
    if (act[q]) {
      n[q][0] = g0[q].x; n[q][1] = g0[q].y; n[q][2] = g1[q].x; c[q][0] = g1[q].y; c[q][1] = g2[q].x; c[q][2] = g2[q].y;
      d[q] = __builtin_bit_cast(float, ga[q].x); radius[q] = __builtin_bit_cast(float, ga[q].y);
      meta[q] = ga[q].z;
      if ((meta[q] >> CAND_LAYER_SHIFT) <= max_layer) {
        const double *oc = L.ctx[owner[q]];
        const double pw[3] = {oc[0], oc[1], oc[2]};
        pass[q] = radius_gate(n[q], c[q], d[q], radius[q], pw, g[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PAIRS; q++) {
    if (!pass[q]) continue;
    if (PAIRS > 1) {
#pragma unroll
      for (int w = 0; w < 5; w++) sv[w] = P2[q][3 + w];
    }
    const double See[6] = {sv[0].x, sv[0].y, sv[1].x, sv[1].y, sv[2].x, sv[2].y};
    const double pv[3] = {sv[3].x, sv[3].y, sv[4].x};
    const double pk = sv[4].y;
    const double *oc = L.ctx[owner[q]];
    double ppc[3], ppi[3], pCb[6];
#pragma unroll
    for (int k = 0; k < 3; k++) { ppc[k] = oc[3 + k]; ppi[k] = oc[6 + k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) pCb[k] = oc[9 + k];
    Best tb; tb.success = false; tb.prob_valid = false; tb.prob = 0.0; tb.dis2 = 0.0; tb.sigma = 1.0; tb.w = 0.0; tb.plane = -1; tb.r = 0.f;
#pragma unroll
    for (int u = 0; u < 6; u++) tb.h[u] = 0.0;
    sigma_gate_and_row(n[q], c[q], See, pv, pk, g[q], meta[q] & CAND_PLANE_MASK, gc.sigma_num(), gc.sn2_lo(), gc.sn2_hi(), ppc, ppi, pCb, st, tb);
    if (tb.success) {
      const double prob = 1.0 / sqrt(tb.sigma) * exp(-0.5 * tb.dis2 / tb.sigma);
      int row = tid;
      if (PAIRS > 1) row = atomicAdd(&L.res_count, 1);
      if (row < BLOCK) {
        acc[q] = true;
        pbits[q] = __builtin_bit_cast(unsigned long long, prob);
        L.slot_row[tid + q * BLOCK] = (uint8_t)row;
        L.oacc[owner[q]] = 1;
        atomicMax(&L.omax[owner[q]], pbits[q]);
        double *o = L.res[row];
        o[0] = prob; o[1] = tb.w;
#pragma unroll
        for (int u = 0; u < 6; u++) o[2 + u] = tb.h[u];
        o[8] = __builtin_bit_cast(double, make_int2(__builtin_bit_cast(int, tb.r), tb.plane));
      } else L.overflow = 1;
    }
  }
  __syncthreads();
  CSTAMP(4);
  if (PAIRS > 1 && L.overflow) { __syncthreads(); return false; }   // (block-uniform; the barrier keeps the flag readable until everyone has seen it)
#pragma unroll
  for (int q = 0; q < PAIRS; q++)
    if (acc[q] && L.omax[owner[q]] == pbits[q]) atomicMin(&L.omin[owner[q]], tid + q * BLOCK);
  __syncthreads();
  if (k_lo < k_hi && L.oacc[tid]) {
    best.success = true;
    const double mp = __builtin_bit_cast(double, L.omax[tid]);
    if (mp > best.prob) {
      const double *o = L.res[L.slot_row[L.omin[tid]]];
      best.prob = mp; best.prob_valid = true; best.w = o[1];
#pragma unroll
      for (int u = 0; u < 6; u++) best.h[u] = o[2 + u];
      const int2 m2 = __builtin_bit_cast(int2, o[8]);
      best.r = __builtin_bit_cast(float, m2.x); best.plane = m2.y;
    }
  }
  __syncthreads();
  CSTAMP(5);
  return true;
}

// Evaluation half: every thread of the block calls it with its plan (W is block-uniform).
template <int BLOCK> __device__ __forceinline__ void coop_run(CoopLds<BLOCK> &L, const DevMap &map, const CoopPlan &pl, int max_layer, const GateConsts &gc,
                                         const StateRefs &st, Best &best COOP_PROF_PARAM) {
#ifdef LIVO2_PHASE_PROF
#define COOP_PROF_FWD , cprof
#else
#define COOP_PROF_FWD
#endif
  for (int base = 0; base < pl.W;) {
    if (pl.W - base > BLOCK && coop_round<2, BLOCK>(L, map, pl, base, max_layer, gc, st, best COOP_PROF_FWD)) { base += 2 * BLOCK; continue; }
    coop_round<1, BLOCK>(L, map, pl, base, max_layer, gc, st, best COOP_PROF_FWD);
    base += BLOCK;
  }
}

// ---- fused per-iteration pass -----------------------------------------------------------------------------------------
// One thread per point.  blockIdx is remapped so that consecutive point chunks land on the same XCD (dispatcher places
// block b on XCD b % 8): neighbouring points share voxel planes, so each XCD's private 4-MiB L2 keeps one spatial slab.
// (pblock, pgrid): this block's index within the scan's own grid and that grid's size — blockIdx/gridDim for a single scan, the
// frame-local values in a batched launch.  Row `pblock` of `partials` receives the block's sums, so a frame reduces in the same
// order whether it is launched alone or inside a batch (for the same BLOCK).
// `ext`: extR (9) and extT (3) of `a` once more, as an ADDRESS (the kernel-argument segment / the batch entry): after T1 only the z == 0 patch of a plane evaluation
// reads them (voxel_map.cpp:352-358), and as values they were 24 SGPRs held — and spilled — for the whole kernel on behalf of a branch almost no lane takes.
// PUBLISH (k_lidar_iteration): the partial row is consumed by another block of the SAME launch — write-through (sc1) stores, HIP guide section 6 guideline 16, recipe R1.
template <int BLOCK, bool PUBLISH = false> __device__ __forceinline__ void lidar_residual_body(const LidarKernelArgs &a, const double *__restrict__ ext, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                    int check_stop, int pblock, int pgrid) {
  if (check_stop && ctl->hdr.stop) return;
  extern __shared__ __attribute__((aligned(16))) double lds_red[];
  const int per_xcd = pgrid >> 3;                                // host launches a multiple of 8 blocks per scan
  const int vb = (pblock & 7) * per_xcd + (pblock >> 3);         // chunk index; chunks past the scan are empty
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = vb * BLOCK + tid;            // position in the sorted scan
  const bool valid = i < a.n;
  const int ic = valid ? i : 0;                    // clamped index: invalid lanes compute on point 0 and contribute nothing
  CoopLds<BLOCK> &coop = *reinterpret_cast<CoopLds<BLOCK> *>(lds_red);       // aliases the block's reduction tiles (used strictly before them)

  PHASE(0);
  // wave-uniform state (scalar loads)
  const double *R = ctl->cur.rot, *t = ctl->cur.pos, *cov = ctl->cur.cov;
  const double plx = a.x[ic], ply = a.y[ic], plz = a.z[ic];
  const int o = a.perm[ic];                                     // per-point outputs go back to the caller's order
  PointCtx pt;
  double *Cb = pt.Cb, *pi = pt.pi;
  float *pwf = pt.pwf;
  pt.plx = a.x[ic]; pt.ply = a.y[ic]; pt.zpatch = (plz == 0);
#pragma unroll
  for (int e = 0; e < 6; e++) Cb[e] = a.cb[(size_t)e * a.n + ic];
  const GateConsts gc = {&a.sigma_num, (int)((offsetof(LidarKernelArgs, sn2_lo) - offsetof(LidarKernelArgs, sigma_num)) / 8), (int)((offsetof(LidarKernelArgs, sn2_hi) - offsetof(LidarKernelArgs, sigma_num)) / 8)};
  // p_i = extR * p_l + extT  (un-patched, voxel_map.cpp:522 / 418)
#pragma unroll
  for (int j = 0; j < 3; j++) pi[j] = ((a.ER[j * 3] * plx + a.ER[j * 3 + 1] * ply) + a.ER[j * 3 + 2] * plz) + a.Et[j];
  // p_w = float32( R * p_i + t )   (voxel_map.cpp:522-526)
#pragma unroll
  for (int j = 0; j < 3; j++) pwf[j] = (float)(((R[j * 3] * pi[0] + R[j * 3 + 1] * pi[1]) + R[j * 3 + 2] * pi[2]) + t[j]);
  // voxel key (voxel_map.cpp:665-671): double divide, narrow to float, -1 for negatives, truncate
  float loc[3]; int32_t key[3]; bool in_range = valid;
#pragma unroll
  for (int j = 0; j < 3; j++) loc[j] = (float)((double)pwf[j] * a.inv_voxel_size);       // power-of-two voxel_size: the product IS the quotient
  if (__builtin_expect(a.inv_voxel_size == 0.0, 0)) {                                     // any other voxel_size (HILTI22's 0.4): the true division, ~12 f64 operations per axis
#pragma unroll
    for (int j = 0; j < 3; j++) loc[j] = (float)((double)pwf[j] / a.voxel_size);
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float l = loc[j];
    if (l < 0) l = (float)((double)l - 1.0);
    loc[j] = l;
    in_range = in_range && (l > -2147483000.f) && (l < 2147483000.f);
    key[j] = in_range ? (int32_t)l : 0;
  }
  PHASE(1);
  // T2: both cuckoo slots at once
  RootSlot s1, s2;
  {
    LIVO2_HASH_PAIR(key[0], key[1], key[2], a.map.seed1, a.map.seed2, a.map.mask, h1, h2);
    s1 = load_slot(a.map.slots, h1); s2 = load_slot(a.map.slots, h2);
  }
  // symmetric parts of P[0:3,0:3] and P[3:6,3:6] (a quadratic form only sees the symmetric part): block-uniform, kept in LDS behind the prefetch dump
  // (published by the barrier of the first coop_plan, which every thread passes before any plane is evaluated)
  double *sP = reinterpret_cast<double *>(reinterpret_cast<char *>(lds_red) + LIDAR_LDS_BYTES_OF(BLOCK) + LIDAR_LDS_SP_OFF);
  if (tid < 12) {
    const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
    const int e = tid % 6, o3 = tid < 6 ? 0 : 3;
    int i_ = 0, j_ = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) if (k == e) { i_ = ii[k]; j_ = jj[k]; }
    sP[tid] = 0.5 * (cov[(o3 + i_) * DS + o3 + j_] + cov[(o3 + j_) * DS + o3 + i_]);
  }
  const StateRefs st = {ctl, sP};
  if (a.var && valid) {       // pv.var = R Cb R^T + X Prr X^T + Ptt (voxel_map.cpp:387), only materialised when the caller asks for it
    double pc[3]; point_pc(pt, a.ER, a.Et, pc);
    const double Cbf[9] = {Cb[0], Cb[1], Cb[2], Cb[1], Cb[3], Cb[4], Cb[2], Cb[4], Cb[5]};
    double T[9], RC[9], XP[9], XPX[9];
    mat3_mul(R, Cbf, T); mat3_mul_Bt(T, R, RC);
    const double X[9] = {0.0, -pc[2], pc[1], pc[2], 0.0, -pc[0], -pc[1], pc[0], 0.0};
    const double Prr[9] = {cov[0], cov[1], cov[2], cov[DS], cov[DS + 1], cov[DS + 2], cov[2 * DS], cov[2 * DS + 1], cov[2 * DS + 2]};
    mat3_mul(X, Prr, XP); mat3_mul_Bt(XP, X, XPX);
#pragma unroll
    for (int e = 0; e < 9; e++) { const int r = e / 3, c = e % 3, u = r < c ? r : c, v = r < c ? c : r; a.var[(size_t)o * 9 + e] = RC[u * 3 + v] + XPX[u * 3 + v] + cov[(3 + u) * DS + 3 + v]; }
  }
  if (a.pw && valid) { a.pw[(size_t)o * 3] = pwf[0]; a.pw[(size_t)o * 3 + 1] = pwf[1]; a.pw[(size_t)o * 3 + 2] = pwf[2]; }
  PHASE(2);
  Best best; best.prob = 0.0; best.plane = -1; best.r = 0.f; best.success = false; best.prob_valid = false; best.dis2 = 0.0; best.sigma = 1.0; best.w = 0.0;
#pragma unroll
  for (int u = 0; u < 6; u++) best.h[u] = 0.0;
  const bool f1 = in_range && slot_match(s1, key), f2 = in_range && slot_match(s2, key);
  const bool found = f1 || f2;
  RootRef s = {-1, 0, 0}, nb = {-1, 0, 0};
  int plan1W = 0;
  {
    const RootSlot &sl = f1 ? s1 : s2;
    if (found) s = {sl.val, sl.cand_begin, sl.cand_count};
    // neighbour rule (voxel_map.cpp:682-688): voxel-index units compared with metres, reproduced as is.  Its two slots are
    // requested before the first visit so they travel together with the plane record (T3).
    int32_t nk[3] = {key[0], key[1], key[2]};
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if ((double)loc[j] > (sl.center[j] + (double)sl.quarter)) nk[j] = nk[j] + 1;
      else if ((double)loc[j] < (sl.center[j] - (double)sl.quarter)) nk[j] = nk[j] - 1;
    }
    // (when no axis triggers, the reference re-visits the same voxel with prob = 0 and fails again: nothing to do)
    const bool nbr = found && ((nk[0] != key[0]) || (nk[1] != key[1]) || (nk[2] != key[2]));
    PlaneRec p0;
    SlotHead n1, n2;                                             // only the words a later visit needs: key, val, cand_begin, cand_count
    n1.val = -1; n2.val = -1;
    if (nbr) {
      LIVO2_HASH_PAIR(nk[0], nk[1], nk[2], a.map.seed1, a.map.seed2, a.map.mask, g1, g2);
      n1 = load_slot_head(a.map.slots, g1); n2 = load_slot_head(a.map.slots, g2);
    }
    if (s.val >= 0) load_plane(a.map, s.val, p0);         // issued right behind the neighbour slots: same round trip
    // while that trip is in flight: plan the cooperative visit of the block's non-plane roots (LDS + one barrier only)
    const CoopPlan plan1 = coop_plan(coop, 0, (s.val == -2) ? s.cand_count : 0, s.cand_begin);
    // the neighbour slots return first (in order); keep only the three words a later visit needs
    if (nbr) {
      if (head_match(n1, nk)) nb = {n1.val, n1.cand_begin, n1.cand_count};
      else if (head_match(n2, nk)) nb = {n2.val, n2.cand_begin, n2.cand_count};
    }
    // A plane-root neighbour is visited only if the first visit fails, one dependent trip later: start pulling its two cache
    // lines towards this CU now (a discarded dword per line), so that trip is an L2 hit instead of an HBM miss.
    if (nb.val >= 0) { touch_line<BLOCK>(a.map.planes + (size_t)nb.val * PLANE_HOT_DOUBLES); touch_line<BLOCK>(a.map.plane_aux + nb.val); }
    if (s.val >= 0) visit_plane_root(p0, s.val, gc, pt, ext, ext + 9, st, best);
    plan1W = plan1.W;
    if (plan1.W > 0) {              // block-uniform
      coop_park_ctx(coop, pt, ext, ext + 9);      // (the first barrier inside coop_run orders these rows before the evaluators' reads)
      coop_run(coop, a.map, plan1, a.max_layer, gc, st, best COOP_PROF_ARG1);
      coop_unpark_ctx(coop, pt);
    }
#ifdef LIVO2_PHASE_PROF
    if (a.prof && lane == 0) a.prof[((size_t)blockIdx.x * 4 + wave) * 8 + 7] = (unsigned long long)plan1.W;
#endif
  }
  PHASE(3);
  {
    const bool retry = found && !best.success && nb.val != -1;
    const CoopPlan plan2 = coop_plan(coop, 1, (retry && nb.val == -2) ? nb.cand_count : 0, nb.cand_begin);
    if (retry && nb.val >= 0) {
      PlaneRec p1; load_plane(a.map, nb.val, p1);
      visit_plane_root(p1, nb.val, gc, pt, ext, ext + 9, st, best);
    }
    if (plan2.W > 0) {
      if (plan1W == 0) coop_park_ctx(coop, pt, ext, ext + 9);
      coop_run(coop, a.map, plan2, a.max_layer, gc, st, best COOP_PROF_ARG2);
    }
  }
  PHASE(4);
  double acc[LIDAR_NSUM];
#pragma unroll
  for (int q = 0; q < LIDAR_NSUM; q++) acc[q] = 0.0;
  if (valid) {
    if (a.match_plane) a.match_plane[o] = best.success ? best.plane : -1;
    if (a.dis) a.dis[o] = best.success ? best.r : 0.f;
    if (best.success) {
      if (a.normal_plane) a.normal_plane[o] = best.plane;
      const double zz = -(double)best.r;                        // meas_vec(i) = -dis_to_plane_ (float32 residual, voxel_map.cpp:457)
      int sidx = 0;
#pragma unroll
      for (int u = 0; u < 6; u++) {
        const double hw = best.h[u] * best.w;
#pragma unroll
        for (int v = u; v < 6; v++) { acc[sidx] = hw * best.h[v]; sidx++; }
        acc[21 + u] = hw * zz;
      }
      acc[27] = 1.0;
      acc[28] = fabs((double)best.r);
    }
    if (a.r_inv) a.r_inv[o] = best.success ? best.w : 0.0;
    if (a.h_row) {
#pragma unroll
      for (int u = 0; u < 6; u++) a.h_row[(size_t)o * 6 + u] = best.success ? best.h[u] : 0.0;
    }
  }

  PHASE(5);
  // Block reduction through an LDS transpose (deterministic).  ds_bpermute butterflies over 29 doubles cost ~20 us of this
  // kernel (measured); here every lane stores its 29 values (conflict-free: consecutive lanes -> consecutive 8-B words), then
  // lane (v = lane&31, half = lane>>5) of each wave adds 32 of the 64 columns of value v (row pitch 65 doubles => the 32 lanes of
  // a ds_read_b64 group hit 32 distinct bank pairs), one xor-32 exchange joins the halves, and 4 waves are joined in fixed order.
  __syncthreads();                                     // the cooperative-visit tiles alias the reduction tiles
  double *T = lds_red + (size_t)wave * (32 * 65);
#pragma unroll
  for (int q = 0; q < LIDAR_NSUM; q++) T[q * 65 + lane] = acc[q];
  __syncthreads();
  {
    const int v = lane & 31, half = lane >> 5;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (v < LIDAR_NSUM) {
      const double *row = T + v * 65 + half * 32;
#pragma unroll
      for (int j = 0; j < 32; j += 4) { s0 += row[j]; s1 += row[j + 1]; s2 += row[j + 2]; s3 += row[j + 3]; }
    }
    double tot = (s0 + s1) + (s2 + s3);
    tot += __shfl_xor(tot, 32, 64);
    __syncthreads();                                   // all column reads done before the tile is reused for the wave totals
    if (lane < 32) lds_red[wave * 32 + lane] = tot;
  }
  __syncthreads();
  if (tid < 32) {
    double v = lds_red[tid];
#pragma unroll
    for (int w = 1; w < BLOCK / LIVO2_WAVE; w++) v = v + lds_red[32 * w + tid];
    const double out = (tid < LIDAR_NSUM) ? v : 0.0;
    if (PUBLISH) __hip_atomic_store(&partials[(size_t)pblock * 32 + tid], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partials[(size_t)pblock * 32 + tid] = out;
  }
  PHASE(6);
}

// chunks = number of BLOCK-point chunks of the scan (a multiple of 8).  The grid may be SMALLER than that: block b then works through chunks b, b + gridDim, ...
// (same XCD slab: gridDim is a multiple of 8) — at C4 the scan has 782 chunks against the 512 blocks the chip holds at once, and a second-round block pays the
// dispatcher, the LDS allocation and the scalar state loads again (measured life: 10.7 us against 7.8 us for a first-round block).  Row `chunk` of `partials`
// receives the sums whichever block produced them: the reduction order does not depend on the grid.
// `order` (nullable): launch slot -> chunk.  Blocks are dispatched in blockIdx order, so order[] decides which chunks start in the first round of blocks; the chunk
// index (not the slot) selects the points and the partial row, so results do not depend on it.  `cost` (nullable): cost[chunk] = lifetime of the block that
// worked on the chunk, in 10-ns ticks.
static_assert(offsetof(LidarKernelArgs, Et) == offsetof(LidarKernelArgs, ER) + 72, "extR and extT are read as twelve consecutive doubles");
// the same twelve doubles in the kernel-argument segment (LidarKernelArgs is the first argument of both single-scan kernels)
__device__ __forceinline__ const double *lidar_kernarg_ext() {
  return (const double *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LidarKernelArgs, ER));      // (constant address space -> generic)
}
__device__ __forceinline__ const LidarKernelArgs &lidar_kernarg_args() { return *(const LidarKernelArgs *)__builtin_amdgcn_kernarg_segment_ptr(); }
// `stamps` (nullable, read by address like the other arguments): [chunks][2] start / end of the block that worked on the chunk on the chip-wide 100-MHz clock — the
// kernel's OWN duration (first block's start to last block's end) without a profiler and without the launch-boundary cost an event pair includes; k_lidar_span_acc
// folds them into per-context totals behind every solve of a timed pass (bench.py: roofline.kernel_us_device).
struct LidarResidualKernargs { LidarKernelArgs a; const DevCtl *ctl; double *partials; int32_t check_stop, chunks; const int32_t *order; uint32_t *cost; unsigned long long *stamps; };   // mirror of the parameter list
__device__ __forceinline__ unsigned long long *lidar_kernarg_stamps() {
  return *(unsigned long long *const *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LidarResidualKernargs, stamps));
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LIDAR_WPE))) k_lidar_residual(LidarKernelArgs a, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                                int check_stop, int chunks, const int32_t *__restrict__ order, uint32_t *__restrict__ cost, unsigned long long *stamps_by_address) {
  // One block per chunk, and NO loop around the body: inside a loop every launch-invariant value of the body (kernel arguments, addresses into the control block, the
  // "is this output wanted" conditions) is hoisted in front of it and stays live across the whole body — 145 spilled SGPRs, i.e. ~140 v_writelane at the start of every
  // wave and ~180 v_readlane along its way, a quarter of the VALU instructions a wave issues (the body is bound by instruction issue, profiles/r04_l2_retention_probe.txt).
  static_assert(offsetof(LidarResidualKernargs, stamps) == sizeof(LidarKernelArgs) + 40, "kernel-argument layout");
  if (check_stop && ctl->hdr.stop) return;
  const int pb = order ? order[blockIdx.x] : (int)blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // The body reads the launch arguments BY ADDRESS from the kernel-argument segment (scalar loads where a value is used): as by-value parameters LLVM loads all ~70
  // SGPRs of them in the prologue and keeps them live across the whole body — 70 spilled SGPRs, i.e. v_writelane / v_readlane traffic in a body that is bound by
  // instruction issue.  By address: 30 spills (exec masks of the nested branches), k_lidar_residual 21.1 -> 20.35 us by events, 28.45 -> 27.2 us per iteration
  // (profiles/r05_lidar_args_by_address_ab.txt).
  lidar_residual_body<BLOCK>(lidar_kernarg_args(), lidar_kernarg_ext(), ctl, partials, 0, pb, chunks);
  if (threadIdx.x == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (cost) cost[pb] = (uint32_t)(t1 - t0);
    unsigned long long *st = lidar_kernarg_stamps();
    if (st) { st[2 * pb] = t0; st[2 * pb + 1] = t1; }
  }
}
// acc[0] += last end - first start of the launch whose stamps are in `stamps` (nothing if it exited at its first instruction), acc[1] += 1, acc[2] = longest, acc[3] = shortest span;
// the stamps are emptied again.  One block.
__global__ void __launch_bounds__(256) k_lidar_span_acc(unsigned long long *__restrict__ stamps, int chunks, unsigned long long *__restrict__ acc) {
  __shared__ unsigned long long smin[256], smax[256];
  unsigned long long lo = ~0ull, hi = 0ull;
  for (int c = threadIdx.x; c < chunks; c += 256) {
    const unsigned long long s0 = stamps[2 * c], s1 = stamps[2 * c + 1];
    if (s0 != ~0ull && s1 != ~0ull) { lo = s0 < lo ? s0 : lo; hi = s1 > hi ? s1 : hi; }
    stamps[2 * c] = ~0ull; stamps[2 * c + 1] = ~0ull;
  }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) { if (smin[threadIdx.x + d] < smin[threadIdx.x]) smin[threadIdx.x] = smin[threadIdx.x + d]; if (smax[threadIdx.x + d] > smax[threadIdx.x]) smax[threadIdx.x] = smax[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && smin[0] != ~0ull && smax[0] >= smin[0]) {
    const unsigned long long span = smax[0] - smin[0];
    acc[0] += span; acc[1] += 1;
    if (span > acc[2]) acc[2] = span;
    if (acc[3] == 0 || span < acc[3]) acc[3] = span;
  }
}
// The resident-grid variant (LIVO2_LIDAR_RESIDENT=<blocks>, tools/lidar_resident_probe.py): block b works through chunks b, b + gridDim, ...
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2))) k_lidar_residual_resident(LidarKernelArgs a, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                                int check_stop, int chunks, const int32_t *__restrict__ order, uint32_t *__restrict__ cost) {
  for (int slot = (int)blockIdx.x; slot < chunks; slot += (int)gridDim.x) {
    const int pb = order ? order[slot] : slot;
    const unsigned long long t0 = cost ? __builtin_amdgcn_s_memrealtime() : 0ull;
    lidar_residual_body<BLOCK>(lidar_kernarg_args(), lidar_kernarg_ext(), ctl, partials, check_stop, pb, chunks);
    if (cost && threadIdx.x == 0) cost[pb] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t0);
    if (slot + (int)gridDim.x < chunks) __syncthreads();           // the next chunk's cooperative-visit tiles alias the reduction tiles just read
  }
}

// Batched launch: several independent (scan, state) problems against the resident map in ONE grid.  A single 100k-point scan is
// 392 blocks — fewer than the 512 the chip holds at once — so its launch lasts as long as its slowest block (~2x the mean);
// with several frames in the grid the CUs always have another block to start and the tail is paid once per batch.
struct LidarBatchEntry { LidarKernelArgs a; DevCtl *ctl; double *partials; int32_t block_begin, nblocks; };
__global__ void __launch_bounds__(LIDAR_BLOCK_BATCH) __attribute__((amdgpu_waves_per_eu(2))) k_lidar_residual_batch(const LidarBatchEntry *__restrict__ entries,
                                                                const int32_t *__restrict__ block_frame, int check_stop) {
  const int f = block_frame[blockIdx.x];                         // block-uniform: scalar loads
  const LidarBatchEntry &e = entries[f];
#ifdef LIDAR_BATCH_ARGS_COPY
  const LidarKernelArgs a = e.a;
  lidar_residual_body<LIDAR_BLOCK_BATCH>(a, e.a.ER, e.ctl, e.partials, check_stop, (int)blockIdx.x - e.block_begin, e.nblocks);
#else
  lidar_residual_body<LIDAR_BLOCK_BATCH>(e.a, e.a.ER, e.ctl, e.partials, check_stop, (int)blockIdx.x - e.block_begin, e.nblocks);      // (by address, as in k_lidar_residual)
#endif
}

// Deterministic reduction of per-block partial sums: partials[nblocks][32] -> out[32] (LDS).  SOLVE_THREADS threads = 16 slices
// x 32 values: slice s adds blocks s, s+16, ... in order.  The rows were written by other CUs, so every dependent load is a full
// ~1.5-us round trip (a single wave walking them serially cost ~20 us): each thread therefore issues its loads in batches of 32
// (512 rows = 131k points of a single scan per batch) before the first add of the batch.  The slices are then joined in fixed order.
// Every thread of the block must call this.  (512 threads, not 1024: the register-resident solve of wave 0 needs > 128 VGPRs.)
#define SOLVE_THREADS 512
__device__ inline void reduce_partials_block(const double *__restrict__ partials, int nblocks, double *scratch /*[16][33]*/, double *out /*[32]*/) {
  const int t = threadIdx.x, kidx = t & 31, slice = t >> 5;       // 16 slices
  double acc = 0.0;
  if (nblocks <= 512) {                                              // one pass of 32 loads per thread
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; u++) { const int b = slice + 16 * u; v[u] = (b < nblocks) ? partials[(size_t)b * 32 + kidx] : 0.0; }
#pragma unroll
    for (int u = 0; u < 32; u++) acc += v[u];
  } else {
    for (int base = 0; base < nblocks; base += 1024) {               // 1024 rows per pass: 64 loads per thread in flight before the first add
      double v[64];
#pragma unroll
      for (int u = 0; u < 64; u++) { const int b = base + slice + 16 * u; v[u] = (b < nblocks) ? partials[(size_t)b * 32 + kidx] : 0.0; }
#pragma unroll
      for (int u = 0; u < 64; u++) acc += v[u];
    }
  }
  scratch[slice * 33 + kidx] = acc;
  __syncthreads();
  if (t < 32) {
    double r = scratch[t];
#pragma unroll
    for (int sl = 1; sl < 16; sl++) r += scratch[sl * 33 + t];
    out[t] = r;
  }
  __syncthreads();
}

// ---- reduction + solve + loop control ----------------------------------------------------------------------------------
// mode 0: bare iterate (only reduce and publish sums_l) ; mode 1: full ESIKF iteration `iter` of `max_iter`;
// mode 2: like 1 but never stops (benchmark: fixed iteration count).
#ifdef LIVO2_PHASE_PROF
#define SPHASE(k) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); if (sprof && threadIdx.x == 0) { sprof[k] = __builtin_readcyclecounter(); if ((k) == 0 || (k) == 5) sprof[8 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#define SOLVE_PROF_PARAM , unsigned long long *sprof
#else
#define SPHASE(k) do { } while (0)
#define SOLVE_PROF_PARAM
#endif
// ---- block order of the NEXT k_lidar_residual launch: longest lifetime first ------------------------------------------------------------------------------
// A C4 scan is 784 blocks against the 512 the chip holds at once: the blocks of the second round start at 7-10 us, and when one of them is a cluttered chunk (12-13 us
// instead of 7) the launch ends with it.  Lifetimes are stable from iteration to iteration, so the residual kernel records them per chunk and one otherwise idle
// wave of the solve turns them into a launch order (blocks are dispatched in blockIdx order): counting sort over 64 buckets between the shortest and the longest, descending.  Only the ORDER in
// which chunks start changes; the chunk index selects points and partial row, so every result is unchanged.  Measured at C4: 25.1 -> 21.7 us per launch (events).
struct LptArgs { const uint32_t *cost; int32_t *order; int32_t chunks, pad; };
#define LPT_MAX_CHUNKS 1024                                      // 16 lifetimes per lane of the ordering wave, all in registers (262 144 points per scan; larger scans keep the identity order)
__device__ __forceinline__ void lidar_block_order_load(const LptArgs &lpt, int lane, uint32_t (&cc)[LPT_MAX_CHUNKS / LIVO2_WAVE]) {
#pragma unroll
  for (int u = 0; u < LPT_MAX_CHUNKS / LIVO2_WAVE; u++) { const int c = lane + LIVO2_WAVE * u; cc[u] = (c < lpt.chunks) ? lpt.cost[c] : 0u; }      // one batch of loads, issued with the kernel's other reads
}
__device__ inline void lidar_block_order_wave(const LptArgs &lpt, uint32_t *hist /*[64]*/, uint32_t *fill /*[64]*/, int lane, const uint32_t (&cc)[LPT_MAX_CHUNKS / LIVO2_WAVE]) {
  hist[lane] = 0u;
  // the 64 buckets span [shortest, longest] lifetime of THIS launch (a fixed 160-ns grid clamps every block of a slow device into the last bucket: 1 000 LDS atomics
  // on one address, +4 us, and no order at all)
  uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
  for (int u = 0; u < LPT_MAX_CHUNKS / LIVO2_WAVE; u++) if (lane + LIVO2_WAVE * u < lpt.chunks) { lo = min(lo, cc[u]); hi = max(hi, cc[u]); }
#pragma unroll
  for (int d = 1; d < LIVO2_WAVE; d <<= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, LIVO2_WAVE)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, LIVO2_WAVE)); }
  const uint32_t span = max(hi - lo, 1u);
  uint32_t bk[LPT_MAX_CHUNKS / LIVO2_WAVE];
#pragma unroll
  for (int u = 0; u < LPT_MAX_CHUNKS / LIVO2_WAVE; u++) bk[u] = (uint32_t)(((unsigned long long)(max(cc[u], lo) - lo) * 63ull) / span);
  wave_sync();
#pragma unroll
  for (int u = 0; u < LPT_MAX_CHUNKS / LIVO2_WAVE; u++) if (lane + LIVO2_WAVE * u < lpt.chunks) atomicAdd(&hist[bk[u]], 1u);
  wave_sync();
  const uint32_t v = hist[63 - lane];                            // lane l <-> bucket 63 - l: the exclusive prefix over the lanes is the start of the bucket in descending order
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < LIVO2_WAVE; d <<= 1) { const uint32_t up = __shfl_up(incl, d, LIVO2_WAVE); if (lane >= d) incl += up; }
  fill[63 - lane] = incl - v;
  wave_sync();
#pragma unroll
  for (int u = 0; u < LPT_MAX_CHUNKS / LIVO2_WAVE; u++) {
    const int c = lane + LIVO2_WAVE * u;
    if (c < lpt.chunks) lpt.order[atomicAdd(&fill[bk[u]], 1u)] = c;      // a permutation of 0 .. chunks-1 by construction
  }
}

__device__ __forceinline__ void lidar_solve_algebra(DevCtl *__restrict__ ctl, SolveLds &s, const double *sums, int mode, int iter, int max_iter, int hdr_rematch, const int lane);
__device__ __forceinline__ void lidar_solve_body(DevCtl *__restrict__ ctl, const double *__restrict__ partials, int nblocks, int mode, int iter,
                                                 int max_iter, const LptArgs &lpt, const bool order_block SOLVE_PROF_PARAM) {
  SPHASE(0);
  __shared__ SolveLds s;
  __shared__ double scratch[16 * 33];
  __shared__ double sums[32];
  __shared__ uint32_t lpt_hist[64], lpt_fill[64];
  // The block order of the next residual launch is the work of a SECOND block (blockIdx 1, one wave; the host launches it only when an order is wanted): round 3 ran
  // it on an otherwise idle wave of THIS block, and on some boxes the solve went from 8.7 to 13.3 us (profiles/r03_lidar_block_order_ab.txt) — the sort's ~1 000 LDS
  // atomics and 16 loads per lane share the CU's LDS and its memory pipeline with the wave whose latency chain is the critical path of the iteration.  On its own CU
  // it costs the solve nothing and ends well inside the solve's shadow.
  if (order_block) {
    // (no test of ctl->hdr.stop here: block 0 of this launch may be writing it — advisor, round 4; an order computed for a loop that has just ended is simply not used)
    if (threadIdx.x < LIVO2_WAVE && lpt.order) {
      uint32_t lpt_cc[LPT_MAX_CHUNKS / LIVO2_WAVE];
      lidar_block_order_load(lpt, threadIdx.x, lpt_cc);
      lidar_block_order_wave(lpt, lpt_hist, lpt_fill, threadIdx.x, lpt_cc);
    }
    return;
  }
  // every global read of this kernel is issued here, in one batch: loop-control words, covariance + states, the partial rows
  const int hdr_stop = ctl->hdr.stop, hdr_rematch = ctl->hdr.rematch_num;
  double craw[6];
  if (mode != 0 && threadIdx.x < LIVO2_WAVE) esikf_prefetch_wave(ctl, s, 1.0, threadIdx.x, craw);
  if (mode != 0 && threadIdx.x == LIVO2_WAVE) esikf_log_lane(ctl, s);                   // second wave: overlaps the partial rows
  SPHASE(1);
  reduce_partials_block(partials, nblocks, scratch, sums);
  SPHASE(2);
  if (mode == 1 && hdr_stop) return;
  if (threadIdx.x >= LIVO2_WAVE) return;             // the 19-dim algebra is one wave: wave-local synchronisation only from here on
  SPHASE(3);
  lidar_solve_algebra(ctl, s, sums, mode, iter, max_iter, hdr_rematch, threadIdx.x);
  SPHASE(5);
}

// The 19-dim part of one iteration, ONE wave (wave-local synchronisation only): sums (reduced partial rows, LDS) -> H^T R^-1 H / H^T R^-1 z -> Kalman update of
// ctl->cur -> convergence / rematch / covariance update and, on the stopping iteration, the result block (voxel_map.cpp:464-499).  s.P / s.cur / s.prop
// (esikf_prefetch_wave) and s.vec[0..2] (esikf_log_lane) must be in LDS.  Shared by k_lidar_solve and by the last-arriving block of k_lidar_iteration.
__device__ __forceinline__ void lidar_solve_algebra(DevCtl *__restrict__ ctl, SolveLds &s, const double *sums, int mode, int iter, int max_iter, int hdr_rematch, const int lane) {
  // expand symmetric 21 -> 6x6
  if (lane < 36) {
    int r = lane / 6, c = lane % 6;
    int u = r < c ? r : c, v = r < c ? c : r;
    int idx = u * 6 - (u * (u - 1)) / 2 + (v - u);
    s.hth[lane] = sums[idx];
  }
  if (lane < 6) s.htz[lane] = sums[21 + lane];
  wave_sync();
  livo2_lidar_sums *out = (mode == 0) ? &ctl->sums_l : &ctl->lidar.iter_sums[iter];
  if (lane < 36) out->HtH[lane] = s.hth[lane];
  if (lane < 6) out->Htz[lane] = s.htz[lane];
  if (lane == 0) { out->total_residual = sums[28]; out->n_eff = (int32_t)sums[27]; out->pad = 0; }
  if (mode == 0) return;
  esikf_update_wave<6>(ctl, s, +1, lane);
  if (lane < DS) ctl->lidar.iter_solution[iter][lane] = s.sol[lane];

  // convergence / rematch / covariance update (voxel_map.cpp:475-499)
  const double rq = (s.sol[0] * s.sol[0] + s.sol[1] * s.sol[1]) + s.sol[2] * s.sol[2], tq = (s.sol[3] * s.sol[3] + s.sol[4] * s.sol[4]) + s.sol[5] * s.sol[5];
  const bool conv = esikf_norm_below(rq, 57.3, 0.01) && esikf_norm_below(tq, 100.0, 0.015);
  int rematch = hdr_rematch;
  if (conv || ((rematch == 0) && (iter == (max_iter - 2)))) rematch++;
  const bool stop_now = (rematch >= 2 || (iter == max_iter - 1));
  wave_sync();
  if (stop_now && mode == 1) {
    // cov = (I - G) * cov ; s.P holds cov (meas_cov_scale = 1) and G is zero beyond column 5.  The posterior goes to the result block here, in the launch that
    // ends the loop (round 3: a separate k_lidar_finish launch cost ~5 us per frame); exactly one iteration of an update takes this branch (iter == max_iter - 1 at the latest).
    double *post = reinterpret_cast<double *>(&ctl->lidar.state);
    for (int e = lane; e < DS * DS; e += LIVO2_WAVE) {
      const int r = e / DS, c = e % DS;
      double v = 0.0;
      for (int k = 0; k < DS; k++) {
        const double coef = ((r == k) ? 1.0 : 0.0) - ((k < 6) ? s.G[r * KMAX + k] : 0.0);
        v = v + coef * s.P[k * DS + c];
      }
      ctl->cur.cov[e] = v; post[25 + e] = v;
    }
    if (lane < 9) post[lane] = s.newR[lane]; else if (lane < 25) post[lane] = s.cur[lane] + s.sol[lane - 6];      // the values esikf_commit_wave stored in ctl->cur
    if (lane < 3) ctl->lidar.position_last[lane] = s.cur[9 + lane] + s.sol[3 + lane];
  }
  if (lane == 0) {
    ctl->hdr.rematch_num = rematch;
    ctl->lidar.n_iters = iter + 1;
    ctl->lidar.converged = conv ? 1 : 0;
    if (stop_now && mode == 1) ctl->hdr.stop = 1;
  }
}

__global__ void __launch_bounds__(SOLVE_THREADS) k_lidar_solve(DevCtl *__restrict__ ctl, const double *__restrict__ partials, int nblocks, int mode, int iter,
                                                              int max_iter, LptArgs lpt SOLVE_PROF_PARAM) {
#ifdef LIVO2_PHASE_PROF
  lidar_solve_body(ctl, partials, nblocks, mode, iter, max_iter, lpt, blockIdx.x == 1, sprof);
#else
  lidar_solve_body(ctl, partials, nblocks, mode, iter, max_iter, lpt, blockIdx.x == 1);
#endif
}

// one block per frame of a batch
__global__ void __launch_bounds__(SOLVE_THREADS) k_lidar_solve_batch(const LidarBatchEntry *__restrict__ entries, int mode, int iter, int max_iter) {
  const LidarBatchEntry &e = entries[blockIdx.x];
#ifdef LIVO2_PHASE_PROF
  lidar_solve_body(e.ctl, e.partials, e.nblocks, mode, iter, max_iter, LptArgs{nullptr, nullptr, 0, 0}, false, nullptr);
#else
  lidar_solve_body(e.ctl, e.partials, e.nblocks, mode, iter, max_iter, LptArgs{nullptr, nullptr, 0, 0}, false);
#endif
}

// copies the iterate into the result block after a loop that never takes the stopping branch (mode 0 / 2: fixed iteration counts)
__global__ void __launch_bounds__(LIVO2_WAVE) k_lidar_finish(DevCtl *__restrict__ ctl_base) {
  DevCtl *ctl = ctl_base + blockIdx.x;             // one block per frame (a single update has one)
  const int lane = threadIdx.x;
  const double *src = reinterpret_cast<const double *>(&ctl->cur);
  double *dst = reinterpret_cast<double *>(&ctl->lidar.state);
  for (int e = lane; e < (int)(sizeof(livo2_state) / sizeof(double)); e += LIVO2_WAVE) dst[e] = src[e];
}

// ---- one ESIKF iteration as ONE launch: residual grid + reduction + solve (round 5) -------------------------------------------------------------------------------
// reference src/voxel_map.cpp:372-500 (one turn of the iteration loop of StateEstimation).
// Rounds 1-4 ran every iteration as two launches: the residual grid, then k_lidar_solve on ONE compute unit — 11-15 us of which ~2 are the kernel boundary, ~3 the
// start-up and the partial rows' trip through one CU, and, on some boxes, several more because two blocks alone on the chip between two full-grid launches look like
// an idle chip to the power management (VERDICT r04 "weak" 3).  Here the last block of the residual grid to publish its row does the solve:
//   every block : partial row with write-through (sc1) stores -> the storing wave drains (s_waitcnt vmcnt(0)) -> one lane draws a ticket (relaxed agent-scope
//                 fetch_add) — HIP guide section 6 guideline 16, recipe R1 in its counter form; no fence, no polling of rows.
//   tickets are HIERARCHICAL: 784 fetch_adds on one address are served one after the other at ~30-40 ns each (measured: the flat counter made the launch 31 us
//                 longer); a block draws from the counter of its group (launch slot & 31: ~25 blocks, and — the dispatcher deals slots round-robin over the XCDs —
//                 all on one XCD), the last of a group draws from the top counter, the last there is the last arriver of the launch.
//   last arriver: P / both states / Log(cur^T prop) exactly as k_lidar_solve issues them, all rows with cache-bypassing loads in the order of
//                 reduce_partials_block (16 slices, rows s, s + 16, ... ascending, slices joined in order: bit-identical sums), then lidar_solve_algebra; it leaves
//                 every counter at zero again (nobody else touches them any more) and flips the order selector.
//   second-to-last top ticket : the launch order of the NEXT launch (the counting sort of lidar_block_order_wave) on its own CU, into the order buffer this launch
//                 does NOT read (a block reads order[blockIdx.x] as its first instruction, and the one block that is still missing may not have got there yet).
// The fused arguments are read from the kernel-argument segment where they are needed (scalar loads by address): as by-value parameters they would be ~12 more
// SGPRs live across the whole residual body.
#define LIDAR_TICKET_GROUPS 32
#define LIDAR_TICKET_WORDS (LIDAR_TICKET_GROUPS + 2)             // group counters, top counter, order-buffer selector
struct LidarFuseArgs { int32_t mode, iter, max_iter, order_cap; const uint32_t *cost; int32_t *order_buf; int32_t *tickets; };   // order_buf: [2][order_cap] or null
struct LidarIterationKernargs { LidarKernelArgs a; const DevCtl *ctl; double *partials; int32_t check_stop, chunks; const int32_t *order_buf; uint32_t *cost; LidarFuseArgs fz; };   // mirror of the kernel's parameter list
__device__ __forceinline__ const LidarFuseArgs &lidar_kernarg_fuse() {
  return *(const LidarFuseArgs *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LidarIterationKernargs, fz));
}
#define LIDAR_FUSE_FLAG_OFF 384                                  // behind the reduction tiles: [0,256) prefetch landing area, [256,352) sym(P_rr), sym(P_tt), [384,388) "my top ticket"
#define LIDAR_FUSE_PASS 32                                       // rows per slice and pass of the last arriver: 64 doubles per thread in flight (512 rows per pass)

template <int BLOCK>
__device__ __forceinline__ void lidar_fused_tail(DevCtl *__restrict__ ctl, const double *partials, int chunks, int sel, double *lds) {
  static_assert(BLOCK == 256, "16 reduction slices x 16 lanes");
  const LidarFuseArgs &fz = lidar_kernarg_fuse();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *flag = reinterpret_cast<int *>(reinterpret_cast<char *>(lds) + LIDAR_LDS_BYTES_OF(BLOCK) + LIDAR_FUSE_FLAG_OFF);
  const int ngroups = min(chunks, LIDAR_TICKET_GROUPS);
  if (wave == 0) {                                                // the wave that stored the row (and the lifetime)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const int g = (int)blockIdx.x & (LIDAR_TICKET_GROUPS - 1);
      const int members = chunks / LIDAR_TICKET_GROUPS + (g < chunks % LIDAR_TICKET_GROUPS ? 1 : 0);
      int top = -1;
      if (__hip_atomic_fetch_add(fz.tickets + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
        top = __hip_atomic_fetch_add(fz.tickets + LIDAR_TICKET_GROUPS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = top;
    }
  }
  __syncthreads();
  const int top = *flag;
  if (top < ngroups - 2) return;
  if (top == ngroups - 2) {
    // every lifetime but those of the last group's stragglers has been published (sc1 stores ahead of the tickets): any permutation is a valid order
    if (wave == 0 && fz.order_buf) {
      uint32_t cc[LPT_MAX_CHUNKS / LIVO2_WAVE];
#pragma unroll
      for (int u = 0; u < LPT_MAX_CHUNKS / LIVO2_WAVE; u++) { const int c = lane + LIVO2_WAVE * u; cc[u] = (c < chunks) ? __hip_atomic_load(fz.cost + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; }
      const LptArgs lpt = {fz.cost, fz.order_buf + (size_t)(1 - sel) * fz.order_cap, chunks, 0};
      lidar_block_order_wave(lpt, reinterpret_cast<uint32_t *>(lds), reinterpret_cast<uint32_t *>(lds) + 64, lane, cc);
    }
    return;
  }
  // ---- the last arriver: every row of this launch has been published
  SolveLds &s = *reinterpret_cast<SolveLds *>(lds);
  static_assert(sizeof(SolveLds) <= 1024 * 8, "SolveLds in the first 8 KB of the tiles");
  double *scratch = lds + 1024, *sums = scratch + 16 * 33;
  const int mode = fz.mode, iter = fz.iter, max_iter = fz.max_iter;
  const int hdr_rematch = ctl->hdr.rematch_num;
  double craw[6];
  if (wave == 0) esikf_prefetch_wave(ctl, s, 1.0, lane, craw);
  if (tid == LIVO2_WAVE) esikf_log_lane(ctl, s);
  if (tid >= 2 * LIVO2_WAVE && tid < 2 * LIVO2_WAVE + LIDAR_TICKET_GROUPS + 1) __hip_atomic_store(fz.tickets + (tid - 2 * LIVO2_WAVE), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all tickets are drawn
  if (tid == 3 * LIVO2_WAVE && fz.order_buf) __hip_atomic_store(fz.tickets + LIDAR_TICKET_GROUPS + 1, 1 - sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the next launch reads the buffer the sorter of this one fills
  {
    const int kp = tid & 15, slice = tid >> 4;                    // two of the 32 values of a row per thread; slice s adds rows s, s + 16, ... (reduce_partials_block's order)
    double a0 = 0.0, a1 = 0.0;
    for (int base = 0; base < chunks; base += 16 * LIDAR_FUSE_PASS) {
      double v0[LIDAR_FUSE_PASS], v1[LIDAR_FUSE_PASS];
#pragma unroll
      for (int u = 0; u < LIDAR_FUSE_PASS; u++) {
        const int b = base + slice + 16 * u;
        const double *src = partials + (size_t)(b < chunks ? b : 0) * 32 + 2 * kp;
        v0[u] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v1[u] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b >= chunks) { v0[u] = 0.0; v1[u] = 0.0; }
      }
#pragma unroll
      for (int u = 0; u < LIDAR_FUSE_PASS; u++) { a0 += v0[u]; a1 += v1[u]; }
    }
    scratch[slice * 33 + 2 * kp] = a0; scratch[slice * 33 + 2 * kp + 1] = a1;
  }
  __syncthreads();
  if (tid < 32) {
    double r = scratch[tid];
#pragma unroll
    for (int sl = 1; sl < 16; sl++) r += scratch[sl * 33 + tid];
    sums[tid] = r;
  }
  __syncthreads();
  if (wave != 0) return;
  lidar_solve_algebra(ctl, s, sums, mode, iter, max_iter, hdr_rematch, lane);
}

// order_buf: two launch orders of `fz.order_cap` entries each (null: scan order); word LIDAR_TICKET_GROUPS + 1 of fz.tickets selects the one this launch reads.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2))) k_lidar_iteration(LidarKernelArgs a, const DevCtl *__restrict__ ctl, double *__restrict__ partials,
                                                                int check_stop, int chunks, const int32_t *__restrict__ order_buf, uint32_t *__restrict__ cost, LidarFuseArgs fz_by_address) {
  static_assert(offsetof(LidarIterationKernargs, fz) == sizeof(LidarKernelArgs) + 40, "kernel-argument layout");
  if (check_stop && ctl->hdr.stop) return;                        // the loop has ended: no row, no ticket
  extern __shared__ __attribute__((aligned(16))) double lds_red[];
  int sel = 0, pb = (int)blockIdx.x;
  if (order_buf) {                                                // three independent scalar loads, one round trip
    const LidarFuseArgs &fz = lidar_kernarg_fuse();
    const int o0 = order_buf[blockIdx.x], o1 = order_buf[(size_t)fz.order_cap + blockIdx.x];
    sel = fz.tickets[LIDAR_TICKET_GROUPS + 1] & 1;
    pb = sel ? o1 : o0;
  }
  const unsigned long long t0 = cost ? __builtin_amdgcn_s_memrealtime() : 0ull;
  lidar_residual_body<BLOCK, true>(lidar_kernarg_args(), lidar_kernarg_ext(), ctl, partials, 0, pb, chunks);
  if (cost && threadIdx.x == 0) __hip_atomic_store(&cost[pb], (uint32_t)(__builtin_amdgcn_s_memrealtime() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lidar_fused_tail<BLOCK>(const_cast<DevCtl *>(ctl), partials, chunks, sel, lds_red);
}
