// Device-side data layout and small dense algebra shared by the gfx950 kernels of liblivo2_hip.so.
//
// Precision recipe: every quantity that the reference narrows to float32 or feeds into a
// thresholded decision is evaluated with the reference's operand types and operation order; this translation unit is
// compiled with -ffp-contract=off so the compiler never fuses those.  Where only tolerance-level agreement is
// required (covariance propagation, quadratic forms, the 19x19 algebra) fused multiply-adds are requested explicitly
// through fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/livo2_hip.h"

#define LIVO2_WAVE 64
#define DS LIVO2_DIM_STATE

// ---- VoxelMap snapshot in HBM -------------------------------------------------------------------------------------
// The kernel's critical path is a chain of DEPENDENT memory round trips (hash -> plane -> neighbour hash -> plane), each
// costing ~1 us under load, while every wave of a 100k-point scan is resident at once: the layout below exists to make
// that chain as short as possible, not to save bytes.
//
// Root table: 2-choice cuckoo hash, one 64-B slot per bucket.  A lookup issues BOTH candidate slots at once => exactly one
// round trip, never a probe loop.  The slot carries everything the first visit and the neighbour rule need
// (voxel_center_, quater_length_, reference include/voxel_map.h:139,141), so no second table is touched.
struct __attribute__((aligned(64))) RootSlot {
  int32_t kx, ky, kz;          // VOXEL_LOCATION (int32 range enforced at upload)
  int32_t val;                 // -1 empty ; >=0: root is a plane -> plane index ; -2: non-plane root -> candidate list
  double center[3];            // voxel_center_
  float quarter;               // quater_length_
  int32_t cand_begin;          // first entry in cand[] (non-plane roots)
  int32_t cand_count;          // number of descendant planes in depth-first leaves_[0..7] order
  int32_t pad;
};
// candidate entry meta: plane index | (layer << 28); the all-8-children recursion of build_single_residual (voxel_map.cpp:769-785)
// visits exactly the descendant planes whose ancestors are all non-planes, in depth-first child order, and only down to
// cfg.max_layer: flattening that walk at upload time removes three dependent loads per visited node.
#define CAND_LAYER_SHIFT 28
#define CAND_PLANE_MASK 0x0fffffff

// Plane table, master copy: 256 B per plane — what the map kernels (fit, octree maintenance, export) write and read.
//   [0..2] normal_  [3..5] center_  [6..26] sym(plane_var_) upper triangle row-major (21)  [27] {float d_, float radius_}
//   [28] {int32 layer, 0} (device-built trees)  [29..31] pad
#define PLANE_REC_DOUBLES 32
// What the residual kernel reads per (point, plane) pair: ONE 128-B line + one 16-B word of a side array.
//   With J = [a, -n] (a = p - center_, voxel_map.cpp:733-735 / 437-440) the quadratic form J plane_var_ J^T is
//       a^T See a - 2 a^T v + k,   See = plane_var_[0:3,0:3] (sym),  v = plane_var_[0:3,3:6] n,  k = n^T plane_var_[3:6,3:6] n
//   — v and k depend on the plane only, so they are formed once when the plane is (re)fitted instead of once per pair: 10 doubles instead of 21 travel
//   and stay in registers, and the form costs 15 operations instead of 48.
//   hot[0..2] normal_  [3..5] center_  [6..11] See (00 01 02 11 12 22)  [12..14] v  [15] k ;  aux = {d_, radius_, plane | layer << 28 (candidate copies), 0}
#define PLANE_HOT_DOUBLES 16
struct __attribute__((aligned(16))) PlaneAux { float d, radius; int32_t meta, pad; };
// S = the 21-entry upper triangle at words [6..26] of the master record (row r starts at {0, 6, 11, 15, 18, 20})
__host__ __device__ inline void plane_hot_words(const double *n, const double *c, const double *S, double *hot) {
  hot[0] = n[0]; hot[1] = n[1]; hot[2] = n[2]; hot[3] = c[0]; hot[4] = c[1]; hot[5] = c[2];
  hot[6] = S[0]; hot[7] = S[1]; hot[8] = S[2]; hot[9] = S[6]; hot[10] = S[7]; hot[11] = S[11];
  hot[12] = (S[3] * n[0] + S[4] * n[1]) + S[5] * n[2];
  hot[13] = (S[8] * n[0] + S[9] * n[1]) + S[10] * n[2];
  hot[14] = (S[12] * n[0] + S[13] * n[1]) + S[14] * n[2];
  hot[15] = (n[0] * ((S[15] * n[0] + S[16] * n[1]) + S[17] * n[2]) + n[1] * ((S[16] * n[0] + S[18] * n[1]) + S[19] * n[2])) +
            n[2] * ((S[17] * n[0] + S[19] * n[1]) + S[20] * n[2]);
}

struct DevMap {
  const RootSlot *slots;
  const double *cand_rec;      // [n_cand][16]: hot words of the candidate's plane, in candidate (depth-first) order
  const PlaneAux *cand_aux;    // [n_cand]: d_, radius_, plane | layer << 28
  const double *planes;        // [n_planes][16]: hot words
  const PlaneAux *plane_aux;   // [n_planes]
  uint32_t mask;               // capacity - 1 (power of two)
  uint32_t seed1, seed2;
  int32_t n_planes;
};

// Table hash of a voxel key.  Both probes of a lookup (seed1 / seed2) share the key-dependent part: three multiplies for the pair instead of fourteen — a point makes two
// lookups (its voxel, the neighbour voxel) per iteration, and an integer multiply costs what an f64 FMA costs on this chip (tools/valu_rate_probe.hip:
// v_mul_lo_u32 4 cycles per wave, v_xor / shifts 2.4; profiles/r06_valu_rate_probe.txt).  Only key EQUALITY has to agree with VOXEL_LOCATION::operator==
// (voxel_map.h:103); the hash itself is free.  Add and xor alternate in the base so that it is linear over neither; the finaliser is lowbias32.
#define LIVO2_HASH_PAIR(x, y, z, s1, s2, m, o1, o2) const uint32_t hb_##o1 = voxel_hash_base(x, y, z); const uint32_t o1 = voxel_hash_fin(hb_##o1, s1) & (m), o2 = voxel_hash_fin(hb_##o1, s2) & (m)
__host__ __device__ inline uint32_t voxel_hash_base(int32_t x, int32_t y, int32_t z) {
  uint32_t h = (uint32_t)x * 0x9e3779b1u;
  h = ((h << 13) | (h >> 19)) + (uint32_t)y * 0x85ebca6bu;
  h = ((h << 11) | (h >> 21)) ^ ((uint32_t)z * 0xc2b2ae35u);
  return h;
}
__host__ __device__ inline uint32_t voxel_hash_fin(uint32_t base, uint32_t seed) {
  uint32_t h = base ^ seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t voxel_hash(int32_t x, int32_t y, int32_t z, uint32_t seed) { return voxel_hash_fin(voxel_hash_base(x, y, z), seed); }

// ---- control block (one per ctx, in HBM) ----------------------------------------------------------------------------
struct DevHeader {             // written by the host at the start of every update (one H2D copy together with cur/prop)
  int32_t stop;                // LiDAR EKF_stop_flg / visual EKF_end of the running level
  int32_t rematch_num;
  int32_t reserved;
  float last_error;            // visual
  int32_t n_steps;
  int32_t pad[3];
  double RE[9];                // state_propagat.rot_end * extR_  (voxel_map.cpp:445), constant during one update
};
struct DevCtl {
  livo2_state cur;             // state_ / *state   (the iterate)
  livo2_state prop;            // state_propagat
  DevHeader hdr;
  livo2_state old;             // visual old_state (vio.cpp:1523,1650,1679)
  double G[DS * DS];
  livo2_lidar_result lidar;
  livo2_visual_result visual;
  livo2_lidar_sums sums_l;     // result of a bare iterate call
  livo2_visual_sums sums_v;
  double solve_hth[49], solve_htz[7], solve_solution[DS];   // livo2_esikf_solve scratch
};

// atan out of line: inlined into a kernel with a device-side loop its 20 polynomial coefficients are hoisted above the loop and spilled (see so3_exp_call below)
__device__ __attribute__((noinline)) double atan_call(double x) { return atan(x); }

// ---- camera models of rpg_vikit (third party, unpinned: reference README.md:76-84) as used through cam->world2cam / cam->cam2world ------------------
// model 0: vk::PinholeCamera without distortion; 1: vk::PinholeCamera with the radial-tangential coefficients d0..d4 (config/camera_pinhole.yaml);
// 2: vk::EquidistantCamera with k1..k4 in d[0..3] (config/camera_fisheye_HILTI22.yaml) — the Kannala-Brandt / OpenCV-fisheye polynomial
//    theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8), theta = atan(r).
// (u, v) = the point projected onto z = 1 (vk::project2d).  Operation order = oracle/orc_visual.hpp (PinholeCam::world2cam).
#define LIVO2_CAM_PINHOLE 0
#define LIVO2_CAM_RADTAN 1
#define LIVO2_CAM_EQUIDISTANT 2
__device__ __forceinline__ void cam_project(int model, const double *d, double fx, double fy, double cx, double cy, double u, double v, double &px, double &py) {
  if (model == LIVO2_CAM_PINHOLE) { px = fx * u + cx; py = fy * v + cy; }
  else if (model == LIVO2_CAM_RADTAN) {
    const double x = u, y = v, r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6;
    const double xd = x * cdist + d[2] * a1 + d[3] * a2, yd = y * cdist + d[2] * a3 + d[3] * a1;
    px = xd * fx + cx; py = yd * fy + cy;
  } else {
    const double r = sqrt(u * u + v * v), theta = atan_call(r), t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double thetad = theta * (1 + d[0] * t2 + d[1] * t4 + d[2] * t6 + d[3] * t8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    px = fx * u * scaling + cx; py = fy * v * scaling + cy;
  }
}
// cam2world before the final normalisation: (x, y) with bearing ~ (x, y, 1).  Radtan: OpenCV's undistortPoints as vikit calls it (float32 pixel in, five
// fixed-point iterations in double, float32 out); equidistant: OpenCV-fisheye's undistortPoints scheme (ten fixed-point iterations on theta).
__device__ __forceinline__ void cam_unproject(int model, const double *d, double fx, double fy, double cx, double cy, double u, double v, double &x, double &y) {
  if (model == LIVO2_CAM_PINHOLE) { x = (u - cx) / fx; y = (v - cy) / fy; }
  else if (model == LIVO2_CAM_RADTAN) {
    const double uf = (double)(float)u, vf = (double)(float)v;
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    const double x0 = (uf - cx) * ifx, y0 = (vf - cy) * ify;
    x = x0; y = y0;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = 1.0 / (1.0 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2);
      if (icdist < 0) { x = x0; y = y0; break; }
      const double deltaX = 2.0 * d[2] * x * y + d[3] * (r2 + 2.0 * x * x);
      const double deltaY = d[2] * (r2 + 2.0 * y * y) + 2.0 * d[3] * x * y;
      x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
    }
    x = (double)(float)x; y = (double)(float)y;
  } else {
    const double xd = (u - cx) / fx, yd = (v - cy) / fy;
    const double thetad = sqrt(xd * xd + yd * yd);
    double theta = thetad;
    for (int j = 0; j < 10; j++) {
      const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      theta = thetad / (1 + d[0] * t2 + d[1] * t4 + d[2] * t6 + d[3] * t8);
    }
    const double scaling = (thetad > 1e-8) ? tan(theta) / thetad : 1.0;
    x = xd * scaling; y = yd * scaling;
  }
}

// ---- tiny 3x3 helpers (row-major) ------------------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_mul_Bt(const double *A, const double *B, double *C) {   // C = A * B^T
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i * 3 + j] = (A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1]) + A[i * 3 + 2] * B[j * 3 + 2];
}
__device__ __forceinline__ void mat3t_vec(const double *A, const double *v, double *o) {     // o = A^T v
#pragma unroll
  for (int i = 0; i < 3; i++) o[i] = (A[i] * v[0] + A[3 + i] * v[1]) + A[6 + i] * v[2];
}
// o = A^T v with fused multiply-adds (tolerance-level quantities only)
__device__ __forceinline__ void mat3t_vec_fma(const double *A, const double *v, double *o) {
#pragma unroll
  for (int i = 0; i < 3; i++) o[i] = fma(A[6 + i], v[2], fma(A[3 + i], v[1], A[i] * v[0]));
}
// v^T S v for a symmetric 3x3 given as (xx,xy,xz,yy,yz,zz)
__device__ __forceinline__ double quad3_sym(const double *S, const double *v) {
  double r0 = fma(S[2], v[2], fma(S[1], v[1], S[0] * v[0]));
  double r1 = fma(S[4], v[2], fma(S[3], v[1], S[1] * v[0]));
  double r2 = fma(S[5], v[2], fma(S[4], v[1], S[2] * v[0]));
  return fma(v[2], r2, fma(v[1], r1, v[0] * r0));
}

// SO(3) Exp / Log  (reference include/utils/so3_math.h:44-66)
__device__ inline void so3_exp(double v1, double v2, double v3, double *R) {
  double nrm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  if (nrm > 0.00001) {
    double r0 = v1 / nrm, r1 = v2 / nrm, r2 = v3 / nrm;
    double K[9] = {0.0, -r2, r1, r2, 0.0, -r0, -r1, r0, 0.0};
    double KK[9];
    mat3_mul(K, K, KK);
    double s, c;
    sincos(nrm, &s, &c);                                    // one argument reduction, both polynomials in one instruction stream (sin() and cos() apart: ~0.3 us more on the single lane that runs this)
    const double c1 = 1.0 - c;
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + c1 * KK[i];
  } else {
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
}
// column j of so3_exp(v1, v2, v3): the operations of so3_exp for the three elements of that column, in their order (the same bits) — for callers that spread the
// nine elements of a product with Exp over nine lanes instead of forming K, K^2 and R on one
__device__ inline void so3_exp_col(double v1, double v2, double v3, int j, double *E) {
  const double nrm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  if (nrm > 0.00001) {
    const double r0 = v1 / nrm, r1 = v2 / nrm, r2 = v3 / nrm;
    // K = {0, -r2, r1, r2, 0, -r0, -r1, r0, 0}: its column j, and (K K)[k][j] = (K[k][0] K[0][j] + K[k][1] K[1][j]) + K[k][2] K[2][j] as mat3_mul forms it
    const double k0 = j == 0 ? 0.0 : (j == 1 ? -r2 : r1), k1 = j == 0 ? r2 : (j == 1 ? 0.0 : -r0), k2 = j == 0 ? -r1 : (j == 1 ? r0 : 0.0);
    const double kk0 = (0.0 * k0 + (-r2) * k1) + r1 * k2, kk1 = (r2 * k0 + 0.0 * k1) + (-r0) * k2, kk2 = ((-r1) * k0 + r0 * k1) + 0.0 * k2;
    double s, c;
    sincos(nrm, &s, &c);
    const double c1 = 1.0 - c;
    E[0] = ((j == 0) ? 1.0 : 0.0) + s * k0 + c1 * kk0;
    E[1] = ((j == 1) ? 1.0 : 0.0) + s * k1 + c1 * kk1;
    E[2] = ((j == 2) ? 1.0 : 0.0) + s * k2 + c1 * kk2;
  } else {
    E[0] = (j == 0) ? 1.0 : 0.0; E[1] = (j == 1) ? 1.0 : 0.0; E[2] = (j == 2) ? 1.0 : 0.0;
  }
}
__device__ inline void so3_log(const double *R, double *o) {
  double tr = (R[0] + R[4]) + R[8];
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
  double K0 = R[7] - R[5], K1 = R[2] - R[6], K2 = R[3] - R[1];
  if (fabs(theta) < 0.001) { o[0] = 0.5 * K0; o[1] = 0.5 * K1; o[2] = 0.5 * K2; }
  else { double f = 0.5 * theta / sin(theta); o[0] = f * K0; o[1] = f * K1; o[2] = f * K2; }
}

// Out-of-line copies for kernels that call Exp / Log inside a device-side loop: inlined there, the polynomial coefficients of sin / cos / acos are hoisted out of
// the loop as VGPR constants and, under register pressure, spilled — every use then becomes a dependent scratch load (measured in k_visual_update_persistent).
struct So3Mat { double v[9]; };
struct So3Vec { double v[3]; };
__device__ __attribute__((noinline)) So3Mat so3_exp_call(double v1, double v2, double v3) { So3Mat m; so3_exp(v1, v2, v3, m.v); return m; }
__device__ __attribute__((noinline)) So3Vec so3_exp_col_call(double v1, double v2, double v3, int j) { So3Vec o; so3_exp_col(v1, v2, v3, j, o.v); return o; }
__device__ __attribute__((noinline)) So3Vec so3_log_call(So3Mat R) { So3Vec o; so3_log(R.v, o.v); return o; }

// LDS hand-off inside ONE wave: orders this wave's LDS writes before its later LDS reads without a workgroup barrier (s_barrier counts every wave of the
// block, so a phase that only one wave executes must not use __syncthreads())
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// wave-64 all-reduce (sum) of a double through the LDS crossbar (ds_bpermute, no LDS storage)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
