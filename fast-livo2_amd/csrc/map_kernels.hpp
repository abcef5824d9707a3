// Map maintenance kernels: the arithmetic of VoxelOctoTree::init_plane (reference src/voxel_map.cpp:55-135)
// for a BATCH of voxels — centre, 3x3 covariance, its eigen-decomposition, the plane test, and the 6x6 plane covariance
// plane_var_ = sum_i J_i var_i J_i^T — with the fitted records written straight into the resident map (what
// livo2_map_update_planes does from host arrays).  The octree bookkeeping (which voxel collects which point, when a voxel is
// re-fitted or subdivided: UpdateOctoTree / cut_octo_tree, voxel_map.cpp:163-290) stays with the caller: it decides the groups.
//
// LPG lanes per group: 8 for the <= 64-point voxels UpdateVoxelMap re-fits (the eigen-solve is serial per group: a whole wave per group
// runs it 64x redundantly — measured 456 us for 20k groups against 77 us with 8 lanes), 64 for the large voxels of an initial BuildVoxelMap.  Sums are reduced in a fixed
// (lane-strided, then butterfly) order: deterministic, but not the reference's
// serial order, and covariance_ = E[pp^T] - cc^T cancels ~8 digits at world coordinates of tens of metres, so eigenvalues agree
// with the serial CPU evaluation to ~1e-8 relative, not to the last bit (tests/test_plane_fit_gpu.py states the tolerances).
// The reference's Eigen::EigenSolver is not available here (parity unpinned); like the oracle this uses a cyclic Jacobi
// solver, the same operation sequence as oracle/orc_voxel_map.hpp so that the two agree wherever their inputs do.
#pragma once
#include "livo2_device.hpp"

#define FIT_WAVES 4

struct PlaneFitArgs {
  const double *pw;               // [N][3] point_w
  const double *var;              // [N][9] pointWithVar::var, row-major
  const int32_t *offsets;         // [n_groups + 1]
  const int32_t *list;            // [n_list] the groups this launch handles
  int32_t n_list;
  float planer_threshold;         // VoxelOctoTree::planer_threshold_ (a float member compared against a double eigenvalue)
  livo2_plane_fit *out;           // [n_groups]
  // optional in-place refresh of the resident map
  const int32_t *plane_idx;       // [n_groups] caller's plane index, -1 = do not write ; or null
  const int32_t *plane_internal;  // caller plane index -> row of the device plane table
  const int32_t *plane_cand_pos;  // caller plane index -> position of its copy in the candidate records, or -1
  double *planes;                 // master records [.][32]
  double *planes_hot, *cand;      // what the residual kernel reads: [.][16] by plane row / by candidate position
  PlaneAux *plane_aux, *cand_aux;
};

// symmetric 3x3 eigen-decomposition, cyclic Jacobi; columns of V are unit eigenvectors
__device__ __forceinline__ void eig3_jacobi(const double A[9], double ev[3], double V[9]) {
  double a[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { a[k] = A[k]; V[k] = (k % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = (a[1] * a[1] + a[2] * a[2]) + a[5] * a[5];
    if (off < 1e-300) break;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int q = p + 1; q < 3; q++) {
        if (a[p * 3 + q] == 0.0) continue;
        const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * a[p * 3 + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; k++) { const double akp = a[k * 3 + p], akq = a[k * 3 + q]; a[k * 3 + p] = c * akp - s * akq; a[k * 3 + q] = s * akp + c * akq; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const double apk = a[p * 3 + k], aqk = a[q * 3 + k]; a[p * 3 + k] = c * apk - s * aqk; a[q * 3 + k] = s * apk + c * aqk; }
#pragma unroll
        for (int k = 0; k < 3; k++) { const double vkp = V[k * 3 + p], vkq = V[k * 3 + q]; V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq; }
      }
  }
  ev[0] = a[0]; ev[1] = a[4]; ev[2] = a[8];
}

template <int LPG> __device__ __forceinline__ double group_sum(double v) {      // all-reduce over the LPG lanes of a group
#pragma unroll
  for (int off = LPG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// VoxelOctoTree::init_plane for the points [lo, hi) of (pw, var), evaluated by the LPG lanes of one group (every lane ends up with the same values).
struct FitRes {
  double c[3], cov[9], ev_min, ev_mid, ev_max, vmin[3], vmid[3], vmax[3], pv[36];
  bool is_plane;
};
// SYM: only the upper triangle of plane_var_ is accumulated (pv[r * 6 + u], r <= u; the rest stays 0) — for callers that keep sym(plane_var_) anyway (the
// device octree: 15 accumulators, their products and their group sums less in a kernel that sits at the register limit).  Equal to the symmetrised full sum
// when the point covariances are symmetric, to rounding otherwise.
template <int LPG, bool SYM = false> __device__ __forceinline__ void plane_fit_core(const double *__restrict__ pw, const double *__restrict__ var_, int lo, int hi, int lane, float planer_threshold, FitRes &R) {
  const int n = hi - lo;
  // pass 1: covariance_ += p p^T, center_ += p  (voxel_map.cpp:63-67)
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // sum p (3), then xx xy xz yy yz zz
  for (int i = lo + lane; i < hi; i += LPG) {
    const double x = pw[(size_t)i * 3], y = pw[(size_t)i * 3 + 1], z = pw[(size_t)i * 3 + 2];
    s[0] += x; s[1] += y; s[2] += z;
    s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
  }
#pragma unroll
  for (int k = 0; k < 9; k++) s[k] = group_sum<LPG>(s[k]);
  const double nn = (double)n;
  double *c = R.c, *cov = R.cov;
  c[0] = s[0] / nn; c[1] = s[1] / nn; c[2] = s[2] / nn;                       // voxel_map.cpp:68
  cov[0] = s[3] / nn - c[0] * c[0]; cov[1] = s[4] / nn - c[0] * c[1]; cov[2] = s[5] / nn - c[0] * c[2];      // voxel_map.cpp:69
  cov[3] = s[4] / nn - c[1] * c[0]; cov[4] = s[6] / nn - c[1] * c[1]; cov[5] = s[7] / nn - c[1] * c[2];
  cov[6] = s[5] / nn - c[2] * c[0]; cov[7] = s[7] / nn - c[2] * c[1]; cov[8] = s[8] / nn - c[2] * c[2];
  double ev[3], V[9];
  eig3_jacobi(cov, ev, V);
  int imin = 0, imax = 0;                                                     // minCoeff / maxCoeff: first occurrence (voxel_map.cpp:75-76)
#pragma unroll
  for (int i = 1; i < 3; i++) { if (ev[i] < ev[imin]) imin = i; if (ev[i] > ev[imax]) imax = i; }
  if (imin == imax) { imin = 0; imax = 2; }                                    // all three equal: the reference indexes out of range
  const int imid = 3 - imin - imax;
  // (register arrays are only indexed with compile-time constants: a lane-dependent index would move them to scratch memory)
  auto sel3 = [](double x0, double x1, double x2, int i) { return i == 0 ? x0 : (i == 1 ? x1 : x2); };
  const double ev_min = sel3(ev[0], ev[1], ev[2], imin), ev_mid = sel3(ev[0], ev[1], ev[2], imid), ev_max = sel3(ev[0], ev[1], ev[2], imax);
  const bool is_plane = ev_min < (double)planer_threshold;                     // voxel_map.cpp:85
  const double vmin[3] = {sel3(V[0], V[1], V[2], imin), sel3(V[3], V[4], V[5], imin), sel3(V[6], V[7], V[8], imin)};
  const double vmid[3] = {sel3(V[0], V[1], V[2], imid), sel3(V[3], V[4], V[5], imid), sel3(V[6], V[7], V[8], imid)};
  const double vmax[3] = {sel3(V[0], V[1], V[2], imax), sel3(V[3], V[4], V[5], imax), sel3(V[6], V[7], V[8], imax)};
  R.ev_min = ev_min; R.ev_mid = ev_mid; R.ev_max = ev_max; R.is_plane = is_plane;
#pragma unroll
  for (int k = 0; k < 3; k++) { R.vmin[k] = vmin[k]; R.vmid[k] = vmid[k]; R.vmax[k] = vmax[k]; }
  double *pv = R.pv;
#pragma unroll
  for (int k = 0; k < 36; k++) pv[k] = 0.0;
  if (is_plane) {
    // group constants of F_m (voxel_map.cpp:95-99): 1 / (n (l_min - l_m)) and the symmetric v_m v_min^T + v_min v_m^T
    double den[3], M[3][9];
#pragma unroll
    for (int m = 0; m < 3; m++) {
      den[m] = nn * (ev_min - ev[m]);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) M[m][r * 3 + q] = V[r * 3 + m] * vmin[q] + vmin[r] * V[q * 3 + m];
    }
    const double jq = 1.0 / nn;                                                // J_Q (voxel_map.cpp:82)
    for (int i = lo + lane; i < hi; i += LPG) {
      const double d[3] = {pw[(size_t)i * 3] - c[0], pw[(size_t)i * 3 + 1] - c[1], pw[(size_t)i * 3 + 2] - c[2]};
      double var[9];
#pragma unroll
      for (int k = 0; k < 9; k++) var[k] = var_[(size_t)i * 9 + k];
      double F[9];
#pragma unroll
      for (int m = 0; m < 3; m++) {
        const double l0 = d[0] / den[m], l1 = d[1] / den[m], l2 = d[2] / den[m];
#pragma unroll
        for (int q = 0; q < 3; q++) F[m * 3 + q] = (m != imin) ? ((l0 * M[m][q] + l1 * M[m][3 + q]) + l2 * M[m][6 + q]) : 0.0;
      }
      // J = [evecs F ; J_Q] (6x3), plane_var_ += (J var) J^T (voxel_map.cpp:108-110).  The lower half of J is (1/n) I: its products with
      // exact zeros are dropped, which changes no rounding (0*x + y == y for finite x).
      double Jt[9];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) Jt[r * 3 + q] = (V[r * 3] * F[q] + V[r * 3 + 1] * F[3 + q]) + V[r * 3 + 2] * F[6 + q];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double Tt[3], Tb[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { Tt[q] = (Jt[r * 3] * var[q] + Jt[r * 3 + 1] * var[3 + q]) + Jt[r * 3 + 2] * var[6 + q]; Tb[q] = jq * var[r * 3 + q]; }
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (!SYM || q >= r) pv[r * 6 + q] += (Tt[0] * Jt[q * 3] + Tt[1] * Jt[q * 3 + 1]) + Tt[2] * Jt[q * 3 + 2];
          pv[r * 6 + 3 + q] += Tt[q] * jq;
          if (!SYM) pv[(3 + r) * 6 + q] += (Tb[0] * Jt[q * 3] + Tb[1] * Jt[q * 3 + 1]) + Tb[2] * Jt[q * 3 + 2];
          if (!SYM || q >= r) pv[(3 + r) * 6 + 3 + q] += Tb[q] * jq;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 36; k++) if (!SYM || (k % 6) >= (k / 6)) pv[k] = group_sum<LPG>(pv[k]);
  }
}

template <int LPG> __global__ void __launch_bounds__(FIT_WAVES *LIVO2_WAVE) __attribute__((amdgpu_waves_per_eu(2))) k_plane_fit(PlaneFitArgs a) {
  const int lane = threadIdx.x & (LPG - 1);                      // lane within the group
  const int slot = (blockIdx.x * (FIT_WAVES * LIVO2_WAVE) + threadIdx.x) / LPG;
  if (slot >= a.n_list) return;                                  // (a whole group's lanes leave together)
  const int g = a.list[slot];
  const int lo = a.offsets[g], hi = a.offsets[g + 1];
  const int n = hi - lo;
  livo2_plane_fit *o = a.out + g;
  if (n <= 0) {                                   // the reference never fits an empty voxel; report "no plane"
    double *z = reinterpret_cast<double *>(o);
    for (int k = lane; k < (int)(sizeof(livo2_plane_fit) / 8); k += LPG) z[k] = 0.0;
    return;
  }
  FitRes R;
  plane_fit_core<LPG>(a.pw, a.var, lo, hi, lane, a.planer_threshold, R);
  const bool is_plane = R.is_plane;
  const double *c = R.c, *cov = R.cov, *pv = R.pv, *vmin = R.vmin, *vmid = R.vmid, *vmax = R.vmax;
  const double ev_min = R.ev_min, ev_mid = R.ev_mid, ev_max = R.ev_max;
  const double nrm[3] = {is_plane ? vmin[0] : 0.0, is_plane ? vmin[1] : 0.0, is_plane ? vmin[2] : 0.0};
  const float radius = is_plane ? (float)sqrt(ev_max) : 0.f;                 // radius_ is a float member (voxel_map.h:77)
  const float dd = is_plane ? (float)(-((nrm[0] * c[0] + nrm[1] * c[1]) + nrm[2] * c[2])) : 0.f;
  // every lane of the group holds the same values; the lanes share the stores (entry k by lane k mod LPG)
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (lane == k) {
      o->center[k] = c[k]; o->normal[k] = nrm[k];
      o->y_normal[k] = is_plane ? vmid[k] : 0.0; o->x_normal[k] = is_plane ? vmax[k] : 0.0;
    }
#pragma unroll
  for (int k = 0; k < 9; k++) if (lane == (k + 3) % LPG) o->covariance[k] = cov[k];
#pragma unroll
  for (int k = 0; k < 36; k++) if (lane == (k + 12) % LPG) o->plane_var[k] = pv[k];
  if (lane == 0) {
    o->radius = radius; o->d = dd;
    // eigenvalue members are floats and only assigned for planes (voxel_map.cpp:113-115); their initial value is 1 (voxel_map.h:78-80)
    o->min_eigen_value = is_plane ? (float)ev_min : 1.f; o->mid_eigen_value = is_plane ? (float)ev_mid : 1.f; o->max_eigen_value = is_plane ? (float)ev_max : 1.f;
    o->points_size = n; o->is_plane = is_plane ? 1 : 0; o->pad = 0;
  }
  // in-place refresh of the resident map record (same packing as pack_plane on the host)
  if (a.plane_idx && is_plane) {
    const int32_t pi = a.plane_idx[g];
    if (pi >= 0) {
      const int32_t row = a.plane_internal[pi];
      double *rec = a.planes + (size_t)row * PLANE_REC_DOUBLES;
      auto put = [&](int k, double v) { if (lane == k % LPG) rec[k] = v; };
#pragma unroll
      for (int k = 0; k < 3; k++) { put(k, nrm[k]); put(3 + k, c[k]); }
      double S[21];
      {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int u = r; u < 6; u++) { S[q] = 0.5 * (pv[r * 6 + u] + pv[u * 6 + r]); put(6 + q, S[q]); q++; }
      }
      put(27, __builtin_bit_cast(double, make_float2(dd, radius)));
      // the residual kernel's view: hot words + side word, in the plane table and in the plane's copy inside a candidate list (whose meta stays)
      double hot[PLANE_HOT_DOUBLES];
      plane_hot_words(nrm, c, S, hot);
      const int32_t gp = a.plane_cand_pos[pi];
#pragma unroll
      for (int k = 0; k < PLANE_HOT_DOUBLES; k++)
        if (lane == k % LPG) { a.planes_hot[(size_t)row * PLANE_HOT_DOUBLES + k] = hot[k]; if (gp >= 0) a.cand[(size_t)gp * PLANE_HOT_DOUBLES + k] = hot[k]; }
      if (lane == 0) {
        a.plane_aux[row].d = dd; a.plane_aux[row].radius = radius;
        if (gp >= 0) { a.cand_aux[gp].d = dd; a.cand_aux[gp].radius = radius; }
      }
#pragma unroll
      for (int k = 28; k < 32; k++) put(k, 0.0);
    }
  }
}
