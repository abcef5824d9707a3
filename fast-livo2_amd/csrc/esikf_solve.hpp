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
// Launch geometry: ONE wave (64 threads) of whatever block calls it; LDS hand-offs inside the wave are ordered by wave_sync(), never by a workgroup barrier.
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
  double hv[8];                 // H_k vec[0:k]: with it the solution row needs 2k FMAs behind the back substitution instead of k x k + 2k (G = K_1 H_k can wait, see DEFER_G)
  double vec[DS + 1];
  double sol[DS + 1];
  double cur[25], prop[25];     // the 25 non-covariance scalars of state_ / state_propagat (rot 9, pos 3, inv_expo, vel, bg, ba, grav)
  double newR[9];               // rot_end * Exp(solution[0:3]), prepared by esikf_solve_wave so that the commit only stores
};

// One Kalman update of ctl->cur given the reduced sums (s.hth: k x k row-major with stride k, s.htz).
// sign=+1: LiDAR form (K1*HTz + vec - G*vec) ; sign=-1: visual form (-K1*HTz + vec - G*vec).
// Leaves the solution in s.sol, G[:,0:k] in s.G (and the zero-padded 19x19 G in ctl->G).
// Part 1 (independent of the measurement sums, so it overlaps the partial-sum loads): P' = cov / scale and the 25 pose / bias scalars
// of both states into LDS.  Every later step of the solve reads LDS only: a global load issued late costs a full ~1.5-us round trip
// on the single wave that is the critical path of the whole iteration.  Call from wave 0 only; no barrier inside.
// (cur_state / prop_state: the iterate and the prior where they are — ctl->cur / ctl->prop, or the LiDAR posterior for an update chained to it on the device)
__device__ inline void esikf_prefetch_wave(const livo2_state *cur_state, const livo2_state *prop_state, SolveLds &s, const double meas_cov_scale, const int lane, double *c /*[6] raw covariance words of this lane*/) {
#pragma unroll
  for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; c[q] = (e < DS * DS) ? cur_state->cov[e] : 0.0; }
  const double *cs = reinterpret_cast<const double *>(cur_state), *ps = reinterpret_cast<const double *>(prop_state);
  double sv = 0.0;
  if (lane < 25) sv = cs[lane]; else if (lane < 50) sv = ps[lane - 25];
#pragma unroll
  for (int q = 0; q < 6; q++) { const int e = lane + q * LIVO2_WAVE; if (e < DS * DS) s.P[e] = c[q] / meas_cov_scale; }
  if (lane < 25) s.cur[lane] = sv; else if (lane < 50) s.prop[lane - 25] = sv;
}

__device__ inline void esikf_prefetch_wave(const DevCtl *ctl, SolveLds &s, const double meas_cov_scale, const int lane, double *c) { esikf_prefetch_wave(&ctl->cur, &ctl->prop, s, meas_cov_scale, lane, c); }

// rotation part of vec = state_propagat [-] state: Log(cur^T prop) (common_lib.h:196-197) -> s.vec[0..2].  ~1.5 us of dependent f64
// transcendental code on one lane, so the solve kernels run it on a second wave while the partial rows are in flight.
__device__ inline void esikf_log_lane(const DevCtl *ctl, SolveLds &s) {
  double rc[9], rp[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { rc[i] = ctl->cur.rot[i]; rp[i] = ctl->prop.rot[i]; }
  double rotd[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) rotd[i * 3 + j] = (rc[i] * rp[j] + rc[3 + i] * rp[3 + j]) + rc[6 + i] * rp[6 + j];   // cur^T * prop
  double l[3]; so3_log(rotd, l);
  s.vec[0] = l[0]; s.vec[1] = l[1]; s.vec[2] = l[2];
}

// value held by lane SRC (compile-time constant) as a wave-uniform scalar: two v_readlane_b32
template <int SRC> __device__ __forceinline__ double esikf_bcast(double v) {
  const int2 w = __builtin_bit_cast(int2, v);
  return __builtin_bit_cast(double, make_int2(__builtin_amdgcn_readlane(w.x, SRC), __builtin_amdgcn_readlane(w.y, SRC)));
}

// elimination step C of the column-per-lane scheme (see esikf_solve_wave): lane C holds column C of A, finds the pivot row and forms the multipliers
template <int C, int k> __device__ __forceinline__ void esikf_eliminate(double (&col)[k], double (&pinv)[k]) {
  if constexpr (C < k) {
    int piv = C; double best = fabs(col[C]);
#pragma unroll
    for (int i = C + 1; i < k; i++) { const double v = fabs(col[i]); if (v > best) { best = v; piv = i; } }
    piv = __builtin_amdgcn_readlane(piv, C);
#pragma unroll
    for (int i = C + 1; i < k; i++)
      if (piv == i) { const double t = col[C]; col[C] = col[i]; col[i] = t; }     // wave-uniform row swap
    const double inv = 1.0 / col[C];
    pinv[C] = inv;                                            // (lane C's copy is the reciprocal of the pivot U[C][C]: the back substitution multiplies by it)
#pragma unroll
    for (int i = C + 1; i < k; i++) {
      const double li = esikf_bcast<C>(col[i] * inv);         // l = A[i][C] * inv, formed in lane C
      col[i] = fma(-li, col[C], col[i]);
    }
    esikf_eliminate<C + 1, k>(col, pinv);
  }
}
// t -= U[I][J] x[J] for J = I+1 .. k-1 (ascending), U[I][J] = entry I of lane J's column
template <int I, int J, int k> __device__ __forceinline__ void esikf_back_row(const double (&col)[k], const double (&x)[k], double &t) {
  if constexpr (J < k) { t = fma(-esikf_bcast<J>(col[I]), x[J], t); esikf_back_row<I, J + 1, k>(col, x, t); }
}
template <int I, int k> __device__ __forceinline__ void esikf_back(const double (&col)[k], const double (&pinv)[k], double (&x)[k]) {
  if constexpr (I >= 0) {
    double t = col[I];
    esikf_back_row<I, I + 1, k>(col, x, t);
    x[I] = t * esikf_bcast<I>(pinv[I]);                       // (round 6: the reciprocal the elimination formed, instead of a second f64 division per row: 7 dependent divisions less per solve)
    esikf_back<I - 1, k>(col, pinv, x);
  }
}

// Part 2: needs s.P / s.cur / s.prop (esikf_prefetch_wave), s.vec[0..2] (esikf_log_lane), the sums in s.hth / s.htz, and a barrier.
//   S = I + H_k P'_kk is built once in LDS; K_1[r, 0:k] (r < 19) is the solution x of  S^T x = P'[r, 0:k]^T : Gaussian elimination with partial pivoting
//   on the augmented matrix [S^T | B], B = the 19 right-hand sides, with ONE COLUMN PER LANE (lanes 0..k-1: the columns of S^T, lanes k..k+18: the right-hand
//   sides): the pivot search and the multipliers of step c are formed in lane c and broadcast with v_readlane, every lane then updates its own column; the back
//   substitution broadcasts the U entries the same way.  Every element sees exactly the operations (and their order) of a per-lane LU with partial pivoting of
//   the full system — the form this replaces held the whole k x k matrix redundantly in every lane: ~250 VGPRs, which no kernel that inlines the solve next to
//   other work could afford (k_visual_update_persistent) — and the results are bit-identical to it.  G[r, :], the Kalman solution entry and the new rotation follow
//   without further LDS round trips.
// DEFER_G: K_1[:, 0:k] is left in s.Kc and s.G is NOT formed — the caller forms G = K_1 H_k (same FMA order: bit-identical) when and where it needs it (the resident
// visual grid: a second wave, on accepted steps only).
// HV: the solution row from hv = H_k vec[0:k] (2k FMAs behind the back substitution) instead of from G — the visual update's forms (its solve is on the step's critical
// path and the resident grid defers G); the LiDAR solve forms G anyway and keeps the fused loop (with hv it measured 9.10 instead of 8.51 us per launch).
template <int k, bool MATH_CALLS = false, bool DEFER_G = false, bool HV = DEFER_G>
__device__ inline void esikf_solve_wave(SolveLds &s, const int sign, const int lane) {
  static_assert(HV || !DEFER_G, "a deferred G needs the hv form of the solution");
  if (lane < k * k) {                                        // S = I + H_k P'_kk, row-major stride k
    const int i = lane / k, j = lane % k;
    double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < k; m++) v = fma(s.hth[i * k + m], s.P[m * DS + j], v);
    s.aug[lane] = v;
  }
  if (lane >= 9 && lane < 25) s.vec[lane - 6] = s.prop[lane] - s.cur[lane];     // pos, inv_expo, vel, bg, ba, grav parts of vec
  if (HV && lane >= 56 && lane < 56 + k) {                   // hv = H_k vec[0:k] on lanes that idle here (vec[c >= 3] re-formed from the same operands: the same values)
    const int m = lane - 56;
    double h = 0.0;
#pragma unroll
    for (int c = 0; c < k; c++) h = fma(s.hth[m * k + c], (c < 3) ? s.vec[c] : (s.prop[c + 6] - s.cur[c + 6]), h);
    s.hv[m] = h;
  }
  wave_sync();
  const int r = (lane < k) ? 0 : (lane - k < DS ? lane - k : DS - 1);             // right-hand side owned by this lane (lanes < k own a matrix column)
  double col[k];
#pragma unroll
  for (int i = 0; i < k; i++) col[i] = (lane < k) ? s.aug[lane * k + i] : s.P[r * DS + i];   // column `lane` of A = S^T is row `lane` of S
  double pinv[k];
  esikf_eliminate<0, k>(col, pinv);                          // forward elimination, one pivot column at a time
  double x[k];                                               // back substitution: x = K_1[r, 0:k] in the lanes that own a right-hand side
  esikf_back<k - 1, k>(col, pinv, x);
  if (HV) {
    double kz = 0.0, gv = 0.0;                               // solution[r] = +-K_1[r, 0:k] H^T z + vec[r] - G[r, 0:k] vec[0:k], with G vec = K_1 (H_k vec)
#pragma unroll
    for (int c = 0; c < k; c++) { kz = fma(x[c], s.htz[c], kz); gv = fma(x[c], s.hv[c], gv); }
    if (lane >= k && lane < k + DS) s.sol[r] = ((sign > 0) ? kz : -kz) + s.vec[r] - gv;
  }
  if (DEFER_G) {
    if (lane >= k && lane < k + DS) {
#pragma unroll
      for (int c = 0; c < KMAX; c++) s.Kc[r * KMAX + c] = (c < k) ? x[c] : 0.0;
    }
  } else {                                                   // G[r, 0:k] = K_1[r, 0:k] H_k
    double kz = 0.0, gv = 0.0;
    double g[KMAX];
#pragma unroll
    for (int c = 0; c < k; c++) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < k; m++) t = fma(x[m], s.hth[m * k + c], t);
      g[c] = t;
      kz = fma(x[c], s.htz[c], kz);
      gv = fma(t, s.vec[c], gv);
    }
    if (lane >= k && lane < k + DS) {
#pragma unroll
      for (int c = 0; c < KMAX; c++) s.G[r * KMAX + c] = (c < k) ? g[c] : 0.0;
      if (!HV) s.sol[r] = ((sign > 0) ? kz : -kz) + s.vec[r] - gv;
    }
  }
  wave_sync();
  {                                                          // state.rot_end * Exp(delta theta)  (common_lib.h:184): lane e < 9 forms element (e / 3, e % 3) of the product —
    // the operations of so3_exp and mat3_mul for that element, in their order (same bits as one lane forming K, K K, R and the product: 0.7-1.1 us there)
    const int e9 = lane < 9 ? lane : 0, i = e9 / 3, j = e9 - 3 * i;
    const double v1 = s.sol[0], v2 = s.sol[1], v3 = s.sol[2];
    const double c0 = s.cur[i * 3], c1 = s.cur[i * 3 + 1], c2 = s.cur[i * 3 + 2];
    double E[3];
    if (MATH_CALLS) { const So3Vec m = so3_exp_col_call(v1, v2, v3, j); E[0] = m.v[0]; E[1] = m.v[1]; E[2] = m.v[2]; }
    else so3_exp_col(v1, v2, v3, j, E);
    const double rn = (c0 * E[0] + c1 * E[1]) + c2 * E[2];
    if (lane < 9) s.newR[lane] = rn;
  }
  wave_sync();
}

// fl(fl(sqrt(q)) * scale) < bound — the reference's convergence tests (voxel_map.cpp:475-478, vio.cpp:1675) — decided from q itself outside a band of 1e-9 around the
// threshold (the predicate is monotone in q and its flip lies within a few ulps of (bound / scale)^2); inside the band, the expression as the reference writes it.  Two
// f64 square roots (~0.3 us on one lane) leave the tail of every solve.
__device__ __forceinline__ bool esikf_norm_below(double q, double scale, double bound) {
  const double t = bound / scale, t2 = t * t;
  if (q < t2 * (1.0 - 1e-9)) return true;
  if (q > t2 * (1.0 + 1e-9)) return false;
  return sqrt(q) * scale < bound;
}

// Part 3: publish G (zero-padded 19x19 in ctl->G) and apply  state += solution  (common_lib.h:182-192) to ctl->cur, from the LDS copies.
__device__ inline void esikf_commit_wave(DevCtl *ctl, SolveLds &s, const int lane) {
  if (lane < DS) {
#pragma unroll
    for (int c = 0; c < KMAX; c++) ctl->G[lane * DS + c] = s.G[lane * KMAX + c];      // columns >= KMAX of ctl->G stay zero
  }
  if (lane < 9) {
    ctl->cur.rot[lane] = s.newR[lane];
  } else if (lane >= 9 && lane < 25) {
    reinterpret_cast<double *>(&ctl->cur)[lane] = s.cur[lane] + s.sol[lane - 6];
  }
}

template <int k>
__device__ inline void esikf_update_wave(DevCtl *ctl, SolveLds &s, const int sign, const int lane) {
  esikf_solve_wave<k>(s, sign, lane);
  esikf_commit_wave(ctl, s, lane);
}
