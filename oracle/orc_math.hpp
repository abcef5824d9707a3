// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// Minimal fixed-size dense algebra standing in for the Eigen types the reference uses
// (reference: include/utils/types.h:17-27 `V3D/M3D/MD(a,b)/VD(a)`; Eigen itself is not in this image).
// All sums are evaluated strictly left-to-right, one rounding per operation; the golden build uses
// -ffp-contract=off so no FMA contraction takes place.  Eigen's own evaluation order (and the reference
// binary's contraction pattern under -O3 -march=native, CMakeLists.txt:34) is build-dependent and unpinned,
// so this is ONE valid realisation of the reference arithmetic — see DESIGN.md "parity unpinned".
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>

namespace orc {

template <int R, int C> struct Mat {
  double a[R * C];
  double &operator()(int i, int j) { return a[i * C + j]; }
  const double &operator()(int i, int j) const { return a[i * C + j]; }
  double &operator[](int i) { return a[i]; }
  const double &operator[](int i) const { return a[i]; }
  static Mat Zero() { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = 0.0; return m; }
  static Mat Identity() { Mat m = Zero(); for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = 1.0; return m; }
  Mat<C, R> T() const { Mat<C, R> t; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t(j, i) = (*this)(i, j); return t; }
};

typedef Mat<3, 1> V3;
typedef Mat<3, 3> M3;

// ---- summation-order model of the two long reductions of the path (voxel_map.cpp:464-466, vio.cpp:1660-1662) -------------------------
// The reference evaluates them with Eigen (GEMM / GEMV), whose order of additions depends on the Eigen version, the SIMD width and the
// cache-derived blocking of the build.  Mode 0 (the parity checker): strict left-to-right, one rounding per operation.  Mode 1
// ("Eigen-like", tests/sweeps/oracle_sensitivity.py only): GEMM = the depth is cut into panels of `kc` (Eigen's gebp blocking), a panel is a
// sequential FMA chain, panels are added in order; column-major GEMV = one sequential FMA chain; row-major GEMV = `lanes` interleaved
// FMA chains (a SIMD packet) reduced pairwise, the tail added sequentially.  Used to bound how far the reference's OWN build can move
// the answer (profiles/r02_oracle_sensitivity.txt); never by the parity tests.
struct SumModel { int mode = 0, kc = 256, lanes = 4; };
inline SumModel &sum_model() { static SumModel m; return m; }
inline double long_dot_gemm(const double *a, size_t sa, const double *b, size_t sb, size_t n) {
  const SumModel &m = sum_model();
  double s = 0.0;
  if (m.mode == 0) { for (size_t i = 0; i < n; i++) s += a[i * sa] * b[i * sb]; return s; }
  for (size_t p0 = 0; p0 < n; p0 += (size_t)m.kc) {
    const size_t p1 = p0 + (size_t)m.kc < n ? p0 + (size_t)m.kc : n;
    double p = 0.0;
    for (size_t i = p0; i < p1; i++) p = std::fma(a[i * sa], b[i * sb], p);
    s = s + p;
  }
  return s;
}
inline double long_dot_gemv_colmajor(const double *a, size_t sa, const double *b, size_t sb, size_t n) {
  double s = 0.0;
  if (sum_model().mode == 0) { for (size_t i = 0; i < n; i++) s += a[i * sa] * b[i * sb]; return s; }
  for (size_t i = 0; i < n; i++) s = std::fma(a[i * sa], b[i * sb], s);
  return s;
}
inline double long_dot_gemv_rowmajor(const double *a, size_t sa, const double *b, size_t sb, size_t n) {
  const SumModel &m = sum_model();
  double s = 0.0;
  if (m.mode == 0) { for (size_t i = 0; i < n; i++) s += a[i * sa] * b[i * sb]; return s; }
  double lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const size_t L = (size_t)m.lanes, nb = n / L * L;
  for (size_t i = 0; i < nb; i += L) for (size_t l = 0; l < L; l++) lane[l] = std::fma(a[(i + l) * sa], b[(i + l) * sb], lane[l]);
  for (size_t w = L / 2; w >= 1; w /= 2) for (size_t l = 0; l < w; l++) lane[l] = lane[l] + lane[l + w];     // predux: (p0+p2)+(p1+p3) for 4 lanes
  s = lane[0];
  for (size_t i = nb; i < n; i++) s = std::fma(a[i * sa], b[i * sb], s);
  return s;
}

template <int R, int K, int C> inline Mat<R, C> operator*(const Mat<R, K> &A, const Mat<K, C> &B) {
  Mat<R, C> o;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) {
      double s = A(i, 0) * B(0, j);
      for (int k = 1; k < K; k++) s = s + A(i, k) * B(k, j);
      o(i, j) = s;
    }
  return o;
}
template <int R, int C> inline Mat<R, C> operator+(const Mat<R, C> &A, const Mat<R, C> &B) { Mat<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = A.a[i] + B.a[i]; return o; }
template <int R, int C> inline Mat<R, C> operator-(const Mat<R, C> &A, const Mat<R, C> &B) { Mat<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = A.a[i] - B.a[i]; return o; }
template <int R, int C> inline Mat<R, C> operator-(const Mat<R, C> &A) { Mat<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = -A.a[i]; return o; }
template <int R, int C> inline Mat<R, C> operator*(const Mat<R, C> &A, double s) { Mat<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = A.a[i] * s; return o; }
template <int R, int C> inline Mat<R, C> operator*(double s, const Mat<R, C> &A) { Mat<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = s * A.a[i]; return o; }
template <int R, int C> inline Mat<R, C> operator/(const Mat<R, C> &A, double s) { Mat<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = A.a[i] / s; return o; }

inline V3 vec3(double x, double y, double z) { V3 v; v[0] = x; v[1] = y; v[2] = z; return v; }
inline double dot(const V3 &a, const V3 &b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 cross(const V3 &a, const V3 &b) { return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]); }
// SKEW_SYM_MATRX (reference: include/utils/so3_math.h:7)
inline M3 skew(const V3 &v) {
  M3 m;
  m(0, 0) = 0.0;   m(0, 1) = -v[2]; m(0, 2) = v[1];
  m(1, 0) = v[2];  m(1, 1) = 0.0;   m(1, 2) = -v[0];
  m(2, 0) = -v[1]; m(2, 1) = v[0];  m(2, 2) = 0.0;
  return m;
}

// Exp(v1,v2,v3)  (reference: include/utils/so3_math.h:44-58; threshold 1e-5)
inline M3 Exp(double v1, double v2, double v3) {
  double nrm = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  M3 Eye3 = M3::Identity();
  if (nrm > 0.00001) {
    V3 r = vec3(v1 / nrm, v2 / nrm, v3 / nrm);
    M3 K = skew(r);
    return Eye3 + std::sin(nrm) * K + (1.0 - std::cos(nrm)) * (K * K);   // note: (1-cos)*K*K groups as ((1-cos)*K)*K in Eigen; value-equivalent to rounding
  }
  return Eye3;
}
// Log(R)  (reference: include/utils/so3_math.h:61-66)
inline V3 Log(const M3 &R) {
  double tr = (R(0, 0) + R(1, 1)) + R(2, 2);
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  V3 K = vec3(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
  return (std::abs(theta) < 0.001) ? (0.5 * K) : ((0.5 * theta / std::sin(theta)) * K);
}

// General n x n inverse by partial-pivot LU followed by a solve against the identity — the algorithm class
// Eigen's `inverse()` uses for fixed sizes > 4 (PartialPivLU; reference call sites voxel_map.cpp:468,
// vio.cpp:1495,1661).  Eigen's blocked update order is not reproduced (unpinned).  Returns false if singular.
template <int N> inline bool inverse_lu(const Mat<N, N> &Ain, Mat<N, N> &inv) {
  Mat<N, N> lu = Ain;
  int perm[N];
  for (int i = 0; i < N; i++) perm[i] = i;
  for (int k = 0; k < N; k++) {
    int piv = k; double best = std::fabs(lu(k, k));
    for (int i = k + 1; i < N; i++) { double v = std::fabs(lu(i, k)); if (v > best) { best = v; piv = i; } }
    if (best == 0.0) return false;
    if (piv != k) { for (int j = 0; j < N; j++) { double t = lu(k, j); lu(k, j) = lu(piv, j); lu(piv, j) = t; } int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; }
    for (int i = k + 1; i < N; i++) {
      lu(i, k) = lu(i, k) / lu(k, k);
      double l = lu(i, k);
      for (int j = k + 1; j < N; j++) lu(i, j) = lu(i, j) - l * lu(k, j);
    }
  }
  for (int c = 0; c < N; c++) {
    double y[N];
    for (int i = 0; i < N; i++) {          // forward: L y = P e_c
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s = s - lu(i, j) * y[j];
      y[i] = s;
    }
    for (int i = N - 1; i >= 0; i--) {     // backward: U x = y
      double s = y[i];
      for (int j = i + 1; j < N; j++) s = s - lu(i, j) * inv(j, c);
      inv(i, c) = s / lu(i, i);
    }
  }
  return true;
}

} // namespace orc
