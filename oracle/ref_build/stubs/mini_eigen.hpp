// TEST INFRASTRUCTURE ONLY — stand-in for the Eigen3 headers, which are absent from this image.
//
// Purpose: let /root/reference/src/{voxel_map,vio,frame,visual_point}.cpp compile TEXTUALLY UNMODIFIED (oracle/ref_build/Makefile), so
// that the hand-written restatement in oracle/*.hpp can be checked against the reference's own statements (tests/test_ref_pin_cpu.py).
// Only the part of the Eigen API those translation units touch exists here.  Everything is evaluated eagerly: every operator returns a
// plain Matrix; products and sums run in ascending index order, one rounding per operation (compile with -ffp-contract=off).
// What this cannot reproduce, because Eigen itself is not here: the association order of Eigen's vectorised
// reductions / GEMM kernels, Eigen's blocked PartialPivLU update order, and EigenSolver's eigenvector sign / order conventions.
// Written from the documented behaviour of the Eigen API; no Eigen source was available or used.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 3

namespace Eigen {

const int Dynamic = -1;
typedef std::ptrdiff_t Index;

template <class T, int R, int C> class Matrix;
template <class M, int BR, int BC> class Block;
template <class D> class CommaInitializer;
template <class T> using aligned_allocator = std::allocator<T>;

namespace internal {
constexpr int pick(int a, int b) { return a != Dynamic ? a : b; }
template <class D> struct traits;
template <class T, int R, int C> struct traits<Matrix<T, R, C>> { typedef T Scalar; enum { Rows = R, Cols = C }; };
template <class M, int BR, int BC> struct traits<Block<M, BR, BC>> { typedef typename traits<M>::Scalar Scalar; enum { Rows = BR, Cols = BC }; };
template <class T> struct real_of { typedef T type; };
template <class T> struct real_of<std::complex<T>> { typedef T type; };
template <class T> inline T real_part(const T &v) { return v; }
template <class T> inline T real_part(const std::complex<T> &v) { return v.real(); }
template <class S> using if_arith = typename std::enable_if<std::is_arithmetic<S>::value, int>::type;
} // namespace internal

// ------------------------------------------------------------------------------------------------------------------------------
// read interface (CRTP): the derived class supplies rows(), cols(), coeff(i,j)
template <class D> class DenseBase {
public:
  typedef typename internal::traits<D>::Scalar Scalar;
  typedef typename internal::real_of<Scalar>::type RealScalar;
  typedef Eigen::Index Index;
  enum { RowsAtCompileTime = internal::traits<D>::Rows, ColsAtCompileTime = internal::traits<D>::Cols };
  typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;

  const D &derived() const { return *static_cast<const D *>(this); }
  D &derived() { return *static_cast<D *>(this); }
  Index rows() const { return derived().rows_(); }
  Index cols() const { return derived().cols_(); }
  Index size() const { return rows() * cols(); }
  Scalar coeff(Index i, Index j) const { return derived().coeff_(i, j); }
  Scalar lin(Index k) const { return cols() == 1 ? coeff(k, 0) : (rows() == 1 ? coeff(0, k) : coeff(k % rows(), k / rows())); }

  Scalar operator()(Index i, Index j) const { return coeff(i, j); }
  Scalar operator()(Index k) const { return lin(k); }
  Scalar operator[](Index k) const { return lin(k); }
  Scalar x() const { return lin(0); }
  Scalar y() const { return lin(1); }
  Scalar z() const { return lin(2); }
  Scalar w() const { return lin(3); }

  PlainObject eval() const { return PlainObject(*this); }
  Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() const {
    Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> r; r.resize(cols(), rows());
    for (Index i = 0; i < rows(); i++) for (Index j = 0; j < cols(); j++) r.coeffRef(j, i) = coeff(i, j);
    return r;
  }
  Scalar sum() const { Scalar s = Scalar(0); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) s = s + coeff(i, j); return s; }
  Scalar trace() const { Scalar s = coeff(0, 0); for (Index i = 1; i < rows(); i++) s = s + coeff(i, i); return s; }
  Scalar squaredNorm() const { Scalar s = Scalar(0); for (Index k = 0; k < size(); k++) s = s + lin(k) * lin(k); return s; }
  Scalar norm() const { return std::sqrt(squaredNorm()); }
  PlainObject normalized() const { PlainObject r(*this); Scalar n = norm(); if (n > Scalar(0)) for (Index k = 0; k < r.size(); k++) r.linRef(k) = r.lin(k) / n; return r; }
  template <class O> Scalar dot(const DenseBase<O> &o) const { assert(size() == o.size()); Scalar s = Scalar(0); for (Index k = 0; k < size(); k++) s = s + lin(k) * o.lin(k); return s; }
  template <class O> Matrix<Scalar, 3, 1> cross(const DenseBase<O> &o) const {
    Matrix<Scalar, 3, 1> r;
    r.coeffRef(0, 0) = lin(1) * o.lin(2) - lin(2) * o.lin(1);
    r.coeffRef(1, 0) = lin(2) * o.lin(0) - lin(0) * o.lin(2);
    r.coeffRef(2, 0) = lin(0) * o.lin(1) - lin(1) * o.lin(0);
    return r;
  }
  template <class U> Matrix<U, RowsAtCompileTime, ColsAtCompileTime> cast() const {
    Matrix<U, RowsAtCompileTime, ColsAtCompileTime> r; r.resize(rows(), cols());
    for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) r.coeffRef(i, j) = static_cast<U>(coeff(i, j));
    return r;
  }
  Matrix<RealScalar, RowsAtCompileTime, ColsAtCompileTime> real() const {
    Matrix<RealScalar, RowsAtCompileTime, ColsAtCompileTime> r; r.resize(rows(), cols());
    for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) r.coeffRef(i, j) = internal::real_part(coeff(i, j));
    return r;
  }
  Matrix<Scalar, internal::pick(RowsAtCompileTime, ColsAtCompileTime), internal::pick(RowsAtCompileTime, ColsAtCompileTime)> asDiagonal() const {   // of a vector
    const Index n = size();
    Matrix<Scalar, internal::pick(RowsAtCompileTime, ColsAtCompileTime), internal::pick(RowsAtCompileTime, ColsAtCompileTime)> r; r.resize(n, n);
    for (Index j = 0; j < n; j++) for (Index i = 0; i < n; i++) r.coeffRef(i, j) = (i == j) ? lin(i) : Scalar(0);
    return r;
  }
  Matrix<Scalar, RowsAtCompileTime, 1> diagonal() const { Matrix<Scalar, RowsAtCompileTime, 1> r; r.resize(rows(), 1); for (Index i = 0; i < rows(); i++) r.coeffRef(i, 0) = coeff(i, i); return r; }

  template <int BR, int BC> Matrix<Scalar, BR, BC> block(Index i0, Index j0) const {
    Matrix<Scalar, BR, BC> r;
    for (Index j = 0; j < BC; j++) for (Index i = 0; i < BR; i++) r.coeffRef(i, j) = coeff(i0 + i, j0 + j);
    return r;
  }
  Matrix<Scalar, Dynamic, Dynamic> block(Index i0, Index j0, Index nr, Index nc) const {
    Matrix<Scalar, Dynamic, Dynamic> r(nr, nc);
    for (Index j = 0; j < nc; j++) for (Index i = 0; i < nr; i++) r.coeffRef(i, j) = coeff(i0 + i, j0 + j);
    return r;
  }
  Matrix<Scalar, 1, ColsAtCompileTime> row(Index i) const { Matrix<Scalar, 1, ColsAtCompileTime> r; r.resize(1, cols()); for (Index j = 0; j < cols(); j++) r.coeffRef(0, j) = coeff(i, j); return r; }
  Matrix<Scalar, RowsAtCompileTime, 1> col(Index j) const { Matrix<Scalar, RowsAtCompileTime, 1> r; r.resize(rows(), 1); for (Index i = 0; i < rows(); i++) r.coeffRef(i, 0) = coeff(i, j); return r; }

  struct RowwiseOp {
    const DenseBase &m;
    Matrix<Scalar, RowsAtCompileTime, 1> sum() const {
      Matrix<Scalar, RowsAtCompileTime, 1> r; r.resize(m.rows(), 1);
      for (Index i = 0; i < m.rows(); i++) { Scalar s = m.coeff(i, 0); for (Index j = 1; j < m.cols(); j++) s = s + m.coeff(i, j); r.coeffRef(i, 0) = s; }
      return r;
    }
  };
  RowwiseOp rowwise() const { return RowwiseOp{*this}; }

  template <class I> Scalar minCoeff(I *idx) const { Index b = 0; for (Index k = 1; k < size(); k++) if (lin(k) < lin(b)) b = k; *idx = static_cast<I>(b); return lin(b); }
  template <class I> Scalar maxCoeff(I *idx) const { Index b = 0; for (Index k = 1; k < size(); k++) if (lin(k) > lin(b)) b = k; *idx = static_cast<I>(b); return lin(b); }
  Scalar minCoeff() const { Index b; return minCoeff(&b); }
  Scalar maxCoeff() const { Index b; return maxCoeff(&b); }

  Scalar determinant() const {
    const Index n = rows(); assert(n == cols());
    if (n == 1) return coeff(0, 0);
    if (n == 2) return coeff(0, 0) * coeff(1, 1) - coeff(1, 0) * coeff(0, 1);
    if (n == 3)
      return coeff(0, 0) * (coeff(1, 1) * coeff(2, 2) - coeff(1, 2) * coeff(2, 1)) - coeff(0, 1) * (coeff(1, 0) * coeff(2, 2) - coeff(1, 2) * coeff(2, 0)) +
             coeff(0, 2) * (coeff(1, 0) * coeff(2, 1) - coeff(1, 1) * coeff(2, 0));
    std::vector<Scalar> a((size_t)(n * n)); for (Index i = 0; i < n; i++) for (Index j = 0; j < n; j++) a[(size_t)(i * n + j)] = coeff(i, j);
    Scalar det = Scalar(1);
    for (Index k = 0; k < n; k++) {
      Index p = k; for (Index i = k + 1; i < n; i++) if (std::abs(a[(size_t)(i * n + k)]) > std::abs(a[(size_t)(p * n + k)])) p = i;
      if (a[(size_t)(p * n + k)] == Scalar(0)) return Scalar(0);
      if (p != k) { for (Index j = 0; j < n; j++) std::swap(a[(size_t)(k * n + j)], a[(size_t)(p * n + j)]); det = -det; }
      det = det * a[(size_t)(k * n + k)];
      for (Index i = k + 1; i < n; i++) { Scalar l = a[(size_t)(i * n + k)] / a[(size_t)(k * n + k)]; for (Index j = k + 1; j < n; j++) a[(size_t)(i * n + j)] = a[(size_t)(i * n + j)] - l * a[(size_t)(k * n + j)]; }
    }
    return det;
  }

  // inverse(): closed forms up to 3x3 (documented Eigen behaviour for fixed sizes <= 4: cofactors / determinant), partial-pivot LU with a
  // solve against the identity above that (the algorithm class of PartialPivLU; Eigen's blocked update order is not reproduced).
  PlainObject inverse() const {
    const Index n = rows(); assert(n == cols());
    PlainObject r; r.resize(n, n);
    if (n == 1) { r.coeffRef(0, 0) = Scalar(1) / coeff(0, 0); return r; }
    if (n == 2) {
      Scalar invdet = Scalar(1) / determinant();
      r.coeffRef(0, 0) = coeff(1, 1) * invdet; r.coeffRef(1, 0) = -coeff(1, 0) * invdet; r.coeffRef(0, 1) = -coeff(0, 1) * invdet; r.coeffRef(1, 1) = coeff(0, 0) * invdet;
      return r;
    }
    if (n == 3) {
      auto cof = [&](int i, int j) { int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return coeff(i1, j1) * coeff(i2, j2) - coeff(i1, j2) * coeff(i2, j1); };
      Scalar c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
      Scalar det = (c00 * coeff(0, 0) + c10 * coeff(1, 0)) + c20 * coeff(2, 0);
      Scalar invdet = Scalar(1) / det;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.coeffRef(j, i) = cof(i, j) * invdet;
      return r;
    }
    std::vector<Scalar> lu((size_t)(n * n)); std::vector<Index> perm((size_t)n);
    auto LU = [&](Index i, Index j) -> Scalar & { return lu[(size_t)(i * n + j)]; };
    for (Index i = 0; i < n; i++) { perm[(size_t)i] = i; for (Index j = 0; j < n; j++) LU(i, j) = coeff(i, j); }
    for (Index k = 0; k < n; k++) {
      Index piv = k; RealScalar best = std::abs(LU(k, k));
      for (Index i = k + 1; i < n; i++) { RealScalar v = std::abs(LU(i, k)); if (v > best) { best = v; piv = i; } }
      if (piv != k) { for (Index j = 0; j < n; j++) std::swap(LU(k, j), LU(piv, j)); std::swap(perm[(size_t)k], perm[(size_t)piv]); }
      for (Index i = k + 1; i < n; i++) { LU(i, k) = LU(i, k) / LU(k, k); Scalar l = LU(i, k); for (Index j = k + 1; j < n; j++) LU(i, j) = LU(i, j) - l * LU(k, j); }
    }
    std::vector<Scalar> yv((size_t)n);
    for (Index c = 0; c < n; c++) {
      for (Index i = 0; i < n; i++) { Scalar s = (perm[(size_t)i] == c) ? Scalar(1) : Scalar(0); for (Index j = 0; j < i; j++) s = s - LU(i, j) * yv[(size_t)j]; yv[(size_t)i] = s; }
      for (Index i = n - 1; i >= 0; i--) { Scalar s = yv[(size_t)i]; for (Index j = i + 1; j < n; j++) s = s - LU(i, j) * r.coeff(j, c); r.coeffRef(i, c) = s / LU(i, i); }
    }
    return r;
  }

  // eulerAngles(a0,a1,a2): the documented Eigen convention (first angle in [0,pi], others in [-pi,pi]); only reached through
  // voxel_map.cpp:478 -> geoQuat_, which is not on the parity path.
  Matrix<Scalar, 3, 1> eulerAngles(Index a0, Index a1, Index a2) const {
    Matrix<Scalar, 3, 1> res;
    const Index odd = ((a0 + 1) % 3 == a1) ? 0 : 1;
    const Index i = a0, j = (a0 + 1 + odd) % 3, k = (a0 + 2 - odd) % 3;
    const Scalar pi = Scalar(3.14159265358979323846);
    if (a0 == a2) {
      res[0] = std::atan2(coeff(j, i), coeff(k, i));
      if ((odd && res[0] < Scalar(0)) || ((!odd) && res[0] > Scalar(0))) {
        if (res[0] > Scalar(0)) res[0] -= pi; else res[0] += pi;
        Scalar s2 = std::sqrt(coeff(j, i) * coeff(j, i) + coeff(k, i) * coeff(k, i));
        res[1] = -std::atan2(s2, coeff(i, i));
      } else {
        Scalar s2 = std::sqrt(coeff(j, i) * coeff(j, i) + coeff(k, i) * coeff(k, i));
        res[1] = std::atan2(s2, coeff(i, i));
      }
      Scalar s1 = std::sin(res[0]), c1 = std::cos(res[0]);
      res[2] = std::atan2(c1 * coeff(j, k) - s1 * coeff(k, k), c1 * coeff(j, j) - s1 * coeff(k, j));
    } else {
      res[0] = std::atan2(coeff(j, k), coeff(k, k));
      Scalar c2 = std::sqrt(coeff(i, i) * coeff(i, i) + coeff(i, j) * coeff(i, j));
      if ((odd && res[0] < Scalar(0)) || ((!odd) && res[0] > Scalar(0))) {
        if (res[0] > Scalar(0)) res[0] -= pi; else res[0] += pi;
        res[1] = std::atan2(-coeff(i, k), -c2);
      } else
        res[1] = std::atan2(-coeff(i, k), c2);
      Scalar s1 = std::sin(res[0]), c1 = std::cos(res[0]);
      res[2] = std::atan2(s1 * coeff(k, i) - c1 * coeff(j, i), c1 * coeff(j, j) - s1 * coeff(k, j));
    }
    if (!odd) { res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2]; }
    return res;
  }

  template <class O> bool operator==(const DenseBase<O> &o) const {
    if (rows() != o.rows() || cols() != o.cols()) return false;
    for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) if (!(coeff(i, j) == o.coeff(i, j))) return false;
    return true;
  }
  template <class O> bool operator!=(const DenseBase<O> &o) const { return !(*this == o); }
};

// ------------------------------------------------------------------------------------------------------------------------------
// write interface: the derived class additionally supplies coeffRef_(i,j) and (for plain matrices) resize
template <class D> class DenseWritable : public DenseBase<D> {
public:
  typedef DenseBase<D> Base;
  typedef typename Base::Scalar Scalar;
  typedef Eigen::Index Index;
  using Base::block; using Base::col; using Base::cols; using Base::derived; using Base::row; using Base::rows; using Base::size; using Base::diagonal;
  using Base::operator(); using Base::operator[]; using Base::x; using Base::y; using Base::z; using Base::w;

  Scalar &coeffRef(Index i, Index j) { return derived().coeffRef_(i, j); }
  Scalar &linRef(Index k) { return cols() == 1 ? coeffRef(k, 0) : (rows() == 1 ? coeffRef(0, k) : coeffRef(k % rows(), k / rows())); }
  Scalar &operator()(Index i, Index j) { return coeffRef(i, j); }
  Scalar &operator()(Index k) { return linRef(k); }
  Scalar &operator[](Index k) { return linRef(k); }
  Scalar &x() { return linRef(0); }
  Scalar &y() { return linRef(1); }
  Scalar &z() { return linRef(2); }
  Scalar &w() { return linRef(3); }

  template <class O> D &assign(const DenseBase<O> &o) {
    derived().resizeLike_(o.rows(), o.cols());
    if (rows() == o.rows() && cols() == o.cols()) {
      for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = o.coeff(i, j);
    } else {                      // Eigen transposes vectors implicitly on assignment (row <- column and vice versa)
      assert(size() == o.size() && (rows() == 1 || cols() == 1) && (o.rows() == 1 || o.cols() == 1));
      for (Index k = 0; k < size(); k++) linRef(k) = o.lin(k);
    }
    return derived();
  }
  template <class O> D &operator+=(const DenseBase<O> &o) { assert(size() == o.size()); if (rows() == o.rows()) { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = this->coeff(i, j) + o.coeff(i, j); } else for (Index k = 0; k < size(); k++) linRef(k) = this->lin(k) + o.lin(k); return derived(); }
  template <class O> D &operator-=(const DenseBase<O> &o) { assert(size() == o.size()); if (rows() == o.rows()) { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = this->coeff(i, j) - o.coeff(i, j); } else for (Index k = 0; k < size(); k++) linRef(k) = this->lin(k) - o.lin(k); return derived(); }
  template <class S, internal::if_arith<S> = 0> D &operator*=(const S &s) { const Scalar v = static_cast<Scalar>(s); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = this->coeff(i, j) * v; return derived(); }
  template <class S, internal::if_arith<S> = 0> D &operator/=(const S &s) { const Scalar v = static_cast<Scalar>(s); for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = this->coeff(i, j) / v; return derived(); }

  D &setZero() { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = Scalar(0); return derived(); }
  D &setIdentity() { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0); return derived(); }
  D &setConstant(const Scalar &v) { for (Index j = 0; j < cols(); j++) for (Index i = 0; i < rows(); i++) coeffRef(i, j) = v; return derived(); }
  void normalize() { Scalar n = this->norm(); if (n > Scalar(0)) for (Index k = 0; k < size(); k++) linRef(k) = this->lin(k) / n; }

  template <int BR, int BC> Block<D, BR, BC> block(Index i0, Index j0) { return Block<D, BR, BC>(derived(), i0, j0, BR, BC); }
  Block<D, Dynamic, Dynamic> block(Index i0, Index j0, Index nr, Index nc) { return Block<D, Dynamic, Dynamic>(derived(), i0, j0, nr, nc); }
  Block<D, 1, Base::ColsAtCompileTime> row(Index i) { return Block<D, 1, Base::ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
  Block<D, Base::RowsAtCompileTime, 1> col(Index j) { return Block<D, Base::RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }

  struct DiagonalRef {                                        // m.diagonal() = v
    D &m;
    template <class O> DiagonalRef &operator=(const DenseBase<O> &o) { assert(o.size() == std::min(m.rows(), m.cols())); for (Index i = 0; i < o.size(); i++) m.coeffRef(i, i) = o.lin(i); return *this; }
    operator Matrix<Scalar, Base::RowsAtCompileTime, 1>() const { return static_cast<const D &>(m).diagonal(); }
  };
  DiagonalRef diagonal() { return DiagonalRef{derived()}; }

  template <class S, internal::if_arith<S> = 0> CommaInitializer<D> operator<<(const S &s) { CommaInitializer<D> c(derived()); c.put(static_cast<Scalar>(s)); return c; }
  template <class O> CommaInitializer<D> operator<<(const DenseBase<O> &o) { CommaInitializer<D> c(derived()); c.put(o); return c; }
};

template <class D> class CommaInitializer {
  D &m_; Eigen::Index row_ = 0, col_ = 0, blockRows_ = 1;
public:
  typedef typename internal::traits<D>::Scalar Scalar;
  explicit CommaInitializer(D &m) : m_(m) {}
  void put(const Scalar &s) {
    if (col_ == m_.cols()) { row_ += blockRows_; col_ = 0; blockRows_ = 1; }
    assert(row_ < m_.rows() && col_ < m_.cols());
    m_.coeffRef(row_, col_++) = s;
  }
  template <class O> void put(const DenseBase<O> &o) {
    if (col_ == m_.cols()) { row_ += blockRows_; col_ = 0; blockRows_ = o.rows(); }
    else if (col_ == 0) blockRows_ = o.rows();
    assert(row_ + o.rows() <= m_.rows() && col_ + o.cols() <= m_.cols());
    for (Eigen::Index j = 0; j < o.cols(); j++) for (Eigen::Index i = 0; i < o.rows(); i++) m_.coeffRef(row_ + i, col_ + j) = o.coeff(i, j);
    col_ += o.cols();
  }
  template <class S, internal::if_arith<S> = 0> CommaInitializer &operator,(const S &s) { put(static_cast<Scalar>(s)); return *this; }
  template <class O> CommaInitializer &operator,(const DenseBase<O> &o) { put(o); return *this; }
};

// ------------------------------------------------------------------------------------------------------------------------------
namespace internal {
template <class T, int R, int C, bool Fixed = (R != Dynamic && C != Dynamic)> struct Storage;
template <class T, int R, int C> struct Storage<T, R, C, true> {
  T a[R * C];
  Storage() { for (int k = 0; k < R * C; k++) a[k] = T(); }
  Index rows() const { return R; }
  Index cols() const { return C; }
  void resize(Index r, Index c) { assert(r == R && c == C); (void)r; (void)c; }
  T *data() { return a; }
  const T *data() const { return a; }
};
template <class T, int R, int C> struct Storage<T, R, C, false> {
  std::vector<T> a; Index r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  void resize(Index r, Index c) { assert((R == Dynamic || r == R) && (C == Dynamic || c == C)); if (r != r_ || c != c_) { r_ = r; c_ = c; a.assign((size_t)(r * c), T()); } }
  T *data() { return a.data(); }
  const T *data() const { return a.data(); }
};
} // namespace internal

namespace internal {
// inner products convert to their scalar (Eigen: 1x1 expressions are implicitly convertible)
template <class D, class T, bool OneByOne> struct ScalarConvertible {};
template <class D, class T> struct ScalarConvertible<D, T, true> { operator T() const { return static_cast<const D *>(this)->coeff_(0, 0); } };
} // namespace internal

template <class T, int R, int C> class Matrix : public DenseWritable<Matrix<T, R, C>>, public internal::ScalarConvertible<Matrix<T, R, C>, T, R == 1 && C == 1> {
  internal::Storage<T, R, C> s_;
public:
  typedef DenseWritable<Matrix<T, R, C>> Base;
  typedef T Scalar;
  typedef Eigen::Index Index;
  enum { IsFixed = (R != Dynamic && C != Dynamic), IsVector = (R == 1 || C == 1) };

  Index rows_() const { return s_.rows(); }
  Index cols_() const { return s_.cols(); }
  T coeff_(Index i, Index j) const { assert(i >= 0 && i < s_.rows() && j >= 0 && j < s_.cols()); return s_.data()[i + j * s_.rows()]; }
  T &coeffRef_(Index i, Index j) { assert(i >= 0 && i < s_.rows() && j >= 0 && j < s_.cols()); return s_.data()[i + j * s_.rows()]; }
  void resizeLike_(Index r, Index c) {
    if (IsFixed) return;
    if (R == Dynamic && C == Dynamic) { s_.resize(r, c); return; }
    if ((R == 1 && c == 1 && r != 1) || (C == 1 && r == 1 && c != 1)) { s_.resize(R == Dynamic ? c : R, C == Dynamic ? r : C); return; }   // vector <- transposed vector
    s_.resize(r, c);
  }
  void resize(Index r, Index c) { s_.resize(r, c); }
  void resize(Index n) { if (C == 1) s_.resize(n, 1); else s_.resize(1, n); }
  T *data() { return s_.data(); }
  const T *data() const { return s_.data(); }

  Matrix() {}
  Matrix(const Matrix &o) : s_(o.s_) {}
  Matrix &operator=(const Matrix &o) { s_ = o.s_; return *this; }
  template <class O> Matrix(const DenseBase<O> &o) { this->assign(o); }
  template <class O> Matrix &operator=(const DenseBase<O> &o) { return this->assign(o); }

  // one argument: a size (dynamic vectors) — never a value
  template <class S, internal::if_arith<S> = 0> explicit Matrix(const S &n) { static_assert(!IsFixed || (R * C == 1), "size constructor on a fixed matrix"); if (IsFixed) s_.data()[0] = static_cast<T>(n); else resize(static_cast<Index>(n)); }
  // two arguments: (rows, cols) for dynamic matrices, (x, y) for fixed 2-vectors
  template <class A, class B, internal::if_arith<A> = 0, internal::if_arith<B> = 0> Matrix(const A &a, const B &b) {
    if (IsFixed) { assert(R * C == 2); s_.data()[0] = static_cast<T>(a); s_.data()[1] = static_cast<T>(b); }
    else s_.resize(static_cast<Index>(a), static_cast<Index>(b));
  }
  template <class A, class B, class D3, internal::if_arith<A> = 0, internal::if_arith<B> = 0, internal::if_arith<D3> = 0> Matrix(const A &a, const B &b, const D3 &c) {
    static_assert(IsFixed && R * C == 3, "three-value constructor needs a fixed 3-vector");
    s_.data()[0] = static_cast<T>(a); s_.data()[1] = static_cast<T>(b); s_.data()[2] = static_cast<T>(c);
  }
  template <class A, class B, class D3, class E, internal::if_arith<A> = 0, internal::if_arith<B> = 0, internal::if_arith<D3> = 0, internal::if_arith<E> = 0>
  Matrix(const A &a, const B &b, const D3 &c, const E &d) {
    static_assert(IsFixed && R * C == 4, "four-value constructor needs a fixed 4-vector");
    s_.data()[0] = static_cast<T>(a); s_.data()[1] = static_cast<T>(b); s_.data()[2] = static_cast<T>(c); s_.data()[3] = static_cast<T>(d);
  }

  static Matrix Zero() { Matrix m; m.setZero(); return m; }
  static Matrix Zero(Index r, Index c) { Matrix m; m.resize(r, c); m.setZero(); return m; }
  static Matrix Zero(Index n) { Matrix m; m.resize(n); m.setZero(); return m; }
  static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
  static Matrix Identity(Index r, Index c) { Matrix m; m.resize(r, c); m.setIdentity(); return m; }
  static Matrix Ones() { Matrix m; m.setConstant(T(1)); return m; }
  static Matrix Constant(const T &v) { Matrix m; m.setConstant(v); return m; }
};

template <class M, int BR, int BC> class Block : public DenseWritable<Block<M, BR, BC>> {
  M &m_; Eigen::Index i0_, j0_, nr_, nc_;
public:
  typedef typename internal::traits<M>::Scalar Scalar;
  typedef Eigen::Index Index;
  Block(M &m, Index i0, Index j0, Index nr, Index nc) : m_(m), i0_(i0), j0_(j0), nr_(nr), nc_(nc) { assert(i0 >= 0 && j0 >= 0 && i0 + nr <= m.rows() && j0 + nc <= m.cols()); }
  Block(const Block &) = default;
  Index rows_() const { return nr_; }
  Index cols_() const { return nc_; }
  Scalar coeff_(Index i, Index j) const { return static_cast<const M &>(m_).coeff(i0_ + i, j0_ + j); }
  Scalar &coeffRef_(Index i, Index j) { return m_.coeffRef(i0_ + i, j0_ + j); }
  void resizeLike_(Index, Index) {}
  Block &operator=(const Block &o) { Matrix<Scalar, BR, BC> tmp(o); return this->assign(tmp); }
  template <class O> Block &operator=(const DenseBase<O> &o) { Matrix<Scalar, internal::traits<O>::Rows, internal::traits<O>::Cols> tmp(o); return this->assign(tmp); }
};

// ------------------------------------------------------------------------------------------------------------------------------
// arithmetic (eager, ascending index order)
template <class A, class B>
Matrix<typename DenseBase<A>::Scalar, internal::pick(DenseBase<A>::RowsAtCompileTime, DenseBase<B>::RowsAtCompileTime), internal::pick(DenseBase<A>::ColsAtCompileTime, DenseBase<B>::ColsAtCompileTime)>
operator+(const DenseBase<A> &a, const DenseBase<B> &b) {
  static_assert(std::is_same<typename DenseBase<A>::Scalar, typename DenseBase<B>::Scalar>::value, "mixed scalar types");
  assert(a.rows() == b.rows() && a.cols() == b.cols());
  Matrix<typename DenseBase<A>::Scalar, internal::pick(DenseBase<A>::RowsAtCompileTime, DenseBase<B>::RowsAtCompileTime), internal::pick(DenseBase<A>::ColsAtCompileTime, DenseBase<B>::ColsAtCompileTime)> r;
  r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) r.coeffRef(i, j) = a.coeff(i, j) + b.coeff(i, j);
  return r;
}
template <class A, class B>
Matrix<typename DenseBase<A>::Scalar, internal::pick(DenseBase<A>::RowsAtCompileTime, DenseBase<B>::RowsAtCompileTime), internal::pick(DenseBase<A>::ColsAtCompileTime, DenseBase<B>::ColsAtCompileTime)>
operator-(const DenseBase<A> &a, const DenseBase<B> &b) {
  static_assert(std::is_same<typename DenseBase<A>::Scalar, typename DenseBase<B>::Scalar>::value, "mixed scalar types");
  assert(a.rows() == b.rows() && a.cols() == b.cols());
  Matrix<typename DenseBase<A>::Scalar, internal::pick(DenseBase<A>::RowsAtCompileTime, DenseBase<B>::RowsAtCompileTime), internal::pick(DenseBase<A>::ColsAtCompileTime, DenseBase<B>::ColsAtCompileTime)> r;
  r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) r.coeffRef(i, j) = a.coeff(i, j) - b.coeff(i, j);
  return r;
}
template <class A> typename DenseBase<A>::PlainObject operator-(const DenseBase<A> &a) {
  typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) r.coeffRef(i, j) = -a.coeff(i, j);
  return r;
}
template <class A, class B>
Matrix<typename DenseBase<A>::Scalar, DenseBase<A>::RowsAtCompileTime, DenseBase<B>::ColsAtCompileTime> operator*(const DenseBase<A> &a, const DenseBase<B> &b) {
  static_assert(std::is_same<typename DenseBase<A>::Scalar, typename DenseBase<B>::Scalar>::value, "mixed scalar types");
  typedef typename DenseBase<A>::Scalar T;
  assert(a.cols() == b.rows());
  Matrix<T, DenseBase<A>::RowsAtCompileTime, DenseBase<B>::ColsAtCompileTime> r; r.resize(a.rows(), b.cols());
  const Index K = a.cols();
  for (Index j = 0; j < b.cols(); j++)
    for (Index i = 0; i < a.rows(); i++) {
      if (K == 0) { r.coeffRef(i, j) = T(0); continue; }
      T s = a.coeff(i, 0) * b.coeff(0, j);
      for (Index k = 1; k < K; k++) s = s + a.coeff(i, k) * b.coeff(k, j);
      r.coeffRef(i, j) = s;
    }
  return r;
}
// scalar operands are first converted to the matrix scalar type (Eigen's promote_scalar_arg)
template <class A, class S, internal::if_arith<S> = 0> typename DenseBase<A>::PlainObject operator*(const DenseBase<A> &a, const S &s) {
  const typename DenseBase<A>::Scalar v = static_cast<typename DenseBase<A>::Scalar>(s);
  typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) r.coeffRef(i, j) = a.coeff(i, j) * v;
  return r;
}
template <class A, class S, internal::if_arith<S> = 0> typename DenseBase<A>::PlainObject operator*(const S &s, const DenseBase<A> &a) {
  const typename DenseBase<A>::Scalar v = static_cast<typename DenseBase<A>::Scalar>(s);
  typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) r.coeffRef(i, j) = v * a.coeff(i, j);
  return r;
}
template <class A, class S, internal::if_arith<S> = 0> typename DenseBase<A>::PlainObject operator/(const DenseBase<A> &a, const S &s) {
  const typename DenseBase<A>::Scalar v = static_cast<typename DenseBase<A>::Scalar>(s);
  typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); j++) for (Index i = 0; i < a.rows(); i++) r.coeffRef(i, j) = a.coeff(i, j) / v;
  return r;
}
template <class A> std::ostream &operator<<(std::ostream &os, const DenseBase<A> &a) {
  for (Index i = 0; i < a.rows(); i++) { for (Index j = 0; j < a.cols(); j++) os << (j ? " " : "") << a.coeff(i, j); if (i + 1 < a.rows()) os << "\n"; }
  return os;
}

typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 2, 1> Vector2f; typedef Matrix<float, 3, 1> Vector3f; typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<int, 2, 1> Vector2i; typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 2, 2> Matrix2f; typedef Matrix<float, 3, 3> Matrix3f; typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd; typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<float, Dynamic, Dynamic> MatrixXf; typedef Matrix<float, Dynamic, 1> VectorXf;
typedef Matrix<std::complex<double>, 3, 3> Matrix3cd; typedef Matrix<std::complex<double>, 3, 1> Vector3cd;

// ------------------------------------------------------------------------------------------------------------------------------
// Quaternion from a rotation matrix (documented Eigen conversion); only used by publishing code that is never called here.
template <class T> class Quaternion {
  T w_, x_, y_, z_;
public:
  Quaternion() : w_(1), x_(0), y_(0), z_(0) {}
  Quaternion(T w, T x, T y, T z) : w_(w), x_(x), y_(y), z_(z) {}
  template <class D> explicit Quaternion(const DenseBase<D> &m) {
    T t = m.trace();
    if (t > T(0)) { t = std::sqrt(t + T(1)); w_ = T(0.5) * t; t = T(0.5) / t; x_ = (m(2, 1) - m(1, 2)) * t; y_ = (m(0, 2) - m(2, 0)) * t; z_ = (m(1, 0) - m(0, 1)) * t; }
    else {
      int i = 0; if (m(1, 1) > m(0, 0)) i = 1; if (m(2, 2) > m(i, i)) i = 2;
      int j = (i + 1) % 3, k = (j + 1) % 3; T q[3];
      t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + T(1)); q[i] = T(0.5) * t; t = T(0.5) / t;
      w_ = (m(k, j) - m(j, k)) * t; q[j] = (m(j, i) + m(i, j)) * t; q[k] = (m(k, i) + m(i, k)) * t;
      x_ = q[0]; y_ = q[1]; z_ = q[2];
    }
  }
  T w() const { return w_; } T x() const { return x_; } T y() const { return y_; } T z() const { return z_; }
};
typedef Quaternion<double> Quaterniond; typedef Quaternion<float> Quaternionf;

// ------------------------------------------------------------------------------------------------------------------------------
// EigenSolver<Matrix3d>: the reference only feeds it symmetric 3x3 covariances (voxel_map.cpp:69-72), whose eigenvalues are real.
// Cyclic Jacobi on the symmetrised input; unit eigenvectors as columns.  Eigen's own order and signs of the eigenvectors (a property of
// its Hessenberg/QR implementation) cannot be reproduced without Eigen — comparisons of plane normals must allow a sign.
template <class M> class EigenSolver {
  typedef typename M::Scalar T;
  Matrix<std::complex<T>, M::RowsAtCompileTime, M::ColsAtCompileTime> vecs_;
  Matrix<std::complex<T>, M::RowsAtCompileTime, 1> vals_;
public:
  template <class D> explicit EigenSolver(const DenseBase<D> &in) {
    const Index n = in.rows();
    std::vector<T> a((size_t)(n * n)), v((size_t)(n * n), T(0));
    auto A = [&](Index i, Index j) -> T & { return a[(size_t)(i * n + j)]; };
    auto V = [&](Index i, Index j) -> T & { return v[(size_t)(i * n + j)]; };
    for (Index i = 0; i < n; i++) { V(i, i) = T(1); for (Index j = 0; j < n; j++) A(i, j) = T(0.5) * (in.coeff(i, j) + in.coeff(j, i)); }
    for (int sweep = 0; sweep < 64; sweep++) {
      T off = T(0); for (Index i = 0; i < n; i++) for (Index j = i + 1; j < n; j++) off += A(i, j) * A(i, j);
      if (off == T(0)) break;
      for (Index p = 0; p < n; p++)
        for (Index q = p + 1; q < n; q++) {
          if (A(p, q) == T(0)) continue;
          T theta = (A(q, q) - A(p, p)) / (T(2) * A(p, q));
          T t = (theta >= T(0) ? T(1) : T(-1)) / (std::abs(theta) + std::sqrt(theta * theta + T(1)));
          T c = T(1) / std::sqrt(t * t + T(1)), s = t * c;
          for (Index k = 0; k < n; k++) { T akp = A(k, p), akq = A(k, q); A(k, p) = c * akp - s * akq; A(k, q) = s * akp + c * akq; }
          for (Index k = 0; k < n; k++) { T apk = A(p, k), aqk = A(q, k); A(p, k) = c * apk - s * aqk; A(q, k) = s * apk + c * aqk; }
          for (Index k = 0; k < n; k++) { T vkp = V(k, p), vkq = V(k, q); V(k, p) = c * vkp - s * vkq; V(k, q) = s * vkp + c * vkq; }
        }
    }
    vecs_.resize(n, n); vals_.resize(n, 1);
    for (Index j = 0; j < n; j++) {
      vals_.coeffRef(j, 0) = std::complex<T>(A(j, j), T(0));
      T nr = T(0); for (Index i = 0; i < n; i++) nr += V(i, j) * V(i, j); nr = std::sqrt(nr);
      for (Index i = 0; i < n; i++) vecs_.coeffRef(i, j) = std::complex<T>(V(i, j) / nr, T(0));
    }
  }
  const Matrix<std::complex<T>, M::RowsAtCompileTime, M::ColsAtCompileTime> &eigenvectors() const { return vecs_; }
  const Matrix<std::complex<T>, M::RowsAtCompileTime, 1> &eigenvalues() const { return vals_; }
};

} // namespace Eigen
