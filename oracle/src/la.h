// oracle/src/la.h — TEST INFRASTRUCTURE (CPU oracle). Small dense row-major linear algebra (f64).
#pragma once
#include <vector>
#include <cmath>
#include <cassert>
#include <algorithm>
#include <stdexcept>

struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
  static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
};
typedef std::vector<double> Vec;

inline Mat matmul(const Mat& A, const Mat& B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; ++i)
    for (int k = 0; k < A.c; ++k) { const double aik = A(i, k); if (aik == 0.0) continue; for (int j = 0; j < B.c; ++j) C(i, j) += aik * B(k, j); }
  return C;
}
inline Mat transpose(const Mat& A) { Mat T(A.c, A.r); for (int i = 0; i < A.r; ++i) for (int j = 0; j < A.c; ++j) T(j, i) = A(i, j); return T; }
inline Mat matmulTN(const Mat& A, const Mat& B) { return matmul(transpose(A), B); }   // Aᵀ B
inline Mat add(const Mat& A, const Mat& B) { assert(A.r == B.r && A.c == B.c); Mat C = A; for (size_t i = 0; i < C.a.size(); ++i) C.a[i] += B.a[i]; return C; }
inline Mat sub(const Mat& A, const Mat& B) { assert(A.r == B.r && A.c == B.c); Mat C = A; for (size_t i = 0; i < C.a.size(); ++i) C.a[i] -= B.a[i]; return C; }
inline Mat scaled(const Mat& A, double s) { Mat C = A; for (auto& v : C.a) v *= s; return C; }
inline Vec matvec(const Mat& A, const Vec& x) { assert(A.c == (int)x.size()); Vec y(A.r, 0.0); for (int i = 0; i < A.r; ++i) { double s = 0; for (int j = 0; j < A.c; ++j) s += A(i, j) * x[j]; y[i] = s; } return y; }
inline Vec matvecT(const Mat& A, const Vec& x) { assert(A.r == (int)x.size()); Vec y(A.c, 0.0); for (int i = 0; i < A.r; ++i) for (int j = 0; j < A.c; ++j) y[j] += A(i, j) * x[i]; return y; }
inline Vec vadd(const Vec& a, const Vec& b) { Vec c = a; for (size_t i = 0; i < c.size(); ++i) c[i] += b[i]; return c; }
inline Vec vsub(const Vec& a, const Vec& b) { Vec c = a; for (size_t i = 0; i < c.size(); ++i) c[i] -= b[i]; return c; }
inline Vec vscaled(const Vec& a, double s) { Vec c = a; for (auto& v : c) v *= s; return c; }
inline double vdot(const Vec& a, const Vec& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }
inline Mat vstack(const Mat& A, const Mat& B) {
  if (A.r == 0) return B;
  if (B.r == 0) return A;
  assert(A.c == B.c);
  Mat C(A.r + B.r, A.c);
  std::copy(A.a.begin(), A.a.end(), C.a.begin());
  std::copy(B.a.begin(), B.a.end(), C.a.begin() + A.a.size());
  return C;
}
inline Vec vcat(const Vec& a, const Vec& b) { Vec c = a; c.insert(c.end(), b.begin(), b.end()); return c; }

// Cholesky A = L Lᵀ (lower); returns false if not positive definite
inline bool cholesky(const Mat& A, Mat& L) {
  const int n = A.r; L = Mat(n, n);
  for (int j = 0; j < n; ++j) {
    double d = A(j, j); for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0)) return false;
    L(j, j) = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) { double s = A(i, j); for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k); L(i, j) = s / L(j, j); }
  }
  return true;
}
// Cholesky that survives an indefinite matrix the way [upstream, recalled] BLASFEO's dpotrf kernels do (what HPIPM's Riccati factorisation runs on): a pivot that is
// not positive gets a ZERO diagonal entry and a zero reciprocal, i.e. its whole column of L is zero — the variable drops out of the factorisation and of every later
// pivot's update.  Together with cholSolveInPlace below (a zero diagonal yields a zero component) the solve returns x_j = 0 for such a j and, for the others, the solution of
// the system with row and column j deleted.  Returns the number of pivots treated that way (0: A is positive definite and L is the ordinary factor).
inline int choleskyZeroPivots(const Mat& A, Mat& L) {
  const int n = A.r; L = Mat(n, n); int nz = 0;
  for (int j = 0; j < n; ++j) {
    double d = A(j, j); for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0)) { ++nz; continue; }                      // L(j, j) = 0 and L(i, j) = 0 for i > j
    L(j, j) = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) { double s = A(i, j); for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k); L(i, j) = s / L(j, j); }
  }
  return nz;
}
inline void cholSolveInPlace(const Mat& L, double* b) {   // solves L Lᵀ x = b; a zero diagonal entry (choleskyZeroPivots) stands for a zero reciprocal: that component is 0
  const int n = L.r;
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L(i, k) * b[k]; b[i] = (L(i, i) != 0.0) ? s / L(i, i) : 0.0; }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L(k, i) * b[k]; b[i] = (L(i, i) != 0.0) ? s / L(i, i) : 0.0; }
}
inline Mat cholSolve(const Mat& L, const Mat& B) {
  Mat X(B.r, B.c); Vec col(B.r);
  for (int j = 0; j < B.c; ++j) { for (int i = 0; i < B.r; ++i) col[i] = B(i, j); cholSolveInPlace(L, col.data()); for (int i = 0; i < B.r; ++i) X(i, j) = col[i]; }
  return X;
}
inline Vec cholSolve(const Mat& L, const Vec& b) { Vec x = b; cholSolveInPlace(L, x.data()); return x; }

// Householder QR of A (m x n, m >= n): returns Q (m x m) explicitly and R (n x n upper) : A = Q[:, :n] R
inline void householderQR(const Mat& A, Mat& Q, Mat& R) {
  const int m = A.r, n = A.c;
  Mat W = A; Q = Mat::identity(m);
  for (int k = 0; k < n; ++k) {
    double norm = 0; for (int i = k; i < m; ++i) norm += W(i, k) * W(i, k); norm = std::sqrt(norm);
    if (norm == 0.0) continue;
    const double alpha = W(k, k) > 0 ? -norm : norm;
    Vec v(m, 0.0); for (int i = k; i < m; ++i) v[i] = W(i, k); v[k] -= alpha;
    double vn = 0; for (int i = k; i < m; ++i) vn += v[i] * v[i];
    if (vn == 0.0) continue;
    for (int j = k; j < n; ++j) { double s = 0; for (int i = k; i < m; ++i) s += v[i] * W(i, j); s *= 2.0 / vn; for (int i = k; i < m; ++i) W(i, j) -= s * v[i]; }
    for (int j = 0; j < m; ++j) { double s = 0; for (int i = k; i < m; ++i) s += Q(j, i) * v[i]; s *= 2.0 / vn; for (int i = k; i < m; ++i) Q(j, i) -= s * v[i]; }
  }
  R = Mat(n, n); for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) R(i, j) = W(i, j);
}

// LU with partial pivoting solve (general square)
inline Vec luSolve(Mat A, Vec b) {
  const int n = A.r;
  for (int k = 0; k < n; ++k) {
    int p = k; for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, k)) > std::fabs(A(p, k))) p = i;
    if (A(p, k) == 0.0) throw std::runtime_error("luSolve: singular");
    if (p != k) { for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j)); std::swap(b[k], b[p]); }
    for (int i = k + 1; i < n; ++i) { const double f = A(i, k) / A(k, k); if (f == 0.0) continue; for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j); b[i] -= f * b[k]; }
  }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int j = i + 1; j < n; ++j) s -= A(i, j) * b[j]; b[i] = s / A(i, i); }
  return b;
}

// Null-space basis of A (m x n): orthonormal columns spanning ker(A), via Householder QR with column
// pivoting of Aᵀ.  (Reference: Eigen FullPivLU::kernel() at qm_wbc/src/HoQp.cpp:126-133 — the basis is
// not unique; x is basis independent)
inline Mat nullSpace(const Mat& A, double tol = 1e-9) {
  const int m = A.r, n = A.c;
  Mat W = transpose(A);            // n x m
  Mat Q = Mat::identity(n);
  std::vector<int> perm(m); for (int i = 0; i < m; ++i) perm[i] = i;
  int rank = 0; double maxnorm0 = 0;
  const int steps = std::min(n, m);
  for (int k = 0; k < steps; ++k) {
    int best = k; double bn = -1;
    for (int j = k; j < m; ++j) { double s = 0; for (int i = k; i < n; ++i) s += W(i, j) * W(i, j); if (s > bn) { bn = s; best = j; } }
    if (k == 0) maxnorm0 = std::sqrt(bn);
    if (std::sqrt(bn) <= tol * std::max(1.0, maxnorm0)) break;
    if (best != k) for (int i = 0; i < n; ++i) std::swap(W(i, k), W(i, best));
    double norm = std::sqrt(bn);
    const double alpha = W(k, k) > 0 ? -norm : norm;
    Vec v(n, 0.0); for (int i = k; i < n; ++i) v[i] = W(i, k); v[k] -= alpha;
    double vn = 0; for (int i = k; i < n; ++i) vn += v[i] * v[i];
    if (vn > 0) {
      for (int j = k; j < m; ++j) { double s = 0; for (int i = k; i < n; ++i) s += v[i] * W(i, j); s *= 2.0 / vn; for (int i = k; i < n; ++i) W(i, j) -= s * v[i]; }
      for (int j = 0; j < n; ++j) { double s = 0; for (int i = k; i < n; ++i) s += Q(j, i) * v[i]; s *= 2.0 / vn; for (int i = k; i < n; ++i) Q(j, i) -= s * v[i]; }
    }
    ++rank;
  }
  Mat Z(n, n - rank);
  for (int i = 0; i < n; ++i) for (int j = rank; j < n; ++j) Z(i, j - rank) = Q(i, j);
  return Z;
}
