// oracle/src/ad.h — TEST INFRASTRUCTURE (CPU oracle). Forward-mode automatic differentiation.
//
// The reference obtains every Jacobian on the MPC path from CppAD / CppADCodeGen applied to templated
// Pinocchio + OCS2 code (qm_interface/src/dynamics/QMDynamicsAD.cpp:15-33,
// qm_interface/src/QMInterface.cpp:363-379).  The oracle restates that mechanism with a plain
// dual-number scalar: value + N directional derivatives.  The product kernels use hand-derived
// analytic derivatives instead, so the two are independent.
#pragma once
#include <cmath>

template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  static Dual seed(double c, int k) { Dual r(c); r.d[k] = 1.0; return r; }
};

template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; const double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N>& a) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N>& a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, double b) { a.v += b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, double b) { a = a * b; return a; }
template <int N> inline bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
template <int N> inline bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
template <int N> inline bool operator<(const Dual<N>& a, double b) { return a.v < b; }
template <int N> inline bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> inline bool operator>=(const Dual<N>& a, double b) { return a.v >= b; }

template <int N> inline Dual<N> sin(const Dual<N>& a) { Dual<N> r; r.v = std::sin(a.v); const double c = std::cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i]; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& a) { Dual<N> r; r.v = std::cos(a.v); const double s = -std::sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> inline Dual<N> sqrt(const Dual<N>& a) { Dual<N> r; r.v = std::sqrt(a.v); const double s = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> inline Dual<N> acos(const Dual<N>& a) { Dual<N> r; r.v = std::acos(a.v); const double s = -1.0 / std::sqrt(1.0 - a.v * a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> inline Dual<N> log(const Dual<N>& a) { Dual<N> r; r.v = std::log(a.v); const double s = 1.0 / a.v; for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }

inline double value_of(double a) { return a; }
template <int N> inline double value_of(const Dual<N>& a) { return a.v; }
using std::sin; using std::cos; using std::sqrt; using std::acos; using std::log;
