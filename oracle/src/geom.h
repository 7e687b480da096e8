// oracle/src/geom.h — TEST INFRASTRUCTURE (CPU oracle). Templated 3-D algebra + model container.
#pragma once
#include <vector>
#include <cstring>
#include "ad.h"
#include "../../include/qmhip_layout.h"

template <class T> struct V3 { T x[3]; T& operator[](int i) { return x[i]; } const T& operator[](int i) const { return x[i]; } };
template <class T> struct M3 { T m[9]; T& operator()(int i, int j) { return m[3 * i + j]; } const T& operator()(int i, int j) const { return m[3 * i + j]; } };

template <class T> inline V3<T> v3(const T& a, const T& b, const T& c) { V3<T> r; r[0] = a; r[1] = b; r[2] = c; return r; }
template <class T> inline V3<T> v3d(const double* p) { V3<T> r; for (int i = 0; i < 3; ++i) r[i] = T(p[i]); return r; }
template <class T> inline M3<T> m3d(const double* p) { M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = T(p[i]); return r; }
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { V3<T> r; for (int i = 0; i < 3; ++i) r[i] = a[i] + b[i]; return r; }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { V3<T> r; for (int i = 0; i < 3; ++i) r[i] = a[i] - b[i]; return r; }
template <class T> inline V3<T> operator*(const V3<T>& a, const T& s) { V3<T> r; for (int i = 0; i < 3; ++i) r[i] = a[i] * s; return r; }
template <class T> inline V3<T> scale(const V3<T>& a, double s) { V3<T> r; for (int i = 0; i < 3; ++i) r[i] = a[i] * s; return r; }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) { return v3<T>(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]); }
template <class T> inline V3<T> operator*(const M3<T>& A, const V3<T>& b) { V3<T> r; for (int i = 0; i < 3; ++i) r[i] = A(i, 0) * b[0] + A(i, 1) * b[1] + A(i, 2) * b[2]; return r; }
template <class T> inline M3<T> operator*(const M3<T>& A, const M3<T>& B) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j); return r; }
template <class T> inline M3<T> transpose(const M3<T>& A) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = A(j, i); return r; }
template <class T> inline M3<T> skew(const V3<T>& v) { M3<T> r; r(0, 0) = T(0.0); r(0, 1) = -v[2]; r(0, 2) = v[1]; r(1, 0) = v[2]; r(1, 1) = T(0.0); r(1, 2) = -v[0]; r(2, 0) = -v[1]; r(2, 1) = v[0]; r(2, 2) = T(0.0); return r; }
template <class T> inline M3<T> scale(const M3<T>& A, double s) { M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = A.m[i] * s; return r; }
template <class T> inline M3<T> inverse(const M3<T>& A) {   // cofactor inverse (Eigen's fixed 3x3 .inverse())
  M3<T> c;
  c(0, 0) = A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1); c(0, 1) = A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2); c(0, 2) = A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1);
  c(1, 0) = A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2); c(1, 1) = A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0); c(1, 2) = A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2);
  c(2, 0) = A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0); c(2, 1) = A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1); c(2, 2) = A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0);
  T det = A(0, 0) * c(0, 0) + A(0, 1) * c(1, 0) + A(0, 2) * c(2, 0);
  T id = T(1.0) / det;
  M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = c.m[i] * id; return r;
}

// R = Rz(z) Ry(y) Rx(x)   [upstream ocs2_robotic_tools getRotationMatrixFromZyxEulerAngles]
template <class T> inline M3<T> rotZyx(const T& z, const T& y, const T& x) {
  T sz = sin(z), cz = cos(z), sy = sin(y), cy = cos(y), sx = sin(x), cx = cos(x);
  M3<T> R;
  R(0, 0) = cz * cy; R(0, 1) = cz * sy * sx - sz * cx; R(0, 2) = cz * sy * cx + sz * sx;
  R(1, 0) = sz * cy; R(1, 1) = sz * sy * sx + cz * cx; R(1, 2) = sz * sy * cx - cz * sx;
  R(2, 0) = -sy;     R(2, 1) = cy * sx;                R(2, 2) = cy * cx;
  return R;
}
// omega_world = E(theta) * theta_dot   [upstream getMappingFromEulerAnglesZyxDerivativeToGlobalAngularVelocity]
template <class T> inline M3<T> eulerZyxE(const T& z, const T& y) {
  T sz = sin(z), cz = cos(z), sy = sin(y), cy = cos(y);
  M3<T> E;
  E(0, 0) = T(0.0); E(0, 1) = -sz; E(0, 2) = cy * cz;
  E(1, 0) = T(0.0); E(1, 1) = cz;  E(1, 2) = cy * sz;
  E(2, 0) = T(1.0); E(2, 1) = T(0.0); E(2, 2) = -sy;
  return E;
}
// rotation about a unit axis (Rodrigues)
template <class T> inline M3<T> axisAngle(const double* a, const T& q) {
  T s = sin(q), c = cos(q); T oc = T(1.0) - c;
  M3<T> R;
  R(0, 0) = c + oc * (a[0] * a[0]);        R(0, 1) = oc * (a[0] * a[1]) - s * a[2]; R(0, 2) = oc * (a[0] * a[2]) + s * a[1];
  R(1, 0) = oc * (a[1] * a[0]) + s * a[2]; R(1, 1) = c + oc * (a[1] * a[1]);        R(1, 2) = oc * (a[1] * a[2]) - s * a[0];
  R(2, 0) = oc * (a[2] * a[0]) - s * a[1]; R(2, 1) = oc * (a[2] * a[1]) + s * a[0]; R(2, 2) = c + oc * (a[2] * a[2]);
  return R;
}

struct Model {
  double mb[MB_SIZE];
  double st[ST_SIZE];
  int parent(int j) const { return (int)mb[MB_PARENT + j]; }
  const double* jR(int j) const { return mb + MB_JR + 9 * j; }
  const double* jp(int j) const { return mb + MB_JP + 3 * j; }
  const double* axis(int j) const { return mb + MB_AXIS + 3 * j; }
  double mass(int b) const { return mb[MB_MASS + b]; }
  const double* com(int b) const { return mb + MB_COM + 3 * b; }
  const double* inertia(int b) const { return mb + MB_INERTIA + 9 * b; }
  int fparent(int f) const { return (int)mb[MB_FPARENT + f]; }
  const double* fR(int f) const { return mb + MB_FR + 9 * f; }
  const double* fp(int f) const { return mb + MB_FP + 3 * f; }
  double robotMass() const { return mb[MB_ROBOTMASS]; }
};

// forward kinematics of the fixed tree; q(24) = [p_base, zyx, joints]
template <class T> struct Kin {
  M3<T> R[QM_NB]; V3<T> p[QM_NB];      // body (joint) frames in world
  M3<T> fR[QM_NF]; V3<T> fp[QM_NF];    // frames of interest in world
};
template <class T> inline void forwardKinematics(const Model& M, const T* q, Kin<T>& k) {
  k.R[0] = rotZyx(q[3], q[4], q[5]);
  k.p[0] = v3<T>(q[0], q[1], q[2]);
  for (int j = 0; j < QM_NJ; ++j) {
    const int par = M.parent(j);
    k.R[j + 1] = k.R[par] * (m3d<T>(M.jR(j)) * axisAngle<T>(M.axis(j), q[6 + j]));
    k.p[j + 1] = k.p[par] + k.R[par] * v3d<T>(M.jp(j));
  }
  for (int f = 0; f < QM_NF; ++f) {
    const int b = M.fparent(f);
    k.fR[f] = k.R[b] * m3d<T>(M.fR(f));
    k.fp[f] = k.p[b] + k.R[b] * v3d<T>(M.fp(f));
  }
}
