// Stand-in for the slice of Eigen / Sophus that the reference's src/ORBmatcher.cc uses (TEST INFRASTRUCTURE; see
// oracle/Makefile target `ref`): fixed-size float vectors / 3x3 matrices and SE3f / Sim3f with a unit quaternion +
// translation, with the arithmetic the real libraries perform:
//   * SE3f * point = q._transformVector(p) + t, Eigen's formula  uv = 2 (q.vec x p);  p + w uv + q.vec x uv
//   * SE3f * SE3f  = (q1 q2 renormalised, q1 * t2 + t1);  inverse = (q^-1, -(q^-1 * t))
//   * rotationMatrix() = Eigen's Quaternion::toRotationMatrix()
// fp32 throughout, compiled without contraction like the rest of the oracle.
#pragma once
#include <math.h>

namespace Eigen {
template <int N> struct Vec {
  float v[N];
  Vec() { for (int i = 0; i < N; ++i) v[i] = 0.f; }
  Vec(float a, float b) { static_assert(N == 2, ""); v[0] = a; v[1] = b; }
  Vec(float a, float b, float c) { static_assert(N == 3, ""); v[0] = a; v[1] = b; v[2] = c; }
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  Vec operator-(const Vec& o) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
  Vec operator+(const Vec& o) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i]; return r; }
  Vec operator-() const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = -v[i]; return r; }
  Vec operator/(float s) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] / s; return r; }
  Vec operator*(float s) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] * s; return r; }
  float dot(const Vec& o) const { float s = v[0] * o.v[0]; for (int i = 1; i < N; ++i) s += v[i] * o.v[i]; return s; }
  float squaredNorm() const { return dot(*this); }
  float norm() const { return sqrtf(squaredNorm()); }
  Vec cross(const Vec& o) const {
    static_assert(N == 3, "");
    return Vec(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
  }
};
template <int N> inline Vec<N> operator*(float s, const Vec<N>& a) { return a * s; }
typedef Vec<2> Vector2f;
typedef Vec<3> Vector3f;

struct Matrix3f {
  float m[9];  // row-major
  Matrix3f() { for (int i = 0; i < 9; ++i) m[i] = 0.f; }
  static Matrix3f Identity() { Matrix3f r; r.m[0] = r.m[4] = r.m[8] = 1.f; return r; }
  float& operator()(int i, int j) { return m[3 * i + j]; }
  float operator()(int i, int j) const { return m[3 * i + j]; }
  Vector3f operator*(const Vector3f& p) const {
    return Vector3f(m[0] * p(0) + m[1] * p(1) + m[2] * p(2), m[3] * p(0) + m[4] * p(1) + m[5] * p(2), m[6] * p(0) + m[7] * p(1) + m[8] * p(2));
  }
  // fixed-size lazy product: a coefficient is the redux of three products, which Eigen's unroller evaluates as p0 + (p1 + p2)
  Matrix3f operator*(const Matrix3f& o) const {
    Matrix3f r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const float p0 = m[3 * i] * o.m[j], p1 = m[3 * i + 1] * o.m[3 + j], p2 = m[3 * i + 2] * o.m[6 + j];
        r.m[3 * i + j] = p0 + (p1 + p2);
      }
    return r;
  }
  // internal::compute_inverse<Matrix3f, Matrix3f, 3>: cofactors, the determinant from the cofactors of column 0
  Matrix3f inverse() const {
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
    };
    const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    const float id = 1.0f / det;
    Matrix3f r;
    r.m[0] = c0 * id; r.m[1] = c1 * id; r.m[2] = c2 * id;
    r.m[3] = cof(0, 1) * id; r.m[4] = cof(1, 1) * id; r.m[5] = cof(2, 1) * id;
    r.m[6] = cof(0, 2) * id; r.m[7] = cof(1, 2) * id; r.m[8] = cof(2, 2) * id;
    return r;
  }
  Matrix3f transpose() const { Matrix3f r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = m[3 * j + i]; return r; }
};

struct Quaternionf {
  // coefficients behind accessor functions, as in Eigen (the shims read unit_quaternion().x() etc.)
  float qx, qy, qz, qw;
  Quaternionf() : qx(0), qy(0), qz(0), qw(1) {}
  Quaternionf(float w_, float x_, float y_, float z_) : qx(x_), qy(y_), qz(z_), qw(w_) {}
  float& x() { return qx; } float& y() { return qy; } float& z() { return qz; } float& w() { return qw; }
  float x() const { return qx; } float y() const { return qy; } float z() const { return qz; } float w() const { return qw; }
  Vector3f vec() const { return Vector3f(qx, qy, qz); }
  // QuaternionBase::_transformVector
  Vector3f operator*(const Vector3f& p) const {
    Vector3f uv = vec().cross(p);
    uv = uv + uv;
    return p + qw * uv + vec().cross(uv);
  }
  // quaternion product (Eigen's internal::quat_product, scalar path)
  Quaternionf operator*(const Quaternionf& b) const {
    return Quaternionf(qw * b.qw - qx * b.qx - qy * b.qy - qz * b.qz, qw * b.qx + qx * b.qw + qy * b.qz - qz * b.qy,
                       qw * b.qy + qy * b.qw + qz * b.qx - qx * b.qz, qw * b.qz + qz * b.qw + qx * b.qy - qy * b.qx);
  }
  Quaternionf conjugate() const { return Quaternionf(qw, -qx, -qy, -qz); }
  float squaredNorm() const { return qx * qx + qy * qy + qz * qz + qw * qw; }
  void normalize() { const float n = sqrtf(squaredNorm()); qx /= n; qy /= n; qz /= n; qw /= n; }
  // QuaternionBase::toRotationMatrix
  Matrix3f toRotationMatrix() const {
    Matrix3f r;
    const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
    const float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy,
                tzz = tz * qz;
    r(0, 0) = 1.f - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1.f - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1.f - (txx + tyy);
    return r;
  }
  // Quaternion(Matrix3) : internal::quaternionbase_assign_impl (Shepperd's method as Eigen writes it)
  static Quaternionf FromMatrix(const Matrix3f& a) {
    Quaternionf q;
    float t = a(0, 0) + a(1, 1) + a(2, 2);
    if (t > 0.f) {
      t = sqrtf(t + 1.f);
      q.qw = 0.5f * t;
      t = 0.5f / t;
      q.qx = (a(2, 1) - a(1, 2)) * t; q.qy = (a(0, 2) - a(2, 0)) * t; q.qz = (a(1, 0) - a(0, 1)) * t;
    } else {
      int i = 0;
      if (a(1, 1) > a(0, 0)) i = 1;
      if (a(2, 2) > a(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = sqrtf(a(i, i) - a(j, j) - a(k, k) + 1.f);
      float c[3];
      c[i] = 0.5f * t;
      t = 0.5f / t;
      q.qw = (a(k, j) - a(j, k)) * t;
      c[j] = (a(j, i) + a(i, j)) * t;
      c[k] = (a(k, i) + a(i, k)) * t;
      q.qx = c[0]; q.qy = c[1]; q.qz = c[2];
    }
    return q;
  }
};
}  // namespace Eigen

namespace Sophus {
template <class T> struct SO3;
template <> struct SO3<float> {
  // SO3::hat: the skew-symmetric matrix of a 3-vector
  static Eigen::Matrix3f hat(const Eigen::Vector3f& w) {
    Eigen::Matrix3f r;
    r(0, 1) = -w(2); r(0, 2) = w(1); r(1, 0) = w(2); r(1, 2) = -w(0); r(2, 0) = -w(1); r(2, 1) = w(0);
    return r;
  }
};
typedef SO3<float> SO3f;
template <class T> class SE3;
template <> class SE3<float> {
 public:
  SE3() {}
  SE3(const Eigen::Matrix3f& R, const Eigen::Vector3f& t) : q_(Eigen::Quaternionf::FromMatrix(R)), t_(t) { q_.normalize(); }
  SE3(const Eigen::Quaternionf& q, const Eigen::Vector3f& t) : q_(q), t_(t) { q_.normalize(); }
  const Eigen::Quaternionf& unit_quaternion() const { return q_; }
  Eigen::Matrix3f rotationMatrix() const { return q_.toRotationMatrix(); }
  const Eigen::Vector3f& translation() const { return t_; }
  Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return q_ * p + t_; }
  SE3 operator*(const SE3& o) const {
    SE3 r;
    r.q_ = q_ * o.q_;
    // SO3 product: renormalise only when the squared norm drifted (Sophus so3.hpp operator*)
    const float sn = r.q_.squaredNorm();
    if (sn != 1.f) { const float s = 2.f / (1.f + sn); r.q_.qx *= s; r.q_.qy *= s; r.q_.qz *= s; r.q_.qw *= s; }
    r.t_ = q_ * o.t_ + t_;
    return r;
  }
  SE3 inverse() const {
    SE3 r;
    r.q_ = q_.conjugate();
    r.t_ = r.q_ * (-t_);
    return r;
  }
 private:
  Eigen::Quaternionf q_;
  Eigen::Vector3f t_;
};
typedef SE3<float> SE3f;

template <class T> class Sim3;
template <> class Sim3<float> {
 public:
  Sim3() : s_(1.f) {}
  Sim3(float s, const Eigen::Quaternionf& q, const Eigen::Vector3f& t) : q_(q), t_(t), s_(s) {}
  Eigen::Matrix3f rotationMatrix() const { return q_.toRotationMatrix(); }
  const Eigen::Vector3f& translation() const { return t_; }
  float scale() const { return s_; }
  Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return s_ * (q_ * p) + t_; }
  Sim3 inverse() const {
    Sim3 r;
    r.q_ = q_.conjugate();
    r.s_ = 1.f / s_;
    r.t_ = -(r.s_ * (r.q_ * t_));
    return r;
  }
 private:
  Eigen::Quaternionf q_;
  Eigen::Vector3f t_;
  float s_;
};
typedef Sim3<float> Sim3f;
}  // namespace Sophus
