// Vector types of the public API.  With Eigen3 installed (the reference's only hard dependency,
// CMakeLists.txt:30) the real types are used and the node compiles unchanged; without it (this build
// image has no Eigen) a minimal value-type fallback with the handful of members the API needs is provided
// so the library, the replay harness and the tests still build.
#ifndef NDTPSO_SLAM_AMD_LINALG_H
#define NDTPSO_SLAM_AMD_LINALG_H

// Which of the two it is decides the layout of NDTCell and NDTFrame (Eigen's Vector2d is 16-byte aligned, the fallback
// 8), so the library and everything compiled against its headers must agree.  The choice is therefore
//   - forced by NDTPSO_USE_EIGEN=0/1 when that is defined (the CMake build bakes it into the target's PUBLIC compile
//     definitions, so consumers inherit the library's choice), else taken from __has_include, and
//   - recorded in the library as a symbol, ndtpso_slam_abi_with_eigen or ndtpso_slam_abi_without_eigen, that every
//     translation unit including this header references: a node built with Eigen against a library built without
//     (or the reverse) fails at LINK time with that name in the message instead of corrupting `cells` silently.
#if defined(NDTPSO_USE_EIGEN)
#if NDTPSO_USE_EIGEN
#define NDTPSO_HAVE_EIGEN 1
#endif
#elif defined(__has_include)
#if __has_include(<eigen3/Eigen/Core>)
#define NDTPSO_HAVE_EIGEN 1
#endif
#endif

#ifdef NDTPSO_HAVE_EIGEN
#define NDTPSO_ABI_TAG ndtpso_slam_abi_with_eigen
#else
#define NDTPSO_ABI_TAG ndtpso_slam_abi_without_eigen
#endif
extern "C" int NDTPSO_ABI_TAG;  // defined by libndtpso_slam (host/src/device.cpp) for the choice IT was built with
namespace ndtpso_abi {
__attribute__((used)) static int* const linalg_choice = &NDTPSO_ABI_TAG;
}

#ifdef NDTPSO_HAVE_EIGEN
#include <eigen3/Eigen/Core>
#else
#include <cmath>
#include <cstddef>

namespace Eigen {

template <int N>
struct SmallVec {
  double v[N];
  SmallVec() { for (int i = 0; i < N; ++i) v[i] = 0.; }
  SmallVec(double a, double b) { static_assert(N == 2, "2 coefficients"); v[0] = a; v[1] = b; }
  SmallVec(double a, double b, double c) { static_assert(N == 3, "3 coefficients"); v[0] = a; v[1] = b; v[2] = c; }
  static SmallVec Zero() { return SmallVec(); }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { static_assert(N >= 3, "no z"); return v[2]; }
  double& x() { return v[0]; }
  double& y() { return v[1]; }
  double& z() { static_assert(N >= 3, "no z"); return v[2]; }
  double operator[](std::size_t i) const { return v[i]; }
  double& operator[](std::size_t i) { return v[i]; }
  double operator()(std::size_t i) const { return v[i]; }
  SmallVec operator+(const SmallVec& o) const { SmallVec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i]; return r; }
  SmallVec operator-(const SmallVec& o) const { SmallVec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
  SmallVec operator*(double s) const { SmallVec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] * s; return r; }
  SmallVec operator/(double s) const { SmallVec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] / s; return r; }
  SmallVec& operator+=(const SmallVec& o) { for (int i = 0; i < N; ++i) v[i] += o.v[i]; return *this; }
  SmallVec& operator-=(const SmallVec& o) { for (int i = 0; i < N; ++i) v[i] -= o.v[i]; return *this; }
  const SmallVec& array() const { return *this; }
  SmallVec abs() const { SmallVec r; for (int i = 0; i < N; ++i) r.v[i] = std::fabs(v[i]); return r; }
  bool isZero(double prec = 1e-12) const { for (int i = 0; i < N; ++i) if (std::fabs(v[i]) > prec) return false; return true; }
};

using Vector2d = SmallVec<2>;
using Vector3d = SmallVec<3>;
using Array2d = SmallVec<2>;
using Array3d = SmallVec<3>;

}  // namespace Eigen
#endif

#endif
