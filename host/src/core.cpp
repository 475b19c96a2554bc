// Free functions of core.h, forwarded to the frame's device-backed implementation.
#include "ndtpso_slam/core.h"

#include <cmath>

// reference: pso_optimization, lib/ndtpso_slam/core.cpp:50-116
Vector3d pso_optimization(Vector3d initial_guess, NDTFrame* ref_frame, const NDTFrame* const new_frame,
                          const Array3d& deviation, const PSOConfig& pso_conf) {
  return ref_frame->optimize(initial_guess, new_frame, Vector3d(deviation[0], deviation[1], deviation[2]), pso_conf);
}

// reference: cost_function, lib/ndtpso_slam/core.cpp:26-48
double cost_function(Vector3d trans, NDTFrame* const ref_frame, const NDTFrame* const new_frame) {
  return ref_frame->cost(trans, new_frame);
}

// geometry helpers of the public header (reference: core.h:28-31, :40-42, :45-47)
Vector2d transform_point(const Vector2d& point, const Vector3d& trans) {
  // one sincos() call, as GCC compiles the reference's cos / sin of the same argument (glibc's sincos is not always
  // bit-identical to its cos and sin)
  double c, s;
  ::sincos(trans.z(), &s, &c);
  return Vector2d(point.x() * c - point.y() * s + trans.x(), point.x() * s + point.y() * c + trans.y());
}

float index_to_angle(unsigned int idx, float step, float min_angle) { return idx * step + min_angle; }

Vector2d laser_to_point(float r, float theta) {
  const double t = double(theta);
  double c, s;
  ::sincos(t, &s, &c);
  return Vector2d(double(r) * c, double(r) * s);
}

vector<double> origin_at(Vector2d& point, double& cell_side) {
  vector<double> corner(2);
  corner[0] = std::floor(point.x() / cell_side) * cell_side;
  corner[1] = std::floor(point.y() / cell_side) * cell_side;
  return corner;
}
