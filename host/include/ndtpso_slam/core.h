// Free functions of the reference's include/ndtpso_slam/core.h:16-50, backed by the GPU path.
#ifndef NDTPSO_SLAM_AMD_CORE_H
#define NDTPSO_SLAM_AMD_CORE_H

#include <cmath>
#include <vector>

#include "ndtpso_slam/config.h"
#include "ndtpso_slam/ndtframe.h"

using Eigen::Array3d;
using Eigen::Vector2d;
using Eigen::Vector3d;
using std::vector;

// PSO over (x, y, theta) minimising cost_function; the reference's single-thread order on the std::rand() stream
Vector3d pso_optimization(Vector3d initial_guess, NDTFrame* ref_frame, const NDTFrame* const new_frame,
                          const Array3d& deviation = Array3d(0, 0, 0), const PSOConfig& pso_conf = PSOConfig());

// -sum of the reference-frame cell Gaussians at the transformed points of new_frame
double cost_function(Vector3d trans, NDTFrame* const ref_frame, const NDTFrame* const new_frame);

// small inline helpers that are part of the public header of the reference (core.h:28-47)
inline Vector2d transform_point(const Vector2d& point, const Vector3d& trans) {
  const double c = std::cos(trans.z()), s = std::sin(trans.z());
  return Vector2d(point.x() * c - point.y() * s + trans.x(), point.x() * s + point.y() * c + trans.y());
}
inline float index_to_angle(unsigned int idx, float step, float min_angle) { return idx * step + min_angle; }
inline Vector2d laser_to_point(float r, float theta) {
  return Vector2d(double(r) * std::cos(double(theta)), double(r) * std::sin(double(theta)));
}

#endif
