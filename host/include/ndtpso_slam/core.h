// Free functions of the reference's include/ndtpso_slam/core.h:16-50, backed by the GPU path.
// Declarations only: the bodies live in host/src/core.cpp (the two optimiser entry points forward to the
// device through NDTFrame; the three geometry helpers are one-liners kept for source compatibility).
#ifndef NDTPSO_SLAM_AMD_CORE_H
#define NDTPSO_SLAM_AMD_CORE_H

#include <vector>

#include "ndtpso_slam/config.h"
#include "ndtpso_slam/ndtframe.h"

using Eigen::Array3d;
using Eigen::Vector2d;
using Eigen::Vector3d;
using std::vector;

// PSO over (x, y, theta) that minimises cost_function, in the reference's single-thread order on std::rand()
Vector3d pso_optimization(Vector3d initial_guess, NDTFrame* ref_frame, const NDTFrame* const new_frame,
                          const Array3d& deviation = Array3d(0, 0, 0), const PSOConfig& pso_conf = PSOConfig());

// minus the sum of the reference-frame cell Gaussians at the transformed points of new_frame
double cost_function(Vector3d trans, NDTFrame* const ref_frame, const NDTFrame* const new_frame);

// rigid 2-D motion of `point` by `trans` = (tx, ty, theta)
Vector2d transform_point(const Vector2d& point, const Vector3d& trans);
// beam index -> beam angle (fp32, as LaserScan carries it)
float index_to_angle(unsigned int idx, float step, float min_angle);
// polar -> cartesian in fp64
Vector2d laser_to_point(float r, float theta);
// lower-left corner of the cell_side-aligned square that contains `point` (reference core.h:33-36; unused by the library)
vector<double> origin_at(Vector2d& point, double& cell_side);
// glir_pso_optimization (reference core.h:21-23, marked "UNTESTED" there and never called) is not provided.

#endif
