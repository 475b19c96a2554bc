// Free functions of core.h, forwarded to the frame's device-backed implementation.
#include "ndtpso_slam/core.h"

// reference: pso_optimization, lib/ndtpso_slam/core.cpp:50-116
Vector3d pso_optimization(Vector3d initial_guess, NDTFrame* ref_frame, const NDTFrame* const new_frame,
                          const Array3d& deviation, const PSOConfig& pso_conf) {
  return ref_frame->optimize(initial_guess, new_frame, Vector3d(deviation[0], deviation[1], deviation[2]), pso_conf);
}

// reference: cost_function, lib/ndtpso_slam/core.cpp:26-48
double cost_function(Vector3d trans, NDTFrame* const ref_frame, const NDTFrame* const new_frame) {
  return ref_frame->cost(trans, new_frame);
}
