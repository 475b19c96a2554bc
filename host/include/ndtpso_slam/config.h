// Drop-in configuration header of the MI355X build of libndtpso_slam.
//
// Same macro names, struct names and field order as the reference's include/ndtpso_slam/config.h
// (macros :5-25, PSOConfig :27-38, NDTPSOConfig :40-45) so that ndtpso_slam_node.cpp compiles against it
// unchanged (it prints PSO_* defaults and fills ndtpso_conf_.psoConfig.* at ndtpso_slam_node.cpp:30-49).
#ifndef NDTPSO_SLAM_AMD_CONFIG_H
#define NDTPSO_SLAM_AMD_CONFIG_H

// ---- cell statistics -------------------------------------------------------------------------
#define NDT_WINDOW_SIZE 100          // slots of the per-cell sliding window
#define NDT_MAX_POINTS_PER_CELL 50   // a build() that sees more points than this in the open slot closes it
#define LASER_IGNORE_EPSILON 0.1f    // beams shorter than 10 cm are dropped at load

// ---- behaviour switches the reference exposes as macros ---------------------------------------
#define TRANSFORM_POINTS_AT_LOAD true
#define TRANSFORM_POSE_AFTER_ALIGN (!TRANSFORM_POINTS_AT_LOAD)
#define PREFER_FRONTAL_POINTS false
#define BUILD_OCCUPANCY_GRID true    // keeps the trailing occupancy_grid_cell_size / save_occupancy_grid arguments
#define USE_LOGGER false

// ---- PSO defaults -----------------------------------------------------------------------------
#define PSO_ITERATIONS 50
#define PSO_POPULATION_SIZE 30
#define PSO_W .8
#define PSO_C1 2.
#define PSO_C2 2.
#define PSO_W_DUMPING_COEF 1.

struct PSOConfig {
  int iterations{PSO_ITERATIONS};
  int populationSize{PSO_POPULATION_SIZE};
  int num_threads{-1};  // host threads of the reference's OpenMP loop; meaningless on the GPU, kept for layout

  struct {
    double w{PSO_W};
    double c1{PSO_C1};
    double c2{PSO_C2};
    double w_dumping{PSO_W_DUMPING_COEF};
  } coeff;
};

struct NDTPSOConfig {
  PSOConfig psoConfig;
  float laserIgnoreEpsilon{LASER_IGNORE_EPSILON};
};

#endif
