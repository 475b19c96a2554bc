// Configuration surface of the MI355X build of libndtpso_slam.
//
// ndtpso_slam_node.cpp is compiled against this header unchanged, so every name it touches keeps the spelling
// and meaning it has in the reference's include/ndtpso_slam/config.h: the PSO_* / NDT_* / BUILD_OCCUPANCY_GRID
// macros it prints or tests (ndtpso_slam_node.cpp:30-49) and the PSOConfig / NDTPSOConfig aggregates it fills
// (config.h:27-45 there; field order matters because the structs cross the library boundary by value).
// The numeric defaults live in one constexpr table; the macros are thin aliases of it.
#ifndef NDTPSO_SLAM_AMD_CONFIG_H
#define NDTPSO_SLAM_AMD_CONFIG_H

namespace ndtpso_defaults {
// per-cell sliding window
constexpr int window_slots = 100;        // closed slots remembered per cell
constexpr int slot_capacity = 50;        // a build() that finds more points than this in the open slot closes it
constexpr float min_beam_range = 0.1f;   // metres; shorter beams are dropped at load
// particle swarm
constexpr int swarm_iterations = 50;
constexpr int swarm_size = 30;
constexpr double inertia = .8;
constexpr double cognitive = 2.;
constexpr double social = 2.;
constexpr double inertia_damping = 1.;
}  // namespace ndtpso_defaults

// preprocessor switches (tested with #if by the node and by ndtframe.h)
#define BUILD_OCCUPANCY_GRID 1        // keeps the trailing occupancy-grid arguments of NDTFrame / dumpMap
#define TRANSFORM_POINTS_AT_LOAD 1    // scans are moved by the frame's own pose when loaded
#define TRANSFORM_POSE_AFTER_ALIGN 0  // == !TRANSFORM_POINTS_AT_LOAD
#define PREFER_FRONTAL_POINTS 0
#define USE_LOGGER 0

// value macros
#define NDT_WINDOW_SIZE (ndtpso_defaults::window_slots)
#define NDT_MAX_POINTS_PER_CELL (ndtpso_defaults::slot_capacity)
#define LASER_IGNORE_EPSILON (ndtpso_defaults::min_beam_range)
#define PSO_ITERATIONS (ndtpso_defaults::swarm_iterations)
#define PSO_POPULATION_SIZE (ndtpso_defaults::swarm_size)
#define PSO_W (ndtpso_defaults::inertia)
#define PSO_C1 (ndtpso_defaults::cognitive)
#define PSO_C2 (ndtpso_defaults::social)
#define PSO_W_DUMPING_COEF (ndtpso_defaults::inertia_damping)

struct PSOConfig {
  int iterations = ndtpso_defaults::swarm_iterations;
  int populationSize = ndtpso_defaults::swarm_size;
  int num_threads = -1;  // the reference's OpenMP width; no meaning on the GPU, kept so the layout matches
  struct Coefficients {
    double w = ndtpso_defaults::inertia;
    double c1 = ndtpso_defaults::cognitive;
    double c2 = ndtpso_defaults::social;
    double w_dumping = ndtpso_defaults::inertia_damping;
  } coeff;
};

struct NDTPSOConfig {
  PSOConfig psoConfig;
  float laserIgnoreEpsilon = ndtpso_defaults::min_beam_range;
};

#endif
