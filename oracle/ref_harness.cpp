// ref_harness.cpp -- drives the PUBLIC API of libndtpso_slam (ndtpso_slam/ndtframe.h, ndtpso_slam/core.h) over the golden
// fixtures' cases G1-G5.  TEST INFRASTRUCTURE, like everything under oracle/.
//
// It contains no algorithm of its own: every number it returns comes out of the library it is linked with.  Two builds:
//   oracle/build_ref.sh      -> oracle/_ref/libndtpso_ref.so   the REFERENCE itself: /root/reference/lib/ndtpso_slam/
//                               {core,ndtcell,ndtframe}.cpp compiled unmodified where they lie, against a REAL Eigen3
//                               (EIGEN3_INCLUDE_DIR).  Eigen3 is not in this image, so here the script reports that and
//                               exits 3; on any box with Eigen it pins the oracle in one command (tests/test_ref_parity.py).
//   host/Makefile (replay/ref_harness_dropin.so)               the same source against the repo's drop-in libndtpso_slam
//                               (host/): the reference's own API on the HIP path, diffed against the same fixtures.
//
// Only public members are used (NDTFrame::cells, NDTCell::points_vector / mean / built / created /
// normalDistribution), so that the same file compiles against both.  The inverse covariance is private
// (ndtcell.h:66): it is recovered from three normalDistribution probes around the mean, exp(-q/2) with q = d' S d.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

#include "ndtpso_slam/core.h"
#include "ndtpso_slam/ndtframe.h"

namespace {

std::vector<float> scan(const float* r, uint32_t n) { return std::vector<float>(r, r + n); }

void single_thread() { omp_set_num_threads(1); }  // pso_optimization takes omp_get_max_threads() (core.cpp:72-79): the
                                                  // reproducible order is the single-thread one

}  // namespace

extern "C" {

// G1: loadLaser into a one-cell frame (what ndtpso_slam_node.cpp:229-230 allocates per scan); the kept points in order
int refh_scan_points(const float* ranges, uint32_t n, float amin, float ainc, float rmax, uint32_t frame_m, double* xy,
                     uint32_t cap) {
  NDTFrame f(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, double(frame_m), false);
  f.loadLaser(scan(ranges, n), amin, ainc, rmax);
  uint32_t k = 0;
  for (auto& c : f.cells)
    for (auto& p : c.points_vector[0]) {
      if (k >= cap) return -1;
      xy[2 * k] = p.x();
      xy[2 * k + 1] = p.y();
      ++k;
    }
  return (int)k;
}

// G2: loadLaser + build of a fresh frame; per created cell (ascending index): index, points in slot 0, built, mean,
// and {S00, S01 + S10, S11} of the inverse covariance (probed)
int refh_cells(const float* ranges, uint32_t n, float amin, float ainc, float rmax, uint32_t frame_m, double cell_side,
               int32_t* index, int32_t* n_slot0, int8_t* built, double* mean, double* icov3, uint32_t cap) {
  NDTFrame f(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, cell_side, true);
  f.loadLaser(scan(ranges, n), amin, ainc, rmax);
  f.build();
  uint32_t k = 0;
  for (unsigned i = 0; i < f.numOfCells; ++i) {
    NDTCell& c = f.cells[i];
    if (!c.created) continue;
    if (k >= cap) return -1;
    index[k] = (int32_t)i;
    n_slot0[k] = (int32_t)c.points_vector[0].size();
    built[k] = c.built ? 1 : 0;
    mean[2 * k] = mean[2 * k + 1] = 0.;
    icov3[3 * k] = icov3[3 * k + 1] = icov3[3 * k + 2] = 0.;
    if (c.built) {
      mean[2 * k] = c.mean.x();
      mean[2 * k + 1] = c.mean.y();
      // q(d) = d' S d = -2 ln normalDistribution(mean + d).  Per axis the step is shrunk (or grown) until q is of order
      // one -- a thin cell's S reaches 1e6 / m^2, exp(-q/2) would underflow with a fixed step --, and d is what the
      // library itself will subtract: (mean + h) - mean, not h.
      auto q = [&](double dx, double dy) {
        Vector2d p(c.mean.x() + dx, c.mean.y() + dy);
        return -2. * std::log(c.normalDistribution(p));
      };
      auto axis_step = [&](bool along_x) {
        double h = cell_side / 5.;
        for (int it = 0; it < 60; ++it) {
          const double v = along_x ? q(h, 0.) : q(0., h);
          if (!(v < 8.)) h /= 4.;
          else if (v < 0.25 && h < 64. * cell_side) h *= 2.;
          else break;
        }
        return h;
      };
      const double hx = axis_step(true), hy = axis_step(false);
      const double dx = (c.mean.x() + hx) - c.mean.x(), dy = (c.mean.y() + hy) - c.mean.y();
      const double qx = q(hx, 0.), qy = q(0., hy), qxy = q(hx, hy);
      icov3[3 * k] = qx / (dx * dx);
      icov3[3 * k + 2] = qy / (dy * dy);
      icov3[3 * k + 1] = (qxy - qx - qy) / (dx * dy);
    }
    ++k;
  }
  return (int)k;
}

// G3: cost_function over a list of poses (reference frame built from scan A at the origin, scan B in a one-cell frame)
int refh_costs(const float* ref_ranges, const float* new_ranges, uint32_t n, float amin, float ainc, float rmax,
               uint32_t frame_m, double cell_side, const double* poses, uint32_t n_poses, double* out) {
  NDTFrame ref(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, cell_side, true);
  ref.loadLaser(scan(ref_ranges, n), amin, ainc, rmax);
  NDTFrame nw(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, double(frame_m), false);
  nw.loadLaser(scan(new_ranges, n), amin, ainc, rmax);
  for (uint32_t i = 0; i < n_poses; ++i)
    out[i] = cost_function(Vector3d(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]), &ref, &nw);
  return 0;
}

// G4: pso_optimization on the srand(seed) stream, single thread; the pose and its cost
int refh_pso(const float* ref_ranges, const float* new_ranges, uint32_t n, float amin, float ainc, float rmax, uint32_t frame_m,
             double cell_side, const double* guess, const double* deviation, int iterations, int population, uint32_t seed,
             double* pose_out, double* cost_out) {
  single_thread();
  NDTFrame ref(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, cell_side, true);
  ref.loadLaser(scan(ref_ranges, n), amin, ainc, rmax);
  NDTFrame nw(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, double(frame_m), false);
  nw.loadLaser(scan(new_ranges, n), amin, ainc, rmax);
  PSOConfig cfg;
  cfg.iterations = iterations;
  cfg.populationSize = population;
  cfg.num_threads = 1;
  std::srand(seed);
  const Vector3d g(guess[0], guess[1], guess[2]);
  const Array3d dev(deviation[0], deviation[1], deviation[2]);
  const Vector3d pose = pso_optimization(g, &ref, &nw, dev, cfg);
  pose_out[0] = pose.x();
  pose_out[1] = pose.y();
  pose_out[2] = pose.z();
  *cost_out = cost_function(pose, &ref, &nw);
  return 0;
}

// G5: the node's per-scan sequence (ndtpso_slam_node.cpp:177-244) -- first scan straight into the map, every later scan
// loadLaser -> NDTFrame::align (its own deviation rule and the default PSOConfig, ndtframe.cpp:251-266) -> update --
// on ONE srand(seed) stream; the pose of every scan
int refh_sequence(const float* ranges, uint32_t n_scans, uint32_t n_beams, float amin, float ainc, float rmax,
                  uint32_t frame_m, double cell_side, uint32_t seed, double* poses_out) {
  single_thread();
  NDTFrame ref(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, cell_side, true);
  Vector3d prev = Vector3d::Zero();
  std::srand(seed);
  for (uint32_t k = 0; k < n_scans; ++k) {
    NDTFrame cur(Vector3d::Zero(), (unsigned short)frame_m, (unsigned short)frame_m, double(frame_m), false);
    cur.loadLaser(scan(ranges + (size_t)k * n_beams, n_beams), amin, ainc, rmax);
    Vector3d pose = prev;
    if (k > 0) pose = ref.align(prev, &cur);
    prev = pose;
    ref.update(pose, &cur);
    poses_out[3 * k] = pose.x();
    poses_out[3 * k + 1] = pose.y();
    poses_out[3 * k + 2] = pose.z();
  }
  return 0;
}

}  // extern "C"
