// Exercises the public NDTFrame / core.h API beyond what the node uses (multi-cell frames loaded directly,
// addPoint, explicit build, cost_function, pso_optimization, a second loadLaser into the same frame, transform,
// resetCells, dumpMap) and prints every number with full precision.  tests/test_host_library.py runs it with
// resident and host-side frames and requires identical output, and checks the first part against the oracle.
//   usage: frame_api_check scans.bin dump_prefix       (file format: see node_replay.cpp; needs >= 3 scans)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndtpso_slam/core.h"
#include "ndtpso_slam/ndtframe.h"

static void print_cells(const char* tag, NDTFrame& f) {
  f.syncHostView();
  unsigned created = 0, built = 0;
  double sx = 0., sy = 0.;
  for (const NDTCell& c : f.cells) {
    created += c.created ? 1u : 0u;
    if (c.built) {
      ++built;
      sx += c.mean.x();
      sy += c.mean.y();
    }
  }
  std::vector<double> xy;
  f.collectPoints(xy);
  std::printf("%s created %u built %u mean_sum %.17g %.17g slot0_points %zu\n", tag, created, built, sx, sy, xy.size() / 2);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t n_scans = 0, n_beams = 0;
  float amin = 0, ainc = 0, rmax = 0;
  if (std::fread(&n_scans, 4, 1, f) != 1 || std::fread(&n_beams, 4, 1, f) != 1 || std::fread(&amin, 4, 1, f) != 1 ||
      std::fread(&ainc, 4, 1, f) != 1 || std::fread(&rmax, 4, 1, f) != 1 || n_scans < 3)
    return 2;
  std::vector<std::vector<float>> scans((size_t)n_scans, std::vector<float>((size_t)n_beams));
  for (auto& s : scans)
    if (std::fread(s.data(), 4, (size_t)n_beams, f) != (size_t)n_beams) return 2;
  std::fclose(f);
  ndtpso_slam_device_init();

  NDTFrame a(Vector3d::Zero(), 60, 60, 0.5, true, NDTPSOConfig(), 0.2);  // multi-cell frame, loaded directly
  a.loadLaser(scans[0], amin, ainc, rmax);
  a.build();
  print_cells("A after scan 0", a);

  NDTFrame b(Vector3d::Zero(), 60, 60, 60, false);  // one-cell per-scan frame
  b.loadLaser(scans[1], amin, ainc, rmax);
  const Vector3d probe(0.05, -0.03, 0.01);
  std::printf("cost %.17g\n", cost_function(probe, &a, &b));

  std::srand(5);
  PSOConfig cfg;
  cfg.iterations = 25;
  cfg.populationSize = 20;
  const Vector3d pose = pso_optimization(Vector3d::Zero(), &a, &b, Array3d(.1, .1, 3.1415E-3), cfg);
  std::printf("pso %.17g %.17g %.17g\n", pose.x(), pose.y(), pose.z());

  b.setTrans(Vector3d(0.2, -0.1, 0.02));
  b.loadLaser(scans[2], amin, ainc, rmax);  // a second scan into the same one-cell frame, through s_trans
  a.update(pose, &b);
  a.build();
  print_cells("A after update", a);
  print_cells("B (two scans)", b);

  Vector2d p1(1.25, -2.5), p2(-40., 3.), p3(0.1, 0.2);  // one outside the frame
  b.addPoint(p1);
  b.addPoint(p2);
  b.addPoint(p3);
  print_cells("B after addPoint", b);
  a.update(Vector3d(0., 0., 0.), &b);
  print_cells("A before build", a);
  const Vector3d again = a.align(pose, &b);  // builds lazily; consumes rand() on from the stream above
  std::printf("align %.17g %.17g %.17g\n", again.x(), again.y(), again.z());
  print_cells("A after align", a);

  a.transform(Vector3d(0.1, 0.2, 0.05));
  a.build();
  print_cells("A transformed", a);
  a.addPose(0.5, pose);
  a.addPose(1.0, again);
  a.dumpMap(argv[2], true, true, true, 10, true);

  a.resetCells();
  a.loadLaser(scans[1], amin, ainc, rmax);
  a.build();
  print_cells("A reset + scan 1", a);

  // addScan(pose, scan) == loadLaser into a one-cell per-scan frame + update with it (the two lines must be identical)
  const Vector3d at(0.4, -0.3, 0.07);
  NDTFrame c1(Vector3d::Zero(), 60, 60, 0.5, true), c2(Vector3d::Zero(), 60, 60, 0.5, true);
  c1.addScan(Vector3d::Zero(), scans[0], amin, ainc, rmax);
  c1.addScan(at, scans[2], amin, ainc, rmax);
  c1.build();
  {
    NDTFrame s0(Vector3d::Zero(), 60, 60, 60, false), s2(Vector3d::Zero(), 60, 60, 60, false);
    s0.loadLaser(scans[0], amin, ainc, rmax);
    c2.update(Vector3d::Zero(), &s0);
    s2.loadLaser(scans[2], amin, ainc, rmax);
    c2.update(at, &s2);
  }
  c2.build();
  print_cells("addScan", c1);
  print_cells("addScan", c2);
  std::printf("errors %lu\n", ndtpso_slam_error_count());
  return 0;
}
