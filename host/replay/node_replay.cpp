// ROS-free replay of NDTPSONode::scan_matcher_ (reference src/ndtpso_slam_node.cpp:177-244) against the drop-in
// library: same call sequence -- loadLaser, align (except on the first scan), update, re-allocation of the
// per-scan frame as a one-cell frame (:229-230).  Reads a binary scan file, prints one pose per scan.
//
//   file: int32 n_scans, int32 n_beams, float angle_min, float angle_inc, float range_max, then n_scans*n_beams floats
//   usage: node_replay scans.bin frame_size cell_side iterations population [srand_seed [og_cell_size dump_prefix
//          [pixels_per_metre]]]
//   NODE_REPLAY_PACE_HZ=f in the environment delivers the scans at f Hz like a sensor would (the time a scan takes is
//   then the latency a robot sees, with whatever the library does between scans off the clock)
// With a dump prefix the shutdown export of the node (:141-172) runs too: a one-cell global map collects every 10th
// scan (SAVE_DATA_TO_FILE_EACH_NUM_ITERS, ndtpso_slam_node.hpp:18; NODE_REPLAY_SAVE_EACH=n overrides) and every
// pose and is dumped as <prefix>.{pose.csv,map.csv,gnuplot,png}; the reference frame (with its occupancy grid) as
// <prefix>-ref-frame.*, plus the raw grid as <prefix>-ref-frame.og.bin for the tests.
#include <algorithm>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndtpso_slam/core.h"
#include "ndtpso_slam/ndtframe.h"

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s scans.bin frame_size cell_side iterations population [seed]\n", argv[0]);
    return 2;
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t n_scans = 0, n_beams = 0;
  float amin = 0, ainc = 0, rmax = 0;
  if (std::fread(&n_scans, 4, 1, f) != 1 || std::fread(&n_beams, 4, 1, f) != 1 || std::fread(&amin, 4, 1, f) != 1 ||
      std::fread(&ainc, 4, 1, f) != 1 || std::fread(&rmax, 4, 1, f) != 1)
    return 2;
  const unsigned short frame_size = (unsigned short)std::atoi(argv[2]);
  const double cell_side = std::atof(argv[3]);
  NDTPSOConfig conf;
  conf.psoConfig.iterations = std::atoi(argv[4]);
  conf.psoConfig.populationSize = std::atoi(argv[5]);
  ndtpso_slam_device_init();  // before srand(): keep runtime start-up out of the rand() stream
  if (argc > 6) std::srand((unsigned)std::atoi(argv[6]));

  const double og_cell_size = argc > 8 ? std::atof(argv[7]) : 0.;
  const char* dump_prefix = argc > 8 ? argv[8] : nullptr;
  const short density = argc > 9 ? (short)std::atoi(argv[9]) : 100;

  const Vector3d initial_pose = Vector3d::Zero();
  // ndtpso_slam_node.cpp:64-78
  NDTFrame* ref_frame = new NDTFrame(Vector3d::Zero(), frame_size, frame_size, cell_side, true, conf, og_cell_size);
  NDTFrame* global_map = dump_prefix ? new NDTFrame(Vector3d::Zero(), frame_size, frame_size, frame_size, false) : nullptr;
  NDTFrame* current_frame = new NDTFrame(initial_pose, frame_size, frame_size, cell_side, false);
  Vector3d previous_pose = initial_pose, current_pose = initial_pose;
  bool first_iteration = true;
  unsigned iter_num = 0;
  const unsigned kSaveEachNumIters = std::getenv("NODE_REPLAY_SAVE_EACH") ? (unsigned)std::max(1, std::atoi(std::getenv("NODE_REPLAY_SAVE_EACH"))) : 10u;
  std::vector<float> ranges((size_t)n_beams);
  double busy_s = 0., part_s[4] = {0., 0., 0., 0.};  // loadLaser, align, update, between scans (NODE_REPLAY_TIMING)
  auto t_prev_end = std::chrono::steady_clock::now();
  const double pace_hz = std::getenv("NODE_REPLAY_PACE_HZ") ? std::atof(std::getenv("NODE_REPLAY_PACE_HZ")) : 0.;
  auto next_scan = std::chrono::steady_clock::now();
  for (int k = 0; k < n_scans; ++k) {
    if (std::fread(ranges.data(), 4, (size_t)n_beams, f) != (size_t)n_beams) return 2;
    if (pace_hz > 0.) {
      std::this_thread::sleep_until(next_scan);
      next_scan += std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(1. / pace_hz));
    }
    const auto t0 = std::chrono::steady_clock::now();
    current_frame->loadLaser(ranges, amin, ainc, rmax);                       // :186
    const auto t1 = std::chrono::steady_clock::now();
    if (first_iteration)
      current_pose = previous_pose;                                            // :188-189
    else
      current_pose = ref_frame->align(previous_pose, current_frame);           // :194
    previous_pose = current_pose;
    const auto t2 = std::chrono::steady_clock::now();
    ref_frame->update(current_pose, current_frame);                            // :198
    const auto t3 = std::chrono::steady_clock::now();
    if (k > 0) {
      busy_s += std::chrono::duration<double>(t3 - t0).count();
      part_s[0] += std::chrono::duration<double>(t1 - t0).count();
      part_s[1] += std::chrono::duration<double>(t2 - t1).count();
      part_s[2] += std::chrono::duration<double>(t3 - t2).count();
      if (k > 1) part_s[3] += std::chrono::duration<double>(t0 - t_prev_end).count();
    }
    t_prev_end = t3;
    if (global_map) {                                                          // :200-206
      // the node merges every SAVE_DATA_TO_FILE_EACH_NUM_ITERS-th scan (10, ndtpso_slam_node.hpp:18) into the global
      // map, starting with the first, and records every pose
      if (iter_num == 0) global_map->update(current_pose, current_frame);
      iter_num = (iter_num + 1) % kSaveEachNumIters;
      global_map->addPose(0.025 * k, current_pose);
    }
    std::printf("%d %.17g %.17g %.17g\n", k, current_pose.x(), current_pose.y(), current_pose.z());
    delete current_frame;                                                      // :228-230
    current_frame = new NDTFrame(initial_pose, frame_size, frame_size, frame_size, false);
    first_iteration = false;
  }
  std::fclose(f);
  if (n_scans > 1 && std::getenv("NODE_REPLAY_TIMING"))
    std::fprintf(stderr, "per scan (us): loadLaser %.1f  align %.1f  update %.1f  between scans %.1f\n", 1e6 * part_s[0] / (n_scans - 1),
                 1e6 * part_s[1] / (n_scans - 1), 1e6 * part_s[2] / (n_scans - 1), 1e6 * part_s[3] / (n_scans - 1));
  if (n_scans > 1)  // the node's own metric ("matching rate", ndtpso_slam_node.cpp:239): loadLaser + align + update per scan
    std::fprintf(stderr, "matching rate: %.1f Hz (%.3f ms per scan)\n", (n_scans - 1) / busy_s, 1e3 * busy_s / (n_scans - 1));
  if (dump_prefix) {  // :141-172
    char name[1024];
    global_map->dumpMap(dump_prefix, true, true, true, density, true);
    std::snprintf(name, sizeof(name), "%s-ref-frame", dump_prefix);
    ref_frame->dumpMap(name, false, true, true, density, true);
    uint32_t dims[6] = {0, 0, 0, 0, 0, 0};
    const std::vector<int8_t>& og = ref_frame->occupancyGrid(&dims[0], &dims[1], &dims[2]);
    std::snprintf(name, sizeof(name), "%s-ref-frame.og.bin", dump_prefix);
    if (FILE* o = std::fopen(name, "wb")) {
      std::fwrite(dims, 4, 6, o);
      std::fwrite(og.data(), 1, og.size(), o);
      std::fclose(o);
    }
    delete global_map;
  }
  delete current_frame;
  delete ref_frame;
  return 0;
}
