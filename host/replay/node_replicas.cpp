// R replicas of the live sequence in ONE process: R host threads, each replaying NDTPSONode::scan_matcher_ (reference
// src/ndtpso_slam_node.cpp:177-244: loadLaser, align, update, a fresh one-cell per-scan frame) over the same recorded scans
// with its own frames, its own device context and stream (the library gives every host thread one, host/src/device.h) and
// its own random stream (ndtpso_slam_thread_srand(seed + r): what srand(seed + r) would give a process of its own).
// Replica r's pose log is therefore what `node_replay scans.bin ... (seed + r)` prints, whatever the other replicas do.
//
//   usage: node_replicas scans.bin frame_size cell_side iterations population seed replicas [out_prefix]
//   writes <out_prefix>.<r>.poses ("k x y theta" per scan, %.17g) and prints one JSON line: aggregate scans per second,
//   per-replica mean / p95 / max time per scan (loadLaser + align + update, the node's "matching rate" interval).
//   The first NODE_REPLICAS_WARM scans (default 10: map and scan buffers get allocated, which serialises the whole device) are
//   replayed but not timed: the replicas meet at a barrier behind them and the clock starts there.
//   A process's streams share GPU_MAX_HW_QUEUES hardware queues (ROCm's default: 4), and streams that share one run their
//   kernels one after the other: unless the environment says otherwise this harness asks for one queue per replica (up to 16:
//   with more the hardware scheduler time-slices the queues and a cluster's workgroups begin to wait milliseconds for each
//   other -- 16 replicas 14 k scans/s on 16 queues, 32 replicas 9 k on 16 and 7 k on 24) before the runtime starts.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include <sys/resource.h>

#include "ndtpso_slam/core.h"
#include "ndtpso_slam/ndtframe.h"
#include "ndtpso_slam/status.h"

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s scans.bin frame_size cell_side iterations population seed replicas [out_prefix]\n", argv[0]);
    return 2;
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t n_scans = 0, n_beams = 0;
  float amin = 0, ainc = 0, rmax = 0;
  if (std::fread(&n_scans, 4, 1, f) != 1 || std::fread(&n_beams, 4, 1, f) != 1 || std::fread(&amin, 4, 1, f) != 1 ||
      std::fread(&ainc, 4, 1, f) != 1 || std::fread(&rmax, 4, 1, f) != 1)
    return 2;
  std::vector<float> all((size_t)n_scans * (size_t)n_beams);
  if (std::fread(all.data(), 4, all.size(), f) != all.size()) return 2;
  std::fclose(f);
  const unsigned short frame_size = (unsigned short)std::atoi(argv[2]);
  const double cell_side = std::atof(argv[3]);
  NDTPSOConfig conf;
  conf.psoConfig.iterations = std::atoi(argv[4]);
  conf.psoConfig.populationSize = std::atoi(argv[5]);
  const unsigned seed = (unsigned)std::atoi(argv[6]);
  const int R = std::max(1, std::atoi(argv[7]));
  const char* prefix = argc > 8 ? argv[8] : nullptr;
  const int warm = std::min(n_scans - 1, std::getenv("NODE_REPLICAS_WARM") ? std::atoi(std::getenv("NODE_REPLICAS_WARM")) : 10);
  {
    char q[16];
    std::snprintf(q, sizeof(q), "%d", std::min(std::max(R, 4), 16));
    setenv("GPU_MAX_HW_QUEUES", q, 0);  // (no HIP call has been made yet: the library's first is in ndtpso_slam_device_init)
  }

  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<std::vector<double>> per_scan_ms((size_t)R);
  std::vector<std::vector<std::array<double, 4>>> parts((size_t)R);  // diagnostics: start (ms since the clock started), loadLaser, align, update
  std::vector<int> failed((size_t)R, 0);
  std::vector<double> between_ms((size_t)R, 0.);  // time between a scan's end and the next one's start (frame re-allocation, :228-230)
  std::vector<std::thread> threads;
  std::chrono::steady_clock::time_point t_start;
  for (int r = 0; r < R; ++r)
    threads.emplace_back([&, r] {
      ndtpso_slam_device_init();              // this thread's context
      ndtpso_slam_thread_srand(seed + (unsigned)r);
      FILE* out = nullptr;
      if (prefix) out = std::fopen((std::string(prefix) + "." + std::to_string(r) + ".poses").c_str(), "w");
      const Vector3d initial_pose = Vector3d::Zero();
      NDTFrame* ref_frame = new NDTFrame(Vector3d::Zero(), frame_size, frame_size, cell_side, true, conf);           // :64-66
      NDTFrame* current_frame = new NDTFrame(initial_pose, frame_size, frame_size, cell_side, false);             // :73
      Vector3d previous_pose = initial_pose, current_pose = initial_pose;
      std::vector<float> ranges((size_t)n_beams);
      std::chrono::steady_clock::time_point t_prev_end{};
      for (int k = 0; k < n_scans; ++k) {
        if (k == warm) {  // everybody's buffers exist: meet, start the clock
          ++ready;  // (asleep, not spinning: R threads in a yield loop spend a control group's CPU quota, and the period's
          while (!go.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));  // throttle then falls on the first timed scans)
        }
        std::copy(all.begin() + (size_t)k * n_beams, all.begin() + (size_t)(k + 1) * n_beams, ranges.begin());
        const auto t0 = std::chrono::steady_clock::now();
        if (k > warm && k > 1) between_ms[(size_t)r] += std::chrono::duration<double, std::milli>(t0 - t_prev_end).count();
        current_frame->loadLaser(ranges, amin, ainc, rmax);                                                         // :186
        const auto ta = std::chrono::steady_clock::now();
        current_pose = k == 0 ? previous_pose : ref_frame->align(previous_pose, current_frame);                     // :188-194
        const auto tb = std::chrono::steady_clock::now();
        if (k > 0 && !ref_frame->lastAlignOk()) ++failed[(size_t)r];
        previous_pose = current_pose;
        ref_frame->update(current_pose, current_frame);                                                             // :198
        const auto t1 = std::chrono::steady_clock::now();
        t_prev_end = t1;
        if (k >= warm && k > 0) {
          per_scan_ms[(size_t)r].push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
          auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
          parts[(size_t)r].push_back({ms(t_start, t0), ms(t0, ta), ms(ta, tb), ms(tb, t1)});
        }
        if (out) std::fprintf(out, "%d %.17g %.17g %.17g\n", k, current_pose.x(), current_pose.y(), current_pose.z());
        delete current_frame;                                                                                       // :228-230
        current_frame = new NDTFrame(initial_pose, frame_size, frame_size, frame_size, false);
      }
      delete current_frame;
      delete ref_frame;
      if (out) std::fclose(out);
    });
  while (ready.load() < R) std::this_thread::sleep_for(std::chrono::microseconds(200));
  std::this_thread::sleep_for(std::chrono::milliseconds(150));  // (a fresh quota period for everybody)
  rusage ru0;
  getrusage(RUSAGE_SELF, &ru0);
  t_start = std::chrono::steady_clock::now();
  go = true;
  for (auto& t : threads) t.join();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  rusage ru1;
  getrusage(RUSAGE_SELF, &ru1);
  auto secs = [](const timeval& a, const timeval& b) { return (double)(b.tv_sec - a.tv_sec) + 1e-6 * (double)(b.tv_usec - a.tv_usec); };
  const double cpu_user = secs(ru0.ru_utime, ru1.ru_utime), cpu_sys = secs(ru0.ru_stime, ru1.ru_stime);
  std::vector<double> every;
  int n_failed = 0;
  for (int r = 0; r < R; ++r) {
    every.insert(every.end(), per_scan_ms[(size_t)r].begin(), per_scan_ms[(size_t)r].end());
    n_failed += failed[(size_t)r];
  }
  if (std::getenv("NODE_REPLICAS_SLOWEST")) {  // diagnostics: the slowest scans, with replica and scan number
    std::vector<std::pair<double, std::pair<int, int>>> slow;
    for (int r = 0; r < R; ++r)
      for (size_t i = 0; i < per_scan_ms[(size_t)r].size(); ++i) slow.push_back({per_scan_ms[(size_t)r][i], {r, (int)i + std::max(warm, 1)}});
    std::sort(slow.begin(), slow.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    for (size_t i = 0; i < std::min<size_t>(slow.size(), 24); ++i)
    {
      const auto& q = parts[(size_t)slow[i].second.first][(size_t)(slow[i].second.second - std::max(warm, 1))];
      std::fprintf(stderr, "slow: %.3f ms replica %d scan %d  at %.2f ms: loadLaser %.3f align %.3f update %.3f\n", slow[i].first,
                   slow[i].second.first, slow[i].second.second, q[0], q[1], q[2], q[3]);
    }
  }
  std::sort(every.begin(), every.end());
  double mean = 0.;
  for (double v : every) mean += v;
  mean /= std::max<size_t>(every.size(), 1);
  const double p95 = every.empty() ? 0. : every[(size_t)(0.95 * (every.size() - 1))];
  std::printf("{\"replicas\": %d, \"timed_scans_per_replica\": %d, \"wall_s\": %.6f, \"aggregate_scans_per_s\": %.1f, "
              "\"ms_per_scan_mean\": %.4f, \"ms_per_scan_p95\": %.4f, \"ms_per_scan_max\": %.4f, \"failed_alignments\": %d, "
              "\"device_errors\": %lu, \"cluster_timeouts\": %lu, \"host_cpus_busy_user\": %.2f, \"host_cpus_busy_sys\": %.2f, \"ms_between_scans_mean\": %.4f}\n",
              R, n_scans - warm, wall, (double)R * (n_scans - warm) / wall, mean, p95, every.empty() ? 0. : every.back(), n_failed,
              ndtpso_slam_error_count(), ndtpso_slam_cluster_timeouts(), cpu_user / wall, cpu_sys / wall,
              [&] { double t = 0.; for (double v : between_ms) t += v; return t / std::max(1., (double)R * (n_scans - warm - 1)); }());
  return 0;
}
