// batch_sharded -- C++ driver of the multi-device batch path: one process, one batch of recorded scan pairs, G devices.
//
//   batch_sharded <batch file> <poses out> [devices, e.g. 0,1,2,3 | all] [repetitions]
//
// Reads a batch (format below), calls ndtpso_align_pairs_sharded (include/ndtpso_hip.h: contiguous index ranges, one
// context and stream per device, ONE ncclAllGather of the poses over xGMI) and writes the poses + costs as raw doubles
// ([n][3] then [n]).  With repetitions > 1 it also times the call (host buffers in and out, so the figure includes the
// scatter over PCIe) and prints alignments per second.  Uses the C-ABI only: this is what a C or C++ caller with a
// recorded run would write; the Python benchmark (bench.py, one process per GPU over torch.distributed) is the other
// way to the same kernels.
//
// Batch file (little endian): magic "NDTB", u32 version = 1, u32 n_pairs, u32 n_beams, f32 min_angle, angle_increment,
// max_range, laser_ignore_epsilon, u32 grid width, height (metres), f64 cell_side, i32 iterations, population,
// f64 w, c1, c2, w_damping, i32 score_mode, then f32 ref[n][beams], f32 new[n][beams], f64 guess[n][3],
// f64 deviation[n][3], u32 seeds[n].  tests/test_host_library.py writes one from the synthetic run.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <algorithm>

#include "include/ndtpso_hip.h"

namespace {
template <class T>
bool rd(std::FILE* f, T* v, size_t n = 1) { return std::fread(v, sizeof(T), n, f) == n; }
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <batch file> <poses out> [devices a,b,c | all] [repetitions]\n", argv[0]);
    return 2;
  }
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  char magic[4];
  uint32_t version = 0, n = 0, beams = 0, gw = 0, gh = 0;
  float geomf[4];
  double cs = 0., coeff[4];
  int32_t it_pop[2], mode = 0;
  bool ok = rd(f, magic, 4) && !std::memcmp(magic, "NDTB", 4) && rd(f, &version) && version == 1 && rd(f, &n) && rd(f, &beams) &&
            rd(f, geomf, 4) && rd(f, &gw) && rd(f, &gh) && rd(f, &cs) && rd(f, it_pop, 2) && rd(f, coeff, 4) && rd(f, &mode);
  std::vector<float> ref((size_t)n * beams), nw((size_t)n * beams);
  std::vector<double> guess(3 * (size_t)n), dev(3 * (size_t)n), pose(3 * (size_t)n), cost(n);
  std::vector<uint32_t> seeds(n);
  ok = ok && rd(f, ref.data(), ref.size()) && rd(f, nw.data(), nw.size()) && rd(f, guess.data(), guess.size()) &&
       rd(f, dev.data(), dev.size()) && rd(f, seeds.data(), seeds.size());
  std::fclose(f);
  if (!ok) { std::fprintf(stderr, "%s: not a batch file\n", argv[1]); return 2; }

  std::vector<int> devices;
  const std::string spec = argc > 3 ? argv[3] : "all";
  if (spec == "all") {
    for (int d = 0; d < 64; ++d) {  // every device a context can be made on
      ndtpso_ctx* c = nullptr;
      if (ndtpso_ctx_create(d, &c) != NDTPSO_OK) break;
      ndtpso_ctx_destroy(c);
      devices.push_back(d);
    }
  } else {
    for (size_t p = 0; p < spec.size();) {
      size_t q = spec.find(',', p);
      if (q == std::string::npos) q = spec.size();
      devices.push_back(std::atoi(spec.substr(p, q - p).c_str()));
      p = q + 1;
    }
  }
  const int reps = argc > 4 ? std::max(1, std::atoi(argv[4])) : 1;
  if (devices.empty()) { std::fprintf(stderr, "no HIP device: nothing is computed on the CPU instead\n"); return 3; }

  ndtpso_shard_group* g = nullptr;
  if (ndtpso_shard_group_create(devices.data(), (int)devices.size(), &g) != NDTPSO_OK) {
    std::fprintf(stderr, "ndtpso_shard_group_create failed for %zu device(s) (device or RCCL missing)\n", devices.size());
    return 3;
  }
  ndtpso_scan_geom geom{beams, geomf[0], geomf[1], geomf[2], geomf[3]};
  ndtpso_grid grid{(uint16_t)gw, (uint16_t)gh, cs};
  ndtpso_pso_config cfg{};
  cfg.iterations = it_pop[0];
  cfg.population = it_pop[1];
  cfg.num_threads = -1;
  cfg.w = coeff[0];
  cfg.c1 = coeff[1];
  cfg.c2 = coeff[2];
  cfg.w_damping = coeff[3];
  std::vector<ndtpso_align_stats> stats(n);
  double best = 1e30, call_us[3] = {0, 0, 0};
  std::vector<double> per_dev(3 * devices.size(), 0.);  // per device: start after the call's entry, uploads, launches (us)
  for (int r = 0; r < reps; ++r) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = ndtpso_align_pairs_sharded(g, n, ref.data(), nw.data(), &geom, &grid, guess.data(), dev.data(), &cfg,
                                              seeds.data(), nullptr, mode, pose.data(), cost.data(), stats.data());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != NDTPSO_OK) {
      std::fprintf(stderr, "ndtpso_align_pairs_sharded: %d (%s)\n", rc, ndtpso_shard_last_error(g));
      ndtpso_shard_group_destroy(g);
      return 4;
    }
    if ((r > 0 || reps == 1) && dt < best) {
      best = dt;
      ndtpso_shard_last_timing(g, per_dev.data(), call_us);
    }
  }
  uint32_t flagged = 0;
  for (const ndtpso_align_stats& s : stats) flagged += NDTPSO_STATUS_FLAGS(s.status) ? 1u : 0u;
  // what the group was and whether the collective demonstrably moved every shard's poses to every shard
  ndtpso_shard_info info{};
  int ranks_seen = 0;
  (void)ndtpso_shard_group_describe(g, &info);
  (void)ndtpso_shard_verify_gather(g, &ranks_seen);
  double up_max = 0, up_sum = 0, launch_max = 0;
  for (size_t d = 0; d < devices.size(); ++d) {
    up_max = std::max(up_max, per_dev[3 * d + 1]);
    up_sum += per_dev[3 * d + 1];
    launch_max = std::max(launch_max, per_dev[3 * d + 2]);
  }
  std::printf("{\"pairs\": %u, \"devices\": %zu, \"repetitions\": %d, \"best_call_s\": %.6f, \"alignments_per_s\": %.1f, "
              "\"flagged\": %u, \"shards\": %d, \"ranks_seen\": %d, \"comm_ranks\": %d, \"rccl_version\": %d, \"gather\": \"%s\", "
              "\"includes\": \"host-to-device scatter, the all-gather and the copy back\", "
              "\"host_us\": {\"uploads_slowest_device\": %.1f, \"uploads_all_devices_summed\": %.1f, \"launches_slowest_device\": %.1f, "
              "\"all_enqueued\": %.1f, \"collective_enqueue\": %.1f, \"call\": %.1f}}\n",
              n, devices.size(), reps, best, n / best, flagged, info.n_shards, ranks_seen, info.comm_ranks, info.rccl_version,
              info.gather_kind ? "host-staged (test)" : "ncclAllGather (RCCL)", up_max, up_sum, launch_max, call_us[0], call_us[1], call_us[2]);
  ndtpso_shard_group_destroy(g);
  std::FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror(argv[2]); return 2; }
  std::fwrite(pose.data(), 8, pose.size(), o);
  std::fwrite(cost.data(), 8, cost.size(), o);
  std::fclose(o);
  return flagged ? 5 : 0;
}
