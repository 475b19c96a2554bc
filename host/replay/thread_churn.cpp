// Short-lived host threads (a callback pool, a thread per request) each touching frames: the library gives a thread a device
// context at its first frame operation (host/src/device.h) -- and, since round 6, takes the context of a thread that has ended
// and that no frame is bound to any more over instead of creating another.  N threads one after the other, each loading a scan
// into a frame of its own and aligning against a frame the MAIN thread owns (bound to the main thread's context): prints how
// many contexts the process ended up with.
//   usage: thread_churn n_threads
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ndtpso_slam/ndtframe.h"
#include "ndtpso_slam/status.h"

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 40;
  std::vector<float> ranges(361);
  for (int i = 0; i < 361; ++i) ranges[(size_t)i] = 4.f + 0.01f * (float)(i % 37);
  NDTFrame ref(Vector3d::Zero(), 20, 20, 0.5, true);
  ref.loadLaser(ranges, -1.5f, 3.0f / 360.f, 30.f);
  int failed = 0;
  for (int i = 0; i < n; ++i) {
    std::thread([&] {
      NDTFrame scan(Vector3d::Zero(), 20, 20, 20., false);
      scan.loadLaser(ranges, -1.5f, 3.0f / 360.f, 30.f);
      (void)ref.align(Vector3d::Zero(), &scan);
      if (!ref.lastAlignOk()) ++failed;
    }).join();
  }
  std::printf("{\"threads\": %d, \"contexts\": %lu, \"failed\": %d, \"device_errors\": %lu}\n", n, ndtpso_slam_context_count(), failed,
              ndtpso_slam_error_count());
  return 0;
}
