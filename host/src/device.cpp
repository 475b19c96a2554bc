#include "device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace ndtpso_host {

namespace {
ndtpso_ctx* g_ctx = nullptr;
std::once_flag g_once;
}  // namespace

ndtpso_ctx* device() {
  std::call_once(g_once, [] {
    int dev = 0;
    if (const char* e = std::getenv("NDTPSO_DEVICE")) dev = std::atoi(e);
    const int rc = ndtpso_ctx_create(dev, &g_ctx);
    if (rc != NDTPSO_OK || !g_ctx) {
      std::fprintf(stderr, "libndtpso_slam (MI355X build): no usable HIP device %d (error %d); there is no CPU path\n",
                   dev, rc);
      std::abort();
    }
    std::atexit([] {
      if (g_ctx) ndtpso_ctx_destroy(g_ctx);
      g_ctx = nullptr;
    });
  });
  return g_ctx;
}

const void*& table_owner() {
  static const void* owner = nullptr;
  return owner;
}

int score_mode() {
  const char* e = std::getenv("NDTPSO_SCORE");
  return (e && std::strcmp(e, "f64") == 0) ? NDTPSO_SCORE_F64 : NDTPSO_SCORE_F32;
}

void check(int rc, const char* what) {
  if (rc == NDTPSO_OK) return;
  std::fprintf(stderr, "libndtpso_slam (MI355X build): %s failed: %d %s\n", what, rc, ndtpso_last_error(g_ctx));
  std::abort();
}

}  // namespace ndtpso_host

extern "C" void ndtpso_slam_device_init(void) { (void)ndtpso_host::device(); }
