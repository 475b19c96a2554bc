#include "device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace ndtpso_host {

namespace {
ndtpso_ctx* g_ctx = nullptr;
std::once_flag g_once;
bool g_alive = false;
std::vector<std::pair<ndtpso_points*, uint32_t>> g_scan_pool;
std::mutex g_pool_mutex;
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
    g_alive = true;
    std::atexit([] {
      g_alive = false;
      for (auto& e : g_scan_pool) ndtpso_points_destroy(e.first);
      g_scan_pool.clear();
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

bool alive() { return g_alive; }

bool resident_default() {
  const char* e = std::getenv("NDTPSO_RESIDENT");
  return !(e && e[0] == '0');
}

ndtpso_points* acquire_scan(uint32_t capacity) {
  ndtpso_ctx* c = device();
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t i = 0; i < g_scan_pool.size(); ++i)
      if (g_scan_pool[i].second >= capacity) {
        ndtpso_points* p = g_scan_pool[i].first;
        g_scan_pool.erase(g_scan_pool.begin() + (long)i);
        return p;
      }
  }
  ndtpso_points* p = nullptr;
  check(ndtpso_points_create(c, capacity, &p), "scan buffer");
  return p;
}

void release_scan(ndtpso_points* p, uint32_t capacity) {
  if (!p) return;
  if (!g_alive) return;  // the context is gone, and the device memory with it
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  if (g_scan_pool.size() < 8)
    g_scan_pool.emplace_back(p, capacity);
  else
    ndtpso_points_destroy(p);
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
