#include "device.h"

#include "ndtpso_slam/linalg.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace ndtpso_host {

namespace {
ndtpso_ctx* g_ctx = nullptr;
std::once_flag g_once;
bool g_alive = false;
std::vector<std::pair<ndtpso_points*, uint32_t>> g_scan_pool;
std::mutex g_pool_mutex;

// error state of the library (ndtpso_slam/status.h)
std::mutex g_err_mutex;
std::string g_last_error;            // text of the most recent failure
std::set<std::string> g_err_logged;  // call sites already reported on stderr
std::atomic<unsigned long> g_err_count{0};

bool abort_on_error() {
  static const bool v = [] {
    const char* e = std::getenv("NDTPSO_ABORT_ON_ERROR");
    return e && e[0] == '1';
  }();
  return v;
}

void record_error(const char* what, int rc, const char* text) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "%s failed: %d %s", what, rc, text ? text : "");
  bool first;
  {
    std::lock_guard<std::mutex> lock(g_err_mutex);
    g_last_error = buf;
    first = g_err_logged.insert(what).second;
  }
  ++g_err_count;
  if (first || abort_on_error())
    std::fprintf(stderr, "libndtpso_slam (MI355X build): %s%s\n", buf,
                 abort_on_error() ? "" : " -- the call is skipped (further failures of this call are counted, not printed; "
                                         "see ndtpso_slam_last_error())");
  if (abort_on_error()) std::abort();
}
}  // namespace

ndtpso_ctx* device() {
  std::call_once(g_once, [] {
    int dev = 0;
    if (const char* e = std::getenv("NDTPSO_DEVICE")) dev = std::atoi(e);
    const int rc = ndtpso_ctx_create(dev, &g_ctx);
    if (rc != NDTPSO_OK || !g_ctx) {
      g_ctx = nullptr;  // every later device call fails with NDTPSO_E_ARG and is skipped: there is no CPU path
      char what[96];
      std::snprintf(what, sizeof(what), "creating a context on HIP device %d", dev);
      record_error(what, rc, "no usable HIP device");
      return;
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
  if (!check(ndtpso_points_create(c, capacity, &p), "scan buffer")) return nullptr;
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

// ---- std::rand() in bulk ------------------------------------------------------------------------------------------
// One alignment consumes 3 + 3P + 6PI outputs of std::rand() (9093 at the default 30 x 50) -- Eigen's Random() in the
// reference, core.cpp:14,84.  Through rand() that is 80 us of call + lock overhead per scan, a sixth of the live
// path.  glibc's generator (TYPE_3 additive feedback: r[i] = r[i-31] + r[i-3], output >> 1) keeps its state in a plain
// int32 array that setstate() hands out, with the rear index stored in the word in front of it; so the state is
// advanced in place here, n steps at once, and handed back.  The result is indistinguishable from n calls of rand():
// same outputs, same state afterwards.  A self-test on a private state (the caller's stream is not touched) decides
// once whether this glibc behaves as expected; otherwise, and for generator types other than TYPE_3, rand() is called.
namespace {
#if defined(__GLIBC__)
constexpr int kRandTypes = 5, kRandType3 = 3, kRandDeg = 31, kRandSep = 3;

bool fast_fill(int32_t* out, size_t n) {
  static char scratch[128];
  static bool scratch_ready = false;
  if (!scratch_ready) {  // a valid state to park the generator on while its real state is edited
    char* prev = initstate(1u, scratch, sizeof(scratch));
    setstate(prev);
    scratch_ready = true;
  }
  int32_t* w = reinterpret_cast<int32_t*>(setstate(scratch));  // w[0]: type + 5 * rear, w[1..31]: the table
  if (!w) return false;
  const int type = w[0] % kRandTypes;
  if (type != kRandType3) {
    setstate(reinterpret_cast<char*>(w));
    return false;
  }
  uint32_t* t = reinterpret_cast<uint32_t*>(w + 1);
  int r = w[0] / kRandTypes, f = (r + kRandSep) % kRandDeg;
  for (size_t i = 0; i < n; ++i) {
    t[f] += t[r];
    out[i] = (int32_t)(t[f] >> 1);
    if (++f == kRandDeg) f = 0;
    if (++r == kRandDeg) r = 0;
  }
  w[0] = r * kRandTypes + kRandType3;
  setstate(reinterpret_cast<char*>(w));
  return true;
}

bool fast_fill_verified() {
  static const bool ok = [] {
    if (const char* e = std::getenv("NDTPSO_SLOW_RAND"))
      if (e[0] == '1') return false;
    static char a[128], b[128];
    constexpr int kN = 200;
    int32_t want[kN], got[kN];
    char* user = initstate(20240521u, a, sizeof(a));  // the caller's state is parked, untouched
    for (int i = 0; i < kN; ++i) want[i] = std::rand();
    const int32_t want_next = std::rand();
    initstate(20240521u, b, sizeof(b));
    bool good = fast_fill(got, kN);
    for (int i = 0; good && i < kN; ++i) good = got[i] == want[i];
    good = good && std::rand() == want_next;  // the state after the bulk draw continues the same stream
    setstate(user);
    return good;
  }();
  return ok;
}
#endif
}  // namespace

void draw_rand(int32_t* out, size_t n) {
#if defined(__GLIBC__)
  if (fast_fill_verified() && fast_fill(out, n)) return;
#endif
  for (size_t i = 0; i < n; ++i) out[i] = std::rand();
}

// Default: NDTPSO_SCORE_EXACT -- the fp64 score's poses (the reference's arithmetic, ndtcell.cpp:70-78) at close to the
// fp32 score's speed.  NDTPSO_SCORE=f32 selects the plain fp32 score (a tolerance mode: a comparison of two costs closer
// than its rounding error can fall the other way, measured on 1 of 4096 scan pairs), =f64 the fp64 score throughout.
int score_mode() {
  const char* e = std::getenv("NDTPSO_SCORE");
  if (e && std::strcmp(e, "f64") == 0) return NDTPSO_SCORE_F64;
  if (e && std::strcmp(e, "f32") == 0) return NDTPSO_SCORE_F32;
  return NDTPSO_SCORE_EXACT;
}

bool check(int rc, const char* what) {
  if (rc == NDTPSO_OK) return true;
  record_error(what, rc, g_ctx ? ndtpso_last_error(g_ctx) : "no device context");
  return false;
}

}  // namespace ndtpso_host

extern "C" {
// the Eigen-or-fallback choice this library was compiled with (ndtpso_slam/linalg.h): consumers reference the symbol
// of THEIR choice, so a mismatch is an undefined reference at link time
int NDTPSO_ABI_TAG = 1;

void ndtpso_slam_device_init(void) { (void)ndtpso_host::device(); }

// ndtpso_slam/status.h
const char* ndtpso_slam_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lock(ndtpso_host::g_err_mutex);
  copy = ndtpso_host::g_last_error;
  return copy.c_str();
}
unsigned long ndtpso_slam_error_count(void) { return ndtpso_host::g_err_count.load(); }
void ndtpso_slam_clear_error(void) {
  std::lock_guard<std::mutex> lock(ndtpso_host::g_err_mutex);
  ndtpso_host::g_last_error.clear();
  ndtpso_host::g_err_logged.clear();
  ndtpso_host::g_err_count = 0;
}
}
