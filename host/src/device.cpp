#include "device.h"

#include "ndtpso_slam/linalg.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace ndtpso_host {
struct Ctx {
  ndtpso_ctx* c = nullptr;
  std::recursive_mutex mu;
  const void* table_owner = nullptr;
  std::vector<std::pair<ndtpso_points*, uint32_t>> scan_pool;
  std::atomic<int> frames{0};        // frames bound to this context (NDTFrame::dev() ... ~NDTFrame)
  std::atomic<bool> orphan{false};   // the thread it was made for has ended
};

namespace {
std::mutex g_registry_mutex;
std::vector<Ctx*> g_registry;  // every context ever made (they outlive their threads: frames may)
bool g_alive = false, g_atexit = false;
thread_local Ctx* t_own = nullptr;     // this thread's context
thread_local Ctx* t_active = nullptr;  // the innermost Use's
// A context outlives its thread while frames are bound to it.  One whose thread has ended AND that no frame refers to is taken
// over by the next thread that needs one (a process that serves requests on short-lived threads would otherwise gather a
// stream, workspaces and pinned buffers per thread it ever had, released only at exit).
struct Owner {
  Ctx* x = nullptr;
  ~Owner() {
    if (x) x->orphan.store(true);
  }
};
thread_local Owner t_owner;

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

unsigned long context_count() {
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  return (unsigned long)g_registry.size();
}
void bind_frame(Ctx* x) {
  if (x) x->frames.fetch_add(1);
}
void unbind_frame(Ctx* x) {
  if (x) x->frames.fetch_sub(1);
}

Ctx* thread_ctx() {
  if (t_own) return t_own;
  {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (Ctx* q : g_registry)
      if (q->c && q->frames.load() == 0 && q->orphan.exchange(false)) {  // (exchange: two new threads never adopt the same one)
        if (q->frames.load() != 0) {  // bound in the meantime by a frame handed on from its old thread: leave it be
          q->orphan.store(true);
          continue;
        }
        t_own = q;
        t_owner.x = q;
        return q;
      }
  }
  Ctx* x = new Ctx();
  int dev = 0;
  if (const char* e = std::getenv("NDTPSO_DEVICE")) dev = std::atoi(e);
  const int rc = ndtpso_ctx_create(dev, &x->c);
  if (rc != NDTPSO_OK || !x->c) {
    x->c = nullptr;  // every later device call fails with NDTPSO_E_ARG and is skipped: there is no CPU path
    char what[96];
    std::snprintf(what, sizeof(what), "creating a context on HIP device %d", dev);
    record_error(what, rc, "no usable HIP device");
  }
  {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    g_registry.push_back(x);
    if (x->c) g_alive = true;
    if (!g_atexit) {
      g_atexit = true;
      std::atexit([] {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        g_alive = false;
        for (Ctx* q : g_registry) {
          for (auto& e : q->scan_pool) ndtpso_points_destroy(e.first);
          q->scan_pool.clear();
          if (q->c) ndtpso_ctx_destroy(q->c);
          q->c = nullptr;
        }
      });
    }
  }
  t_own = x;
  t_owner.x = x;
  return x;
}

Use::Use(Ctx* ctx) : ctx_(ctx ? ctx : (t_active ? t_active : thread_ctx())), prev_(t_active) {
  ctx_->mu.lock();
  t_active = ctx_;
}
Use::~Use() {
  t_active = prev_;
  ctx_->mu.unlock();
}

static Ctx* active_ctx() { return t_active ? t_active : thread_ctx(); }

ndtpso_ctx* device() { return active_ctx()->c; }

const void*& table_owner() { return active_ctx()->table_owner; }

bool alive() { return g_alive; }

bool resident_default() {
  const char* e = std::getenv("NDTPSO_RESIDENT");
  return !(e && e[0] == '0');
}

ndtpso_points* acquire_scan(uint32_t capacity) {
  Ctx* x = active_ctx();
  {
    std::lock_guard<std::recursive_mutex> lock(x->mu);
    for (size_t i = 0; i < x->scan_pool.size(); ++i)
      if (x->scan_pool[i].second >= capacity) {
        ndtpso_points* p = x->scan_pool[i].first;
        x->scan_pool.erase(x->scan_pool.begin() + (long)i);
        return p;
      }
  }
  ndtpso_points* p = nullptr;
  if (!check(ndtpso_points_create(x->c, capacity, &p), "scan buffer")) return nullptr;
  return p;
}

void release_scan(ndtpso_points* p, uint32_t capacity) {  // (under a Use of the context the buffer was acquired from)
  if (!p) return;
  if (!g_alive) return;  // the contexts are gone, and the device memory with them
  Ctx* x = active_ctx();
  std::lock_guard<std::recursive_mutex> lock(x->mu);
  if (x->scan_pool.size() < 8)
    x->scan_pool.emplace_back(p, capacity);
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

// A thread's PRIVATE generator (ndtpso_slam_thread_srand): glibc's TYPE_3 algorithm restated -- srandom_r's seeding (Park-Miller
// by Schrage's method into 31 words, 310 outputs discarded), then r[i] = r[i-31] + r[i-3], output >> 1 -- on state of its own, so
// that replicas of the live sequence in one process each see the stream srand(seed) + rand() would have given them alone
// (the process-wide generator is one stream for everybody: with two threads drawing from it neither is reproducible, in
// the reference as here).  host/replay/rand_check.cpp holds it to the libc's.
namespace {
struct PrivateRand {
  bool on = false;
  uint32_t t[31];
  int f = 3, r = 0;
  void seed(unsigned s) {
    int32_t word = (int32_t)(s ? s : 1u);
    t[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      const int32_t hi = word / 127773, lo = word % 127773;
      word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      t[i] = (uint32_t)word;
    }
    f = 3;
    r = 0;
    int32_t sink;
    for (int i = 0; i < 310; ++i) fill(&sink, 1);
    on = true;
  }
  void fill(int32_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      t[f] += t[r];
      out[i] = (int32_t)(t[f] >> 1);
      if (++f == 31) f = 0;
      if (++r == 31) r = 0;
    }
  }
};
thread_local PrivateRand t_rand;
std::mutex g_rand_mutex;  // the process-wide generator's state is edited in place (fast_fill)
}  // namespace

void draw_rand(int32_t* out, size_t n) {
  if (t_rand.on) {
    t_rand.fill(out, n);
    return;
  }
  std::lock_guard<std::mutex> lock(g_rand_mutex);
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
  // (the error text of whatever context is at hand: reporting a failure must not create one)
  Ctx* x = t_active ? t_active : t_own;
  ndtpso_ctx* c = x ? x->c : nullptr;
  record_error(what, rc, c ? ndtpso_last_error(c) : "no device context");
  return false;
}

}  // namespace ndtpso_host

extern "C" {
// the Eigen-or-fallback choice this library was compiled with (ndtpso_slam/linalg.h): consumers reference the symbol
// of THEIR choice, so a mismatch is an undefined reference at link time
int NDTPSO_ABI_TAG = 1;

void ndtpso_slam_device_init(void) {
  ndtpso_ctx* c = ndtpso_host::device();
  // ... and, where the exact mode will be used, its once-per-process self check (ndtpso_exact_check, ~30 ms) now rather
  // than inside the first align()
  if (c && ndtpso_host::score_mode() == NDTPSO_SCORE_EXACT) (void)ndtpso_exact_check(c, nullptr, nullptr, nullptr, nullptr);
}
void ndtpso_slam_thread_srand(unsigned seed) { ndtpso_host::t_rand.seed(seed); }
int ndtpso_slam_thread_rand(void) {
  int32_t v;
  ndtpso_host::draw_rand(&v, 1);
  return (int)v;
}

// ndtpso_slam/status.h
const char* ndtpso_slam_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lock(ndtpso_host::g_err_mutex);
  copy = ndtpso_host::g_last_error;
  return copy.c_str();
}
unsigned long ndtpso_slam_error_count(void) { return ndtpso_host::g_err_count.load(); }
unsigned long ndtpso_slam_context_count(void) { return ndtpso_host::context_count(); }
unsigned long ndtpso_slam_cluster_timeouts(void) {
  uint64_t v[1] = {0};
  return ndtpso_process_counters(v, 1) == NDTPSO_OK ? (unsigned long)v[0] : 0ul;
}
void ndtpso_slam_clear_error(void) {
  std::lock_guard<std::mutex> lock(ndtpso_host::g_err_mutex);
  ndtpso_host::g_last_error.clear();
  ndtpso_host::g_err_logged.clear();
  ndtpso_host::g_err_count = 0;
}
}
