// Device contexts of the host library (one per host thread).  No CPU fallback: when the HIP library cannot create a context, or
// a device call fails, the error is recorded (ndtpso_slam_last_error(), ndtpso_slam/status.h) and logged once per call
// site, and the API call degrades -- align() returns its initial guess, update() / loadLaser() / build() do nothing --
// so that a robot process survives a transient device fault the reference could never have had.  Nothing is ever
// computed on the CPU instead.  NDTPSO_ABORT_ON_ERROR=1 restores abort-at-first-error (tests, debugging).
#ifndef NDTPSO_HOST_DEVICE_H
#define NDTPSO_HOST_DEVICE_H

#include <cstddef>
#include <cstdint>

#include "../../include/ndtpso_hip.h"

namespace ndtpso_host {
// One device context PER HOST THREAD (its own HIP stream, workspaces, staged reference table and scan pool): frames used by
// different threads never share one, so two robots -- or R replicas of the live sequence -- can run in one process, each on
// its own stream (SURVEY 8b "one context per host thread / HIP stream").  A frame is bound to the context of the thread that
// first used it on the device (NDTFrame::dev()) and keeps it: every device-touching member opens a `Use` of that context,
// which locks it (recursively: members call each other) and makes it the one device() & co. below refer to.  A frame handed
// to another thread therefore still works, serialised against its context's other users.
struct Ctx;
Ctx* thread_ctx();                    // the calling thread's own context, created on HIP device $NDTPSO_DEVICE (default 0) at first use
                                      // (or taken over from a thread that has ended, once no frame is bound to it any more)
void bind_frame(Ctx* ctx);            // a frame now refers to the context / no longer does: a context is recycled only at zero
void unbind_frame(Ctx* ctx);
unsigned long context_count();        // contexts ever made and still held
class Use {
 public:
  explicit Use(Ctx* ctx);             // nullptr: the calling thread's own
  ~Use();
  Use(const Use&) = delete;
  Use& operator=(const Use&) = delete;

 private:
  Ctx* ctx_;
  Ctx* prev_;
};
ndtpso_ctx* device();                 // the innermost Use's context (else the calling thread's own); nullptr if there is no device
int score_mode();                     // $NDTPSO_SCORE = exact (default: fp64 results, fp32 speed) | f32 | f64
bool check(int rc, const char* what); // true if rc == NDTPSO_OK; otherwise records + logs the C-ABI error text, returns false
const void*& table_owner();           // which frame's cell table currently sits in that context
bool resident_default();              // $NDTPSO_RESIDENT != 0
bool alive();                         // false once the contexts have been torn down (atexit)
// device scan buffers are recycled: the node allocates a fresh per-scan frame for every scan (ndtpso_slam_node.cpp:228-230)
// n outputs of std::rand(), in order, with the process-wide generator advanced exactly as n calls would leave it -- or, on a
// thread that called ndtpso_slam_thread_srand(), of that thread's private generator (same algorithm, own state)
void draw_rand(int32_t* out, size_t n);
ndtpso_points* acquire_scan(uint32_t capacity);
void release_scan(ndtpso_points* p, uint32_t capacity);
}  // namespace ndtpso_host

#endif
