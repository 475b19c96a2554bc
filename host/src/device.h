// Process-wide device context of the host library.  No CPU fallback: when the HIP library cannot create a context, or
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
ndtpso_ctx* device();                 // lazily created on HIP device $NDTPSO_DEVICE (default 0); nullptr if there is none
int score_mode();                     // $NDTPSO_SCORE = exact (default: fp64 results, fp32 speed) | f32 | f64
bool check(int rc, const char* what); // true if rc == NDTPSO_OK; otherwise records + logs the C-ABI error text, returns false
const void*& table_owner();           // which frame's cell table currently sits in the device context
bool resident_default();              // $NDTPSO_RESIDENT != 0
bool alive();                         // false once the process-wide context has been torn down (atexit)
// device scan buffers are recycled: the node allocates a fresh per-scan frame for every scan (ndtpso_slam_node.cpp:228-230)
// n outputs of std::rand(), in order, with the process-wide generator advanced exactly as n calls would leave it
void draw_rand(int32_t* out, size_t n);
ndtpso_points* acquire_scan(uint32_t capacity);
void release_scan(ndtpso_points* p, uint32_t capacity);
}  // namespace ndtpso_host

#endif
