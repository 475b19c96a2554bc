// Process-wide device context of the host library.  No CPU fallback: if the HIP library cannot create a
// context the process reports the error and aborts.
#ifndef NDTPSO_HOST_DEVICE_H
#define NDTPSO_HOST_DEVICE_H

#include <cstddef>
#include <cstdint>

#include "../../include/ndtpso_hip.h"

namespace ndtpso_host {
ndtpso_ctx* device();                 // lazily created on HIP device $NDTPSO_DEVICE (default 0)
int score_mode();                     // $NDTPSO_SCORE = f32 (default) | f64
void check(int rc, const char* what); // abort with the C-ABI error text unless rc == NDTPSO_OK
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
