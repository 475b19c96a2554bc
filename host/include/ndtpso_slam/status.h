// Error state of the MI355X build of libndtpso_slam.
//
// The reference library cannot fail at run time (host memory and arithmetic only: no status codes, no exceptions,
// SURVEY 8b "Errors").  This one drives a GPU, and a device call can fail -- no device, a transient HIP fault, the
// resident map's point pool exhausted.  The C++ API keeps the reference's signatures, so it reports nothing through
// them: a failing call is SKIPPED (align() returns its initial guess, update() / loadLaser() / build() leave the frame
// as it was, cost_function() returns 0), the failure is printed once per call site on stderr, counted, and its text kept
// here.  Nothing is ever computed on the CPU instead.  A node that wants to react polls these after align():
//
//     Vector3d pose = ref.align(guess, &scan);
//     if (ndtpso_slam_error_count()) { ROS_ERROR("%s", ndtpso_slam_last_error()); ... }
//
// NDTPSO_ABORT_ON_ERROR=1 in the environment turns the first failure into abort() instead (tests, debugging).
#ifndef NDTPSO_SLAM_AMD_STATUS_H
#define NDTPSO_SLAM_AMD_STATUS_H

#ifdef __cplusplus
extern "C" {
#endif

/* text of the most recent failed device call ("" if none); valid until the calling thread's next call of this function */
const char* ndtpso_slam_last_error(void);
/* number of failed (skipped) device calls since start-up or the last ndtpso_slam_clear_error() */
unsigned long ndtpso_slam_error_count(void);
void ndtpso_slam_clear_error(void);
/* creates the process-wide device context now instead of at the first frame operation */
void ndtpso_slam_device_init(void);

#ifdef __cplusplus
}
#endif
#endif
