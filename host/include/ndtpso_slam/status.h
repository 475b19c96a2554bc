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
/* not an error, but worth a node's attention: alignments that ran on a cluster of workgroups whose members were not scheduled
 * together (device shared with other work, more streams than hardware queues), hit the exchange's bounded wait (20 ms) and
 * were redone on one workgroup -- same pose, a latency spike.  Process-wide, since start-up. */
unsigned long ndtpso_slam_cluster_timeouts(void);
/* creates the calling thread's device context now instead of at its first frame operation.  Every host thread that uses
 * frames gets a context of its own (stream, workspaces, staged table): frames used by different threads run concurrently on
 * the device.  A frame belongs to the context of the thread that first used it and may be handed to another thread (its
 * calls are then serialised against that context's other users). */
void ndtpso_slam_device_init(void);
/* device contexts the process holds (one per host thread that has used frames; a thread that has ended leaves its context to
 * the next one that needs it, once no frame is bound to it) */
unsigned long ndtpso_slam_context_count(void);
/* The reference draws its PSO's random numbers from the process-wide std::rand() (Eigen's Random(), core.cpp:14,84): ONE
 * stream for every thread, so two matchers in one process disturb each other's streams -- in the reference as here.
 * ndtpso_slam_thread_srand(seed) gives the CALLING THREAD a private generator instead (glibc's own algorithm on private
 * state: the numbers srand(seed) + rand() would produce if the thread were alone); from then on the alignments that thread
 * runs draw from it and leave std::rand() untouched.  ndtpso_slam_thread_rand() is its rand() (std::rand() on a thread
 * that never seeded one).  An addition of this build: replicas of the live sequence in one process stay reproducible. */
void ndtpso_slam_thread_srand(unsigned seed);
int ndtpso_slam_thread_rand(void);

#ifdef __cplusplus
}
#endif
#endif
