/*
 * ndtpso_oracle.h -- CPU restatement of the reference NDT-PSO alignment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ndtpso_slam_amd/,
 * include/, the C-ABI library) may include, link or call this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it,
 * and there only as the checker / the CPU number printed beside the GPU one.
 *
 * PARITY UNPINNED.  The reference (abougouffa/ndtpso_slam) ships no tests, no
 * golden vectors and no recorded data (src/test/ndtpso_slam_test.cpp:21-25 is
 * an empty main), and it cannot be compiled in this image: every library TU
 * includes <eigen3/Eigen/Core> (include/ndtpso_slam/core.h:6, ndtcell.h:5,
 * ndtframe.h:5) and Eigen3 is not installed.  This oracle is therefore a
 * line-by-line restatement of the reference's arithmetic that has NOT been
 * checked against reference outputs.  Third-party arithmetic it restates:
 *   - Eigen3 (unpinned by the reference; 3.3.4 / 3.3.7 on the ROS distros its
 *     README names): DenseBase::Random() = x + (y-x)*double(rand())/double(RAND_MAX)
 *     with x=-1,y=1 (Eigen/src/Core/MathFunctions.h, random_default_impl<double>),
 *     coefficient-wise +,-,*,/ on 2- and 3-vectors, 2x2 determinant, and
 *     EigenSolver<Matrix2d>::pseudoEigenvalueMatrix().diagonal() (ndtcell.cpp:96-97),
 *     restated step by step from Eigen 3.3.7's RealSchur / JacobiRotation sources
 *     (orc_eigen_eigenvalues_2x2; rounds 1-2 used a closed form that differs from it
 *     by a few ulp -- kept as variant 2 so that the difference stays measurable).
 *   - glibc rand()/srand() (TYPE_3 additive feedback generator, RAND_MAX 2^31-1).
 *
 * Every function cites the reference file:line (relative to the reference
 * repository root) it follows.
 */
#ifndef NDTPSO_ORACLE_H
#define NDTPSO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/ndtpso_slam/config.h:5-8 */
#define ORC_NDT_MAX_POINTS_PER_CELL 50
#define ORC_NDT_WINDOW_SIZE 100

/* include/ndtpso_slam/config.h:27-38 (PSOConfig), field for field */
typedef struct {
  int iterations;   /* PSO_ITERATIONS 50 */
  int population;   /* PSO_POPULATION_SIZE 30 */
  int num_threads;  /* -1; the oracle is always the OMP_NUM_THREADS=1 semantics */
  double w;         /* .8 */
  double c1;        /* 2. */
  double c2;        /* 2. */
  double w_damping; /* 1. */
} orc_pso_config;

/* Source of the std::rand() stream consumed by Eigen's Random() (core.cpp:14,84).
 * table != NULL: draws are table[cursor++] (values must be glibc rand() outputs);
 * table == NULL: live libc rand() is called, exactly as the reference does. */
typedef struct {
  const int32_t *table;
  size_t n;
  size_t cursor;
} orc_rand;

typedef struct {
  uint64_t cost_evals;
  uint64_t pbest_updates;
  uint64_t gbest_updates;
  uint64_t rand_draws;
} orc_pso_stats;

typedef struct orc_frame orc_frame; /* mirrors NDTFrame (include/ndtpso_slam/ndtframe.h:12-72) */

/* one exported row per `created` cell, ascending linear cell index */
typedef struct {
  int32_t index;    /* linear cell index ix + W*iy (ndtframe.cpp:244-245) */
  int32_t count;    /* s_global_count after build() */
  int32_t built;    /* NDTCell::built */
  int32_t n_slot0;  /* points_vector[0].size() */
  int32_t window_id;     /* s_current_window_id */
  int32_t current_count; /* s_current_count */
  double mean[2];   /* NDTCell::mean */
  double icov[4];   /* s_inv_covar, row-major (0,0),(0,1),(1,0),(1,1) */
} orc_cell_row;

void orc_pso_config_default(orc_pso_config *c);

/* glibc srand(seed); rand() x n, restated (stdlib/random_r.c, TYPE_3). */
void orc_glibc_rand_fill(uint32_t seed, int32_t *out, size_t n);

/* EigenSolver<Matrix2d>(M).pseudoEigenvalueMatrix().diagonal() for M = {M00, M01, M10, M11} (ndtcell.cpp:96-97).
 * variant 0: Eigen 3.3.7's operation order (scaled RealSchur, Givens rotation) -- what the oracle uses;
 * variant 1: the same without RealSchur::compute's scaling (older 3.3.x); variant 2: closed form (rounds 1-2). */
void orc_eigen_eigenvalues_2x2(const double m[4], int variant, double ev[2]);
/* Which variant NDTCell::s_calc_covar_inverse uses from now on (process-wide; default 0). */
void orc_set_eigen_variant(int variant);
int orc_get_eigen_variant(void);
/* s_calc_covar_inverse (ndtcell.cpp:93-111) on n covariance matrices {c00,c01,c10,c11}; out: 8 doubles per matrix =
 * {large, small, det used, degenerate branch (0/1), inv00, inv01, inv10, inv11}. */
void orc_covar_inverse_batch(const double *m, size_t n, int variant, double *out);
/* number of rand() draws one pso_optimization call consumes: 3 + 3P + 6PI */
size_t orc_pso_rand_draws(const orc_pso_config *c);

/* NDTFrame::NDTFrame, ndtframe.cpp:19-66 (occupancy grid omitted: size 0 path) */
orc_frame *orc_frame_create(const double trans[3], unsigned short width, unsigned short height,
                            double cell_side, float laser_ignore_epsilon);
void orc_frame_destroy(orc_frame *f);
/* NDTFrame::loadLaser, ndtframe.cpp:144-185 */
void orc_frame_load_laser(orc_frame *f, const float *ranges, unsigned n, float min_angle,
                          float angle_increment, float max_range);
/* NDTFrame::addPoint, ndtframe.cpp:215-235 */
void orc_frame_add_point(orc_frame *f, double x, double y);
/* NDTFrame::update, ndtframe.cpp:187-198 */
void orc_frame_update(orc_frame *ref, const double trans[3], const orc_frame *new_frame);
/* NDTFrame::build, ndtframe.cpp:68-117 (occupancy-grid branch omitted) */
void orc_frame_build(orc_frame *f);
/* NDTFrame::getCellIndex, ndtframe.cpp:240-249 */
int orc_frame_get_cell_index(const orc_frame *f, double x, double y);
/* NDTFrame::resetCells, ndtframe.cpp:208-212 (NDTCell::reset, ndtcell.cpp:80-91) */
void orc_frame_reset_cells(orc_frame *f);
/* NDTFrame::setTrans, ndtframe.h:51 */
void orc_frame_set_trans(orc_frame *f, const double trans[3]);
/* NDTFrame::align, ndtframe.cpp:251-266.  use_frame_config=0 reproduces the
 * reference exactly (PSOConfig() default is used, ndtframe.cpp:257); 1 uses cfg. */
void orc_frame_align(orc_frame *ref, const double guess[3], const orc_frame *new_frame,
                     const orc_pso_config *cfg, orc_rand *rng, double out_pose[3]);

/* cost_function, core.cpp:26-48.  cell_idx (optional, length = number of new
 * points in iteration order) receives the reference-frame cell index each
 * transformed point scored against, or -1 (outside frame) / -2 (cell not built). */
double orc_cost_function(const double trans[3], orc_frame *ref, const orc_frame *new_frame,
                         int32_t *cell_idx);
/* pso_optimization, core.cpp:50-116 (single-thread semantics: asynchronous gbest) */
void orc_pso_optimization(const double guess[3], orc_frame *ref, const orc_frame *new_frame,
                          const double deviation[3], const orc_pso_config *cfg, orc_rand *rng,
                          double out_pose[3], double *out_cost, orc_pso_stats *stats);

/* pso_optimization in the reference's parallel shape (core.cpp:72-109: OpenMP over the particles of an iteration,
 * racy global best, live rand()): irreproducible like the original -- TIMING ONLY, never used as a checker. */
void orc_pso_optimization_omp(const double guess[3], orc_frame *ref, const orc_frame *new_frame,
                              const double deviation[3], const orc_pso_config *cfg, int n_threads,
                              double out_pose[3], double *out_cost);

/* occupancy grid of NDTFrame (ndtframe.h:22-29, ctor ndtframe.cpp:32-46, rasterised in build() :79-112).
 * Enable before build(); og is width_cells x height_cells int8 (index x + height*y as the reference writes it). */
void orc_frame_enable_occupancy_grid(orc_frame *f, double og_cell_size);
const int8_t *orc_frame_occupancy_grid(const orc_frame *f, uint32_t *og_width, uint32_t *og_height,
                                       uint32_t minmax[4] /* min_x, max_x, min_y, max_y */);

/* accessors used by the tests */
unsigned orc_frame_num_points(const orc_frame *f);              /* sum of points_vector[0] sizes */
unsigned orc_frame_get_points(const orc_frame *f, double *xy);   /* cells order then insertion order (core.cpp:33-36) */
unsigned long orc_frame_get_points_all(const orc_frame *f, double *xy, unsigned long max_points); /* all window slots */
unsigned orc_frame_num_created(const orc_frame *f);
unsigned orc_frame_export_cells(const orc_frame *f, orc_cell_row *rows, unsigned max_rows);
void orc_frame_dims(const orc_frame *f, int32_t *width_cells, int32_t *height_cells);

/* Batched scan pairs (BASELINE configs 3/4), timed by bench.py as the CPU
 * baseline: for each pair b: ref frame <- ref scan at identity, new 1-cell
 * frame <- new scan (ndtpso_slam_node.cpp:229-230), pso_optimization with
 * srand(seeds[b]) stream.  OpenMP across pairs only; each alignment is the
 * sequential reference algorithm.  Returns number of threads used. */
int orc_align_pairs(int n_pairs, const float *ref_ranges, const float *new_ranges, unsigned n_beams,
                    float min_angle, float angle_increment, float max_range, float laser_ignore_epsilon,
                    unsigned short width, unsigned short height, double cell_side,
                    const double *guess /*[n_pairs*3]*/, const double *deviation /*[n_pairs*3]*/,
                    const orc_pso_config *cfg, const uint32_t *seeds, int n_threads,
                    double *out_pose /*[n_pairs*3]*/, double *out_cost /*[n_pairs]*/);

/* the host libm's exp / sincos over arrays (the device's math library is compared with them, tests/test_gpu_exp.py) */
void orc_libm_exp(const double *x, size_t n, double *out);
void orc_libm_sincos(const double *x, size_t n, double *s, double *c);

#ifdef __cplusplus
}
#endif
#endif
