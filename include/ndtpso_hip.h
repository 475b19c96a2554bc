/*
 * ndtpso_hip.h -- C-ABI of the MI355X (gfx950) NDT-PSO scan-alignment path.
 *
 * The reference (abougouffa/ndtpso_slam) has no FFI/plugin layer: its boundary
 * is the C++ class API of libndtpso_slam consumed by ndtpso_slam_node
 * (CMakeLists.txt:155-160,183-187).  This header is the thin C-ABI that the
 * drop-in host library (host/ndtpso_slam/ *.h, same class names and
 * signatures as include/ndtpso_slam/ndtframe.h:37-70 of the reference) calls
 * into; each entry point names the reference code it replaces.  Plain
 * pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns 0 (NDTPSO_OK) or a negative NDTPSO_E_* code;
 *     ndtpso_last_error(ctx) gives the text.  Nothing throws across the ABI
 *     (the reference API never throws either; SURVEY 8b "Errors").
 *   - `_dev` variants take DEVICE pointers, enqueue on the context stream and
 *     return without synchronising; the others take HOST pointers and return
 *     after the results are back on the host.
 *   - there is no CPU fallback: with no usable HIP device every call fails
 *     with NDTPSO_E_HIP.
 *   - a context, and the resident scans / maps created from it, keep scratch
 *     buffers between calls and are not safe for concurrent use: one thread at
 *     a time per context (the reference's NDTFrame is single-threaded too);
 *     independent threads use independent contexts.
 *   - a frame may have at most 65535 cells per side (the reference keeps
 *     widthNumOfCells / heightNumOfCells in uint16_t, ndtframe.h:32); larger
 *     grids are refused with NDTPSO_E_ARG.
 */
#ifndef NDTPSO_HIP_H
#define NDTPSO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDTPSO_OK 0
#define NDTPSO_E_HIP (-1)      /* HIP runtime error / no device */
#define NDTPSO_E_ARG (-2)      /* invalid argument */
#define NDTPSO_E_CAPACITY (-3) /* problem does not fit the on-chip (LDS) layout */
#define NDTPSO_E_STATE (-4)    /* call order (e.g. align before a reference table is set) */

/* score arithmetic after the fp64 transform + cell lookup */
#define NDTPSO_SCORE_F32 0 /* Mahalanobis + exp in fp32, fp64 accumulation (BASELINE config 2: "fp32") */
#define NDTPSO_SCORE_F64 1 /* everything in fp64, reference operation order */
/* NDTPSO_SCORE_EXACT: the results of NDTPSO_SCORE_F64 at (nearly) the speed of NDTPSO_SCORE_F32.  The PSO only ever
 * COMPARES costs (core.cpp:63,94,97), so the fp32 score decides every comparison whose two sides are further apart
 * than the margin tau = max(5e-6 x |gbest cost|, 7e-7 x points of the scan) -- the fp32 form's rounding error is below
 * 2.95e-7 per point whatever the cells look like (derivation: ndtpso_kernels.hpp verify_pose_wave), i.e.
 * below tau / 2 for ANY input, so a comparison outside the margin falls as it would in fp64 by proof, not by luck
 * (tests/test_gpu_margin.py checks the inequality for every evaluation with a diagnostic build) -- and the rest
 * ("near-ties", about 0.5 per 70 x 70 alignment) are arbitrated with the fp64 score of the poses involved, evaluated exactly as SCORE_F64 evaluates them;
 * the returned cost is the fp64 score of the returned pose.  Pose and cost equal SCORE_F64's bit for bit (tests:
 * every pair of BASELINE configs 3, 4 and 5).  Where the fp32 kernel cannot arbitrate (tables too large for the dense
 * LDS form, degenerate overlaps, swarms of near-identical costs) the fp64-score kernel does the alignment.
 * ndtpso_cost_batch / ndtpso_map_cost treat it as SCORE_F64. */
#define NDTPSO_SCORE_EXACT 2

typedef struct ndtpso_ctx ndtpso_ctx;

/* PSOConfig, include/ndtpso_slam/config.h:27-38 (field for field; num_threads is ignored on the device) */
typedef struct {
  int32_t iterations;
  int32_t population;
  int32_t num_threads;
  double w, c1, c2, w_damping;
} ndtpso_pso_config;

/* NDTFrame grid geometry: width/height in metres (uint16, ndtframe.h:32), cell_side (ndtframe.h:36) */
typedef struct {
  uint16_t width, height;
  double cell_side;
} ndtpso_grid;

/* arguments of NDTFrame::loadLaser (ndtframe.h:47-48) + NDTPSOConfig::laserIgnoreEpsilon (config.h:44) */
typedef struct {
  uint32_t n_beams;
  float min_angle, angle_increment, max_range;
  float laser_ignore_epsilon;
} ndtpso_scan_geom;

/* one row per created cell (ascending linear index), for inspection / parity tests */
typedef struct {
  int32_t index; /* ix + W*iy, NDTFrame::getCellIndex (ndtframe.cpp:244-245) */
  int32_t count; /* points accumulated */
  int32_t built; /* NDTCell::built (count > 2, ndtcell.cpp:43) */
  int32_t reserved;
  double mean[2];
  double icov[4]; /* row-major inverse covariance (ndtcell.cpp:109-110) */
} ndtpso_cell_row;

typedef struct {
  uint32_t n_points;      /* new-scan points that entered the PSO */
  uint32_t n_built;       /* built reference cells */
  uint32_t cost_evals;    /* particle evaluations incl. those thrown away behind a gbest update (the one-workgroup kernels deal an
                             iteration's items by ticket: how many were in flight then depends on timing -- poses and costs do not) */
  uint32_t rounds;        /* evaluation rounds (iterations + replays) */
  uint32_t gbest_updates; /* in-iteration gbest improvements */
  uint32_t status;        /* 0 ok; bit0: a reference point fell outside the staging window; bit1: more built cells
                             than record capacity; bit2: fp32 costs underflowed and the
                             fp64 redo did not fit; bit3: dense-table overflow and the bitmap redo did not fit.
                             Bits 16-31 (NDTPSO_SCORE_EXACT): number of pbest / gbest comparisons of this alignment
                             that were arbitrated with the fp64 score (saturating); flags = status & 0xffff */
  uint32_t t_start, t_end; /* low 32 bits of the device's 100 MHz real-time counter when the alignment's workgroup
                              started / finished (fused pairs kernel; load-balance diagnostics) */
} ndtpso_align_stats;
/* `status` is two fields in one word: test NDTPSO_STATUS_FLAGS(s) != 0 for "something went wrong", never the raw word
 * (in NDTPSO_SCORE_EXACT the upper half counts arbitrated comparisons of a perfectly good alignment). */
#define NDTPSO_STATUS_FLAGS(status) ((uint32_t)(status) & 0xffffu)
#define NDTPSO_STATUS_ARBITRATED(status) ((uint32_t)(status) >> 16)

/* ---- context --------------------------------------------------------- */
int ndtpso_ctx_create(int device, ndtpso_ctx **out);
void ndtpso_ctx_destroy(ndtpso_ctx *ctx);
const char *ndtpso_last_error(const ndtpso_ctx *ctx);
/* hipStream_t to enqueue on (NULL = the context's own stream) */
int ndtpso_set_stream(ndtpso_ctx *ctx, void *hip_stream);
int ndtpso_synchronize(ndtpso_ctx *ctx);
/* Process-wide telemetry of the single-alignment path (ndtpso_align, ndtpso_map_align -- the live node, or R replicas of it on
 * R host threads): out[0] alignments whose cluster of workgroups was not co-resident, ran into the exchange's bounded wait and
 * was redone on one workgroup (a latency spike each: a device shared with other work, or more streams than hardware queues);
 * out[1] waits for a kernel's result that slept between looks instead of spinning (more waiting threads than half the CPUs
 * the process may use: affinity mask cut by the control group's quota); out[2] that CPU budget; out[3] threads waiting now;
 * out[4] alignments kept on ONE workgroup because 16 (NDTPSO_CLUSTER_MAX_INFLIGHT) were already in flight in the process -- more
 * clusters than the device's hardware queues run side by side would wait for each other; out[5] (the batch path,
 * ndtpso_align_pairs*) batches whose few flagged pairs -- a cell table outgrown, an fp32 score in its underflow regime -- were
 * redone on clusters of workgroups instead of one workgroup each.  n: how many to write (1 .. 6).  The reference has nothing of
 * the kind (single-threaded caller, ndtpso_slam_node.cpp:182). */
int ndtpso_process_counters(uint64_t *out, int n);
/* Batches in flight.  depth 1 (default): every call is enqueued on the context's stream, one after the other.
 * depth 2: consecutive ndtpso_align_pairs_dev calls (batches of more pairs than half the device's compute units) run
 * on two internal streams with their own workspaces, so that the next batch's workgroups fill the compute units the
 * previous launch's tail leaves idle (one launch lasts as long as its slowest alignment).  Each call still starts
 * after everything enqueued on the context's stream before it (its inputs), but its OUTPUTS are ordered before later
 * work on the context's stream only by ndtpso_pipeline_flush / ndtpso_synchronize -- until then neither the inputs
 * nor the outputs of a call in flight may be touched, and two calls in flight need distinct output buffers.
 * The results do not depend on the depth (each alignment is its own workgroup).  Reference: none -- the reference
 * aligns one pair at a time (ndtpso_slam_node.cpp:194); this is the batch path of BASELINE configs 3/4. */
int ndtpso_set_pipeline_depth(ndtpso_ctx *ctx, int depth);
/* The context's stream waits (device-side, the host does not block) for the calls in flight, except the
 * `keep_newest` most recent ones (0: for all of them). */
int ndtpso_pipeline_flush(ndtpso_ctx *ctx, int keep_newest);
/* number of std::rand() draws one alignment consumes: 3 + 3P + 6PI (core.cpp:14,84) */
size_t ndtpso_rand_draws(const ndtpso_pso_config *cfg);

/* ---- K3 companions: scan ingest and cell statistics ------------------- */
/* NDTFrame::loadLaser's beam filter + index_to_angle + laser_to_point (+ s_trans transform)
 * (ndtframe.cpp:144-185, core.h:40-47).  xy_out: 2*n_beams doubles; *n_out = surviving points, beam order. */
int ndtpso_scan_to_points(ndtpso_ctx *ctx, const float *ranges, const ndtpso_scan_geom *geom,
                          const double trans[3], double *xy_out, uint32_t *n_out);

/* Reference table of a FRESH frame from points: NDTFrame::addPoint binning (ndtframe.cpp:215-235) +
 * NDTCell::build / s_calc_covar_inverse (ndtcell.cpp:36-68,93-111).  The table stays on the device
 * as the context's reference table. */
int ndtpso_ref_from_points(ndtpso_ctx *ctx, const ndtpso_grid *grid, const double *xy, uint32_t n_points);
/* same, starting from a LaserScan (loadLaser at pose `trans`, then build) */
int ndtpso_ref_from_scan(ndtpso_ctx *ctx, const ndtpso_grid *grid, const float *ranges,
                         const ndtpso_scan_geom *geom, const double trans[3]);
/* Reference table from host-side cell statistics (an accumulated map kept by the host NDTFrame):
 * built cells only; index[] ascending or not, mean 2*n, icov 4*n row-major. */
int ndtpso_ref_set_cells(ndtpso_ctx *ctx, const ndtpso_grid *grid, uint32_t n_cells, const int32_t *index,
                         const double *mean, const double *icov);
/* created cells of the table built by ndtpso_ref_from_points/_scan (parity inspection) */
int ndtpso_ref_get_cells(ndtpso_ctx *ctx, ndtpso_cell_row *rows, uint32_t max_rows, uint32_t *n_rows);

/* NDTFrame::update / NDTFrame::addPoint binning for a list of points (ndtframe.cpp:187-198, 215-235):
 * xy_out[i] = transform_point(xy[i], trans) (trans == NULL: identity, the loadLaser case),
 * cell_idx[i] = getCellIndex(xy_out[i]) or -1.  xy_out may alias xy. */
int ndtpso_points_to_cells(ndtpso_ctx *ctx, const ndtpso_grid *grid, const double *xy, uint32_t n_points,
                           const double trans[3], double *xy_out, int32_t *cell_idx);

/* NDTFrame::loadLaser in one launch: beam filter + polar->xy (+ s_trans) as ndtpso_scan_to_points, then the
 * binning of ndtpso_points_to_cells for every surviving point.  xy_out: 2*n_beams doubles, cell_idx: n_beams. */
int ndtpso_scan_to_cells(ndtpso_ctx *ctx, const float *ranges, const ndtpso_scan_geom *geom, const double trans[3],
                         const ndtpso_grid *grid, double *xy_out, int32_t *cell_idx, uint32_t *n_out);

/* The sliding-window state of one NDTCell that NDTCell::build reads and writes (ndtcell.h:65-68,
 * ndtcell.cpp:36-68); the host frame keeps it, the device does the arithmetic. */
typedef struct {
  double global_sum[2];       /* s_global_sum */
  double global_covar_sum[4]; /* s_global_covar_sum, row-major */
  double slot_sum[2];         /* s_partial_sums[s_current_window_id] */
  double slot_covar[4];       /* s_partial_covars[s_current_window_id] */
  double mean[2];             /* NDTCell::mean (out) */
  double icov[4];             /* s_inv_covar (out) */
  int32_t global_count;       /* s_global_count */
  int32_t slot_count;         /* s_partial_counts[s_current_window_id] */
  int32_t current_count;      /* s_current_count (in) */
  int32_t built;              /* NDTCell::built (in: before, out: after) */
} ndtpso_cell_window;

/* NDTCell::build + s_calc_covar_inverse for n_cells created cells at once (NDTFrame::build's loop,
 * ndtframe.cpp:73-76).  pts_offset[n_cells+1] delimits each cell's points_vector[s_current_window_id] (insertion
 * order) in pts_xy -- with current_count == 0 these are the stale points of the window's previous lap, which the
 * reference's covariance loop still visits (ndtcell.cpp:49) while its running sum is zero; cells[] is updated in
 * place.  The slot advance (ndtcell.cpp:61-65) is bookkeeping left to the host. */
int ndtpso_cells_build_windowed(ndtpso_ctx *ctx, uint32_t n_cells, ndtpso_cell_window *cells,
                                const uint32_t *pts_offset, const double *pts_xy);

/* Occupancy-grid rasterisation of NDTFrame::build (ndtframe.cpp:69-71,79-112; map export): for each
 * of n_cells BUILT cells and each of its k x k sub-cells (k = floor(cell_side / og_cell_size), j outer, k inner as
 * the reference loops), values[c*k*k + j*k + kk] = int8(100 * normalDistribution(sub-cell centre)), or -1 when the
 * Gaussian is exactly 0 there (the reference then leaves the grid untouched).  The row index of a cell is
 * index / heightNumOfCells, as in the reference (:81). */
int ndtpso_occupancy_values(ndtpso_ctx *ctx, const ndtpso_grid *grid, double og_cell_size, uint32_t n_cells,
                            const int32_t *index, const double *mean, const double *icov, int8_t *values);

/* ---- K1: cost_function (core.cpp:26-48) for M candidate poses ---------- */
/* xy: n_points new-frame points; poses: 3*M; costs: M; cell_idx (optional, M*n_points):
 * linear cell index scored against, -1 outside frame, -2 cell not built. */
int ndtpso_cost_batch(ndtpso_ctx *ctx, const double *xy, uint32_t n_points, const double *poses, uint32_t m,
                      int score_mode, double *costs, int32_t *cell_idx);

/* ---- K2: pso_optimization (core.cpp:50-116) --------------------------- */
/* rand_table: the std::rand() outputs the reference would draw (ndtpso_rand_draws() values), or NULL
 * to replay glibc's srand(seed) stream on the device. */
int ndtpso_align(ndtpso_ctx *ctx, const double *xy, uint32_t n_points, const double guess[3],
                 const double deviation[3], const ndtpso_pso_config *cfg, uint32_t seed,
                 const int32_t *rand_table, int score_mode, double out_pose[3], double *out_cost,
                 ndtpso_align_stats *stats);

/* ---- device-resident reference frame ---------------
 * The live node's path (ndtpso_slam_node.cpp:177-244) without the host touching a point: the map (NDTFrame with its
 * 100-slot sliding-window cells, ndtcell.h:62-70) and the loaded scan stay in HBM; per scan the host sends the ranges
 * and the std::rand() table and receives the pose.
 *   ndtpso_points_*      the node's one-cell per-scan frame (ndtpso_slam_node.cpp:229-230): a point list on the device
 *   ndtpso_map_insert    the addPoint loop of NDTFrame::update (ndtframe.cpp:190-197; pose != NULL: transform_point
 *                        first) / of loadLaser and addPoint on a multi-cell frame (:144-185, 215-235; pose == NULL):
 *                        NDTCell::addPoint (ndtcell.cpp:21-34) per point.  Like NDTFrame::addPoint it clears the
 *                        frame's `built` flag only if a point lands in the frame (:219-224)
 *   ndtpso_map_mark_unbuilt   the unconditional `built = false` at the top of update (:188) and loadLaser (:145) --
 *                        it decides whether the next cost_function builds again, and a build repeated on unchanged
 *                        cells is not a no-op in floating point (WINDOW_ADD, ndtcell.h:13-15)
 *   ndtpso_map_build     NDTFrame::build (:68-117): NDTCell::build (ndtcell.cpp:36-68) on every created cell, the
 *                        occupancy grid (:79-112), and the alignment table of the built cells
 *   ndtpso_map_align     NDTFrame::align's pso_optimization (core.cpp:50-116) against the map, building it first when
 *                        points were inserted since the last build (cost_function's lazy build, core.cpp:27-28)
 * insert and build only enqueue work; align and the get_* calls synchronise. */
#define NDTPSO_WINDOW_SIZE 100        /* NDT_WINDOW_SIZE, config.h:8 */
#define NDTPSO_MAX_POINTS_PER_CELL 50 /* NDT_MAX_POINTS_PER_CELL, config.h:5 */

typedef struct ndtpso_points ndtpso_points;
typedef struct ndtpso_map ndtpso_map;

typedef struct {
  uint32_t n_created, n_built; /* cells with points / cells in the alignment table (as of the last build) */
  uint32_t status;             /* bit 0: point pool exhausted, bit 1: an occupancy write fell outside the grid */
  uint32_t n_words;
  int32_t x0, x1, y0, y1;      /* bounding box of the built cells, cell coordinates */
  uint32_t og_min_x, og_max_x, og_min_y, og_max_y; /* s_occupancy_grid.{min,max}_{x,y}_ind (ndtframe.h:23-25) */
  int32_t pool_bump, pool_free; /* 512-byte point chunks handed out / on the free stack */
  uint64_t n_points;           /* points ever inserted */
} ndtpso_map_info;

int ndtpso_points_create(ndtpso_ctx *ctx, uint32_t capacity, ndtpso_points **out);
void ndtpso_points_destroy(ndtpso_points *pts);
/* loadLaser into a one-cell frame of `clip`'s size (NULL: unbounded); trans = the frame's s_trans (NULL = zero);
 * append != 0 keeps the points already loaded (a second loadLaser into the same frame).  Asynchronous. */
int ndtpso_points_load_scan(ndtpso_points *pts, const float *ranges, const ndtpso_scan_geom *geom, const double trans[3],
                            const ndtpso_grid *clip, int append);
int ndtpso_points_set(ndtpso_points *pts, const double *xy, uint32_t n);
int ndtpso_points_get(ndtpso_points *pts, double *xy, uint32_t max_points, uint32_t *n_points);

/* og_cell_size 0: no occupancy grid.  pool_bytes 0: 1 GiB of point storage (512 B per 32 points of one window slot).
 * Limit (the reference's `unsigned int numOfCells`, ndtframe.h:35, has none): fewer than 2^21 cells per frame (NDTPSO_E_ARG; a
 * frame costs 6.5 KB of HBM per cell: 9 GB for 300 m at 0.25 m).  The built cells may lie anywhere in it: a bounding box of more
 * than ~650 000 cells has its table assembled in HBM instead of LDS (build 0.24 ms instead of 0.05, alignment 0.5 instead of 0.36). */
int ndtpso_map_create(ndtpso_ctx *ctx, const ndtpso_grid *grid, double og_cell_size, uint64_t pool_bytes,
                      ndtpso_map **out);
void ndtpso_map_destroy(ndtpso_map *map);
/* NDTFrame::resetCells (ndtframe.cpp:208-212): NDTCell::reset (ndtcell.cpp:80-91) on every created cell -- sums, counts
 * and point vectors are cleared, the window's partial terms and created / built / mean / inverse covariance are kept */
int ndtpso_map_reset(ndtpso_map *map);
int ndtpso_map_clear(ndtpso_map *map); /* back to a freshly constructed frame */
int ndtpso_map_mark_unbuilt(ndtpso_map *map);
int ndtpso_map_insert(ndtpso_map *map, const ndtpso_points *pts, const double pose[3]);
int ndtpso_map_insert_host(ndtpso_map *map, const double *xy, uint32_t n, const double pose[3]);
int ndtpso_map_build(ndtpso_map *map);
/* Builds the cells and packs the alignment table NOW, ahead of the ndtpso_map_align (or ndtpso_map_build) that will ask
 * for them, so that they are off that call's critical path -- without changing what any call observes: everything the
 * build overwrites is logged, and an insert / reset / export arriving first puts the map back before it runs (the
 * reference builds lazily, core.cpp:27-28; a frame that is only ever updated is never built).  The occupancy grid is
 * rasterised when the speculation is committed.  A caller uses it after NDTFrame::update on a frame it aligns against. */
int ndtpso_map_speculate_build(ndtpso_map *map);
int ndtpso_map_align(ndtpso_map *map, const ndtpso_points *new_points, const double guess[3], const double deviation[3],
                     const ndtpso_pso_config *cfg, uint32_t seed, const int32_t *rand_table, int score_mode,
                     double out_pose[3], double *out_cost, ndtpso_align_stats *stats);
/* cost_function (core.cpp:26-48) of a loaded scan against the map at n_poses candidate poses (3 doubles each) */
int ndtpso_map_cost(ndtpso_map *map, ndtpso_points *new_points, const double *poses, uint32_t n_poses, int score_mode,
                    double *costs);
int ndtpso_map_get_info(ndtpso_map *map, ndtpso_map_info *info);
/* created cells in ascending index order (row.reserved = the cell's current window slot) */
int ndtpso_map_get_cells(ndtpso_map *map, ndtpso_cell_row *rows, uint32_t max_rows, uint32_t *n_rows);
/* stored points in cell order, window-slot order, insertion order (dumpMap's order, ndtframe.cpp:314-328);
 * slot0_only: the points cost_function visits when the frame is the NEW frame (core.cpp:33-36) */
int ndtpso_map_get_points(ndtpso_map *map, int slot0_only, double *xy, uint64_t max_points, uint64_t *n_points);
/* og[x + og_height * y] as the reference indexes it; extent = {min_x, max_x, min_y, max_y} */
int ndtpso_map_get_occupancy(ndtpso_map *map, int8_t *og, uint64_t og_bytes, uint32_t *og_width, uint32_t *og_height,
                             uint32_t extent[4]);

/* ---- one alignment on several compute units --------------------------
 * ndtpso_align, ndtpso_map_align and small batches of ndtpso_align_pairs* spread an alignment over K workgroups (each
 * keeps the table, the points and the whole swarm and runs the identical control flow; the cost evaluations of a round
 * are divided one item per wave and exchanged once per round).  Results are bit-identical to one workgroup; a cluster
 * whose workgroups cannot run together gives up after a bounded wait and the alignment is redone on one workgroup
 * (status bit 16 is internal and never returned).  Environment: NDTPSO_CLUSTER=0|K, NDTPSO_CLUSTER_WAVES=w. */

/* ---- fused batched scan pairs (BASELINE configs 3/4) ------------------- */
/* For pair b: reference frame <- ref scan loaded at identity and built; new frame <- new scan
 * (ndtpso_slam_node.cpp:186,229-230); pose_b = pso_optimization(guess_b, ref, new, deviation_b, cfg)
 * under the srand(seeds[b]) stream (or rand_tables + b*ndtpso_rand_draws(cfg) when not NULL).
 * ranges are [n_pairs][n_beams] row-major; guess/deviation/out_pose [n_pairs][3].
 * A pair the fused kernels cannot hold -- an occupied box beyond the largest cell table of a workgroup whose bitmap form does not
 * fit LDS either: cells of 0.125 m in a 60 m frame, a dozen pairs in 600 -- comes out of them with NDTPSO_STATUS_FLAGS set.  This
 * entry (host buffers, synchronous) then aligns it through a resident frame (ndtpso_map_*: table in HBM), same pose bit for bit --
 * so does ndtpso_align_pairs_sharded, on its first device; ndtpso_align_pairs_dev / ndtpso_align_pairs_sharded_dev, asynchronous
 * on device buffers, leave the flag to the caller. */
int ndtpso_align_pairs(ndtpso_ctx *ctx, uint32_t n_pairs, const float *ref_ranges, const float *new_ranges,
                       const ndtpso_scan_geom *geom, const ndtpso_grid *grid, const double *guess,
                       const double *deviation, const ndtpso_pso_config *cfg, const uint32_t *seeds,
                       const int32_t *rand_tables, int score_mode, double *out_pose, double *out_cost,
                       ndtpso_align_stats *stats);
/* same with DEVICE pointers; asynchronous on the context stream */
int ndtpso_align_pairs_dev(ndtpso_ctx *ctx, uint32_t n_pairs, const float *d_ref_ranges,
                           const float *d_new_ranges, const ndtpso_scan_geom *geom, const ndtpso_grid *grid,
                           const double *d_guess, const double *d_deviation, const ndtpso_pso_config *cfg,
                           const uint32_t *d_seeds, const int32_t *d_rand_tables, int score_mode,
                           double *d_out_pose, double *d_out_cost, ndtpso_align_stats *d_stats);
/* LDS bytes and workgroup size the fused kernel would be launched with (occupancy study; 0 if it does not fit) */
int ndtpso_align_pairs_footprint(const ndtpso_scan_geom *geom, const ndtpso_grid *grid,
                                 const ndtpso_pso_config *cfg, uint32_t *lds_bytes, uint32_t *block_threads);

/* How ndtpso_align_pairs would run a configuration (LDS cell-table sizing / occupancy study, BASELINE config 5), for a
 * batch with more pairs than the device has compute units.  Smaller batches use 16-wave workgroups, and below half the
 * compute units several workgroups share a pair (cluster mode): same layout per workgroup, different shape. */
typedef struct {
  uint32_t lds_bytes;        /* dynamic LDS per workgroup (0 if the configuration does not fit) */
  uint32_t block_threads;    /* workgroup size */
  uint32_t workgroups_per_cu;/* by LDS and by the 16-waves-per-CU register budget */
  uint32_t table_form;       /* 0 bitmap + true division, 1 bitmap + power-of-two cells, 2 dense u16 table (fp32 records),
                                8 / 9 dense u16 table with fp64 records (fp64 score; true division / power-of-two cells) */
  uint32_t swarm_in_hbm;     /* 1: the swarm state lives in an HBM workspace instead of LDS */
  uint32_t window_w, window_h; /* staging window in cells */
  uint32_t table_bytes;      /* LDS bytes of the cell index + records */
} ndtpso_pairs_plan;
int ndtpso_align_pairs_describe(const ndtpso_scan_geom *geom, const ndtpso_grid *grid, const ndtpso_pso_config *cfg,
                                int score_mode, uint32_t n_pairs, ndtpso_pairs_plan *out);

/* ---- one batch on several devices, one process (BASELINE config 4) ---------------------------------
 * The pairs are independent (each is one NDTFrame::align of a recorded pair, ndtpso_slam_node.cpp:194; the reference's
 * own parallelism is an OpenMP loop over particles, core.cpp:81), so the batch is partitioned by contiguous index
 * range -- device d of G takes ndtpso_shard_range(n_pairs, d, G) -- and each device runs ndtpso_align_pairs_dev on its
 * shard in a context and stream of its own, with no traffic between devices.  The only exchange is ONE ncclAllGather
 * (RCCL over xGMI, single-process communicator from ncclCommInitAll) of pose + cost, 4 doubles per pair, after which
 * every device holds the poses of the whole batch; device 0's copy is returned.  Results are those of
 * ndtpso_align_pairs on the same arguments, bit for bit, whatever the number of devices.
 * RCCL is loaded at run time: ndtpso_shard_group_create returns NDTPSO_E_HIP when it (or a listed device) is absent.
 * A group is used by one host thread at a time.  Every device has a host thread of its own inside the group: the uploads
 * of device d (blocking staged copies when the caller's memory is pageable), its launch and the copy of its statistics are
 * issued by that thread, all devices at once; the calling thread joins the ENQUEUES, issues the collective, and only then
 * waits for the devices.  Two flavours: host buffers (the scatter is part of the call) and `_dev` (each device's shard
 * already resident: per-device arrays of device pointers, entry d = the slice [first_d, last_d) on device d, complete
 * before the call).  ndtpso_shard_last_timing reports, for the last call, per device {start after the call's entry, uploads,
 * launches} and per call {until every device's work was enqueued, the collective's enqueue, total} in microseconds of
 * host time; ndtpso_shard_gathered(group, d) is device d's copy of the gathered batch, G blocks of [pose (M x 3) |
 * cost (M)] doubles with M = ceil(n_pairs / G), valid until the group's next call. */
typedef struct ndtpso_shard_group ndtpso_shard_group;
int ndtpso_shard_group_create(const int *devices, int n_devices, ndtpso_shard_group **out);
void ndtpso_shard_group_destroy(ndtpso_shard_group *group);
int ndtpso_shard_group_size(const ndtpso_shard_group *group);
const char *ndtpso_shard_last_error(const ndtpso_shard_group *group);
/* [first, last) of device `rank` of `n_devices`: contiguous, sizes differing by at most one */
void ndtpso_shard_range(uint32_t n_pairs, int rank, int n_devices, uint32_t *first, uint32_t *last);
int ndtpso_align_pairs_sharded(ndtpso_shard_group *group, uint32_t n_pairs, const float *ref_ranges,
                               const float *new_ranges, const ndtpso_scan_geom *geom, const ndtpso_grid *grid,
                               const double *guess, const double *deviation, const ndtpso_pso_config *cfg,
                               const uint32_t *seeds, const int32_t *rand_tables, int score_mode, double *out_pose,
                               double *out_cost, ndtpso_align_stats *stats);
int ndtpso_align_pairs_sharded_dev(ndtpso_shard_group *group, uint32_t n_pairs, const float *const *d_ref_ranges,
                                   const float *const *d_new_ranges, const ndtpso_scan_geom *geom,
                                   const ndtpso_grid *grid, const double *const *d_guess,
                                   const double *const *d_deviation, const ndtpso_pso_config *cfg,
                                   const uint32_t *const *d_seeds, const int32_t *const *d_rand_tables, int score_mode,
                                   double *out_pose /* may be NULL: results stay on the devices */, double *out_cost,
                                   ndtpso_align_stats *stats);
int ndtpso_shard_last_timing(const ndtpso_shard_group *group, double *per_device /* [G][3] or NULL */,
                             double *call /* [3] or NULL */);
/* the last call's collective on shard 0's stream, by device events: from the end of shard 0's own launch to the end of its
 * share of the gather (the wait for the slowest shard included); microseconds, 0 if the call failed */
double ndtpso_shard_last_gather_device_us(const ndtpso_shard_group *group);
const double *ndtpso_shard_gathered(const ndtpso_shard_group *group, int index);
/* What the group is, for a caller's record: how many shards, over what the poses are gathered, which RCCL, how many ranks
 * its communicator holds.  gather_kind 0: ONE ncclAllGather over RCCL (the product path); 1: staged through the host -- the
 * TEST mode NDTPSO_SHARD_VIRTUAL=G in the environment, in which a group has G shards whatever the length of the device
 * list (shard i on devices[i mod n], a context, stream and host thread each) so that a box with one device can run the
 * partition, the thread-per-shard scatter and the error paths of a G-device call; never a measurement of the collective.
 * ndtpso_shard_verify_gather (after a successful call): the number of shards whose send block is found bit for bit in
 * EVERY shard's gathered copy, i.e. the ranks the collective has demonstrably moved; G when all is well. */
typedef struct {
  int32_t n_shards;
  int32_t gather_kind;   /* 0 RCCL ncclAllGather, 1 host-staged (test mode) */
  int32_t rccl_version;  /* ncclGetVersion's code, 0 if RCCL is not loaded */
  int32_t comm_ranks;    /* ncclCommCount of the group's communicator, 0 in the test mode */
  int32_t devices[64];   /* device of shard i */
} ndtpso_shard_info;
int ndtpso_shard_group_describe(const ndtpso_shard_group *group, ndtpso_shard_info *out);
int ndtpso_shard_verify_gather(ndtpso_shard_group *group, int *ranks_seen);

/* ---- probes of the device arithmetic the parity claims rest on (tests/test_gpu_exp.py) --------------------------
 * NDTCell::normalDistribution ends in exp() (ndtcell.cpp:76) and transform_point takes the cosine and sine of the pose's
 * heading (core.h:28-31): on the device those are the ROCm math library's exp / sincos, and the fp64 score's own spelling
 * of exp(-q / 2) (exp_neg_half, ndtpso_kernels.hpp: the library's operations without its two range guards).
 * ndtpso_selftest_exp: for every binade b in [exp2_lo, exp2_hi], per_binade pseudo-random arguments x = -(1 + u) 2^b
 * (positive != 0: +), the library's exp(x) against exp_neg_half(-2 x) bit for bit (two NaNs count as equal); returns the
 * number checked, the number that differ and one differing argument (NaN if none).
 * ndtpso_device_math: kind 0: out0 = exp(x) (library); 1: out0 = exp_neg_half(-2 x); 2: out0 = sin(x), out1 = cos(x) from
 * one sincos(x) -- HOST pointers, n arguments. */
int ndtpso_selftest_exp(ndtpso_ctx *ctx, int exp2_lo, int exp2_hi, uint32_t per_binade, uint32_t seed, int positive,
                        uint64_t *checked, uint64_t *mismatched, double *first_bad);
int ndtpso_device_math(ndtpso_ctx *ctx, int kind, const double *x, uint32_t n, double *out0, double *out1);
/* The start-up known-answer check of NDTPSO_SCORE_EXACT.  The first request for the exact mode on a device (any entry
 * point, any context of the process) first sends a fixed small problem through EVERY kernel instantiation that arbitrates
 * and that the library's dispatchers can reach -- the fused pairs kernels over table-entry form x {one workgroup per
 * alignment without / with clipping trips, clusters of workgroups} x swarm in LDS / in HBM x plain / box-guard copies, and the
 * lone alignment's kernel (ndtpso_align, ndtpso_map_align) on one workgroup and on a cluster: 16 families, each forced by a
 * plan override and confirmed by the instantiation its launch recorded -- in the exact mode and in NDTPSO_SCORE_F64, and
 * compares poses and costs bit for bit (and checks, per family, that comparisons were in fact arbitrated).  If any family
 * differs the device is refused the exact mode: every later request for it runs NDTPSO_SCORE_F64 (the same results by
 * definition, at its speed), and ndtpso_last_error explains.  A check that could not run (no memory for its context, a HIP
 * error) is no verdict: that request runs NDTPSO_SCORE_F64 and the next one runs the check again (three attempts).
 * ndtpso_exact_check runs the check if it has not run and reports: state 1 passed, 2 refused; the comparisons the check
 * arbitrated in the pairs / lone-alignment families; its duration in ms.  ndtpso_exact_check_report writes one JSON object
 * with a line per family (name, passed / refused / not reachable on this device, alignments, arbitrated, mismatched) into
 * buf (at most cap - 1 characters) and returns the length the whole text needs.  NDTPSO_EXACT_CHECK=0 in the environment
 * skips the check. */
int ndtpso_exact_check(ndtpso_ctx *ctx, int *state, uint32_t *arbitrated_batch, uint32_t *arbitrated_single, double *ms);
int ndtpso_exact_check_report(ndtpso_ctx *ctx, char *buf, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
