// ndtpso_kernels.hpp -- device code of the gfx950 NDT-PSO alignment path.
//
// Hand-written for CDNA4 (wave64, 160 KiB LDS/CU).  No MFMA: the path is
// transform / hash / gather / exp work, bounded by VALU + LDS, not by a dense
// contraction.  One workgroup owns one alignment: the reference cell table
// (bitmap index + 64-byte records), the new scan's points and the whole swarm
// live in LDS for the 70 x 70 PSO; HBM sees only the two raw scans and 32 bytes
// of result per alignment.
//
// Built with -ffp-contract=off: every fp64 operation that must follow the
// reference's rounding (PSO update core.cpp:83-90, index arithmetic
// ndtframe.cpp:240-249, cell statistics ndtcell.cpp:36-111) is written as
// separate operations; fused multiply-adds appear only where spelled fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndtpso {

constexpr int kScoreF32 = 0;
constexpr int kScoreF64 = 1;
// NDTPSO_SCORE_EXACT (2) is not a third kernel family: it is the fp32-score kernel with an exact (fp64) table image at
// hand (EvalCtx::arb), which arbitrates every comparison its fp32 costs cannot decide -- see "arbitration" below.
constexpr int kWave = 64;
constexpr int kImageHeaderBytes = 64;
__host__ __device__ constexpr int align16_c(int x) { return (x + 15) & ~15; }
__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }
constexpr int kPointPad = 64;  // the point list is padded to whole waves with out-of-frame sentinels

// ---- uniform parameter blocks (kernel arguments -> SGPRs) ---------------

struct GridP {
  double hw, hh;  // width/2., height/2.  (frame bounds are (-hw, hw) x (-hh, hh), ndtframe.cpp:57-65)
  double cs;      // cell_side
  double inv_cs;  // 1/cell_side, used only when cs is a power of two (the division is then exact)
  int cs_pow2;
  int W, H;       // widthNumOfCells, heightNumOfCells (ndtframe.cpp:27-28)
};

// staging window: the sub-rectangle of the cell grid whose built cells are indexed in LDS
struct WinP {
  int x0, y0, w, h;
  int n_words;  // ceil(w*h/32) bitmap words
  int rec_cap;  // record capacity
};

struct ScanP {
  int n_beams;
  float amin, ainc, rmax, eps;
};

struct PsoP {
  int P, I;
  int G;  // particles evaluated per round: 2 x waves - 1 with a light wave (below), else a multiple of the wave count
  int light;  // k >= 2: wave 0 takes ONE item of a round, every other wave k -- wave 0 is the wave that commits the round
              // before and replays glibc's generator, and with a full share it kept the other waves waiting at the
              // round's barrier for exactly that long (one-workgroup kernels; a cluster deals its items differently)
  double w, c1, c2, wdamp;
};

// Reference cell table as the score loop reads it: structure of 16-byte arrays indexed by record slot
// (ds_read_b128 gathers: two slots collide on LDS banks only when they are congruent mod 16).
//   mean[slot] = NDTCell::mean                                 (both score paths)
//   ab[slot], cd[slot] = s_inv_covar rows (0,0),(0,1) / (1,0),(1,1)   (fp64 score path)
//   chol[slot] = {l11, l21, l22, 0}: with M = 0.5*log2(e)*s_inv_covar = L L^T (Cholesky),
//       exp(-d^T S d / 2) = exp2(-((l11 d0 + l21 d1)^2 + (l22 d1)^2)); no cancellation between large
//       terms for thin, rotated Gaussians, which is what makes fp32 sufficient here  (fp32 score path)
struct TableView {
  const uint2* bm;      // bitmap words {bits, exclusive prefix popcount} over the staging window
  const double2* mean;
  const double2* ab;
  const double2* cd;
  const float4* chol;
};
struct TableOut {  // same arrays, writable (table build); null = not wanted
  uint2* bm;
  double2* mean;
  double2* ab;
  double2* cd;
  float4* chol;
};

// Dense form used by the fp32-score fast path (power-of-two cell side): a u16 entry per cell of the
// staging window (plus one empty border row/column on the low sides) holding the LDS address / 16 of the
// cell's 32-byte record; cells that are not built point at a null record whose exponent is -inf, so a miss
// needs no mask, select or compare.  The table sits at LDS offset 0: entry address = cell index * 2.
// Record of a built cell in the dense form (round 3).  With (L11, L21, L22) the Cholesky factor of 0.5 log2(e) s_inv_covar
// in cell units and (mgx, mgy) the cell's mean in window cell coordinates, the exponent of a point at table coordinates
// (gx, gy) is q = a^2 + b^2 with
//     a = L11 d0 + L21 d1 = L11 (d0 + r d1),   r = L21 / L11      ->  a = l11 * float(fma(r, gy, gx) + alpha),  alpha = -(mgx + r mgy)
//     b = L22 d1                                                  ->  b = l22 * float(gy - mgy)
// The inner sum d0 + r d1 is formed in fp64 straight from the table coordinates and only then rounded to fp32; the
// magnitudes l11, l22 are fp32.  Rounds 1-2 rounded d0, d1 and all three factors to fp32 and formed `a` in fp32: in a
// thin cell that is rotated against the axes the two products of `a` cancel, so its rounding error was relative to
// |L11 d0| + |L21 d1|, up to a hundred times |a|, and the score's error depended on the shape of the cells.  Here the
// cancellation happens in fp64 -- the DIRECTION (1, r) of the thin axis is kept in fp64, only the scale is fp32 -- so
// `a` and `b` are each off by three fp32 roundings of themselves and the error of a term is bounded whatever the cell
// looks like (eval_items / verify_pose_wave / DESIGN 3.6).  |r| <= sqrt(1000): the determinant clamp of
// s_calc_covar_inverse (ndtcell.cpp:103-105) bounds the form's condition number.
// Same footprint as before: two 16-byte parts, A = {r, alpha} and B = {mgy, l11, l22}; one more fp64 operation and one
// fp32 operation less per point.  Records live in blocks of sixteen -- sixteen parts A, then the sixteen parts B 256
// bytes behind (an immediate offset) -- so that the 16-lane groups of a ds_read_b128 see sixteen different 16-byte
// positions of a 256-byte LDS row.
// Record 0 is the null record every non-built cell points at: alpha = +inf, l11 = 1 -> a = +inf -> exp2(-inf) = 0, so
// a miss needs no mask, select or compare.  A cell whose inverse covariance has no Cholesky factor (make_chol) gets
// alpha = NaN: a pose that touches it scores NaN and is handed to the fp64 form.
struct __attribute__((aligned(16))) DenseRecA {
  double r, alpha;
};
struct __attribute__((aligned(16))) DenseRecB {
  double mgy;
  float l11, l22;
};
static_assert(sizeof(DenseRecA) == 16 && sizeof(DenseRecB) == 16, "dense record parts must be 16 bytes");
constexpr unsigned kDenseRecBOff = 256;
__host__ __device__ inline unsigned dense_rec_pos(unsigned k) { return ((k >> 4) << 9) + ((k & 15u) << 4); }
__host__ __device__ inline unsigned dense_rec_index(unsigned pos) { return ((pos >> 9) << 4) + ((pos >> 4) & 15u); }
__host__ __device__ inline int dense_rec_bytes(int n_records) { return 512 * ((n_records + 15) / 16); }
struct DenseP {
  int dw, dh;    // dense window size in cells, = wn.w + 1, wn.h + 1
  int ox, oy;    // grid coordinates of dense cell (0,0), = wn.x0 - 1, wn.y0 - 1
  int rec_off;   // LDS byte offset of the rec_cap + 1 records (dense_rec_pos; record 0 = null), 16-byte aligned
  int clip;      // 1: W*cs > width or H*cs > height -- the last cells overhang the frame, test the upper bounds
  double xmax, ymax;  // frame's upper bounds in window cell coordinates: width/cs - ox, height/cs - oy
};
__host__ __device__ inline void dense_set_limits(DenseP& d, double hw, double hh, double inv_cs) {
  d.xmax = (2. * hw) * inv_cs - (double)d.ox;
  d.ymax = (2. * hh) * inv_cs - (double)d.oy;
}
// The dense table holds (dw + 1) x (dh + 1) u16 entries: dw x dh is the staging window with its empty low border
// column / row; the extra high column and row are always null, so that a clamped coordinate needs no range test.
__host__ __device__ inline int dense_stride(int dw) { return dw + 1; }
__host__ __device__ inline int dense_entries(int dw, int dh) { return (dw + 1) * (dh + 1); }
__host__ __device__ inline int dense_tab_bytes(int dw, int dh) { return align16_c(dense_entries(dw, dh) * 2); }
constexpr int kRecImageBytes = 64;  // per record in the HBM image: mean, ab, cd, chol

struct ImageHeader {
  uint32_t n_built, n_created, status, pad[13];
};
static_assert(sizeof(ImageHeader) == kImageHeaderBytes, "header must be 64 bytes");

struct CellRow {  // == ndtpso_cell_row
  int32_t index, count, built, reserved;
  double mean[2];
  double icov[4];
};

struct AlignStats {  // == ndtpso_align_stats
  uint32_t n_points, n_built, cost_evals, rounds, gbest_updates, status, t_start, t_end;
};


__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// table image in HBM: [header | bitmap words | mean[cap] | ab[cap] | cd[cap] | chol[cap]] (all 16-B elements)
__host__ __device__ inline int image_mean_offset(int n_words) { return kImageHeaderBytes + align16(n_words * 8); }
__host__ __device__ inline int image_ab_offset(int n_words, int cap) { return image_mean_offset(n_words) + 16 * cap; }
__host__ __device__ inline int image_cd_offset(int n_words, int cap) { return image_mean_offset(n_words) + 32 * cap; }
__host__ __device__ inline int image_chol_offset(int n_words, int cap) { return image_mean_offset(n_words) + 48 * cap; }
__host__ __device__ inline int image_bytes(int n_words, int cap) { return image_mean_offset(n_words) + 64 * cap; }

// Cholesky factor of 0.5*log2(e)*[[a, (b+c)/2], [(b+c)/2, d]] for the fp32 score path, times `scale`
// (1 for records in metres, cell_side for records in cell units), rounded to fp32 once.
// Only a positive semi-definite form has one.  An inverse covariance that is not -- NaN entries (three coincident
// points: covariance 0, determinant 0, ndtcell.cpp:104-110), or the indefinite leftovers of resetCells -- gets
// out[3] = NaN instead of 0: that term is added to every exponent, so a pose that touches the cell scores NaN in
// the fp32 form, and the callers hand NaN scores to the fp64 form, which evaluates the reference's expression as
// it stands (NaN, or an exponential above 1).  Negative values at rounding level (a wall: rank-one covariance,
// sliding-window cancellation) are clamped as before.
// make_chol_d: the factor in fp64 (the dense records keep it so); returns whether the form has one.
__host__ __device__ inline bool make_chol_d(double a, double b, double c, double d, double out[3], double scale = 1.) {
  const double k = 0.72134752044448170368;  // 0.5 * log2(e)
  const double A = k * a, B = k * (0.5 * (b + c)), D = k * d;
  const double T = fabs(A) + fabs(D), tol = 1e-12 * T;
  // (comparisons are false for NaN, and T = inf makes the last one false: both land in `bad`)
  const bool ok = (A >= -tol) && (D >= -tol) && (A * D - B * B >= -tol * T) && (T < 1.7e308);
  double l11 = 0., l21 = 0.;
  if (A > 0.) {
    l11 = sqrt(A);
    l21 = B / l11;
  }
  const double rem = D - l21 * l21;
  const double l22 = rem > 0. ? sqrt(rem) : 0.;
  out[0] = l11 * scale;
  out[1] = l21 * scale;
  out[2] = l22 * scale;
  return ok;
}
__host__ __device__ inline void make_chol(double a, double b, double c, double d, float out[4], double scale = 1.) {
  double l[3];
  const bool ok = make_chol_d(a, b, c, d, l, scale);
  out[0] = (float)l[0];
  out[1] = (float)l[1];
  out[2] = (float)l[2];
  out[3] = ok ? 0.f : __builtin_nanf("");
}

// ---- NDTCell::s_calc_covar_inverse (ndtcell.cpp:93-111) -------------------
// The eigenvalues come from EigenSolver<Matrix2d>(covar).pseudoEigenvalueMatrix().diagonal() in the reference
// (ndtcell.cpp:96-97).  Restated here in the operation order of Eigen 3.3.7 for a 2x2 -- RealSchur::compute (scale by
// the largest |coefficient|, the Hessenberg step is the identity), computeFromHessenberg / findSmallSubdiagEntry,
// splitOffTwoRows (p, q, z, one Givens rotation applied on the left and on the right), unscale, EigenSolver::compute --
// so that the larger eigenvalue, which becomes the determinant of a thin cell (.001 * large^2), carries Eigen's
// roundings and not those of a closed form (a few ulp apart in 11 % of the cells of the synthetic world).  The tests'
// CPU restatement of the reference carries the same function, line by line.  fp64 division and sqrt are correctly
// rounded on the device and the library is built with -ffp-contract=off.
__device__ __forceinline__ void eigen_givens(double p, double q, double& c, double& s) {  // JacobiRotation::makeGivens, real case
  if (q == 0.) {
    c = p < 0. ? -1. : 1.;
    s = 0.;
  } else if (p == 0.) {
    c = 0.;
    s = q < 0. ? 1. : -1.;
  } else if (fabs(p) > fabs(q)) {
    const double t = q / p;
    double u = sqrt(1. + t * t);
    if (p < 0.) u = -u;
    c = 1. / u;
    s = -t * c;
  } else {
    const double t = p / q;
    double u = sqrt(1. + t * t);
    if (q < 0.) u = -u;
    s = -1. / u;
    c = -t * s;
  }
}

__device__ __forceinline__ void covar_inverse_eigen(double c00, double c01, double c10, double c11, double* inv) {
  constexpr double kDblMin = 2.2250738585072014e-308, kDblEps = 2.220446049250313e-16;
  double e0 = 0., e1 = 0.;
  // RealSchur::compute: scale = matrix.cwiseAbs().maxCoeff()
  double scale = fabs(c00);
  if (fabs(c01) > scale) scale = fabs(c01);
  if (fabs(c10) > scale) scale = fabs(c10);
  if (fabs(c11) > scale) scale = fabs(c11);
  if (!(scale < kDblMin)) {
    double t00 = c00 / scale, t01 = c01 / scale, t10 = c10 / scale, t11 = c11 / scale;
    const double norm = (fabs(t00) + fabs(t10)) + (fabs(t01) + fabs(t11));  // computeNormOfT (only tested against 0)
    if (norm != 0.) {
      double sd = (fabs(t00) + fabs(t11)) * kDblEps;  // findSmallSubdiagEntry(1)
      if (!(sd > kDblMin)) sd = kDblMin;
      if (fabs(t10) <= sd) {
        t10 = 0.;
      } else {  // splitOffTwoRows(1)
        const double p = 0.5 * (t00 - t11);
        const double q = p * p + t10 * t01;
        if (q >= 0.) {
          const double z = sqrt(fabs(q));
          double c, sn;
          eigen_givens(p >= 0. ? p + z : p - z, t10, c, sn);
          if (!(c == 1. && sn == 0.)) {
            const double ms = -sn;
            // applyOnTheLeft(0, 1, rot.adjoint()): rows, rotation (c, -s)
            double x = t00, y = t10;
            t00 = c * x + ms * y;
            t10 = -ms * x + c * y;
            x = t01, y = t11;
            t01 = c * x + ms * y;
            t11 = -ms * x + c * y;
            // applyOnTheRight(0, 1, rot): columns, rotation rot.transpose() = (c, -s)
            x = t00, y = t01;
            t00 = c * x + ms * y;
            t01 = -ms * x + c * y;
            x = t10, y = t11;
            t10 = c * x + ms * y;
            t11 = -ms * x + c * y;
          }
          t10 = 0.;
        }
      }
    }
    t00 *= scale;  // m_matT *= scale
    t10 *= scale;
    t11 *= scale;
    if (t10 == 0.) {
      e0 = t00;
      e1 = t11;
    } else {  // complex pair (not reachable for a symmetric matrix): its real part on both diagonal entries
      e0 = e1 = t11 + 0.5 * (t00 - t11);
    }
  }
  const double large_val = (e0 > e1) ? e0 : e1;  // ndtcell.cpp:100
  const double small_val = (e0 < e1) ? e0 : e1;  // ndtcell.cpp:101
  double det;
  if (small_val < .001 * large_val)
    det = .001 * large_val * large_val;  // ndtcell.cpp:103-105
  else
    det = c00 * c11 - c10 * c01;  // Matrix2d::determinant(), ndtcell.cpp:107
  inv[0] = c11 / det;  // ndtcell.cpp:109-110
  inv[1] = -c01 / det;
  inv[2] = -c10 / det;
  inv[3] = c00 / det;
}

// ---- small device helpers ------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// sum over the 64 lanes, fixed association order, result uniform: 4 DPP steps inside each row of 16
// lanes (quad xor 1, quad xor 2, half-row mirror, row mirror), then the four row sums via readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int l = __double2loint(v), h = __double2hiint(v);  // every lane is a valid source for these controls
  const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// lane 15 of a row into every lane of the next row (ROWS 0xa: rows 1 and 3 take it), lane 31 into rows 2 and 3 (0xc); 0. elsewhere
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_bcast_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  // the four row sums r0 .. r3 as (r0 + r1) + (r2 + r3): row 1 takes r0, row 3 takes r2 (row_bcast:15), then row 3 takes
  // row 1's r1 + r0 (row_bcast:31) and holds (r3 + r2) + (r1 + r0) -- the same two additions per level as four readlanes
  // and three scalar-operand adds gave (an addition does not care which operand comes first), in 8 instructions instead of 13
  v += dpp_bcast_f64<0x142, 0xa>(v);  // row_bcast:15
  v += dpp_bcast_f64<0x143, 0xc>(v);  // row_bcast:31
  return readlane_f64(v, 63);
}

// cell coordinates of an in-frame point, following NDTFrame::getCellIndex (ndtframe.cpp:240-249):
// floor((x + width/2.)/cell_side), floor((y + height/2.)/cell_side).  x + hw > 0 inside the
// strict bounds, so truncation equals floor.
template <bool POW2>
__device__ __forceinline__ void cell_coords(const GridP& g, double qx, double qy, int& ix, int& iy) {
  const double ux = qx + g.hw, uy = qy + g.hh;
  if (POW2) {
    ix = (int)(ux * g.inv_cs);
    iy = (int)(uy * g.inv_cs);
  } else {
    ix = (int)(ux / g.cs);
    iy = (int)(uy / g.cs);
  }
}
__device__ __forceinline__ void cell_coords_rt(const GridP& g, double qx, double qy, int& ix, int& iy) {
  if (g.cs_pow2) cell_coords<true>(g, qx, qy, ix, iy); else cell_coords<false>(g, qx, qy, ix, iy);
}

#ifndef NDTPSO_DIAG  // timing diagnostics of the dense score loop (score_trip_dense): extra work per 64-point chunk
#define NDTPSO_DIAG 0
#endif
#ifndef NDTPSO_UNROLL
#define NDTPSO_UNROLL 4
#endif
#ifndef NDTPSO_PRIO_SHARE
#define NDTPSO_PRIO_SHARE 9  // sixteenths of the time the later-dispatched partner on a CU holds the higher priority (8, 10, 11 re-measured in round 2: 9 stays)
#endif
#ifndef NDTPSO_MAD24_INDEX
#define NDTPSO_MAD24_INDEX 1
#endif
#ifndef NDTPSO_ALTERNATE_PRIO
#define NDTPSO_ALTERNATE_PRIO 1
#endif
#ifndef NDTPSO_STREAM
#define NDTPSO_STREAM 1  // one-workgroup kernels: an iteration's items dealt by ticket, one barrier per phase (eval_stream); 0: rounds
#endif

// ---- K1: NDT score of one candidate pose, one wave --------------------------
//
// cost_function (core.cpp:26-48) + transform_point (core.h:28-31) + getCellIndex
// (ndtframe.cpp:240-249) + NDTCell::normalDistribution (ndtcell.cpp:70-78), fused.
// Lanes stride the points (coalesced 16-B LDS reads); the cell lookup is one 8-byte bitmap word
// {bits, prefix} + popcount; the record gather is 2 (fp32 score) or 3 (fp64) ds_read_b128.
// The body is branch-free (misses read slot 0 and are masked) and written phase by phase over U
// points per lane, so the three dependent LDS round trips of U independent points overlap.
// Returns the cost (-sum) on every lane.
template <int MODE, bool POW2, int U, bool DUMP>
__device__ __forceinline__ void score_trip(const GridP& g, const WinP& wn, const TableView& T,
                                           const double2* __restrict__ pts, int base, int n, double c, double s,
                                           double tx, double ty, double (&acc)[4], int32_t* __restrict__ dump) {
  const int lane = lane_id();
  double2 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = pts[base + u * kWave + lane];

  double qx[U], qy[U];
  unsigned lin[U];
  bool ok[U];
  int tagv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (MODE == kScoreF64) {
      qx[u] = (p[u].x * c - p[u].y * s) + tx;  // reference rounding, no fma
      qy[u] = (p[u].x * s + p[u].y * c) + ty;
    } else {
      qx[u] = fma(p[u].x, c, fma(-p[u].y, s, tx));
      qy[u] = fma(p[u].x, s, fma(p[u].y, c, ty));
    }
    const bool inframe = (int)(fabs(qx[u]) < g.hw) & (int)(fabs(qy[u]) < g.hh);  // strict bounds, ndtframe.cpp:242
    int ix, iy;
    cell_coords<POW2>(g, qx[u], qy[u], ix, iy);
    const bool wrap = (ix == g.W);  // fl(x + w/2) == w: the reference's linear index lands in the next row
    if (DUMP) tagv[u] = ix + g.W * iy;
    ix = wrap ? 0 : ix;
    iy = wrap ? iy + 1 : iy;
    if (DUMP) tagv[u] = (!inframe || iy >= g.H) ? -1 : tagv[u];
    const unsigned rx = (unsigned)(ix - wn.x0), ry = (unsigned)(iy - wn.y0);
    ok[u] = (int)inframe & (int)(rx < (unsigned)wn.w) & (int)(ry < (unsigned)wn.h);
    lin[u] = ok[u] ? __umul24(ry, (unsigned)wn.w) + rx : 0u;
  }

  unsigned long long e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = reinterpret_cast<const unsigned long long*>(T.bm)[lin[u] >> 5];

  unsigned slot[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned bits = (unsigned)e[u], pre = (unsigned)(e[u] >> 32), bit = lin[u] & 31u;
    ok[u] = (int)ok[u] & (int)((bits >> bit) & 1u);
    const unsigned sl = pre + __popc(bits & ((1u << bit) - 1u));
    slot[u] = ok[u] ? sl : 0u;
  }

  if constexpr (MODE == kScoreF64) {
    double2 m[U], ab[U], cd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = T.mean[slot[u]];
      ab[u] = T.ab[slot[u]];
      cd[u] = T.cd[slot[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double d0 = qx[u] - m[u].x, d1 = qy[u] - m[u].y;
      const double r0 = d0 * ab[u].x + d1 * cd[u].x;  // (diff^T * inv_covar), ndtcell.cpp:73-75
      const double r1 = d0 * ab[u].y + d1 * cd[u].y;
      const double x = -(r0 * d0 + r1 * d1) / 2.;
      acc[u] += exp(ok[u] ? x : -(double)__builtin_inff());  // exp(-inf) = 0; a NaN exponent stays NaN (select, not fmin)
    }
  } else {
    double2 m[U];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = T.mean[slot[u]];
      f[u] = T.chol[slot[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float d0 = (float)(qx[u] - m[u].x), d1 = (float)(qy[u] - m[u].y);
      const float a = fmaf(f[u].x, d0, f[u].y * d1), b = f[u].z * d1;
      // chol.w is 0 (NaN for a cell whose inverse covariance has no Cholesky factor, see make_chol): folding it in
      // keeps the gather a single ds_read_b128.  Misses are masked through the exponent (-inf; exp2(-inf) = 0) so
      // the whole body stays straight-line code.
      const float x = -fmaf(a, a, fmaf(b, b, f[u].w));
      const float q = ok[u] ? x : -__builtin_inff();  // (a select, not fminf: a NaN exponent must stay NaN)
      acc[u] += (double)__builtin_amdgcn_exp2f(q);
    }
  }
  if (DUMP) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kWave + lane;
      if (i < n) dump[i] = (tagv[u] < 0) ? -1 : (ok[u] ? tagv[u] : -2);
    }
  }
}


// ---- fp32-score fast path: dense table, folded index arithmetic ------------------------------------------
//
// The window-relative cell coordinates come straight out of the transform:
//   gx = x*C - y*S + TX,  C = cos/cs, S = sin/cs, TX = (tx + w/2)/cs - ox      (2 fp64 FMAs per axis)
// and the Mahalanobis form is evaluated in cell units against the dense records (DenseRecA / B: a = l11 * float(fma(r, gy,
// gx) + alpha), b = l22 * float(gy - mgy)).  Per point: 4 + 1 fp64 FMA, 2 fp64 adds, 2 cvt to the index, mad24, 3 LDS
// reads, 2 cvt to fp32, 3 fp32 mul + 1 fma, exp2, and per four points three fp32 adds, one cvt + fp64 add.
// For a power-of-two cell side floor((x + w/2)/cs) = floor(fl(x + w/2) * 2^k) and the scaling commutes with
// rounding; for any other cell side 1/cs is rounded once more.  Either way gx is within ~1e-14 cells of the
// reference's value, so only a point that close to a cell edge can bin differently (probability ~1e-7 per
// alignment); the fp64 score mode keeps the reference's rounding step by step and has no such caveat.
struct DenseItem {  // per-pose constants
  double C, S, TX, TY;
  double XMAX, YMAX;  // frame's upper bounds in window cell coordinates (used when DenseP::clip)
};
__device__ __forceinline__ DenseItem dense_item(const GridP& g, const DenseP& dn, double c, double s, double tx,
                                                double ty) {
  DenseItem it;
  it.C = c * g.inv_cs;
  it.S = s * g.inv_cs;
  it.TX = (tx + g.hw) * g.inv_cs - (double)dn.ox;
  it.TY = (ty + g.hh) * g.inv_cs - (double)dn.oy;
  it.XMAX = dn.xmax;
  it.YMAX = dn.ymax;
  return it;
}

// BYTE: the table entries are the records' LDS byte addresses themselves (possible when every record lies below
// 64 KB: PATH 3, the fused pairs kernel) instead of addresses in 16-byte units -- one shift less per point.
// NOCLAMP: the caller has established that no point of the list leaves the table under this pose (DenseGuard): the
// two clamps go -- v_min_u32 costs 1.5 x a v_fma_f32 on this chip (scripts/ubench_valu.hip), the pair a tenth of the
// loop's issue time -- and nothing else changes, so the result is the clamped form's bit for bit.
// MASKLAST (with NOCLAMP, on the trip that holds the list's last chunk): the padding behind point n - 1 is sent to the
// null entry by index -- the clamps are what used to catch the sentinel coordinates.
// FOLD: how the trip's terms enter the lane accumulator.  The sum's order is the list's: groups of four chunks, each
// folded as (t0 + t1) + (t2 + t3) in fp32 and added in fp64, then the chunks left over one by one -- whatever trips
// the chunks are scored in.  kFoldByU: U = 4 / 8 whole groups, 5 a group and one left-over chunk, 1 a left-over chunk;
// kFoldSingles: left-over chunks only; kFoldGroupSingles: a group and U - 4 left-over chunks; kFoldGroupCarry: a group,
// then the first half of the next group goes out in `carry`; kFoldCarryGroup: `carry` + this trip's first two chunks
// complete that group, a whole group follows.  (Six-chunk trips: three dependent LDS round trips per trip is what a
// wave waits for, so 17 chunks go as 6 + 6 + 5 instead of 4 + 4 + 4 + 5: + 4.8 % on the batch.)
enum { kFoldByU = 0, kFoldSingles, kFoldGroupSingles, kFoldGroupCarry, kFoldCarryGroup };
template <int U, bool DUMP, bool CLIP, bool BYTE = false, bool NOCLAMP = false, bool MASKLAST = false, int FOLD = kFoldByU>
__device__ __forceinline__ void score_trip_dense(const GridP& g, const DenseP& dn, const unsigned char* lds0,
                                                 const double2* __restrict__ pts, int base, int n,
                                                 const DenseItem& it, double (&acc)[4], int32_t* __restrict__ dump,
                                                 float* carry = nullptr) {
  const int lane = lane_id();
  double2 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = pts[base + u * kWave + lane];
  double gx[U], gy[U];
  unsigned lin[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    gx[u] = fma(p[u].x, it.C, fma(-p[u].y, it.S, it.TX));
    gy[u] = fma(p[u].x, it.S, fma(p[u].y, it.C, it.TY));
    // Out-of-window coordinates clamp into the always-null high column / row (negative ones convert to huge
    // unsigned values first; column / row 0 is the empty low border): no range test, no select.
    const unsigned rx = NOCLAMP ? (unsigned)(int)gx[u] : min((unsigned)(int)gx[u], (unsigned)dn.dw),
                   ry = NOCLAMP ? (unsigned)(int)gy[u] : min((unsigned)(int)gy[u], (unsigned)dn.dh);
#if NDTPSO_MAD24_INDEX
    // one v_mad_u32_u24 (+ the shift of the table read below) instead of v_mul_u32_u24 + v_add_lshl_u32: on this chip
    // the 24-bit multiply alone costs as much as the multiply-add (scripts/ubench_valu.hip), the shift half of it
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lin[u]) : "v"(ry), "s"((unsigned)dense_stride(dn.dw)), "v"(rx));
#else
    lin[u] = __umul24(ry, (unsigned)dense_stride(dn.dw)) + rx;
#endif
    // A grid whose last cells overhang the frame (width/cs not an integer): points past the frame's upper
    // bound are rejected by NDTFrame::getCellIndex (ndtframe.cpp:242) although a cell exists there.
    if constexpr (CLIP) lin[u] = ((int)(gx[u] < it.XMAX) & (int)(gy[u] < it.YMAX)) ? lin[u] : 0u;  // cell 0: null
    if constexpr (MASKLAST)
      if (u == U - 1) lin[u] = (base + u * kWave + lane < n) ? lin[u] : 0u;
  }
  // the dense table starts at LDS address 0 and records are addressed absolutely: plain shifts, no base add
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  (void)lds0;
  unsigned e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *(lds_u16_t)(uintptr_t)(lin[u] << 1);
  v2d_t ra[U], rb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned r = BYTE ? e[u] : e[u] << 4;
    ra[u] = *(lds_d2_t)(uintptr_t)r;                     // {r, alpha}
    rb[u] = *(lds_d2_t)(uintptr_t)(r + kDenseRecBOff);   // {mgy, (l11, l22) in the two words of the second double}
  }
  float t[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // (the two fp32 magnitudes are taken out of the register pair as they are: no arithmetic touches them as a double)
    const float l11 = __int_as_float(__double2loint(rb[u].y)), l22 = __int_as_float(__double2hiint(rb[u].y));
    const float a = l11 * (float)(fma(ra[u].x, gy[u], gx[u]) + ra[u].y);
    const float b = l22 * (float)(gy[u] - rb[u].x);
    t[u] = __builtin_amdgcn_exp2f(-fmaf(a, a, b * b));  // null record: alpha = +inf -> 0
#if NDTPSO_DIAG >= 1 && NDTPSO_DIAG <= 16  // timing diagnostics: that many extra independent v_fma_f32 per chunk
    {
      float dz = a;
#pragma unroll
      for (int q = 0; q < NDTPSO_DIAG; ++q) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(dz) : "v"(b));
    }
#elif NDTPSO_DIAG == 17  // one extra (independent) 16-byte LDS read per chunk
    {
      v4f_t dz;
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(dz) : "v"((unsigned)(lane * 16)));
    }
#elif NDTPSO_DIAG == 18  // four extra fp64 FMAs per chunk
    {
      double dz;
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(dz) : "v"(gx[u]));
    }
#endif
  }
  // the U terms (each in [0,1]) are summed in fp32 first, then folded into the fp64 lane accumulator
  if constexpr (FOLD == kFoldSingles) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc[0] += (double)t[u];
  } else if constexpr (FOLD == kFoldGroupSingles) {
    acc[0] += (double)((t[0] + t[1]) + (t[2] + t[3]));
#pragma unroll
    for (int u = 4; u < U; ++u) acc[0] += (double)t[u];
  } else if constexpr (FOLD == kFoldGroupCarry) {
    static_assert(U == 6, "a group and half a group");
    acc[0] += (double)((t[0] + t[1]) + (t[2] + t[3]));
    *carry = t[4] + t[5];
  } else if constexpr (FOLD == kFoldCarryGroup) {
    static_assert(U == 6, "half a group and a group");
    acc[0] += (double)(*carry + (t[0] + t[1]));
    acc[0] += (double)((t[2] + t[3]) + (t[4] + t[5]));
  } else if constexpr (U == 8) {  // two groups of four in flight together, folded in the order two U = 4 trips would be
    acc[0] += (double)((t[0] + t[1]) + (t[2] + t[3]));
    acc[0] += (double)((t[4] + t[5]) + (t[6] + t[7]));
  } else if constexpr (U == 5) {  // the last group of four and the one chunk behind it in one trip, folded as they
    acc[0] += (double)((t[0] + t[1]) + (t[2] + t[3]));  // would be by a U = 4 trip followed by a U = 1 trip
    acc[0] += (double)t[4];
  } else if constexpr (U == 4)
    acc[0] += (double)((t[0] + t[1]) + (t[2] + t[3]));
  else if constexpr (U == 2)
    acc[0] += (double)(t[0] + t[1]);
  else
    acc[0] += (double)t[0];
  if (DUMP) {
    const unsigned null16 = BYTE ? (unsigned)dn.rec_off : (unsigned)dn.rec_off >> 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kWave + lane;
      const double fx = gx[u] + (double)dn.ox, fy = gy[u] + (double)dn.oy;  // frame cell coordinates
      const bool inframe = fx > 0. && fx < (double)g.W && fy > 0. && fy < (double)g.H;
      const int cell = (int)fx + g.W * (int)fy;
      if (i < n) dump[i] = !inframe ? -1 : (e[u] != null16 ? cell : -2);
    }
  }
}

// WIDE: a wave that has its SIMD to itself (cluster mode) keeps eight chunks in flight instead of four to cover the
// LDS latency; the terms are folded in exactly the order of the U = 4 loop, so the sum is the same bit for bit.
template <bool DUMP, bool CLIP, bool WIDE = false, bool BYTE = false>
__device__ __forceinline__ double eval_pose_wave_dense_c(const GridP& g, const DenseP& dn, const unsigned char* lds0,
                                                         const double2* __restrict__ pts, int n, double c, double s,
                                                         double tx, double ty, int32_t* __restrict__ dump) {
  constexpr int U = NDTPSO_UNROLL;
  const DenseItem it = dense_item(g, dn, c, s, tx, ty);
  double acc[4] = {0., 0., 0., 0.};  // the dense trips fold their U terms in fp32 and use acc[0] only
  const int n_pad = round_up(n, kWave);
  int base = 0;
  if constexpr (WIDE && U == 4 && !DUMP)
    for (; base + 8 * kWave <= n_pad; base += 8 * kWave)
      score_trip_dense<8, DUMP, CLIP, BYTE>(g, dn, lds0, pts, base, n, it, acc, dump);
  for (; base + U * kWave <= n_pad; base += U * kWave)
    score_trip_dense<U, DUMP, CLIP, BYTE>(g, dn, lds0, pts, base, n, it, acc, dump);
  for (; base < n_pad; base += kWave) score_trip_dense<1, DUMP, CLIP, BYTE>(g, dn, lds0, pts, base, n, it, acc, dump);
  return -wave_sum(acc[0]);
}
// The no-clamp loop's trips (DenseGuard).  The sum is the clamped sequence's bit for bit (score_trip_dense's FOLD);
// the trip that holds the list's last chunk masks its padding.  A 1081-beam scan is 17 chunks: those go as two
// six-chunk trips and a five-chunk one, straight-line code (three dependent LDS round trips per trip is what a wave
// waits for: + 4 %); every other length in four-chunk trips as before.  This kernel's register allocation is decided
// by everything inlined into it: the same three trips behind a general decomposition into six-, five- and four-chunk
// trips for any length lost the whole gain (219 k against 230 k align/s), a second straight-line case for 16 chunks
// cost 2 %, the same trips in the frame-clipping variant as well another 2 % (so a grid whose last cells overhang
// the frame keeps the four-chunk trips), and moving the clamped and clipping forms into an out-of-line function cost
// more than all of it (215 k: what is live across the call gets pinned).
template <bool CLIP, bool BYTE>
__device__ __forceinline__ void noclamp_trips(const GridP& g, const DenseP& dn, const unsigned char* lds0,
                                              const double2* __restrict__ pts, int n, int n_pad, const DenseItem& it,
                                              double (&acc)[4]) {
  const int chunks = n_pad >> 6, rem = chunks & 3;
  if (chunks == 0) return;  // (an empty list)
  int base = 0;
  if constexpr (!CLIP && !BYTE) {
    if (chunks == 32) {  // 2048 beams (BASELINE config 5), as eval_item_wave_dense's clamped sequence
      float carry;
      score_trip_dense<6, false, CLIP, BYTE, true, false, kFoldGroupCarry>(g, dn, lds0, pts, 0, n, it, acc, nullptr, &carry);
      score_trip_dense<6, false, CLIP, BYTE, true, false, kFoldCarryGroup>(g, dn, lds0, pts, 6 * kWave, n, it, acc, nullptr, &carry);
      score_trip_dense<6, false, CLIP, BYTE, true, false, kFoldGroupCarry>(g, dn, lds0, pts, 12 * kWave, n, it, acc, nullptr, &carry);
      score_trip_dense<6, false, CLIP, BYTE, true, false, kFoldCarryGroup>(g, dn, lds0, pts, 18 * kWave, n, it, acc, nullptr, &carry);
      score_trip_dense<4, false, CLIP, BYTE, true>(g, dn, lds0, pts, 24 * kWave, n, it, acc, nullptr);
      score_trip_dense<4, false, CLIP, BYTE, true, true>(g, dn, lds0, pts, 28 * kWave, n, it, acc, nullptr);
      return;
    }
  }
  if (!CLIP && chunks == 17) {
    float carry;
    score_trip_dense<6, false, CLIP, BYTE, true, false, kFoldGroupCarry>(g, dn, lds0, pts, 0, n, it, acc, nullptr, &carry);
    score_trip_dense<6, false, CLIP, BYTE, true, false, kFoldCarryGroup>(g, dn, lds0, pts, 6 * kWave, n, it, acc, nullptr, &carry);
    score_trip_dense<5, false, CLIP, BYTE, true, true>(g, dn, lds0, pts, 12 * kWave, n, it, acc, nullptr);
    return;
  }
  const bool five = rem == 1 && chunks >= 5;
  const int fours = five ? (chunks - 5) >> 2 : (rem == 0 ? (chunks >> 2) - 1 : chunks >> 2);
  for (int i = 0; i < fours; ++i, base += 4 * kWave)
    score_trip_dense<4, false, CLIP, BYTE, true>(g, dn, lds0, pts, base, n, it, acc, nullptr);
  if (five)
    score_trip_dense<5, false, CLIP, BYTE, true, true>(g, dn, lds0, pts, base, n, it, acc, nullptr);
  else if (rem == 0)
    score_trip_dense<4, false, CLIP, BYTE, true, true>(g, dn, lds0, pts, base, n, it, acc, nullptr);
  else {
    for (; base + kWave < n_pad; base += kWave)
      score_trip_dense<1, false, CLIP, BYTE, true>(g, dn, lds0, pts, base, n, it, acc, nullptr);
    score_trip_dense<1, false, CLIP, BYTE, true, true>(g, dn, lds0, pts, base, n, it, acc, nullptr);
  }
}

// ---- two items per wave (short lists: NDTPSO_PAIR_ITEMS kernels) ---------------------------------------------------------
// An evaluation of a few chunks is mostly what surrounds its trips -- the ticket, the item's record, the lane reduction, the
// decision: 80 of 195 vector instructions at six chunks (profiles/r06_phase_budget_361.json).  Here a wave scores TWO items at
// once, lanes 0-31 the first and lanes 32-63 the second, each half over the whole list 32 points at a time: the trips' work
// per item is what it was (an instruction costs the same with 32 lanes busy or 64 ... both halves are busy), everything around
// them is done once for the two.  The pose constants are per-lane registers in the one-item trips already.  No-clamp form
// only (both items inside the DenseGuard; the caller takes anything else through the one-item forms).  The sum of an item is
// its half's: per lane the list's half-chunks in order, groups of four folded in fp32 like the one-item trips' chunks, then
// the 32 lanes by the first five steps of wave_sum -- another order than the one-item forms' (the fp32 mode's costs differ in
// the last bits between the two kinds of kernel; which kind runs is decided by the scan's length alone).
template <int U, bool BYTE, bool MASKLAST>
__device__ __forceinline__ void score_trip_half(const DenseP& dn, const double2* __restrict__ pts, int base, int n,
                                                const DenseItem& it, double& acc) {
  const int hl = lane_id() & 31;
  double2 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = pts[base + u * 32 + hl];
  double gx[U], gy[U];
  unsigned lin[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    gx[u] = fma(p[u].x, it.C, fma(-p[u].y, it.S, it.TX));
    gy[u] = fma(p[u].x, it.S, fma(p[u].y, it.C, it.TY));
    const unsigned rx = (unsigned)(int)gx[u], ry = (unsigned)(int)gy[u];
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lin[u]) : "v"(ry), "s"((unsigned)dense_stride(dn.dw)), "v"(rx));
    if constexpr (MASKLAST)
      if (u == U - 1) lin[u] = (base + u * 32 + hl < n) ? lin[u] : 0u;
  }
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  unsigned e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *(lds_u16_t)(uintptr_t)(lin[u] << 1);
  v2d_t ra[U], rb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned r = BYTE ? e[u] : e[u] << 4;
    ra[u] = *(lds_d2_t)(uintptr_t)r;
    rb[u] = *(lds_d2_t)(uintptr_t)(r + kDenseRecBOff);
  }
  float t[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float l11 = __int_as_float(__double2loint(rb[u].y)), l22 = __int_as_float(__double2hiint(rb[u].y));
    const float a = l11 * (float)(fma(ra[u].x, gy[u], gx[u]) + ra[u].y);
    const float b = l22 * (float)(gy[u] - rb[u].x);
    t[u] = __builtin_amdgcn_exp2f(-fmaf(a, a, b * b));
  }
  // (fp32 within a trip -- a group of four, then what is left of the trip -- fp64 across trips)
  if constexpr (U == 6) {
    acc += (double)((t[0] + t[1]) + (t[2] + t[3]));
    acc += (double)(t[4] + t[5]);
  } else if constexpr (U == 5) {
    acc += (double)((t[0] + t[1]) + (t[2] + t[3]));
    acc += (double)t[4];
  } else if constexpr (U == 4)
    acc += (double)((t[0] + t[1]) + (t[2] + t[3]));
  else if constexpr (U == 3)
    acc += (double)((t[0] + t[1]) + t[2]);
  else if constexpr (U == 2)
    acc += (double)(t[0] + t[1]);
  else
    acc += (double)t[0];
}
// -> the item's cost in lane 31 (first item) and lane 63 (second item) only
template <bool BYTE>
__device__ __forceinline__ double eval_pair_half(const DenseP& dn, const double2* __restrict__ pts, int n, const DenseItem& it) {
  double acc = 0.;
  const int n_pad = round_up(n, 32), chunks = n_pad >> 5;
  if (chunks > 0) {
    // trips of up to six half-chunks, as even as they come (three dependent LDS round trips per trip are what a wave waits
    // for): 12 -> 6 + 6, 17 -> 6 + 6 + 5, 23 -> 6 + 6 + 6 + 5; the list's last trip masks its padding
    const int trips = (chunks + 5) / 6, small = chunks / trips, larger = chunks - small * trips;  // `larger` trips of small + 1 first
    int base = 0;
    for (int t = 0; t < trips; ++t) {
      const int u = small + (t < larger ? 1 : 0);
      if (t + 1 < trips) {
        if (u == 6) score_trip_half<6, BYTE, false>(dn, pts, base, n, it, acc);
        else if (u == 5) score_trip_half<5, BYTE, false>(dn, pts, base, n, it, acc);
        else score_trip_half<4, BYTE, false>(dn, pts, base, n, it, acc);  // (several trips: none shorter than four)
      } else {
        switch (u) {
          case 6: score_trip_half<6, BYTE, true>(dn, pts, base, n, it, acc); break;
          case 5: score_trip_half<5, BYTE, true>(dn, pts, base, n, it, acc); break;
          case 4: score_trip_half<4, BYTE, true>(dn, pts, base, n, it, acc); break;
          case 3: score_trip_half<3, BYTE, true>(dn, pts, base, n, it, acc); break;
          case 2: score_trip_half<2, BYTE, true>(dn, pts, base, n, it, acc); break;
          default: score_trip_half<1, BYTE, true>(dn, pts, base, n, it, acc); break;
        }
      }
      base += u * 32;
    }
  }
  double v = acc;
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  v += dpp_bcast_f64<0x142, 0xa>(v);  // row_bcast:15: lane 31 = rows 0 + 1, lane 63 = rows 2 + 3
  return -v;
}

// the same with the folded constants already at hand (the PSO keeps them with each proposal)
// NOCLIP: the kernel was chosen by the host for a grid whose cells do not overhang the frame (DenseP::clip == 0), and
// carries none of the clipping variants of the trips.
template <bool WIDE, bool BYTE, bool NOCLAMP = false, bool NOCLIP = false>
__device__ __forceinline__ double eval_item_wave_dense(const GridP& g, const DenseP& dn, const unsigned char* lds0,
                                                       const double2* __restrict__ pts, int n, const DenseItem& it) {
  constexpr int U = NDTPSO_UNROLL;
  double acc[4] = {0., 0., 0., 0.};
  const int n_pad = round_up(n, kWave);
  int base = 0;
  if constexpr (NOCLAMP) {
    static_assert(U == 4 && !WIDE, "the no-clamp form is the one-workgroup kernels'");
    // (a grid whose last cells overhang the frame: the guard keeps scan B's disc below the frame's upper bounds as well --
    // k_align_pairs --, so no point of a pose under it can fail the clip test and the trips need not make it)
    noclamp_trips<false, BYTE>(g, dn, lds0, pts, n, n_pad, it, acc);
    return -wave_sum(acc[0]);
  }
  // a 2048-beam list (BASELINE config 5; 32 chunks, entries in 16-byte units) in six-chunk trips, as the no-clamp loop
  // scores 17 chunks (noclamp_trips): 6 + 6 + 6 + 6 + 4 + 4, straight-line, folded in the four-chunk order
  if constexpr (!BYTE && NOCLIP && U == 4 && !WIDE) {
    if (n_pad == 32 * kWave) {
      float carry;
      score_trip_dense<6, false, false, BYTE, false, false, kFoldGroupCarry>(g, dn, lds0, pts, 0, n, it, acc, nullptr, &carry);
      score_trip_dense<6, false, false, BYTE, false, false, kFoldCarryGroup>(g, dn, lds0, pts, 6 * kWave, n, it, acc, nullptr, &carry);
      score_trip_dense<6, false, false, BYTE, false, false, kFoldGroupCarry>(g, dn, lds0, pts, 12 * kWave, n, it, acc, nullptr, &carry);
      score_trip_dense<6, false, false, BYTE, false, false, kFoldCarryGroup>(g, dn, lds0, pts, 18 * kWave, n, it, acc, nullptr, &carry);
      score_trip_dense<4, false, false, BYTE>(g, dn, lds0, pts, 24 * kWave, n, it, acc, nullptr);
      score_trip_dense<4, false, false, BYTE>(g, dn, lds0, pts, 28 * kWave, n, it, acc, nullptr);
      return -wave_sum(acc[0]);
    }
  }
  // a remainder of exactly five chunks (1081 beams are 17) goes as one trip instead of a trip of four and a lonely one
  if (!NOCLIP && dn.clip) {
    if constexpr (WIDE && U == 4)
      for (; base + 8 * kWave <= n_pad; base += 8 * kWave)
        score_trip_dense<8, false, true, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
    for (; base + U * kWave <= n_pad && (U != 4 || WIDE || n_pad - base != 5 * kWave); base += U * kWave)
      score_trip_dense<U, false, true, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
    if constexpr (U == 4 && !WIDE)
      if (n_pad - base == 5 * kWave) {
        score_trip_dense<5, false, true, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
        base += 5 * kWave;
      }
    for (; base < n_pad; base += kWave) score_trip_dense<1, false, true, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
  } else {
    if constexpr (WIDE && U == 4)
      for (; base + 8 * kWave <= n_pad; base += 8 * kWave)
        score_trip_dense<8, false, false, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
    for (; base + U * kWave <= n_pad && (U != 4 || WIDE || n_pad - base != 5 * kWave); base += U * kWave)
      score_trip_dense<U, false, false, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
    if constexpr (U == 4 && !WIDE)
      if (n_pad - base == 5 * kWave) {
        score_trip_dense<5, false, false, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
        base += 5 * kWave;
      }
    for (; base < n_pad; base += kWave) score_trip_dense<1, false, false, BYTE>(g, dn, lds0, pts, base, n, it, acc, nullptr);
  }
  return -wave_sum(acc[0]);
}

template <bool DUMP, bool WIDE = false, bool BYTE = false>
__device__ __forceinline__ double eval_pose_wave_dense(const GridP& g, const DenseP& dn, const unsigned char* lds0,
                                                       const double2* __restrict__ pts, int n, double c, double s,
                                                       double tx, double ty, int32_t* __restrict__ dump) {
  if (dn.clip) return eval_pose_wave_dense_c<DUMP, true, WIDE, BYTE>(g, dn, lds0, pts, n, c, s, tx, ty, dump);
  return eval_pose_wave_dense_c<DUMP, false, WIDE, BYTE>(g, dn, lds0, pts, n, c, s, tx, ty, dump);
}

__device__ __forceinline__ double uniform_f64(double v) {  // a value every lane holds, moved to scalar registers
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// The fp64 score's trip for a pose under the guard (DenseGuard in metres: every point of scan B lands strictly inside the
// frame and inside the table's window), U chunks without padding, table in LDS.  The terms are score_trip's bit for bit;
// what differs is how the integer side gets there:
//   * power-of-two cell side: (q + hw) * inv_cs is fma(q, inv_cs, hw * inv_cs) -- scaling by a power of two commutes with
//     the rounding of the sum (no overflow; a sum below 2^-1021 truncates to cell 0 either way), one instruction for two;
//   * no frame / wrap / window tests, one subtraction for the window's origin;
//   * the built bit comes out as a mask (v_bfe_i32, width 1) and goes into the exponent's high word with one v_bfi_b32:
//     a miss scores exp(-65536 - something) = +0. exactly as exp(-inf) did (the library returns 0 below -745.2), a hit
//     keeps its exponent untouched, NaN included; the bit-field instructions take the bit number from the low five bits
//     of the cell's linear index themselves;
//   * a miss reads the record its prefix count points at (<= n_built <= rec_cap: inside LDS, value unused) instead of
//     selecting record 0, and every address is one v_lshl_add_u32.
// 9 integer instructions per point for the lookup instead of 17, 2 fp64 ones fewer.
template <bool POW2, int U>
__device__ __forceinline__ void score_trip_guarded(const GridP& g, const WinP& wn, const TableView& T,
                                                   const double2* __restrict__ pts, int base, double c, double s,
                                                   double tx, double ty, double hwi, double hhi, double (&acc)[4]) {
  typedef const unsigned long long __attribute__((address_space(3))) * lds_u64_t;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  const unsigned bm_a = (unsigned)(uintptr_t)(const uint2 __attribute__((address_space(3)))*)T.bm;
  const unsigned mean_a = (unsigned)(uintptr_t)(const double2 __attribute__((address_space(3)))*)T.mean;
  const unsigned ab_a = (unsigned)(uintptr_t)(const double2 __attribute__((address_space(3)))*)T.ab;
  const unsigned cd_a = (unsigned)(uintptr_t)(const double2 __attribute__((address_space(3)))*)T.cd;
  const unsigned origin = (unsigned)(wn.y0 * wn.w + wn.x0);
  const int lane = lane_id();
  double2 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = pts[base + u * kWave + lane];
  double qx[U], qy[U];
  unsigned lin[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    qx[u] = (p[u].x * c - p[u].y * s) + tx;  // reference rounding, no fma
    qy[u] = (p[u].x * s + p[u].y * c) + ty;
    int ix, iy;
    if constexpr (POW2) {
      ix = (int)fma(qx[u], g.inv_cs, hwi);  // hwi = hw * inv_cs, hhi = hh * inv_cs (exact)
      iy = (int)fma(qy[u], g.inv_cs, hhi);
    } else {
      cell_coords<false>(g, qx[u], qy[u], ix, iy);
    }
    unsigned t;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(iy), "s"(wn.w), "v"(ix));
    lin[u] = t - origin;
  }
  unsigned long long e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned a;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a) : "v"(lin[u] >> 5), "s"(bm_a));
    e[u] = *(lds_u64_t)(uintptr_t)a;
  }
  int hit[U];
  v2d_t m[U], ab[U], cd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned bits = (unsigned)e[u], pre = (unsigned)(e[u] >> 32);
    unsigned below;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(hit[u]) : "v"(bits), "v"(lin[u]));   // -1: built
    asm("v_bfe_u32 %0, %1, 0, %2" : "=v"(below) : "v"(bits), "v"(lin[u]));     // the bits below the cell's
    const unsigned slot = pre + __popc(below);
    unsigned a0, a1, a2;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(a0) : "v"(slot), "s"(mean_a));
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(a1) : "v"(slot), "s"(ab_a));
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(a2) : "v"(slot), "s"(cd_a));
    m[u] = *(lds_d2_t)(uintptr_t)a0;
    ab[u] = *(lds_d2_t)(uintptr_t)a1;
    cd[u] = *(lds_d2_t)(uintptr_t)a2;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double d0 = qx[u] - m[u].x, d1 = qy[u] - m[u].y;
    const double r0 = d0 * ab[u].x + d1 * cd[u].x;  // (diff^T * inv_covar), ndtcell.cpp:73-75
    const double r1 = d0 * ab[u].y + d1 * cd[u].y;
    const double x = -(r0 * d0 + r1 * d1) / 2.;
    const int hi = (__double2hiint(x) & hit[u]) | ((int)0xC0F00000 & ~hit[u]);
    acc[u] += exp(__hiloint2double(hi, __double2loint(x)));
  }
}

// ---- fp64 score on a dense cell table (round 5; PATH 8 / 9 of the fused pairs kernel) ------------------------------------
//
// The fp64 score mode -- the reference's operations in the reference's order -- looked its cells up through the bitmap
// table: bitmap word, bit-field extract, popcount, three record addresses; 59 vector instructions per point of which ~30 were
// not fp64 arithmetic, and 21 the math library's exp.  Here:
//   * the dense u16 table of the fp32-score kernels (one entry per cell of a window sized per alignment, a null entry for
//     every cell that is not built), entries = LDS byte addresses of 48-byte fp64 records {mean, s_inv_covar row 0, row 1}
//     read with three 16-byte loads off ONE address (immediate offsets): per point one mad24, one shift-add, one 2-byte read;
//   * record 0, the null record, scores exp(-4.5e5 ... -1.8e6) = +0. for any point inside a frame of up to 65 535 m: a miss
//     adds +0. as the reference's `continue` does (core.cpp:40), with no mask, select or compare;
//   * exp(-q / 2) spelt out (exp_neg_half): the library's reduction, polynomial and ldexp -- same constants, same operations,
//     same order -- without its two range guards and with the factor -1/2 folded into the constants (below).
// 43 vector instructions per point (8 transform, 6 index, 11 differences and quadratic form, 17 exponential, 1 add).
constexpr int kD64RecBytes = 48;
__host__ __device__ constexpr bool path_is_dense64(int path) { return path == 8 || path == 9; }
__host__ __device__ inline int d64_rec_bytes(int n_records) { return kD64RecBytes * n_records; }
constexpr uint32_t kHdrWildCell = 64u;  // ImageHeader::status, internal: d64_cell_tame failed for a built cell
constexpr double kD64NullMean = 1e5, kD64NullScale = 1e-4;  // null record: mean (1e5, 1e5), s_inv_covar 1e-4 I

// exp(x) at x = -q / 2 as __ocml_exp_f64 computes it (ROCm 7: t = x log2(e); n = rint(t); r = fma(-ln2_hi, n, x);
// r = fma(-ln2_lo, n, r); eleven Horner steps; ldexp(p, n)), from q itself:
//   * x = -q / 2 is exact, so t = fl(x L) = fl(q (-L / 2)) and the reduction's two fused steps give R = -2 r exactly from q
//     (a power-of-two factor commutes with every rounding; nothing here comes near the subnormal range -- r is 0 or above
//     1e-17 in magnitude unless n = 0, where a subnormal q leaves the polynomial at exactly 1 either way);
//   * Horner in R = -2 r with coefficient k scaled by (-2)^-k carries P_k = (-2)^-k p_k, every step the library's step times a
//     power of two: P_0 = p_0 bit for bit.  The last two constants are -0.5 and 1 (inline operands).
//   * the library then returns +inf for x > 1024 and 0 for x < -1075.  Between those bounds it relies on v_ldexp_f64 for
//     overflow, gradual underflow and +0. -- so does this form, everywhere: for |x| < 2^40 (n exact, r accurate, p in
//     [0.7, 1.5]) ldexp(p, n) IS +inf above 1024 and +0. below -1075, and a NaN stays a NaN through every step.  Beyond 2^40
//     (an infinite or absurd exponent: a singular cell's inverse covariance) the guards differ from what the arithmetic gives,
//     which is why a table holding such a cell is not scored by this form (build_table_wg: kStatusNeedsBitmap).
// tests/test_gpu_exp.py compares it with the library's exp on the device over every binade of [-1100, 64] and the specials.
__device__ __forceinline__ double exp_neg_half(double q) {
  const double t = q * __longlong_as_double(0xbfe71547652b82feLL);  // -log2(e) / 2
  const double dn = __builtin_rint(t);                              // v_rndne_f64
  double r = fma(dn, __longlong_as_double(0x3ff62e42fefa39efLL), q);  // 2 ln2_hi
  r = fma(dn, __longlong_as_double(0x3c8abc9e3b39803fLL), r);         // 2 ln2_lo
  // c_k (-2)^-k: the library's coefficients (c11 ... c2 = 0x3e5ade156a5dcb37, 0x3e928af3fca7ab0c, 0x3ec71dee623fde64,
  // 0x3efa01997c89e6b0, 0x3f2a01a014761f6e, 0x3f56c16c1852b7b0, 0x3f81111111122322, 0x3fa55555555502a1, 0x3fc5555555555511,
  // 0x3fe000000000000b) with k subtracted from the exponent field and the sign set for odd k
  double p = __longlong_as_double((long long)0xbdaade156a5dcb37ULL);             // c11 / -2048
  p = fma(r, p, __longlong_as_double(0x3df28af3fca7ab0cLL));                      // c10 / 1024
  p = fma(r, p, __longlong_as_double((long long)0xbe371dee623fde64ULL));          // c9 / -512
  p = fma(r, p, __longlong_as_double(0x3e7a01997c89e6b0LL));                      // c8 / 256
  p = fma(r, p, __longlong_as_double((long long)0xbeba01a014761f6eULL));          // c7 / -128
  p = fma(r, p, __longlong_as_double(0x3ef6c16c1852b7b0LL));                      // c6 / 64
  p = fma(r, p, __longlong_as_double((long long)0xbf31111111122322ULL));          // c5 / -32
  p = fma(r, p, __longlong_as_double(0x3f655555555502a1LL));                      // c4 / 16
  p = fma(r, p, __longlong_as_double((long long)0xbf95555555555511ULL));          // c3 / -8
  p = fma(r, p, __longlong_as_double(0x3fc000000000000bLL));                      // c2 / 4
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  int n;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(dn));  // (saturating; 0 for a NaN -- the C cast is undefined there)
  return __builtin_ldexp(p, n);
}

// U chunks of a pose under the guard (every point strictly inside the frame and inside the table's window; DenseGuard in
// metres as for score_trip_guarded).  MASK: the trip's last chunk holds the list's padding -- those lanes read the null entry.
// The terms are score_trip<kScoreF64>'s bit for bit.  tab_a: LDS byte address of the table's entry for frame cell (0, 0)
// (may lie "below" the table: unsigned arithmetic); stride: entries per table row.
template <bool POW2, int U, bool MASK>
__device__ __forceinline__ void score_trip_d64(const GridP& g, unsigned stride, unsigned tab_a, unsigned null_a,
                                               const double2* __restrict__ pts, int base, int n, double c, double s,
                                               double tx, double ty, double hwi, double hhi, double (&acc)[4]) {
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  const int lane = lane_id();
  double2 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = pts[base + u * kWave + lane];
  double qx[U], qy[U];
  unsigned ea[U];
  int ix[U], iy[U];
  [[maybe_unused]] bool amb = false;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    qx[u] = (p[u].x * c - p[u].y * s) + tx;  // transform_point, reference rounding (core.h:28-31): no fma
    qy[u] = (p[u].x * s + p[u].y * c) + ty;
    if constexpr (POW2) {
      ix[u] = (int)fma(qx[u], g.inv_cs, hwi);  // = (int)((qx + hw) * inv_cs): hwi = hw * inv_cs, a power-of-two scaling
      iy[u] = (int)fma(qy[u], g.inv_cs, hhi);
    } else {
      // Any other cell side: floor(fl((x + w/2) / cs)) (ndtframe.cpp:244-245) by reciprocal.  t = fl(u fl(1 / cs)) and the
      // correctly rounded quotient both lie within 2.2e-11 of u / cs (u / cs < 65536 inside the frame), so their floors can
      // differ only if an integer lies that close to t: then -- 2e-10 of all points -- the trip takes the true divisions
      // (below).  Two fp64 divisions per point were 60 of this trip's 110 instructions per point.
      const double ux = (qx[u] + g.hw) * g.inv_cs, uy = (qy[u] + g.hh) * g.inv_cs;
      ix[u] = (int)ux;
      iy[u] = (int)uy;
      amb |= (int)(fabs(ux - __builtin_rint(ux)) < 1e-10) | (int)(fabs(uy - __builtin_rint(uy)) < 1e-10);
    }
  }
  if constexpr (!POW2) {
    if (__builtin_expect(__any((int)amb), 0)) {
#pragma unroll
      for (int u = 0; u < U; ++u) cell_coords<false>(g, qx[u], qy[u], ix[u], iy[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned t;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(iy[u]), "s"(stride), "v"(ix[u]));
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(ea[u]) : "v"(t), "s"(tab_a));
    if constexpr (MASK)
      if (u == U - 1) ea[u] = (base + u * kWave + lane < n) ? ea[u] : null_a;
  }
  unsigned e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *(lds_u16_t)(uintptr_t)ea[u];
  v2d_t m[U], ab[U], cd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    m[u] = *(lds_d2_t)(uintptr_t)e[u];
    ab[u] = *(lds_d2_t)(uintptr_t)(e[u] + 16u);
    cd[u] = *(lds_d2_t)(uintptr_t)(e[u] + 32u);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double d0 = qx[u] - m[u].x, d1 = qy[u] - m[u].y;
    const double r0 = d0 * ab[u].x + d1 * cd[u].x;  // (diff^T * inv_covar), ndtcell.cpp:73-75
    const double r1 = d0 * ab[u].y + d1 * cd[u].y;
    acc[u] += exp_neg_half(r0 * d0 + r1 * d1);
  }
}

// The same for a pose outside the guard: frame, wrap and window tests per point as getCellIndex makes them
// (ndtframe.cpp:240-249); a point that fails one adds +0. (core.cpp:40).
template <bool POW2, int U>
__device__ __forceinline__ void score_trip_d64_tested(const GridP& g, const DenseP& dn, unsigned stride, unsigned tab0,
                                                      const double2* __restrict__ pts, int base, double c, double s,
                                                      double tx, double ty, double (&acc)[4]) {
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  const int lane = lane_id();
  double2 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = pts[base + u * kWave + lane];
  double qx[U], qy[U];
  unsigned lin[U];
  bool in[U];
  int cx[U], cy[U];
  [[maybe_unused]] bool amb = false;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    qx[u] = (p[u].x * c - p[u].y * s) + tx;
    qy[u] = (p[u].x * s + p[u].y * c) + ty;
    if constexpr (POW2) {
      cell_coords<true>(g, qx[u], qy[u], cx[u], cy[u]);
    } else {  // by reciprocal, the true divisions only where an integer lies within 1e-10 of a quotient (score_trip_d64)
      const double ux = (qx[u] + g.hw) * g.inv_cs, uy = (qy[u] + g.hh) * g.inv_cs;
      cx[u] = (int)ux;
      cy[u] = (int)uy;
      amb |= (int)(fabs(ux - __builtin_rint(ux)) < 1e-10) | (int)(fabs(uy - __builtin_rint(uy)) < 1e-10);
    }
  }
  if constexpr (!POW2) {
    if (__builtin_expect(__any((int)amb), 0)) {
#pragma unroll
      for (int u = 0; u < U; ++u) cell_coords<false>(g, qx[u], qy[u], cx[u], cy[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool inframe = (int)(fabs(qx[u]) < g.hw) & (int)(fabs(qy[u]) < g.hh);  // strict bounds, ndtframe.cpp:242
    int ix = cx[u], iy = cy[u];
    const bool wrap = (ix == g.W);  // fl(x + w/2) == w: the reference's linear index lands in the next row
    ix = wrap ? 0 : ix;
    iy = wrap ? iy + 1 : iy;
    const unsigned rx = (unsigned)(ix - dn.ox), ry = (unsigned)(iy - dn.oy);
    in[u] = (int)inframe & (int)(iy < g.H) & (int)(rx <= (unsigned)dn.dw) & (int)(ry <= (unsigned)dn.dh);
    lin[u] = in[u] ? ry * stride + rx : 0u;  // entry 0: the empty low border, null
  }
  unsigned e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *(lds_u16_t)(uintptr_t)(tab0 + (lin[u] << 1));
  v2d_t m[U], ab[U], cd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    m[u] = *(lds_d2_t)(uintptr_t)e[u];
    ab[u] = *(lds_d2_t)(uintptr_t)(e[u] + 16u);
    cd[u] = *(lds_d2_t)(uintptr_t)(e[u] + 32u);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double d0 = qx[u] - m[u].x, d1 = qy[u] - m[u].y;
    const double r0 = d0 * ab[u].x + d1 * cd[u].x;
    const double r1 = d0 * ab[u].y + d1 * cd[u].y;
    const double t = exp_neg_half(r0 * d0 + r1 * d1);
    acc[u] += in[u] ? t : 0.;  // (the coordinates of a point outside the frame may be anything, NaN included)
  }
}

// One pose, one wave: eval_pose_wave_t<kScoreF64>'s trips in its order -- groups of four chunks into the four lane
// accumulators, the chunks behind the last group into the first -- so the sum is that function's bit for bit.
// d64_tab: LDS byte address of the u16 table; dn.rec_off: of record 0 (the null record).
template <bool POW2, bool GUARD>
__device__ __forceinline__ double eval_pose_wave_d64(const GridP& g, const DenseP& dn, unsigned d64_tab,
                                                     const double2* __restrict__ pts, int n, double c, double s,
                                                     double tx, double ty) {
  constexpr int U = 4;
  double acc[4] = {0., 0., 0., 0.};
  const int n_pad = round_up(n, kWave);
  const unsigned stride = (unsigned)dense_stride(dn.dw);
  int base = 0;
  if constexpr (GUARD) {
    const double hwi = uniform_f64(g.hw * g.inv_cs), hhi = uniform_f64(g.hh * g.inv_cs);  // wave-uniform: scalar registers
    const unsigned tab_a = d64_tab - 2u * ((unsigned)dn.oy * stride + (unsigned)dn.ox);
    // (entry 0 of the table -- the low border's corner -- is a null entry: the padding's lanes read it)
    for (; base + U * kWave <= n; base += U * kWave)  // (the steady state: whole groups of four chunks, no padding)
      score_trip_d64<POW2, U, false>(g, stride, tab_a, d64_tab, pts, base, n, c, s, tx, ty, hwi, hhi, acc);
    for (; base + U * kWave <= n_pad; base += U * kWave)  // (at most one: a group whose last chunk holds the padding)
      score_trip_d64<POW2, U, true>(g, stride, tab_a, d64_tab, pts, base, n, c, s, tx, ty, hwi, hhi, acc);
    for (; base + kWave <= n; base += kWave)
      score_trip_d64<POW2, 1, false>(g, stride, tab_a, d64_tab, pts, base, n, c, s, tx, ty, hwi, hhi, acc);
    for (; base < n_pad; base += kWave)
      score_trip_d64<POW2, 1, true>(g, stride, tab_a, d64_tab, pts, base, n, c, s, tx, ty, hwi, hhi, acc);
  } else {
    for (; base + U * kWave <= n_pad; base += U * kWave)
      score_trip_d64_tested<POW2, U>(g, dn, stride, d64_tab, pts, base, c, s, tx, ty, acc);
    for (; base < n_pad; base += kWave)
      score_trip_d64_tested<POW2, 1>(g, dn, stride, d64_tab, pts, base, c, s, tx, ty, acc);
  }
  return -wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

// pts must be padded to a multiple of kPointPad with out-of-frame sentinels (pad_points_wg)
template <int MODE, bool POW2, bool DUMP, bool GUARD = false>
__device__ __forceinline__ double eval_pose_wave_t(const GridP& g, const WinP& wn, const TableView& T,
                                                   const double2* __restrict__ pts, int n, double c, double s,
                                                   double tx, double ty, int32_t* __restrict__ dump) {
  constexpr int U = NDTPSO_UNROLL;
  double acc[4] = {0., 0., 0., 0.};
  const int n_pad = round_up(n, kWave);
  int base = 0;
  static_assert(!GUARD || (MODE == kScoreF64 && !DUMP), "the guarded form is the fp64 score's, without cell dump");
  if constexpr (GUARD) {  // (the same trips in the same order: the guarded form wherever a trip holds no padding)
    const double hwi = uniform_f64(g.hw * g.inv_cs), hhi = uniform_f64(g.hh * g.inv_cs);  // wave-uniform: scalar registers
    for (; base + U * kWave <= n; base += U * kWave)
      score_trip_guarded<POW2, U>(g, wn, T, pts, base, c, s, tx, ty, hwi, hhi, acc);
  }
  for (; base + U * kWave <= n_pad; base += U * kWave)
    score_trip<MODE, POW2, U, DUMP>(g, wn, T, pts, base, n, c, s, tx, ty, acc, dump);
  for (; base < n_pad; base += kWave)
    score_trip<MODE, POW2, 1, DUMP>(g, wn, T, pts, base, n, c, s, tx, ty, acc, dump);
  return -wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
}
template <int MODE, bool POW2>
__device__ __forceinline__ double eval_pose_wave(const GridP& g, const WinP& wn, const TableView& T,
                                                 const double2* __restrict__ pts, int n, double c, double s,
                                                 double tx, double ty) {
  return eval_pose_wave_t<MODE, POW2, false>(g, wn, T, pts, n, c, s, tx, ty, nullptr);
}
template <int MODE, bool POW2>
__device__ __forceinline__ double eval_pose_wave_dump(const GridP& g, const WinP& wn, const TableView& T,
                                                      const double2* __restrict__ pts, int n, double c, double s,
                                                      double tx, double ty, int32_t* __restrict__ dump) {
  return eval_pose_wave_t<MODE, POW2, true>(g, wn, T, pts, n, c, s, tx, ty, dump);
}

// fill pts[n .. round_up(n, kPointPad)) with points no pose can bring into the frame
__device__ inline void pad_points_wg(double2* pts, int n) {
  const int n_pad = round_up(n, kPointPad);
  for (int i = n + threadIdx.x; i < n_pad; i += blockDim.x) pts[i] = make_double2(1e6, 1e6);  // beyond any uint16-metre frame, finite in fp32
}

// ---- K3a: LaserScan -> points (NDTFrame::loadLaser, ndtframe.cpp:144-185) --------------------
//
// Workgroup-cooperative, order preserving.  ranges: global; out: LDS or global (generic).
// Returns the number of surviving points (uniform).  `s_cnt` is a >= 17-int LDS scratch.
// clip_hw/clip_hh > 0: additionally drop points outside that frame, as NDTFrame::addPoint does when the scan is
// loaded into a frame of that size (ndtframe.cpp:215-235; the node's per-scan frame has the map's frame size).
// dirs: (cos, sin) of every beam's angle -- index_to_angle (core.h:40-42) in fp32, then ONE glibc sincos() of the
// widened value -- computed by the host (beam_directions in ndtpso_hip.hip): the device's own sincos is within an ulp
// of glibc's but not always equal, and laser_to_point (core.h:45-47) must give the reference's points bit for bit.
// AHEAD: the next pass's range and direction are asked for before this pass's barrier (the resident path reads the ranges
// from pinned host memory: a second pass of 57 beams otherwise waits for a second trip over the host link).
template <bool AHEAD = false>
__device__ inline int scan_to_points_wg(const float* __restrict__ ranges, const ScanP& sp,
                                        const double2* __restrict__ dirs, bool do_trans,
                                        double tc, double ts, double ttx, double tty, double2* out,
                                        int* s_cnt, double clip_hw = 0., double clip_hh = 0.) {
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id(), n_waves = blockDim.x >> 6;
  int base = 0;
  [[maybe_unused]] float r_ahead = 0.f;
  [[maybe_unused]] double2 d_ahead = make_double2(0., 0.);
  if constexpr (AHEAD) {
    if (tid < sp.n_beams) {
      r_ahead = ranges[tid];
      d_ahead = dirs[tid];
    }
  }
  for (int start = 0; start < sp.n_beams; start += blockDim.x) {
    const int i = start + tid;
    bool valid = false;
    double2 p = make_double2(0., 0.);
    [[maybe_unused]] const float r_now = r_ahead;
    [[maybe_unused]] const double2 d_now = d_ahead;
    if constexpr (AHEAD) {
      const int in = i + (int)blockDim.x;
      if (in < sp.n_beams) {
        r_ahead = ranges[in];
        d_ahead = dirs[in];
      }
    }
    if (i < sp.n_beams) {
      const float r = AHEAD ? r_now : ranges[i];
      // ndtframe.cpp:165
      valid = ((double)r > 0.) && (r < sp.rmax) && (r > sp.eps);
      if (valid) {
        const double2 d = AHEAD ? d_now : dirs[i];     // index_to_angle + the cosine and sine of laser_to_point, from the host
        p.x = (double)r * d.x;         // laser_to_point, core.h:45-47
        p.y = (double)r * d.y;
        if (do_trans) {  // transform_point by s_trans, ndtframe.cpp:175-176
          const double x = p.x * tc - p.y * ts + ttx;
          const double y = p.x * ts + p.y * tc + tty;
          p.x = x;
          p.y = y;
        }
        if (clip_hw > 0.)  // getCellIndex != -1 for the one-cell frame (strict bounds; fl(x + w/2) == w indexes past it)
          valid = fabs(p.x) < clip_hw && fabs(p.y) < clip_hh && (p.x + clip_hw) < 2. * clip_hw &&
                  (p.y + clip_hh) < 2. * clip_hh;
      }
    }
    const unsigned long long bal = __ballot(valid);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < n_waves; ++w) {
      const int cw = s_cnt[w];
      off += (w < wave) ? cw : 0;
      tot += cw;
    }
    if (valid) out[base + off + rank] = p;
    base += tot;
    __syncthreads();
  }
  return base;
}

// Largest range among the beams loadLaser keeps (ndtframe.cpp:165), 0 if none: every point of the scan lies within it
// of the sensor.  `slot`: an int of LDS scratch.  Uniform result; ends with a barrier.
__device__ inline float scan_max_range_wg(const float* __restrict__ ranges, const ScanP& sp, int* slot) {
  if (threadIdx.x == 0) *slot = 0;
  __syncthreads();
  float m = 0.f;
  for (int i = threadIdx.x; i < sp.n_beams; i += blockDim.x) {
    const float r = ranges[i];
    if (((double)r > 0.) && (r < sp.rmax) && (r > sp.eps)) m = fmaxf(m, r);
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, kWave));
  if (lane_id() == 0) atomicMax(slot, __float_as_int(m));  // (non-negative floats order as their bit patterns)
  __syncthreads();
  return __int_as_float(*slot);
}

// Extent of a scan's points under a heading, in 1/256 cells, rounded outwards: box[0..3] = x_lo, x_hi, y_lo, y_hi of
// r_i (cos a_i, sin a_i) turned by (c0, s0) and scaled by inv_cs, over the beams laser_to_point keeps (ndtframe.cpp:165; the
// list the score loop runs over is a subset: the new frame drops what leaves it).  Returns false for a scan without points.
__device__ inline bool scan_extent_wg(const float* __restrict__ ranges, const ScanP& sp, const double2* __restrict__ dirs,
                                      double c0, double s0, double inv_cs, int* box, int (&out)[4]) {
  if (threadIdx.x == 0) {
    box[0] = box[2] = 0x7fffffff;
    box[1] = box[3] = -0x7fffffff;
  }
  __syncthreads();
  int x_lo = 0x7fffffff, x_hi = -0x7fffffff, y_lo = 0x7fffffff, y_hi = -0x7fffffff;
  for (int i = threadIdx.x; i < sp.n_beams; i += blockDim.x) {
    const float r = ranges[i];
    if (((double)r > 0.) && (r < sp.rmax) && (r > sp.eps)) {
      const double2 d = dirs[i];
      const double px = (double)r * d.x, py = (double)r * d.y;
      const double qx = (px * c0 - py * s0) * (256. * inv_cs), qy = (px * s0 + py * c0) * (256. * inv_cs);
      if (fabs(qx) < 1e9 && fabs(qy) < 1e9) {
        x_lo = min(x_lo, (int)floor(qx) - 1);
        x_hi = max(x_hi, (int)ceil(qx) + 1);
        y_lo = min(y_lo, (int)floor(qy) - 1);
        y_hi = max(y_hi, (int)ceil(qy) + 1);
      } else {
        x_lo = y_lo = -0x7fffffff;  // (no box can hold it)
        x_hi = y_hi = 0x7fffffff;
      }
    }
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) {
    x_lo = min(x_lo, __shfl_xor(x_lo, d, kWave));
    x_hi = max(x_hi, __shfl_xor(x_hi, d, kWave));
    y_lo = min(y_lo, __shfl_xor(y_lo, d, kWave));
    y_hi = max(y_hi, __shfl_xor(y_hi, d, kWave));
  }
  if (lane_id() == 0 && x_hi >= x_lo) {
    atomicMin(&box[0], x_lo);
    atomicMax(&box[1], x_hi);
    atomicMin(&box[2], y_lo);
    atomicMax(&box[3], y_hi);
  }
  __syncthreads();
  for (int k = 0; k < 4; ++k) out[k] = box[k];
  __syncthreads();  // (the slots are free again)
  return out[1] >= out[0] && out[0] > -0x7fffffff;
}

// ---- K3b: points -> reference cell table (fresh frame) ------------------------------------------
//
// NDTFrame::addPoint binning (ndtframe.cpp:215-235) + NDTCell::build for a cell whose window is
// empty (ndtcell.cpp:36-68: global_sum == partial sum, global_covar_sum == covariance sum) +
// s_calc_covar_inverse (ndtcell.cpp:93-111).  Sums run in beam order inside every cell, as the
// reference's per-cell vectors do, so the statistics are reproducible bit for bit.
//
// LDS in: pts[n].  LDS out: image (header, bitmap words {bits,prefix} of BUILT cells, records).
// rows (global, optional): one row per created cell.
__device__ inline void prefix_words_wave0(uint2* bm, int n_words, uint32_t* total_out) {
  // exclusive prefix of popcounts over bitmap words; executed by wave 0 only
  const int lane = lane_id();
  const int per = (n_words + kWave - 1) / kWave;
  const int w0 = lane * per, w1 = min(n_words, w0 + per);
  uint32_t sum = 0;
  for (int w = w0; w < w1; ++w) sum += __popc(bm[w].x);
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += t;
  }
  uint32_t run = incl - sum;
  for (int w = w0; w < w1; ++w) {
    bm[w].y = run;
    run += __popc(bm[w].x);
  }
  const uint32_t tot = __shfl(incl, kWave - 1, kWave);
  if (lane == 0) *total_out = tot;
}

// frame cell of a point as NDTFrame::addPoint bins it (ndtframe.cpp:215-235); false = dropped
__device__ __forceinline__ bool point_cell(const GridP& g, double2 p, int& ix, int& iy) {
  if (!(fabs(p.x) < g.hw && fabs(p.y) < g.hh)) return false;
  cell_coords_rt(g, p.x, p.y, ix, iy);
  if (ix == g.W) {  // reference linear-index wrap (see score_point)
    ix = 0;
    iy += 1;
  }
  return iy < g.H;
}

// Staging window of one alignment = bounding box of the cells its reference points fall in (workgroup
// cooperative; box[4] = {min x, max x, min y, max y} in LDS).  Empty input gives the 1 x 1 window at (0, 0).
__device__ inline WinP dynamic_window_wg(const GridP& g, const double2* pts, int n, int* box, int rec_cap) {
  if (threadIdx.x == 0) {
    box[0] = box[2] = 0x7fffffff;
    box[1] = box[3] = -1;
  }
  __syncthreads();
  // per-thread, then per-wave extremes; one LDS atomic per wave and bound (four addresses shared by every point
  // serialised 4 x n atomics: 23 us of a 75 us setup)
  int x_lo = 0x7fffffff, x_hi = -1, y_lo = 0x7fffffff, y_hi = -1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int ix, iy;
    if (point_cell(g, pts[i], ix, iy)) {
      x_lo = min(x_lo, ix);
      x_hi = max(x_hi, ix);
      y_lo = min(y_lo, iy);
      y_hi = max(y_hi, iy);
    }
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) {
    x_lo = min(x_lo, __shfl_xor(x_lo, d, kWave));
    x_hi = max(x_hi, __shfl_xor(x_hi, d, kWave));
    y_lo = min(y_lo, __shfl_xor(y_lo, d, kWave));
    y_hi = max(y_hi, __shfl_xor(y_hi, d, kWave));
  }
  if (lane_id() == 0 && x_hi >= 0) {
    atomicMin(&box[0], x_lo);
    atomicMax(&box[1], x_hi);
    atomicMin(&box[2], y_lo);
    atomicMax(&box[3], y_hi);
  }
  __syncthreads();
  WinP w;
  const bool any = box[1] >= 0;
  w.x0 = any ? box[0] : 0;
  w.y0 = any ? box[2] : 0;
  w.w = any ? box[1] - box[0] + 1 : 1;
  w.h = any ? box[3] - box[2] + 1 : 1;
  w.n_words = (w.w * w.h + 31) / 32;
  w.rec_cap = rec_cap;
  return w;
}

__device__ __forceinline__ unsigned bm_slot(const uint2* bm, int k) {
  const uint2 e = bm[k >> 5];
  return e.y + __popc(e.x & ((1u << (k & 31)) - 1u));
}

// scratch: key[n], cellkey[n], cnt[n] ints (rounded up to 4), plist[n] u16 and bm2[n_words] uint2
// hdr/out: where the table goes (LDS); out.ab/out.cd or out.chol may be null when a kernel needs one score form
// dn/lds0 (optional): also emit the dense form (u16 table at lds0, the records at lds0 + dn->rec_off)
// part A of a cell's dense record from its Cholesky factor (cell units) and mean (window cell coordinates); l11 == 0
// (a form without extent along x: both products of `a` vanish) keeps r = 0.  alpha = NaN marks a form without a factor.
__host__ __device__ inline DenseRecA dense_rec_a(const double l[3], double mgx, double mgy, bool ok) {
  const double r = l[0] > 0. ? l[1] / l[0] : 0.;
  return DenseRecA{r, ok ? -fma(r, mgy, mgx) : (double)__builtin_nanf("")};
}
__device__ inline void dense_clear_wg(const DenseP& dn, unsigned char* lds0, bool byte_entries = false) {
  const unsigned null16 = byte_entries ? (unsigned)dn.rec_off : (unsigned)dn.rec_off >> 4;
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lds0);
  const int n32 = dense_tab_bytes(dn.dw, dn.dh) >> 2;
  for (int i = threadIdx.x; i < n32; i += blockDim.x) t32[i] = null16 | (null16 << 16);
  if (threadIdx.x == 0) {
    *reinterpret_cast<DenseRecA*>(lds0 + dn.rec_off) = DenseRecA{0., (double)__builtin_inff()};
    *reinterpret_cast<DenseRecB*>(lds0 + dn.rec_off + kDenseRecBOff) = DenseRecB{0., 1.f, 0.f};
  }
}
// one built cell -> dense record `slot + 1` and its table entry; (rx, ry) = cell inside the staging window
__device__ __forceinline__ void dense_put(const GridP& g, const DenseP& dn, unsigned char* lds0, unsigned slot, int rx,
                                          int ry, double mx, double my, double ia, double ib, double ic, double id,
                                          bool byte_entries = false) {
  double l[3];
  const bool ok = make_chol_d(ia, ib, ic, id, l, g.cs);  // Cholesky factor in cell units
  const unsigned at = (unsigned)dn.rec_off + dense_rec_pos(slot + 1);
  const double mgx = (mx + g.hw) * g.inv_cs - (double)dn.ox, mgy = (my + g.hh) * g.inv_cs - (double)dn.oy;
  const DenseRecA ra = dense_rec_a(l, mgx, mgy, ok);
  *reinterpret_cast<DenseRecA*>(lds0 + at) = ra;
  *reinterpret_cast<DenseRecB*>(lds0 + at + kDenseRecBOff) = DenseRecB{mgy, (float)l[0], (float)l[2]};
  reinterpret_cast<unsigned short*>(lds0)[(ry + 1) * dense_stride(dn.dw) + (rx + 1)] =
      byte_entries ? (unsigned short)at : (unsigned short)(at >> 4);
}

// ---- the fp64 score's dense form (score_trip_d64): u16 table at lds0 + tab_off, 48-byte records at lds0 + dn.rec_off ----
__device__ inline void d64_clear_wg(const DenseP& dn, unsigned char* lds0, int tab_off) {
  const unsigned null16 = (unsigned)dn.rec_off;  // entries are the records' LDS byte addresses
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lds0 + tab_off);
  const int n32 = dense_tab_bytes(dn.dw, dn.dh) >> 2;
  for (int i = threadIdx.x; i < n32; i += blockDim.x) t32[i] = null16 | (null16 << 16);
  if (threadIdx.x == 0) {
    double* r = reinterpret_cast<double*>(lds0 + dn.rec_off);
    r[0] = r[1] = kD64NullMean;
    r[2] = kD64NullScale;
    r[3] = 0.;
    r[4] = 0.;
    r[5] = kD64NullScale;
  }
}
// Can every exponent this cell produces be left to exp_neg_half?  |x| <= (|a| + |b| + |c| + |d|) max(d0^2, |d0 d1|, d1^2) / 2
// (1 + 1e-15), and a point scored against a cell lies in it, as does the mean of the points it was built from: both
// differences are below the cell side (twice that is assumed).  2^39 leaves a factor of two to the form's own bound of 2^40;
// a NaN or infinite entry fails the comparison.
__device__ __forceinline__ bool d64_cell_tame(double cs, double ia, double ib, double ic, double id) {
  const double S = (fabs(ia) + fabs(ib)) + (fabs(ic) + fabs(id));
  return S * (2. * cs * cs) < 549755813888.;
}
__device__ __forceinline__ void d64_put(const DenseP& dn, unsigned char* lds0, int tab_off, unsigned slot, int rx, int ry,
                                        double mx, double my, double ia, double ib, double ic, double id) {
  const unsigned at = (unsigned)dn.rec_off + (unsigned)kD64RecBytes * (slot + 1u);
  double* r = reinterpret_cast<double*>(lds0 + at);
  r[0] = mx;
  r[1] = my;
  r[2] = ia;
  r[3] = ib;
  r[4] = ic;
  r[5] = id;
  reinterpret_cast<unsigned short*>(lds0 + tab_off)[(ry + 1) * dense_stride(dn.dw) + (rx + 1)] = (unsigned short)at;
}

// d64_tab_off >= 0 (with dn): the dense form is the fp64 score's (d64_put) and a cell exp_neg_half must not score raises
// kHdrWildCell in the header (the kernel hands the alignment to the bitmap form)
__device__ inline void build_table_wg(const GridP& g, const WinP& wn, const double2* pts, int n,
                                      ImageHeader* hdr, const TableOut& out, int* key, int* cellkey, int* cnt,
                                      uint2* bm2, unsigned short* plist, CellRow* rows, uint32_t* n_rows_out,
                                      const DenseP* dn, unsigned char* lds0, bool byte_entries = false,
                                      const TableOut* xout = nullptr /* exact mode: bitmap, mean, ab, cd also to HBM */,
                                      int d64_tab_off = -1) {
  const int tid = threadIdx.x, nt = blockDim.x;
  uint2* bm = out.bm;
  if (dn) {
    if (d64_tab_off >= 0) d64_clear_wg(*dn, lds0, d64_tab_off); else dense_clear_wg(*dn, lds0, byte_entries);
  }

  for (int w = tid; w < wn.n_words; w += nt) {
    bm[w] = make_uint2(0u, 0u);
    bm2[w] = make_uint2(0u, 0u);
  }
  if (tid == 0) {
    hdr->n_built = 0;
    hdr->n_created = 0;
    hdr->status = 0;
  }
  __syncthreads();

#ifdef NDTPSO_PROFILE_SETUP
  unsigned long long bt[8];
  bt[0] = bt[6] = wall_clock64();
#define NDTPSO_BT(i) bt[i] = wall_clock64()
#else
#define NDTPSO_BT(i) do { } while (0)
#endif
  // 1. bin every point (NDTFrame::addPoint -> getCellIndex); mark created cells
  for (int i = tid; i < n; i += nt) {
    const double2 p = pts[i];
    int k = -1;
    if (fabs(p.x) < g.hw && fabs(p.y) < g.hh) {
      int ix, iy;
      cell_coords_rt(g, p.x, p.y, ix, iy);
      if (ix == g.W) {  // reference linear-index wrap (see score_point)
        ix = 0;
        iy += 1;
      }
      if (iy < g.H) {
        const unsigned rx = (unsigned)(ix - wn.x0), ry = (unsigned)(iy - wn.y0);
        if (rx < (unsigned)wn.w && ry < (unsigned)wn.h) {
          k = (int)(ry * (unsigned)wn.w + rx);
          atomicOr(&bm2[k >> 5].x, 1u << (k & 31));
        } else {
          atomicOr(&hdr->status, 1u);  // in frame but outside the staging window
        }
      }
    }
    key[i] = k;
  }
  if (tid < 4 && n + tid < ((n + 3) & ~3)) key[n + tid] = -1;  // pad the key array to a multiple of 4 entries
  __syncthreads();

  NDTPSO_BT(1);
  // 2. created cells -> dense slots in ascending cell order
  if (wave_id() == 0) prefix_words_wave0(bm2, wn.n_words, &hdr->n_created);
  __syncthreads();
  const int n_created = (int)hdr->n_created;
  for (int w = tid; w < wn.n_words; w += nt) {
    uint32_t bits = bm2[w].x;
    uint32_t slot = bm2[w].y;
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      cellkey[slot] = w * 32 + b;
      cnt[slot] = 0;
      ++slot;
    }
  }
  __syncthreads();

  NDTPSO_BT(2);
  // 3. points per cell (integer atomics: order independent); key[i] becomes the point's cell slot
  for (int i = tid; i < n; i += nt) {
    const int k = key[i];
    if (k >= 0) {
      const int slot = (int)bm_slot(bm2, k);
      atomicAdd(&cnt[slot], 1);
      key[i] = slot;
    }
  }
  __syncthreads();

  NDTPSO_BT(3);
  // 4. built cells (count > 2, ndtcell.cpp:43) -> final record slots; per-cell offsets into the point lists
  for (int s = tid; s < n_created; s += nt)
    if (cnt[s] > 2) {
      const int k = cellkey[s];
      atomicOr(&bm[k >> 5].x, 1u << (k & 31));
    }
  __syncthreads();
  if (wave_id() == 0) {
    prefix_words_wave0(bm, wn.n_words, &hdr->n_built);
    // exclusive prefix of the counts, kept in the upper half of cnt[] (count and offset are both <= n < 65536)
    const int lane = lane_id();
    const int per = (n_created + kWave - 1) / kWave;
    const int s0 = lane * per, s1 = min(n_created, s0 + per);
    int sum = 0;
    for (int q = s0; q < s1; ++q) sum += cnt[q];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int t = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += t;
    }
    int run = incl - sum;
    for (int q = s0; q < s1; ++q) {
      const int c = cnt[q];
      cnt[q] = c | (run << 16);
      run += c;
    }
  }
  __syncthreads();
  if (tid == 0 && (int)hdr->n_built > wn.rec_cap) atomicOr(&hdr->status, 2u);

  NDTPSO_BT(4);
  // 5a. Per-cell point lists in beam order (= the reference's insertion order): every point files itself at its rank
  //     among its cell's points, i.e. the number of earlier points with the same slot -- a branch-free scan of the slot
  //     keys (15 us for 1081 points, compare-bound).  Tried instead: each cell's owner scanning all keys and storing
  //     matches conditionally (35 us per owner: load -> branch -> store chains, the original); one wave walking the
  //     points 64 at a time with ballots and a running position per cell (32 us: a serial chain of LDS round trips).
  {
    const int4* k4 = reinterpret_cast<const int4*>(key);
    for (int i = tid; i < n; i += nt) {
      const int s = key[i];
      if (s < 0) continue;
      auto hits = [s](const int4 kk) { return (int)(kk.x == s) + (int)(kk.y == s) + (int)(kk.z == s) + (int)(kk.w == s); };
      const int full = i >> 2;
      int rank = 0, q = 0;
      for (; q + 4 <= full; q += 4) {  // four loads in flight
        const int4 a = k4[q], b = k4[q + 1], c4 = k4[q + 2], d = k4[q + 3];
        rank += (hits(a) + hits(b)) + (hits(c4) + hits(d));
      }
      for (; q < full; ++q) rank += hits(k4[q]);
      for (int j = full << 2; j < i; ++j) rank += (int)(key[j] == s);
      plist[((unsigned)cnt[s] >> 16) + rank] = (unsigned short)i;
    }
  }
  __syncthreads();
  NDTPSO_BT(6);
  // 5b. statistics: one owner thread per created cell visits its list: sums round exactly as the reference's
  //     per-cell vectors do (insertion order).
  for (int s = tid; s < n_created; s += nt) {
    const int mykey = cellkey[s];
    const int c = cnt[s] & 0xffff;
    const int off = (int)((unsigned)cnt[s] >> 16);
    const bool built = c > 2;
    double mx = 0., my = 0., ia = 0., ib = 0., ic = 0., id = 0.;
    if (built) {
      const unsigned short* mine = plist + off;
      // Both passes add in insertion order, one point after the other, as the reference does; only the loads are
      // batched four at a time (index, then point: two dependent LDS round trips that would otherwise be paid per
      // point -- the cell with the most points sets the duration of the whole step).
      double sx = 0., sy = 0.;
      int t = 0;
      for (; t + 4 <= c; t += 4) {
        const double2 p0 = pts[mine[t]], p1 = pts[mine[t + 1]], p2 = pts[mine[t + 2]], p3 = pts[mine[t + 3]];
        sx += p0.x;  // s_current_partial_sum += point, ndtcell.cpp:30
        sy += p0.y;
        sx += p1.x;
        sy += p1.y;
        sx += p2.x;
        sy += p2.y;
        sx += p3.x;
        sy += p3.y;
      }
      for (; t < c; ++t) {
        const double2 p = pts[mine[t]];
        sx += p.x;
        sy += p.y;
      }
      mx = sx / (double)c;  // ndtcell.cpp:44
      my = sy / (double)c;
      double c00 = 0., c01 = 0., c10 = 0., c11 = 0.;
      auto fold = [&](const double2 p) {  // ndtcell.cpp:49-52
        const double d0 = p.x - mx, d1 = p.y - my;
        c00 += d0 * d0;
        c01 += d0 * d1;
        c10 += d1 * d0;
        c11 += d1 * d1;
      };
      for (t = 0; t + 4 <= c; t += 4) {
        const double2 p0 = pts[mine[t]], p1 = pts[mine[t + 1]], p2 = pts[mine[t + 2]], p3 = pts[mine[t + 3]];
        fold(p0);
        fold(p1);
        fold(p2);
        fold(p3);
      }
      for (; t < c; ++t) fold(pts[mine[t]]);
      // s_calc_covar_inverse, ndtcell.cpp:93-111 (eigenvalues as Eigen's EigenSolver computes them)
      const double nn = (double)c;
      c00 = c00 / nn;
      c01 = c01 / nn;
      c10 = c10 / nn;
      c11 = c11 / nn;
      double inv[4];
      covar_inverse_eigen(c00, c01, c10, c11, inv);
      ia = inv[0];
      ib = inv[1];
      ic = inv[2];
      id = inv[3];
      const unsigned slot = bm_slot(bm, mykey);
      if ((int)slot < wn.rec_cap) {
        if (dn) {
          if (d64_tab_off >= 0) {
            d64_put(*dn, lds0, d64_tab_off, slot, mykey % wn.w, mykey / wn.w, mx, my, ia, ib, ic, id);
            if (!d64_cell_tame(g.cs, ia, ib, ic, id)) atomicOr(&hdr->status, kHdrWildCell);
          } else {
            dense_put(g, *dn, lds0, slot, mykey % wn.w, mykey / wn.w, mx, my, ia, ib, ic, id, byte_entries);
          }
        }
        if (xout) {
          xout->mean[slot] = make_double2(mx, my);
          xout->ab[slot] = make_double2(ia, ib);
          xout->cd[slot] = make_double2(ic, id);
        }
        if (out.mean) out.mean[slot] = make_double2(mx, my);
        if (out.ab) {
          out.ab[slot] = make_double2(ia, ib);
          out.cd[slot] = make_double2(ic, id);
        }
        if (out.chol) {
          float l[4];
          make_chol(ia, ib, ic, id, l);
          out.chol[slot] = make_float4(l[0], l[1], l[2], l[3]);
        }
      }
    }
    if (rows) {
      const int rx = mykey % wn.w, ry = mykey / wn.w;
      CellRow r;
      r.index = (wn.x0 + rx) + g.W * (wn.y0 + ry);
      r.count = c;
      r.built = built ? 1 : 0;
      r.reserved = 0;
      r.mean[0] = mx;
      r.mean[1] = my;
      r.icov[0] = ia;
      r.icov[1] = ib;
      r.icov[2] = ic;
      r.icov[3] = id;
      rows[s] = r;
    }
  }
  if (n_rows_out && tid == 0) *n_rows_out = (uint32_t)n_created;
  __syncthreads();
#ifdef NDTPSO_PROFILE_SETUP
  NDTPSO_BT(5);
  if (tid == 0 && blockIdx.x == 0)
    printf("table (us): bin %.1f slots %.1f count %.1f offsets %.1f lists %.1f stats %.1f  (n_created %d)\n",
           (bt[1] - bt[0]) * 0.01, (bt[2] - bt[1]) * 0.01, (bt[3] - bt[2]) * 0.01, (bt[4] - bt[3]) * 0.01,
           (bt[6] - bt[4]) * 0.01, (bt[5] - bt[6]) * 0.01, n_created);
#endif
}

// dense form from a table image in HBM (kernels that stage a prebuilt table): one thread per bitmap word
__device__ inline void dense_from_image_wg(const GridP& g, const WinP& wn, const unsigned char* __restrict__ image,
                                           const DenseP& dn, unsigned char* lds0) {
  dense_clear_wg(dn, lds0);
  __syncthreads();
  const uint2* bm = reinterpret_cast<const uint2*>(image + kImageHeaderBytes);
  const double2* mean = reinterpret_cast<const double2*>(image + image_mean_offset(wn.n_words));
  const double2* ab = reinterpret_cast<const double2*>(image + image_ab_offset(wn.n_words, wn.rec_cap));
  const double2* cd = reinterpret_cast<const double2*>(image + image_cd_offset(wn.n_words, wn.rec_cap));
  for (int w = threadIdx.x; w < wn.n_words; w += blockDim.x) {
    uint2 e = bm[w];
    unsigned slot = e.y;
    while (e.x) {
      const int b = __ffs(e.x) - 1;
      e.x &= e.x - 1;
      const int k = w * 32 + b;
      if ((int)slot < wn.rec_cap) {
        const double2 m = mean[slot], r0 = ab[slot], r1 = cd[slot];
        dense_put(g, dn, lds0, slot, k % wn.w, k / wn.w, m.x, m.y, r0.x, r0.y, r1.x, r1.y);
      }
      ++slot;
    }
  }
}

// ---- glibc rand() replay on the device ------------------------------------------------------
//
// srand(seed); rand() ... as glibc's TYPE_3 generator produces it: r[i] = r[i-31] + r[i-3]
// (mod 2^32), output r[i] >> 1, first output r[344].  Unrolling the recurrence ten times gives
// r[i] = r[i-30] + sum_{m<10} r[i-31-3m], which depends only on values >= 30 back: 30 lanes
// produce 30 consecutive values per step from a 64-entry LDS history ring.
// The ring is touched by one wave only (wave 0).  LDS operations of a wave execute in program order, so a value a
// lane stores is what another lane of the same wave loads afterwards; the wave barriers between steps keep the compiler
// from moving accesses across them.  (The ring used to be `volatile`, which serialised the eleven loads of a step:
// 8 us per PSO iteration with only this wave running -- a fifth of a workgroup's time; now about 1 us.)
// History of glibc's additive generator r[i] = r[i-31] + r[i-3] (mod 2^32), as a ring of 64 words kept TWICE (word i at
// [i & 63] and at [(i & 63) + 64]): every word a step reads then lies at a fixed distance below (i & 63) + 64, so the
// step's eleven loads are one address and eleven immediate offsets (with a single ring each of them cost an add, a mask
// and a shift of its own: 33 of a step's 55 vector instructions, and the generator was a ninth of everything a
// 70-particle alignment issues outside its score loop).
struct RngState {
  uint32_t hist[128];
};

__device__ inline void rng_seed_wave0(RngState* st, uint32_t seed) {
  if (lane_id() == 0) {
    int32_t word = (int32_t)(seed ? seed : 1u);
    st->hist[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      const int32_t hi = word / 127773, lo = word % 127773;
      int32_t t = 16807 * lo - 2836 * hi;
      if (t < 0) t += 2147483647;
      word = t;
      st->hist[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; ++i) st->hist[i] = st->hist[i - 31];
    for (int i = 34; i < 64; ++i) st->hist[i] = st->hist[i - 31] + st->hist[i - 3];
    for (int i = 0; i < 64; ++i) st->hist[i + 64] = st->hist[i];
  }
}

// produce r[t .. t+cnt), cnt <= 30; returns this lane's value (lane < cnt)
__device__ inline uint32_t rng_step_wave0(RngState* st, int t, int cnt) {
  const int lane = lane_id();
  uint32_t v = 0;
  __builtin_amdgcn_wave_barrier();
  const uint32_t* h = st->hist + ((t + lane) & 63) + 64;  // word i of the upper copy: i - 61 .. i - 30 lie below it in one piece
  if (lane < cnt) {
    uint32_t w[11];
    w[0] = h[-30];
#pragma unroll
    for (int m = 0; m < 10; ++m) w[m + 1] = h[-31 - 3 * m];  // eleven independent loads in flight
    v = w[0];
#pragma unroll
    for (int m = 0; m < 10; ++m) v += w[m + 1];  // (mod 2^32: any order gives the same sum)
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < cnt) {
    const_cast<uint32_t*>(h)[0] = v;
    const_cast<uint32_t*>(h)[-64] = v;
  }
  __builtin_amdgcn_wave_barrier();
  return v;
}

// generate `count` rand() outputs into dst[0..count) (wave 0 only); *t_io is the generator position
__device__ inline void rng_fill_wave0(RngState* st, int* t_io, int32_t* dst, int count) {
  int t = *t_io;
  const int lane = lane_id();
  // discard up to r[344) (glibc throws away the first 310 outputs)
  while (t < 344) {
    const int cnt = min(30, 344 - t);
    (void)rng_step_wave0(st, t, cnt);
    t += cnt;
  }
  int done = 0;
  while (done < count) {
    const int cnt = min(30, count - done);
    const uint32_t v = rng_step_wave0(st, t, cnt);
    if (lane < cnt) dst[done + lane] = (int32_t)(v >> 1);
    t += cnt;
    done += cnt;
  }
  *t_io = t;
}

// Eigen DenseBase::Random() coefficient for double (Eigen/src/Core/MathFunctions.h,
// random_default_impl<double>): x + (y-x)*double(rand())/double(RAND_MAX), x=-1, y=1
//
// The quotient is formed as q0 = a * y, r = fma(-q0, D, a), q = fma(r, y, q0) with y = RN(1 / D), D = RAND_MAX: by
// Markstein's theorem that is the correctly rounded a / D -- what the reference's division produces -- and
// tests/test_oracle.py verifies it exhaustively for all 2^31 values of rand().  Three flops instead of the ~35 of a
// generic fp64 division, in the proposal step where one or two waves work and the others wait.
__device__ __forceinline__ double uniform_pm1(int32_t raw) {
  constexpr double D = 2147483647.0, y = 1.0 / 2147483647.0;
  const double a = 2.0 * (double)raw;
  const double q0 = a * y;
  const double r = fma(-q0, D, a);
  return -1.0 + fma(r, y, q0);
}

// ---- K2: the PSO (pso_optimization, core.cpp:50-116), one workgroup per alignment ----------------
//
// Exact-order replay of the reference's single-thread semantics: the global best is updated
// inside the particle loop (core.cpp:94-105), so particle j+1 of the same iteration sees particle
// j's improvement.  All particles of an iteration are evaluated in parallel against the current
// gbest; the first particle (in index order) that improves gbest is found, particles up to and
// including it are committed, and the rest are re-proposed from their pre-iteration state with the
// new gbest and the same random draws (draws are indexed by (iteration, particle, k), so the replay
// is deterministic).  gbest moves ~10-20 times per 70x70 run, i.e. ~12 % extra evaluations.
struct Swarm {  // SoA, stride = P+1 (slot P is the "initial guess" particle of core.cpp:58)
  double* pos;    // [3][S] committed position
  double* vel;    // [3][S]
  double* pb;     // [3][S] best_position
  double* pbc;    // [S]    best_cost
  double* tpos;   // [3][S] proposed position
  double* tvel;   // [3][S]
  double* it;     // [S][4] per proposal {cos, sin, -, -} of its heading -- dense form: the folded transform constants
                  //        {C, S, TX, TY} (DenseItem), computed once where the proposal is made instead of by every wave
                  //        that evaluates it; one record, so that an evaluation fetches it with two 16-byte reads off one address
  double* tcost;  // [S]
  unsigned char* tgd;  // [S][2] ([S][4] in the BOX kernels: + heading, + a 1) fused pairs kernels: TX / TY of the proposal lie inside the DenseGuard (set with them; the
                       //        evaluating wave reads the pair as one 16-bit word instead of comparing four doubles)
  double* pcs;    // [2][S] plain cos, sin of the proposal's heading  } exact mode only: what the fp64 score of a
  double* bcs;    // [2][S] the same for the pbest position            } position takes (exact_tasks), so that the
  unsigned char* pex;  // [S] exact mode: pbc[j] holds the fp64 score of the pbest position (an arbitration put it there)
  int32_t* raw;   // [max(3(P+1), 6P)] rand() outputs of the current phase   arbitration needs no sincos of its own
  int32_t* raw2;  // [6P] the next iteration's outputs, generated by an idle wave behind the current one's last round
};
// exact: the arbitrating kernels keep the headings' cosines and sines (pcs, bcs); raw2: the second draw buffer of the
// overlapped generator -- always with the swarm in its HBM workspace, in LDS only for swarms of up to 256 particles (a
// 512-particle swarm with it no longer fits beside the dense table: 16 266 instead of 26 202 align/s, measured)
__host__ __device__ inline bool swarm_has_raw2(int P, bool swarm_global) { return swarm_global || P <= 256; }
__host__ __device__ inline int swarm_doubles(int P, bool exact) { return (exact ? 27 : 22) * (P + 1); }
__host__ __device__ inline int swarm_raw_ints(int P) { return (6 * P > 3 * (P + 1)) ? 6 * P : 3 * (P + 1); }
__host__ __device__ inline int swarm_bytes(int P, bool exact, bool raw2) {
  return align16(swarm_doubles(P, exact) * 8) + align16(swarm_raw_ints(P) * 4) + (raw2 ? align16(6 * P * 4) : 0);
}

__device__ inline Swarm swarm_carve(unsigned char* base, int P, bool exact, bool raw2) {
  const int S = P + 1;
  double* d = reinterpret_cast<double*>(base);
  Swarm sw;
  sw.pos = d;
  sw.vel = d + 3 * S;
  sw.pb = d + 6 * S;
  sw.pbc = d + 9 * S;
  sw.tpos = d + 10 * S;
  sw.tvel = d + 13 * S;
  sw.it = d + 16 * S;
  sw.tcost = d + 20 * S;
  sw.tgd = reinterpret_cast<unsigned char*>(d + 21 * S);  // (4 S bytes of an S-double slot)
  sw.pcs = exact ? d + 22 * S : nullptr;
  sw.bcs = exact ? d + 24 * S : nullptr;
  sw.pex = exact ? reinterpret_cast<unsigned char*>(d + 26 * S) : nullptr;  // (S bytes of an S-double slot)
  sw.raw = reinterpret_cast<int32_t*>(base + align16(swarm_doubles(P, exact) * 8));
  sw.raw2 = raw2 ? reinterpret_cast<int32_t*>(base + align16(swarm_doubles(P, exact) * 8) + align16(swarm_raw_ints(P) * 4)) : nullptr;
  return sw;
}

// Parameter block of exact_tasks (exact mode), kept in LDS: filled once by setup_exact_wg, so that the fp32-score kernels
// carry none of it in registers (the block used to travel in EvalCtx: 17 more SGPRs live through the whole PSO, the
// spill lanes they needed took VGPRs from the score loop).
struct ExactArgs {
  GridP g;
  int dw, dh, ox, oy;          // the dense cell table (DenseP): the cell lookup of the fp64 score goes through it too
  unsigned null_entry;         // its null entry = record 0's (the record index of an entry: dense_rec_index)
  const double2* xmean;        // fp64 records by slot, in the table image in HBM: NDTCell::mean,
  const double2* xab;          //   s_inv_covar row 0,
  const double2* xcd;          //   s_inv_covar row 1
  unsigned pts_lds;    // LDS byte address of the (padded) point list
  int n;
  const double* tpos;  // [3][S]
  const double* pb;    // [3][S]
  double* tcost;       // [S]
  double* pbc;         // [S]
  const double* gb;    // [3]
  const double* pcs;   // [2][S] cos, sin of the proposals' headings (as the proposal step took them)
  const double* bcs;   // [2][S] of the pbest positions'
  const double* gcs;   // [2]    of the gbest position's
  double* xgbc;
  int S;
  // the arbitration's work list and scratch (exact_tasks_wg)
  unsigned char* pex;      // Swarm::pex
  unsigned xs_lds;         // LDS byte address of the partial-sum scratch: xs_slots x 64 doubles
  int xs_slots;            // units (one lane accumulator of one fp64 score each) that can be in flight
  int gex;                 // the gbest cost in PsoShared::gbc is the fp64 score of the gbest position
  int n_task;              // tasks of the current arbitration
  int gb_task;             // ... one of which rescored the gbest position
  unsigned short task[2 * 16 + 2];  // (item << 2) | kind: 0 proposal -> tcost, 1 pbest -> pbc, 2 gbest -> *xgbc   [kMaxNear = 16]
};
// Dense form, fused pairs kernel: the folded translation (DenseItem::TX, TY) of a pose whose transform keeps EVERY point
// of the list inside the cell table lies in [x_lo, x_hi) x [y_lo, y_hi) -- the points lie within `rho` of the sensor,
// so whatever the heading their table coordinates lie within rho / cell_side of (TX, TY).  The kernel sizes the table
// for that (k_align_pairs); a pose outside the box (or a table that could not be made that large: empty box) takes the
// clamped loop.  In LDS: the evaluating waves read it next to the item's constants.
struct __attribute__((aligned(16))) DenseGuard {
  double x_lo, x_hi, y_lo, y_hi;
};
constexpr int kMaxNear = 16;  // undecidable comparisons one evaluation group may contain before the alignment is handed over
struct PsoShared {  // small control block in LDS
  double gb[3];
  double gbc;
  int jstar[3];  // first improver of a group, rotating by group number (see pso_run_wg)
  int tiny;      // fp32 score mode: some cost of the current group fell in the underflow regime
  int tiny_j;    // NDTPSO_STREAM: the lowest item of the phase whose cost did (INT_MAX: none) -- only an item the phase
                 // establishes, i.e. one up to its first improver, may hand the alignment over: what lies behind is evaluated
                 // or not depending on timing, and is proposed again anyway
  int timed_out; // cluster mode: a workgroup of the cluster did not arrive at an exchange
  int ticket;    // NDTPSO_STREAM: the next item of the phase (eval_stream)
  int jmax;      //                the highest item evaluated in it
  int spare[32]; //                where the lanes that take no ticket add their zeros (take_ticket)
  RngState rng;
  // arbitration (exact mode): items of a group whose fp32 cost is too close to their pbest's or the gbest's to decide
  // the comparison; rotating by group number like jstar
  int near_cnt[3];
  unsigned short near_list[3][kMaxNear];
  double xgbc;  // fp64 score of the gbest position (arbitration scratch)
  double gcs[2];  // cos, sin of the gbest position's heading
  DenseGuard guard;  // (fused pairs kernel, dense form)
  // What the proposal step needs besides the swarm: kept HERE, not in registers.  As kernel arguments and loop-carried
  // values they were seven register pairs live through the whole PSO; the arbitrating kernel had no room for them beside
  // its calls, spilled them, and every proposal step -- where one wave works and seven wait -- began with a round trip to
  // scratch memory (proposals 145 us per alignment against 105 us in the plain fp32 kernel, profiles/r04_phase_budget*).
  double k_w, k_c1, k_c2;             // inertia weight of the current iteration (core.cpp:108), c1, c2
  double k_hw, k_hh, k_inv, k_ox, k_oy;  // dense form: the fold of a position into a DenseItem (dense_item)
  ExactArgs xa;
  // BOX kernels (k_align_pairs): the headings `guard` holds for -- any (the box of scan B's disc) or a window around the guess's
  // (the box of scan B's own extent under that heading, grown by what the window can turn it: scan_extent_wg, box_guard_wg).
  // Behind everything else: the other kernels' offsets stay what they were.
  double g_t_lo, g_t_hi;
};

// PATH: 0 = bitmap table, true division by cell_side; 1 = bitmap table, power-of-two cell side; 2 = dense fast path;
// 4 / 5 = as 0 / 1 with the table read from its HBM image instead of LDS (maps too large to stage)
struct EvalCtx {
  GridP g;
  WinP wn;
  TableView T;
  DenseP dn;
  const unsigned char* lds0;
  int light;  // PsoP::light (the item -> wave deal of eval_items)
  unsigned guard_lds;  // LDS byte address of the DenseGuard, 0: none
  unsigned d64_tab;    // PATH 8 / 9: LDS byte address of the u16 cell table (records at dn.rec_off)
#ifdef NDTPSO_VERIFY_MARGIN
  const struct ExactArgs* xa = nullptr;  // diagnostic builds: every fp32 score is checked against its fp64 value
#endif
};

// ---- fp32 score mode, underflow regime ----------------------------------------------------------------
//
// v_exp_f32 flushes results below 2^-126 to zero.  That is irrelevant while the score is O(1..N), but when
// (almost) nothing overlaps -- a reference with a single tiny cell, a guess far off -- the whole sum can be
// 1e-50 or 1e-200 in the reference's fp64 arithmetic, and its PSO still orders particles by those values.
// The PSO kernels therefore stop as soon as an fp32 cost lands above -kTinyCost and flag the alignment
// (kStatusNeedsF64); the host side re-runs flagged alignments with the fp64-score kernel, gated on that flag.
// ndtpso_cost_batch re-evaluates such a pose in place (eval_pose_wave_tiny: same records, fp64 exponential).
// PATH 2 and 3 are the dense form (3: table entries are byte addresses, see score_trip_dense)
__host__ __device__ constexpr bool path_is_dense(int path) { return path == 2 || path == 3; }
constexpr uint32_t kStatusNeedsF64 = 4u;
#ifdef NDTPSO_WHY_BITS  // diagnostic builds (scripts/shape_diag.py): why an alignment was handed to the fp64-score kernel -- 0x100 / 0x200
#define NDTPSO_WHY(bit) (bit)  // underflow regime / more than kMaxNear near-ties at the swarm's initialisation, 0x400 / 0x800 in an iteration
#else
#define NDTPSO_WHY(bit) 0u
#endif
constexpr uint32_t kStatusNeedsBitmap = 8u;  // dense form: the occupied box exceeds the provisioned cell table
constexpr double kTinyCost = 1e-28;

template <int PATH>
__device__ __forceinline__ double eval_pose_wave_tiny(const EvalCtx& E, const double2* __restrict__ pts, int n, double c,
                                                   double s, double tx, double ty) {
  const int lane = lane_id();
  const int n_pad = round_up(n, kWave);
  double acc = 0.;
  if constexpr (path_is_dense(PATH)) {
    const DenseItem it = dense_item(E.g, E.dn, c, s, tx, ty);
    for (int base = 0; base < n_pad; base += kWave) {
      const double2 p = pts[base + lane];
      const double gx = fma(p.x, it.C, fma(-p.y, it.S, it.TX));
      const double gy = fma(p.x, it.S, fma(p.y, it.C, it.TY));
      const unsigned rx = (unsigned)(int)gx, ry = (unsigned)(int)gy;
      const bool ok = (rx < (unsigned)E.dn.dw) && (ry < (unsigned)E.dn.dh) && (!E.dn.clip || (gx < it.XMAX && gy < it.YMAX));
      const unsigned lin = ok ? ry * (unsigned)dense_stride(E.dn.dw) + rx : 0u;
      const unsigned e = reinterpret_cast<const unsigned short*>(E.lds0)[lin];
      const unsigned at = PATH == 3 ? e : e << 4;
      const DenseRecA* ra = reinterpret_cast<const DenseRecA*>(E.lds0 + at);
      const DenseRecB* rb = reinterpret_cast<const DenseRecB*>(E.lds0 + at + kDenseRecBOff);
      const double a = (double)rb->l11 * (fma(ra->r, gy, gx) + ra->alpha), b = (double)rb->l22 * (gy - rb->mgy);
      acc += exp2(-(a * a + b * b));  // null record: alpha = +inf -> 0
    }
  } else {
    const GridP& g = E.g;
    const WinP& wn = E.wn;
    for (int base = 0; base < n_pad; base += kWave) {
      const double2 p = pts[base + lane];
      const double qx = fma(p.x, c, fma(-p.y, s, tx));
      const double qy = fma(p.x, s, fma(p.y, c, ty));
      double term = 0.;
      if (fabs(qx) < g.hw && fabs(qy) < g.hh) {
        int ix, iy;
        cell_coords<(PATH & 3) == 1>(g, qx, qy, ix, iy);
        if (ix == g.W) {
          ix = 0;
          iy += 1;
        }
        const unsigned rx = (unsigned)(ix - wn.x0), ry = (unsigned)(iy - wn.y0);
        if (rx < (unsigned)wn.w && ry < (unsigned)wn.h) {
          const unsigned lin = ry * (unsigned)wn.w + rx;
          const uint2 e = E.T.bm[lin >> 5];
          const unsigned bit = lin & 31u;
          if ((e.x >> bit) & 1u) {
            const unsigned slot = e.y + __popc(e.x & ((1u << bit) - 1u));
            const double2 m = E.T.mean[slot];
            const float4 f = E.T.chol[slot];
            const double d0 = qx - m.x, d1 = qy - m.y;
            const double a = (double)f.x * d0 + (double)f.y * d1, b = (double)f.z * d1;
            term = exp2(-(a * a + b * b + (double)f.w));
          }
        }
      }
      acc += term;
    }
  }
  return -wave_sum(acc);
}


// ---- exact mode: arbitration of the comparisons the fp32 score cannot decide ---------------------------------
//
// The PSO consumes a cost only through `cost < best_cost` / `cost < global_best.best_cost` (core.cpp:63,94,97); the
// value itself survives only as a later comparison's right-hand side and as the returned gbest cost.  The fp32 score
// is within ~1e-8 relative of the fp64 one (worst case seen 6e-6 absolute on costs of 300..900), so a comparison
// whose two sides differ by more than the margin arb_margin() -- kArbRel * |gbest cost| (2e-3 absolute there, 300 times
// that), and never less than kArbAbsPerPoint per point of the scan -- comes out the same in either arithmetic.  The few that are closer ("near": ~0.5 per 70 x 70 alignment) are ARBITRATED: the
// proposal, the particle's pbest position and the gbest position are scored in fp64 exactly as the fp64-score kernels
// score them (same operation order, same summation order, the bitmap-form table read from its HBM image), the three
// stored costs are replaced by those values and the comparison is redone.  Every decision is then the fp64 mode's, so
// the returned pose is the fp64 mode's bit for bit; the returned cost is the fp64 score of that pose, also its.
// What this rests on: the fp32 score's error staying below half the margin -- bounded for any input (below) and checked
// evaluation by evaluation in the -DNDTPSO_VERIFY_MARGIN build; and the dense form's binning (gx within 1e-14 cells of
// the reference's value, see score_trip_dense).
#ifndef NDTPSO_ARB_REL
#define NDTPSO_ARB_REL 5e-6
#endif
constexpr double kArbRel = NDTPSO_ARB_REL;
// ... and never less than kArbAbsPerPoint x (points of the scan): the fp32 form's error is below 2.95e-7 per point
// whatever the cells look like (verify_pose_wave / DESIGN 3.6: u = 2^-24, at most 4.87 u per term, times 1.01), so half
// of 7e-7 n covers it for ANY input -- in particular when the gbest cost is small against the number of points (poor
// overlap), where the relative margin alone would not.
#ifndef NDTPSO_ARB_ABS
#define NDTPSO_ARB_ABS 7e-7
#endif
constexpr double kArbAbsPerPoint = NDTPSO_ARB_ABS;
__device__ __forceinline__ double arb_margin(double gbest_cost, int n_points) {
  return fmax(kArbRel * fabs(gbest_cost), kArbAbsPerPoint * (double)n_points);
}

__device__ __forceinline__ void near_note(int* near_cnt, unsigned short* near_list, int j) {
#ifdef NDTPSO_X_NODETECT
  return;
#endif
  const int k = atomicAdd(near_cnt, 1);
  if (k < kMaxNear) near_list[k] = (unsigned short)j;
}
__device__ __forceinline__ bool near_tie(double a, double b, double tau) { return fabs(a - b) <= tau; }

// The fp64 scores of an arbitration, one task per wave.  Deliberately NOT inlined and not a template: one copy in the
// library, called from the cold blocks of the fp32-score kernels with its parameter block in LDS -- the fp64 score
// loop inlined into them cost the hot loop its registers (private segment 0 -> 208 bytes, -17 % throughput).
// kind 0: swarm initialisation -- the proposals of the listed items; 1: an evaluation group -- proposal and pbest
// position of every listed item plus the gbest position (into *xgbc); 2: the gbest position only (returned cost).
// The scores replace tcost[j] / pbc[j].  Each is what the fp64-score kernels compute for that pose (sincos of the
// heading as the proposal step takes it, eval_pose_wave<kScoreF64>, table read from its HBM image).
// Called by every thread of the workgroup; ends with a barrier.
// The fp64 score of one pose as the arbitration computes it.  Per point exactly the operations of
// score_trip<kScoreF64> (reference order: transform_point core.h:28-31 without FMA, strict frame bounds and
// floor((x + w/2) / cs) of getCellIndex ndtframe.cpp:240-249 incl. the index wrap, the quadratic form and exp of
// normalDistribution ndtcell.cpp:70-78), the same four lane accumulators over trips of four chunks, the chunks behind
// the last full trip into the first, the same fold: the fp64 kernels' sum bit for bit.
// The cell lookup goes through the dense u16 table in LDS (same cell coordinates, hence the same membership as the
// bitmap of the fp64 kernels; one dependent HBM round trip less), the records come from the table image in HBM.
template <bool BYTE, bool POW2>
__device__ __forceinline__ double eval_pose_wave_exact(const ExactArgs* ap, double c, double s, double tx, double ty) {
  const GridP g = ap->g;
  const int dw = ap->dw, dh = ap->dh, ox = ap->ox, oy = ap->oy;
  const unsigned null_entry = ap->null_entry;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  const unsigned pts_lds = ap->pts_lds;
  const double2* __restrict__ xmean = ap->xmean;
  const double2* __restrict__ xab = ap->xab;
  const double2* __restrict__ xcd = ap->xcd;
  const int chunks = round_up(ap->n, kWave) / kWave;
  const int lane = lane_id();
  auto term_of = [&](int k) -> double {
    const v2d_t p = *(lds_d2_t)(uintptr_t)(pts_lds + (unsigned)(k * kWave + lane) * 16u);
    const double qx = (p.x * c - p.y * s) + tx;  // reference rounding, no fma
    const double qy = (p.x * s + p.y * c) + ty;
    const bool inframe = (int)(fabs(qx) < g.hw) & (int)(fabs(qy) < g.hh);
    int ix, iy;
    cell_coords<POW2>(g, qx, qy, ix, iy);
    const bool wrap = (ix == g.W);
    ix = wrap ? 0 : ix;
    iy = wrap ? iy + 1 : iy;
    const unsigned rx = (unsigned)(ix - ox), ry = (unsigned)(iy - oy);
    const bool inwin = (int)inframe & (int)(rx <= (unsigned)dw) & (int)(ry <= (unsigned)dh);
    const unsigned lin = inwin ? ry * (unsigned)dense_stride(dw) + rx : 0u;  // entry 0: the empty low border, null
    const unsigned e = *(lds_u16_t)(uintptr_t)(lin << 1);
    const bool hit = e != null_entry;
    const unsigned slot = hit ? dense_rec_index(BYTE ? e - null_entry : (e - null_entry) << 4) - 1u : 0u;
    // a miss adds exp(-inf) = +0. in the fp64 kernels, which leaves the accumulator as it was; slot 0 exists whenever
    // any cell is built, and a table without built cells has no hits
    const double2 m = xmean[slot], ab = xab[slot], cd = xcd[slot];
    const double d0 = qx - m.x, d1 = qy - m.y;
    const double r0 = d0 * ab.x + d1 * cd.x;  // (diff^T * inv_covar), ndtcell.cpp:73-75
    const double r1 = d0 * ab.y + d1 * cd.y;
    const double x = -(r0 * d0 + r1 * d1) / 2.;
    return exp(hit ? x : -(double)__builtin_inff());
  };
  double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
  int k = 0;
#ifndef NDTPSO_EXACT_ROLLED
#pragma unroll 1
  for (; k + 4 <= chunks; k += 4) {  // four independent gathers in flight
    const double t0 = term_of(k), t1 = term_of(k + 1), t2 = term_of(k + 2), t3 = term_of(k + 3);
    a0 += t0;
    a1 += t1;
    a2 += t2;
    a3 += t3;
  }
#pragma unroll 1
  for (; k < chunks; ++k) a0 += term_of(k);
#else  // one chunk per trip: fewer registers for the callee to save, more latency per chunk
  const int in_trips = chunks & ~3;
#pragma unroll 1
  for (; k < chunks; ++k) {
    const double t = term_of(k);
    const int u = k < in_trips ? (k & 3) : 0;
    a0 += u == 0 ? t : 0.;  // (adding +0. leaves an accumulator as it is: none is ever -0.)
    a1 += u == 1 ? t : 0.;
    a2 += u == 2 ? t : 0.;
    a3 += u == 3 ? t : 0.;
  }
#endif
  return -wave_sum((a0 + a1) + (a2 + a3));
}

// ---- NDTPSO_VERIFY_MARGIN (diagnostic builds only, tests/test_gpu_margin.py) ------------------------------------
// The exact mode rests on one inequality: every fp32 score the PSO stores differs from the fp64 score of the same pose by
// less than half the arbitration margin, tau / 2 = kArbRel * |gbest cost| / 2 -- then a comparison whose sides are more
// than tau apart falls the same way in either arithmetic, and the closer ones are arbitrated.  This build checks it for
// EVERY evaluation of an alignment, two ways:
//   measured   err = |fp32 score - fp64 score| (the fp64 score is the arbitration's own, eval_pose_wave_exact);
//   derived    B >= err, an a-priori bound of the fp32 form's rounding error evaluated for this pose.  Per point that
//              hits a built cell, with u = 2^-24, s = d0 + r d1 formed in fp64 as fma(r, gy, gx) + alpha (error e64 <=
//              3.4e-16 (|gx| + |r gy| + |alpha|)), a = l11 s, b = l22 d1, q = a^2 + b^2, t = 2^-q:
//                s, d1 are rounded to fp32 once, l11 and l22 are fp32 roundings, the products round
//                                                                     -> |da| <= 3u |a| + |l11| e64,  |db| <= 3u |b|
//                q = fma(a, a, b * b): the product and the fma round   -> |dq| <= 6u a^2 + 7u b^2 + u q + ... <= 8u q + 2 |a l11| e64
//                t = v_exp_f32(-q), 1 ulp                             -> |dt| <= t (ln2 |dq| + 2u)
//                groups of four terms are added in fp32               -> <= 2u t each
//              B = 1.01 * sum_points t (u (5.55 q + 4) + 1.4 |a l11| e64) + n 2^-126       (1.01: second-order terms; the
//              last term: v_exp_f32 flushes results below 2^-126 to zero)
//              t (5.55 q + 4) <= 4.87 for every q >= 0, so B <= 2.95e-7 x (points that hit a cell) WHATEVER the cells look
//              like: that is what kArbAbsPerPoint rests on.  (The form of rounds 1-2 -- differences and factors rounded
//              to fp32, a formed in fp32 -- had |da| <= 4u (|L11 d0| + |L21 d1|), up to a hundred times |a| in a thin
//              rotated cell: its bound exceeded the margin eleven-fold on BASELINE config 3, measured with this build.)
// and counts the points the folded binning of the fp32 loop (gx = fma(x, C, fma(-y, S, TX))) would file under another
// table entry than the reference's floor((x + w/2) / cs).  Per alignment (g_verify[blockIdx.x][8], doubles):
//   0 max err   1 max err / B   2 max B / (tau / 2)   3 max err / (tau / 2)   4 evaluations checked
//   5 points binned differently (entries differ)   6 max B   7 points checked
//   15 points whose folded gx or gy lies within 1e-11 of an integer   16 ... within 1e-9   17 points whose folded CELL differs from
//   the reference's (in frame and window; whatever the two cells hold)   18 max |folded - reference| table coordinate (cells)
//   19 sum of it over both coordinates (18 / 19: points inside frame and window only)   20 coordinates summed
#ifdef NDTPSO_VERIFY_MARGIN
constexpr unsigned kVerifyMaxBlocks = 8192;
constexpr unsigned kVerifyRow = 24;
__device__ double g_verify[kVerifyMaxBlocks * kVerifyRow];
struct VerifyBin {
  double near11, near9, cell_differs, max_d, sum_d, n_d;
};
__device__ __forceinline__ void verify_max(double* slot, double v) {  // non-negative doubles order like their bit patterns
  atomicMax(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(v));
}
template <bool BYTE, bool POW2>
__device__ __forceinline__ void verify_pose_wave(const ExactArgs* ap, double c, double s, double tx, double ty, double* bound_out,
                                                 double* misbinned_out, VerifyBin* bin_out) {
  const GridP g = ap->g;
  const int dw = ap->dw, dh = ap->dh, ox = ap->ox, oy = ap->oy;
  const unsigned null_entry = ap->null_entry;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  const int lane = lane_id();
  const double u = 5.9604644775390625e-08;  // 2^-24
  // the folded constants of this pose, as dense_item / the proposal step form them
  const double C = c * g.inv_cs, S = s * g.inv_cs;
  const double TX = (tx + g.hw) * g.inv_cs - (double)ox, TY = (ty + g.hh) * g.inv_cs - (double)oy;
  const double XMAX = (2. * g.hw) * g.inv_cs - (double)ox, YMAX = (2. * g.hh) * g.inv_cs - (double)oy;
  double bound = 0., mis = 0.;
  VerifyBin vb{0., 0., 0., 0., 0., 0.};
  for (int i = lane; i < ap->n; i += kWave) {
    const v2d_t p = *(lds_d2_t)(uintptr_t)(ap->pts_lds + (unsigned)i * 16u);
    // reference binning (eval_pose_wave_exact)
    const double qx = (p.x * c - p.y * s) + tx, qy = (p.x * s + p.y * c) + ty;
    const bool inframe = (int)(fabs(qx) < g.hw) & (int)(fabs(qy) < g.hh);
    int ix, iy;
    cell_coords<POW2>(g, qx, qy, ix, iy);
    const bool wrap = (ix == g.W);
    ix = wrap ? 0 : ix;
    iy = wrap ? iy + 1 : iy;
    const unsigned rx = (unsigned)(ix - ox), ry = (unsigned)(iy - oy);
    const bool inwin = (int)inframe & (int)(rx <= (unsigned)dw) & (int)(ry <= (unsigned)dh);
    const unsigned lin = inwin ? ry * (unsigned)dense_stride(dw) + rx : 0u;
    const unsigned e = *(lds_u16_t)(uintptr_t)(lin << 1);
    // the fp32 loop's binning (score_trip_dense, clamped form; the no-clamp form is the same wherever its guard holds)
    const double gx = fma(p.x, C, fma(-p.y, S, TX)), gy = fma(p.x, S, fma(p.y, C, TY));
    const unsigned fx = min((unsigned)(int)gx, (unsigned)dw), fy = min((unsigned)(int)gy, (unsigned)dh);
    unsigned flin = fy * (unsigned)dense_stride(dw) + fx;
    if (!((int)(gx < XMAX) & (int)(gy < YMAX))) flin = 0u;  // (DenseP::clip; never true on a grid that does not overhang)
    const unsigned fe = *(lds_u16_t)(uintptr_t)(flin << 1);
    if (fe != e) mis += 1.;
    {  // how close the folded coordinates come to a cell edge, and to the reference's own
      const double ex = fabs(gx - rint(gx)), ey = fabs(gy - rint(gy));
      if (ex < 1e-11 || ey < 1e-11) vb.near11 += 1.;
      if (ex < 1e-9 || ey < 1e-9) vb.near9 += 1.;
      if (inwin) {
        if (flin != lin) vb.cell_differs += 1.;
        // the reference's coordinate in this table's units: fl(q + w/2) / cs - window origin (cs a power of two: exact scaling)
        const double rgx = (qx + g.hw) * g.inv_cs - (double)ox, rgy = (qy + g.hh) * g.inv_cs - (double)oy;
        const double dx = fabs(gx - rgx), dy = fabs(gy - rgy);
        vb.max_d = fmax(vb.max_d, fmax(dx, dy));
        vb.sum_d += dx + dy;
        vb.n_d += 2.;
      }
    }
    if (e != null_entry) {
      const unsigned slot = dense_rec_index(BYTE ? e - null_entry : (e - null_entry) << 4) - 1u;
      const double2 m = ap->xmean[slot], ab = ap->xab[slot], cd = ap->xcd[slot];
      double l[3];
      const bool okc = make_chol_d(ab.x, ab.y, cd.x, cd.y, l, g.cs);
      const double mgx = (m.x + g.hw) * g.inv_cs - (double)ox, mgy = (m.y + g.hh) * g.inv_cs - (double)oy;
      const DenseRecA ra = dense_rec_a(l, mgx, mgy, okc);
      const double sv = fma(ra.r, gy, gx) + ra.alpha;
      const double a = l[0] * sv, b = l[2] * (gy - mgy);
      const double e64 = 3.4e-16 * (fabs(gx) + fabs(ra.r * gy) + fabs(ra.alpha));
      const double q = a * a + b * b, t = exp2(-q);
      bound += t * (u * (5.55 * q + 4.) + 1.4 * fabs(a * l[0]) * e64);
    }
  }
  *bound_out = 1.01 * wave_sum(bound) + (double)ap->n * 1.1754943508222875e-38;
  *misbinned_out = wave_sum(mis);
  bin_out->near11 = wave_sum(vb.near11);
  bin_out->near9 = wave_sum(vb.near9);
  bin_out->cell_differs = wave_sum(vb.cell_differs);
  bin_out->sum_d = wave_sum(vb.sum_d);
  bin_out->n_d = wave_sum(vb.n_d);
  double m = vb.max_d;
  for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  bin_out->max_d = m;
}
template <bool BYTE>
__device__ __forceinline__ void verify_item(const ExactArgs* ap, int j, double cost32, double ref_cost) {
  const double c = ap->pcs[j], s = ap->pcs[ap->S + j], tx = ap->tpos[j], ty = ap->tpos[ap->S + j];
  double cost64, bound, mis;
  VerifyBin vb;
  if (ap->g.cs_pow2) {
    cost64 = eval_pose_wave_exact<BYTE, true>(ap, c, s, tx, ty);
    verify_pose_wave<BYTE, true>(ap, c, s, tx, ty, &bound, &mis, &vb);
  } else {
    cost64 = eval_pose_wave_exact<BYTE, false>(ap, c, s, tx, ty);
    verify_pose_wave<BYTE, false>(ap, c, s, tx, ty, &bound, &mis, &vb);
  }
  if (lane_id() == 0 && blockIdx.x < kVerifyMaxBlocks && cost32 == cost32 && cost64 == cost64) {
    double* o = g_verify + (size_t)blockIdx.x * kVerifyRow;
    atomicAdd(o + 15, vb.near11);
    atomicAdd(o + 16, vb.near9);
    atomicAdd(o + 17, vb.cell_differs);
    verify_max(o + 18, vb.max_d);
    atomicAdd(o + 19, vb.sum_d);
    atomicAdd(o + 20, vb.n_d);
    const double err = fabs(cost32 - cost64);
    const double ref = fabs(ref_cost) > 0. ? fabs(ref_cost) : fabs(cost64);  // (swarm initialisation: relative to the cost itself)
    const double half_tau = 0.5 * arb_margin(ref, ap->n);
    verify_max(o + 0, err);
    if (bound > 0.) verify_max(o + 1, err / bound);
    if (half_tau > 1e-30) {
      verify_max(o + 2, bound / half_tau);
      verify_max(o + 3, err / half_tau);
    }
    atomicAdd(o + 4, 1.);
    atomicAdd(o + 5, mis);
    verify_max(o + 6, bound);
    atomicAdd(o + 7, (double)ap->n);
    // 8 evaluations with err > B   9 iteration-phase evaluations with err > tau / 2   10 max err / (tau / 2), iteration phase
    // 11-13 fp32 score, fp64 score and B of the evaluation with the largest err / B so far   14 max err, iteration phase
    if (err > bound) {
      atomicAdd(o + 8, 1.);
      if (bound > 0. && err / bound >= o[1]) {
        o[11] = cost32;
        o[12] = cost64;
        o[13] = bound;
      }
    }
    if (ref_cost != 0. && half_tau > 1e-30) {
      if (err > half_tau) atomicAdd(o + 9, 1.);
      verify_max(o + 10, err / half_tau);
      verify_max(o + 14, err);
    }
  }
}
#endif

// ---- the arbitration's fp64 scores, split for latency (round 3) ----------------------------------------------------
// One fp64 score used to be one wave's job: 17 chunks in five dependent rounds of gathers (four chunks in flight), 6 us
// with the records in HBM -- and an alignment of the live sequence arbitrates 4 comparisons (11-15 us each with the
// barriers around them: 47 us of a 0.5 ms scan).  The score's sum is four independent lane accumulators -- chunk k of a
// four-chunk trip goes to accumulator k, the chunks behind the last full trip to accumulator 0, then (a0 + a1) + (a2 + a3)
// and the wave reduction (eval_pose_wave_exact) -- so a UNIT of work is one accumulator of one score: four or five
// chunks, all in flight at once, summed in the order the one-wave loop adds them.  Units are dealt to the waves; each
// leaves its 64 lane values in LDS, and after a barrier one wave per score folds the four and reduces: the same bits
// as before, in one round of gathers instead of five.
// Which scores are needed is decided first (exact_task_list): the proposals of the noted items always; a pbest or the
// gbest position only if its stored cost is not already an fp64 score from an earlier arbitration (Swarm::pex,
// ExactArgs::gex -- in a converged swarm the same particles tie round after round).
template <bool BYTE, bool POW2>
__device__ __forceinline__ double exact_partial(const ExactArgs* ap, double c, double s, double tx, double ty, int acc) {
  const GridP g = ap->g;
  const int dw = ap->dw, dh = ap->dh, ox = ap->ox, oy = ap->oy;
  const unsigned null_entry = ap->null_entry;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  typedef const v2d_t __attribute__((address_space(3))) * lds_d2_t;
  typedef const unsigned short __attribute__((address_space(3))) * lds_u16_t;
  const unsigned pts_lds = ap->pts_lds;
  const double2* __restrict__ xmean = ap->xmean;
  const double2* __restrict__ xab = ap->xab;
  const double2* __restrict__ xcd = ap->xcd;
  const int chunks = round_up(ap->n, kWave) / kWave, in_trips = chunks & ~3;
  const int lane = lane_id();
  auto term_of = [&](int k) -> double {  // (eval_pose_wave_exact's, operation for operation)
    const v2d_t p = *(lds_d2_t)(uintptr_t)(pts_lds + (unsigned)(k * kWave + lane) * 16u);
    const double qx = (p.x * c - p.y * s) + tx;
    const double qy = (p.x * s + p.y * c) + ty;
    const bool inframe = (int)(fabs(qx) < g.hw) & (int)(fabs(qy) < g.hh);
    int ix, iy;
    cell_coords<POW2>(g, qx, qy, ix, iy);
    const bool wrap = (ix == g.W);
    ix = wrap ? 0 : ix;
    iy = wrap ? iy + 1 : iy;
    const unsigned rx = (unsigned)(ix - ox), ry = (unsigned)(iy - oy);
    const bool inwin = (int)inframe & (int)(rx <= (unsigned)dw) & (int)(ry <= (unsigned)dh);
    const unsigned lin = inwin ? ry * (unsigned)dense_stride(dw) + rx : 0u;
    const unsigned e = *(lds_u16_t)(uintptr_t)(lin << 1);
    const bool hit = e != null_entry;
    const unsigned slot = hit ? dense_rec_index(BYTE ? e - null_entry : (e - null_entry) << 4) - 1u : 0u;
    const double2 m = xmean[slot], ab = xab[slot], cd = xcd[slot];
    const double d0 = qx - m.x, d1 = qy - m.y;
    const double r0 = d0 * ab.x + d1 * cd.x;
    const double r1 = d0 * ab.y + d1 * cd.y;
    const double x = -(r0 * d0 + r1 * d1) / 2.;
    return exp(hit ? x : -(double)__builtin_inff());
  };
  // this accumulator's chunks, in the order the one-wave loop adds them: acc, acc + 4, ... below in_trips, then (acc 0)
  // the chunks behind the last full trip
  double a = 0.;
  int k = acc;
#pragma unroll 1
  for (; k + 12 < in_trips; k += 16) {  // four of them in flight
    const double t0 = term_of(k), t1 = term_of(k + 4), t2 = term_of(k + 8), t3 = term_of(k + 12);
    a += t0;
    a += t1;
    a += t2;
    a += t3;
  }
#pragma unroll 1
  for (; k < in_trips; k += 4) a += term_of(k);
  if (acc == 0) {
#pragma unroll 1
    for (k = in_trips; k < chunks; ++k) a += term_of(k);
  }
  return a;
}

#ifdef NDTPSO_TRACE_ARB  // diagnostic builds (scripts/units_hbm_diag.py): what every arbitration decided, per workgroup
constexpr unsigned kTraceBlocks = 256, kTraceDoubles = 2048;
__device__ double g_arbtrace[kTraceBlocks * kTraceDoubles];
__device__ __forceinline__ void arb_trace(double v) {  // thread 0 only
  if (blockIdx.x < kTraceBlocks) {
    double* t = g_arbtrace + (size_t)blockIdx.x * kTraceDoubles;
    const unsigned n = (unsigned)t[0];
    if (n + 2 < kTraceDoubles) {
      t[n + 1] = v;
      t[0] = (double)(n + 1);
    }
  }
}
#define NDTPSO_ARB_TRACE(v) arb_trace((double)(v))
#else
#define NDTPSO_ARB_TRACE(v) do { } while (0)
#endif

#ifndef NDTPSO_EXACT_CALL
#define NDTPSO_EXACT_CALL 1
#endif
// unit u = 4 * task + accumulator: computes it and leaves the lane values in scratch slot (u mod xs_slots)
template <bool BYTE>
__device__ __forceinline__ void exact_unit_body(const ExactArgs* ap, int u) {
  const unsigned tk = ap->task[u >> 2];
  const int j = (int)(tk >> 2), kind = (int)(tk & 3u), acc = u & 3;
  double x, y, cn, sn;
  if (kind == 2) {
    x = ap->gb[0];
    y = ap->gb[1];
    cn = ap->gcs[0];
    sn = ap->gcs[1];
  } else {
    const double* src = kind == 1 ? ap->pb : ap->tpos;
    const double* cs = kind == 1 ? ap->bcs : ap->pcs;
    x = src[j];
    y = src[ap->S + j];
    cn = cs[j];
    sn = cs[ap->S + j];
  }
  const double a = ap->g.cs_pow2 ? exact_partial<BYTE, true>(ap, cn, sn, x, y, acc) : exact_partial<BYTE, false>(ap, cn, sn, x, y, acc);
  typedef double __attribute__((address_space(3))) * lds_d_t;
  *(lds_d_t)(uintptr_t)(ap->xs_lds + (unsigned)((u % ap->xs_slots) * kWave + lane_id()) * 8u) = a;
}
// this wave's units of a pass: u0, u0 + stride, ... below u_end.  Out of line (and cold) in the one-workgroup kernels,
// which sit at their register limit: the fp32-score kernels must not carry this code in their hot paths' allocation.
// The cluster kernels have registers to spare and inline it (INL): a call there costs more than it saves -- callee-saved
// registers go through scratch memory on every call, and the live sequence arbitrates four comparisons per scan.
template <bool BYTE>
__device__ __attribute__((noinline, cold)) void exact_unit_call(const ExactArgs* ap, int u) { exact_unit_body<BYTE>(ap, u); }
template <bool BYTE, bool INL>
__device__ __forceinline__ void exact_units(const ExactArgs* ap, int u0, int stride, int u_end) {
#pragma unroll 1
  for (int u = u0; u < u_end; u += stride) {
    if constexpr (INL || !NDTPSO_EXACT_CALL)
      exact_unit_body<BYTE>(ap, u);
    else
      exact_unit_call<BYTE>(ap, u);
  }
}
// the four accumulators of task t (scratch slots of units 4t .. 4t + 3) -> its score, stored where the task says
__device__ __forceinline__ void exact_combine(const ExactArgs* ap, int t) {
  typedef const double __attribute__((address_space(3))) * lds_d_t;
  const unsigned base = ap->xs_lds + (unsigned)(((4 * t) % ap->xs_slots) * kWave + lane_id()) * 8u;
  const double a0 = *(lds_d_t)(uintptr_t)base, a1 = *(lds_d_t)(uintptr_t)(base + kWave * 8u),
               a2 = *(lds_d_t)(uintptr_t)(base + 2u * kWave * 8u), a3 = *(lds_d_t)(uintptr_t)(base + 3u * kWave * 8u);
#ifdef NDTPSO_BREAK_ARBITRATION  // test builds only (tests/test_gpu_exact_check.py): the start-up check must refuse this library
  const double c = -wave_sum((a0 + a1) + (a2 + a3)) * (1. + 0x1p-44);
#else
  const double c = -wave_sum((a0 + a1) + (a2 + a3));
#endif
  if (lane_id() == 0) {
    const unsigned tk = ap->task[t];
    const int j = (int)(tk >> 2), kind = (int)(tk & 3u);
    double* dst = kind == 2 ? ap->xgbc : (kind == 1 ? &ap->pbc[j] : &ap->tcost[j]);
    *dst = c;
  }
}
// kind 0: the swarm initialisation's candidates (proposal scores of the listed items); 1: a round's noted items
// (proposal; pbest and gbest unless their costs are fp64 scores already); 2: the gbest position (the returned cost).
// Every thread of the workgroup calls this; ap is the block in LDS (PsoShared::xa), written by thread 0 only.
// one whole fp64 score by this wave (the cluster kernels' way, see exact_tasks_wg)
template <bool BYTE>
__device__ __forceinline__ void exact_task_body(const ExactArgs* ap, unsigned tk) {
  const int j = (int)(tk >> 2), kd = (int)(tk & 3u);
  double x, y, cn, sn;
  if (kd == 2) {
    x = ap->gb[0];
    y = ap->gb[1];
    cn = ap->gcs[0];
    sn = ap->gcs[1];
  } else {
    const double* src = kd == 1 ? ap->pb : ap->tpos;
    const double* cs = kd == 1 ? ap->bcs : ap->pcs;
    x = src[j];
    y = src[ap->S + j];
    cn = cs[j];
    sn = cs[ap->S + j];
  }
#ifdef NDTPSO_BREAK_ARBITRATION  // test builds only (tests/test_gpu_exact_check.py): the whole-task form broken like the unit form
  const double c = (ap->g.cs_pow2 ? eval_pose_wave_exact<BYTE, true>(ap, cn, sn, x, y) : eval_pose_wave_exact<BYTE, false>(ap, cn, sn, x, y)) * (1. + 0x1p-44);
#else
  const double c = ap->g.cs_pow2 ? eval_pose_wave_exact<BYTE, true>(ap, cn, sn, x, y) : eval_pose_wave_exact<BYTE, false>(ap, cn, sn, x, y);
#endif
  if (lane_id() == 0) {
    double* dst = kd == 2 ? ap->xgbc : (kd == 1 ? &ap->pbc[j] : &ap->tcost[j]);
    *dst = c;
  }
}
// (out of line and cold here as well: with the fp64 score inlined the cluster kernel's own evaluation loop got slower)
template <bool BYTE>
__device__ __attribute__((noinline, cold)) void exact_task_call(const ExactArgs* ap, unsigned tk) { exact_task_body<BYTE>(ap, tk); }

// UNITS: the kernel's launches always come with the units' scratch (the fused pairs kernels that keep the swarm in LDS): the
// whole-task form -- and with it exact_task_call, whose frame was two thirds of those kernels' scratch memory -- is left out.
template <bool BYTE, bool INL = false, bool UNITS = false>
__device__ __forceinline__ void exact_tasks_wg(ExactArgs* ap, const unsigned short* list, int cnt, int kind) {
  if (!UNITS && (INL || ap->xs_slots == 0)) {  // (xs_slots: uniform, set once by enable_arbitration)
    // A cluster's workgroups have four waves (one per SIMD).  A whole score per wave, all of them at once, is as quick
    // there as the split into units (three scores are twelve units, three per wave), and it needs neither the work list
    // nor the second barrier: task slot t is fixed -- kind 0: the proposal of item t; kind 1: 2q the proposal and 2q + 1
    // the pbest of item q, 2 cnt the gbest; kind 2: the gbest -- and a slot whose cost is an fp64 score already is
    // skipped by its wave.  (Measured on the live sequence, same box: 0.502 ms per scan before the units, 0.514 with
    // them, 0.52 through an out-of-line call.)
    const int n_slots = kind == 0 ? cnt : (kind == 1 ? 2 * cnt + 1 : 1), n_waves = (int)(blockDim.x >> 6);
    if (threadIdx.x == 0) ap->gb_task = (kind != 0 && !ap->gex) ? 1 : 0;  // (read by the callers after their barrier)
#pragma unroll 1
    for (int t = wave_id(); t < n_slots; t += n_waves) {
      const bool gb = kind == 2 || (kind == 1 && t == 2 * cnt);
      unsigned tk;
      if (gb) {
        if (ap->gex) continue;
        tk = 2u;
      } else {
        const unsigned j = list[kind == 0 ? t : (t >> 1)];
        const bool pbest = kind == 1 && (t & 1);
        if (pbest && ap->pex[j]) continue;
        tk = (j << 2) | (pbest ? 1u : 0u);
      }
      exact_task_call<BYTE>(ap, tk);
    }
    __syncthreads();
    if (threadIdx.x == 0 && kind == 1) {
      for (int q = 0; q < cnt; ++q) ap->pex[list[q]] = 1;
      ap->gex = 1;
    }
    return;
  }
  if (threadIdx.x == 0) {
    int n = 0;
    ap->gb_task = 0;
    if (kind != 2)
      for (int q = 0; q < cnt; ++q) {
        const unsigned j = list[q];
        ap->task[n++] = (unsigned short)(j << 2);
        if (kind == 1 && !ap->pex[j]) ap->task[n++] = (unsigned short)((j << 2) | 1u);
      }
    if (kind != 0 && !ap->gex) {
      ap->task[n++] = 2;
      ap->gb_task = 1;
    }
    ap->n_task = n;
  }
  __syncthreads();
  const int n_tasks = ap->n_task, n_waves = (int)(blockDim.x >> 6);
  const int per_pass = max(1, ap->xs_slots >> 2);  // tasks whose four partial sums fit in the scratch at once
#pragma unroll 1
  for (int t0 = 0; t0 < n_tasks; t0 += per_pass) {
    const int t1 = min(n_tasks, t0 + per_pass);
    // the pass's units, dealt round robin to the waves; a wave with more than one does them one after the other.  Only
    // the waves that have a unit go in (see exact_unit_call)
    if (4 * t0 + wave_id() < 4 * t1) exact_units<BYTE, INL>(ap, 4 * t0 + wave_id(), n_waves, 4 * t1);
    __syncthreads();
    for (int t = t0 + wave_id(); t < t1; t += n_waves) exact_combine(ap, t);
    __syncthreads();
  }
#ifdef NDTPSO_TRACE_ARB
  if (threadIdx.x == 0) {
    NDTPSO_ARB_TRACE(-1000 - kind);
    NDTPSO_ARB_TRACE(n_tasks);
    for (int t = 0; t < n_tasks; ++t) {
      const unsigned tk = ap->task[t];
      const int j = (int)(tk >> 2), kd = (int)(tk & 3u);
      NDTPSO_ARB_TRACE(tk);
      NDTPSO_ARB_TRACE(kd == 2 ? *ap->xgbc : (kd == 1 ? ap->pbc[j] : ap->tcost[j]));
    }
  }
#endif
  if (threadIdx.x == 0 && kind == 1) {  // what the scores just stored are from now on
    for (int q = 0; q < cnt; ++q) ap->pex[list[q]] = 1;
    ap->gex = 1;
  }
}

// One wave per item.  `improver` (optional): the evaluating wave itself records the lowest item index whose
// cost beats `gbc` (core.cpp:97 under single-thread order), so no separate detection pass is needed.
// KGEN: the light-wave deal for any number k of items per wave (PsoP::light = k); without it k is two.  Only the copies of
// the PSO that keep their swarm in HBM carry it (large swarms: make_pso) -- in the 70-particle kernels the general
// deal's few extra instructions cost 2.4 % (exact) / 1 % (fp32) on config 3, same box.
template <int MODE, int PATH, bool ARB = false, bool NOCLIP = false, bool KGEN = false, bool BOX = false>
__device__ inline void eval_items(const EvalCtx& E, const double2* pts, int n, const Swarm& sw, int S, int first,
                                  int last /*exclusive*/, double gbc, int* improver, int* tiny, int* near_cnt,
                                  unsigned short* near_list) {
  const int n_waves = blockDim.x >> 6;
  int j0, dj, jend;
  if constexpr (KGEN) {
    // E.light = k >= 2 items per other wave: a round of k (n - 1) + 1 items; wave w >= 1 takes items w - 1, w - 1 + (n - 1),
    // ... below k (n - 1), wave 0 the one item behind them.  `light` off, or a longer round (the swarm's
    // initialisation): plain striding.
    const int heavy_items = E.light * (n_waves - 1);
    const bool light = E.light && last - first <= heavy_items + 1;
    j0 = light ? (wave_id() == 0 ? first + heavy_items : first + wave_id() - 1) : first + wave_id();
    dj = light ? (n_waves - 1) : n_waves;
    jend = (light && wave_id() != 0) ? min(last, first + heavy_items) : (light ? min(last, j0 + 1) : last);
  } else {
    // item k of the round goes to wave k + 1 for k < n - 1, to wave k - (n - 1) after that: wave 0 gets one item (k = n - 1)
    // of a round of 2n - 1, the others two.  `light` off, or a longer round (the swarm's initialisation): plain striding.
    const bool light = E.light && last - first <= 2 * n_waves - 1;
    j0 = light ? (wave_id() == 0 ? first + n_waves - 1 : first + wave_id() - 1) : first + wave_id();
    dj = light ? (wave_id() == 0 ? 2 * n_waves : n_waves) : n_waves;
    jend = last;
  }
  for (int j = j0; j < jend; j += dj) {
    const double c = sw.it[4 * j], s = sw.it[4 * j + 1];
    const double pbc_j = sw.pbc[j];  // fetched with the pose, not after the evaluation (garbage during the swarm's
                                     // initialisation, where it is not looked at)
    double cost;
    if constexpr (path_is_dense(PATH)) {
      const DenseItem it{c, s, sw.it[4 * j + 2], sw.it[4 * j + 3], E.dn.xmax, E.dn.ymax};  // folded where the proposal was made
      if constexpr (PATH == 3 || PATH == 2) {  // (the fused pairs kernels: they set up the guard -- E.guard_lds says so for the
                                               // one form k_align shares with them, PATH 2 on a grid that may overhang its frame)
        if ((PATH == 3 || NOCLIP || E.guard_lds) &&
            (BOX ? *reinterpret_cast<const unsigned*>(sw.tgd + 4 * j) == 0x01010101u    // inside the guard: x, y, heading
                 : *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u))  // inside the guard both ways (proposal step)
        {
#ifdef NDTPSO_COUNT_NOCLAMP  // diagnostic builds: evaluations through the no-clamp loop, reported as `gbest_updates`
          if (lane_id() == 0) atomicAdd(tiny + 2, 1);  // PsoShared::timed_out (unused by a single workgroup)
#endif
          cost = eval_item_wave_dense<false, PATH == 3, true, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
        }
        else
          cost = eval_item_wave_dense<false, PATH == 3, false, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
      } else {
        cost = eval_item_wave_dense<false, PATH == 3, false, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
      }
    } else {
      const double tx = sw.tpos[j], ty = sw.tpos[S + j];
      if constexpr (path_is_dense64(PATH)) {  // (fp64 score on the dense table: the same guard)
        static_assert(!path_is_dense64(PATH) || MODE == kScoreF64, "PATH 8 / 9 are the fp64 score's");
        if (BOX ? *reinterpret_cast<const unsigned*>(sw.tgd + 4 * j) == 0x01010101u
                : *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u)
          cost = eval_pose_wave_d64<(PATH & 3) == 1, true>(E.g, E.dn, E.d64_tab, pts, n, c, s, tx, ty);
        else
          cost = eval_pose_wave_d64<(PATH & 3) == 1, false>(E.g, E.dn, E.d64_tab, pts, n, c, s, tx, ty);
      } else if constexpr (MODE == kScoreF64 && PATH < 4) {  // (fp64 score of the batches: the guard in metres, set with the proposal)
        if (E.guard_lds && *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u)
          cost = eval_pose_wave_t<MODE, (PATH & 3) == 1, false, true>(E.g, E.wn, E.T, pts, n, c, s, tx, ty, nullptr);
        else
          cost = eval_pose_wave<MODE, (PATH & 3) == 1>(E.g, E.wn, E.T, pts, n, c, s, tx, ty);
      } else {
        cost = eval_pose_wave<MODE, (PATH & 3) == 1>(E.g, E.wn, E.T, pts, n, c, s, tx, ty);
      }
    }
#ifdef NDTPSO_VERIFY_MARGIN
    if constexpr (ARB && (PATH == 2 || PATH == 3))
      if (E.xa) verify_item<PATH == 3>(E.xa, j, cost, improver ? gbc : 0.);
#endif
    if (lane_id() == 0) {
      sw.tcost[j] = cost;
      // A cost in the fp32 underflow regime is only ambiguous when what it is compared with is there too: an
      // outlier particle that left the map scores ~0 against a pbest of -400 and loses in any arithmetic.
      // improver == nullptr is the swarm initialisation, where the cost becomes the particle's pbest.
      // (a NaN cost -- a cell without a Cholesky factor, make_chol -- goes the same way: the fp64 form evaluates it)
      // (written so that the usual evaluation -- a real cost that does not beat the gbest -- passes two comparisons:
      // these few instructions are issued once per evaluation by every wave, about 1 % of the kernel)
      bool ordinary = true;
      if (MODE == kScoreF32 && !(cost <= -kTinyCost)) {  // NaN, or the underflow regime
        // (exact mode: a cost in the underflow regime lies within the arbitration margin -- never below 7e-7 per point -- of
        // whatever it could be confused with, so the comparison is arbitrated in fp64 like any other near tie and nothing
        // needs handing over; only a NaN, which no comparison notices, still does)
        if (cost != cost || (!ARB && (!improver || pbc_j > -kTinyCost))) {
          *tiny = 1;  // the alignment is handed to the fp64-score kernel (see pso_run_wg)
          ordinary = false;
        }
      }
      // core.cpp:94-104: the gbest test sits inside the pbest test.  The two agree (gbest <= pbest) except for a
      // particle whose pbest cost is NaN (its first position touched a cell with a NaN inverse covariance): that
      // particle never passes `cost < best_cost`, so it never moves the gbest either.
      if (ordinary && improver && cost < gbc)
        if (cost < pbc_j) atomicMin(improver, j);
      // exact mode: a comparison too close to call is noted for arbitration (pso_run_wg)
      if constexpr (ARB) {
        if (improver) {
          const double tau = arb_margin(gbc, n);  // (uniform: scalar-unit work)
          if (near_tie(cost, pbc_j, tau) || near_tie(cost, gbc, tau)) near_note(near_cnt, near_list, j);
        }
      }
    }
  }
}

// ---- items of a whole iteration dealt by ticket (round 4) -------------------------------------------------------------
//
// NDTPSO_STREAM (the one-workgroup kernels; a cluster keeps its rounds): the evaluation rounds of an iteration -- 15 items,
// a barrier, 15 items, a barrier ... -- cost a two-item wave a fifth of its time waiting for the round's slowest wave
// (profiles/r03_phase_budget.json), and 70 particles in rounds of 15 are ten item-times per iteration where 70 / 8 is
// 8.75.  Here a PHASE covers every particle not yet committed: the waves take items in index order from a ticket counter
// in LDS, as fast as each gets through them, and meet once at the end.  The exact-order semantics are the rounds': all
// items of a phase carry proposals against the same gbest; the lowest item that beats it (core.cpp:94-104, nested
// tests; atomicMin into *improver by the evaluating wave) ends the phase's useful part -- items up to it are committed,
// everything behind it is proposed again.  What a round bounded by its size a phase bounds by looking: before it takes
// an item a wave reads *improver, and leaves when its ticket lies behind it (so do all later tickets: *improver only
// falls).  Every item below the final first improver has been evaluated -- a ticket is only ever dropped behind an
// improver that was already known --, the evaluated items are [lo, *jmax], and what is thrown away per gbest update
// is what was in flight: about as much as the tail of a round.
//
// The ticket: lane 0 adds one to the counter, every other lane adds zero to one of 32 spare words -- no branch and no two
// lanes on the counter.  (All 64 lanes adding to one address serialise in the LDS unit, an LDS atomic round trip in front
// of every item of a 15-item round lost 12 % in round 1, and the `if (lane == 0) atomicAdd` + readfirstlane spelling
// produced a loop that never ended with that round's compiler, NOTEBOOK 5.1; a one-lane exec mask in inline assembly
// crashed this compiler's register allocator.)
__device__ __forceinline__ int take_ticket(int* counter, int* spare /* 32 words */) {
  typedef int __attribute__((address_space(3))) * lds_int_t;
  const int lane = lane_id();
  lds_int_t a = lane == 0 ? (lds_int_t)counter : (lds_int_t)spare + (lane & 31);
  const int v = __hip_atomic_fetch_add(a, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int take_two_tickets(int* counter, int* spare /* 32 words */) {  // (the pair kernels: items j, j + 1)
  typedef int __attribute__((address_space(3))) * lds_int_t;
  const int lane = lane_id();
  lds_int_t a = lane == 0 ? (lds_int_t)counter : (lds_int_t)spare + (lane & 31);
  const int v = __hip_atomic_fetch_add(a, lane == 0 ? 2 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(v);
}

template <int MODE, int PATH, bool ARB = false, bool NOCLIP = false, bool BOX = false, bool PAIR = false>
__device__ inline void eval_stream(const EvalCtx& E, const double2* pts, int n, const Swarm& sw, int S, int P, double gbc,
                                   int* ticket, int* spare, int* improver, int* jmax, int* tiny, int* near_cnt,
                                   unsigned short* near_list) {
  typedef int __attribute__((address_space(3))) * lds_int_t;
  int last_done = -1;
  if constexpr (PAIR) {
    // ---- two items per wave (eval_pair_half): tickets j and j + 1 at once ----
    static_assert(!PAIR || (MODE == kScoreF32 && (PATH == 3 || PATH == 2) && NOCLIP), "the pair form: fp32 score, dense table, no clipping trips");
    unsigned late;
    asm("s_cmp_ge_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0"
        : "=s"(late)
        : "s"(__builtin_amdgcn_readfirstlane((unsigned)blockIdx.x)), "s"(__builtin_amdgcn_readfirstlane((unsigned)gridDim.x >> 1))
        : "scc");
    // what lane 0 does with an item's cost in the one-item loop below, by the lane that holds it
    auto decide = [&](int j, double cost, double pbc_j) {
      sw.tcost[j] = cost;
      bool ordinary = true;
      if (!(cost <= -kTinyCost)) {
        if (cost != cost || (!ARB && pbc_j > -kTinyCost)) {
          if constexpr (ARB)
            *(lds_int_t)tiny = 0;
          else
            __hip_atomic_fetch_min((lds_int_t)tiny, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ordinary = false;
        }
      }
      if constexpr (ARB) {
        const double tau = arb_margin(gbc, n);
        const double d1 = cost - pbc_j, d2 = cost - gbc;
        if (ordinary && !(fmin(d1, d2) > tau)) {
          if (cost < gbc)
            if (cost < pbc_j) __hip_atomic_fetch_min((lds_int_t)improver, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (fabs(d1) <= tau || fabs(d2) <= tau) *(lds_int_t)near_cnt = 1;
        }
      } else {
        if (ordinary && cost < gbc)
          if (cost < pbc_j) __hip_atomic_fetch_min((lds_int_t)improver, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    for (;;) {
      const int seen = __hip_atomic_load((lds_int_t)improver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int j = take_two_tickets(ticket, spare);
      if (j >= P || j > seen) break;
      {
        unsigned long long now;
        unsigned turn;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now));
        asm("s_bfe_u32 %0, %1, 0x40009\n\ts_cmp_lt_u32 %0, %2\n\ts_cselect_b32 %0, 1, 0" : "=&s"(turn) : "s"((unsigned)now), "n"(NDTPSO_PRIO_SHARE) : "scc");
        if (turn == late)
          __builtin_amdgcn_s_setprio(1);
        else
          __builtin_amdgcn_s_setprio(0);
      }
      const bool has_b = j + 1 < P && j + 1 <= seen;  // (uniform)
      const bool g_a = BOX ? *reinterpret_cast<const unsigned*>(sw.tgd + 4 * j) == 0x01010101u
                           : *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u;
      const bool g_b = !has_b || (BOX ? *reinterpret_cast<const unsigned*>(sw.tgd + 4 * (j + 1)) == 0x01010101u
                                      : *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * (j + 1)) == 0x0101u);
      last_done = has_b ? j + 1 : j;
      if (g_a && g_b) {  // both inside the guard: one pass of the list for the two
        const bool second = lane_id() >= 32;
        const int jl = (second && has_b) ? j + 1 : j;
        const DenseItem it{sw.it[4 * jl], sw.it[4 * jl + 1], sw.it[4 * jl + 2], sw.it[4 * jl + 3], E.dn.xmax, E.dn.ymax};
        const double pbc_l = sw.pbc[jl];
        const double cost = eval_pair_half<PATH == 3>(E.dn, pts, n, it);
#ifdef NDTPSO_VERIFY_MARGIN
        if constexpr (ARB) {
          if (E.xa) {  // (both items' fp32 scores against their fp64 scores and the bound, as the one-item loop does)
            const double cost_a = readlane_f64(cost, 31), cost_b = readlane_f64(cost, 63);
            verify_item<PATH == 3>(E.xa, j, cost_a, gbc);
            if (has_b) verify_item<PATH == 3>(E.xa, j + 1, cost_b, gbc);
          }
        }
#endif
        if ((lane_id() & 31) == 31 && (!second || has_b)) decide(jl, cost, pbc_l);
      } else {  // an item outside the guard: the one-item forms, one after the other
        for (int q = 0; q < (has_b ? 2 : 1); ++q) {
          const int jq = j + q;
          const DenseItem it{sw.it[4 * jq], sw.it[4 * jq + 1], sw.it[4 * jq + 2], sw.it[4 * jq + 3], E.dn.xmax, E.dn.ymax};
          const double pbc_q = sw.pbc[jq];
          double cost;
          if (q == 0 ? g_a : g_b)
            cost = eval_item_wave_dense<false, PATH == 3, true, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
          else
            cost = eval_item_wave_dense<false, PATH == 3, false, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
          if (lane_id() == 0) decide(jq, cost, pbc_q);
        }
      }
    }
    if (lane_id() == 0 && last_done >= 0) __hip_atomic_fetch_max((lds_int_t)jmax, last_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
#if NDTPSO_ALTERNATE_PRIO
  // (the turn is worked out on the scalar unit, spelt in its own instructions: as C the compiler made a 64-bit vector
  // compare and two selects per item of it -- the clock's value counts as divergent -- whatever was wrapped in readfirstlane)
  unsigned late;
  asm("s_cmp_ge_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0"
      : "=s"(late)
      : "s"(__builtin_amdgcn_readfirstlane((unsigned)blockIdx.x)), "s"(__builtin_amdgcn_readfirstlane((unsigned)gridDim.x >> 1))
      : "scc");
#endif
  for (;;) {
    // (an LDS read in flight together with the ticket; through a generic pointer it was a flat load with system scope)
    const int seen = __hip_atomic_load((lds_int_t)improver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int j = take_ticket(ticket, spare);
    if (j >= P || j > seen) break;
#if NDTPSO_ALTERNATE_PRIO
    // the two workgroups of a compute unit take turns holding the higher priority (see pso_run_wg)
    {
      unsigned long long now;
      unsigned turn;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now));
      asm("s_bfe_u32 %0, %1, 0x40009\n\ts_cmp_lt_u32 %0, %2\n\ts_cselect_b32 %0, 1, 0" : "=&s"(turn) : "s"((unsigned)now), "n"(NDTPSO_PRIO_SHARE) : "scc");
      if (turn == late)
        __builtin_amdgcn_s_setprio(1);
      else
        __builtin_amdgcn_s_setprio(0);
    }
#endif
    const double c = sw.it[4 * j], s = sw.it[4 * j + 1];
    const double pbc_j = sw.pbc[j];
    double cost;
    if constexpr (path_is_dense(PATH)) {
      const DenseItem it{c, s, sw.it[4 * j + 2], sw.it[4 * j + 3], E.dn.xmax, E.dn.ymax};  // folded where the proposal was made
      if constexpr (PATH == 3 || PATH == 2) {  // (the fused pairs kernels: they set up the guard -- E.guard_lds says so for the
                                               // one form k_align shares with them, PATH 2 on a grid that may overhang its frame)
        if ((PATH == 3 || NOCLIP || E.guard_lds) &&
            (BOX ? *reinterpret_cast<const unsigned*>(sw.tgd + 4 * j) == 0x01010101u    // inside the guard: x, y, heading
                 : *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u))  // inside the guard both ways (proposal step)
        {
#ifdef NDTPSO_COUNT_NOCLAMP
          if (lane_id() == 0) atomicAdd(tiny + 1, 1);  // (`tiny` is PsoShared::tiny_j here) PsoShared::timed_out, as eval_items
#endif
          cost = eval_item_wave_dense<false, PATH == 3, true, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
        }
        else
          cost = eval_item_wave_dense<false, PATH == 3, false, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
      } else {
        cost = eval_item_wave_dense<false, PATH == 3, false, NOCLIP>(E.g, E.dn, E.lds0, pts, n, it);
      }
    } else {
      const double tx = sw.tpos[j], ty = sw.tpos[S + j];
      if constexpr (path_is_dense64(PATH)) {  // (fp64 score on the dense table: the same guard)
        static_assert(!path_is_dense64(PATH) || MODE == kScoreF64, "PATH 8 / 9 are the fp64 score's");
        if (BOX ? *reinterpret_cast<const unsigned*>(sw.tgd + 4 * j) == 0x01010101u
                : *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u)
          cost = eval_pose_wave_d64<(PATH & 3) == 1, true>(E.g, E.dn, E.d64_tab, pts, n, c, s, tx, ty);
        else
          cost = eval_pose_wave_d64<(PATH & 3) == 1, false>(E.g, E.dn, E.d64_tab, pts, n, c, s, tx, ty);
      } else if constexpr (MODE == kScoreF64 && PATH < 4) {  // (fp64 score of the batches: the guard in metres, set with the proposal)
        if (E.guard_lds && *reinterpret_cast<const unsigned short*>(sw.tgd + 2 * j) == 0x0101u)
          cost = eval_pose_wave_t<MODE, (PATH & 3) == 1, false, true>(E.g, E.wn, E.T, pts, n, c, s, tx, ty, nullptr);
        else
          cost = eval_pose_wave<MODE, (PATH & 3) == 1>(E.g, E.wn, E.T, pts, n, c, s, tx, ty);
      } else {
        cost = eval_pose_wave<MODE, (PATH & 3) == 1>(E.g, E.wn, E.T, pts, n, c, s, tx, ty);
      }
    }
#ifdef NDTPSO_VERIFY_MARGIN
    if constexpr (ARB && (PATH == 2 || PATH == 3))
      if (E.xa) verify_item<PATH == 3>(E.xa, j, cost, gbc);
#endif
    last_done = j;
    if (lane_id() == 0) {  // (as eval_items, with an improver to look for)
      sw.tcost[j] = cost;
      bool ordinary = true;
      if (MODE == kScoreF32 && !(cost <= -kTinyCost)) {  // NaN, or the underflow regime (exact mode: NaN only, see eval_items)
        if (cost != cost || (!ARB && pbc_j > -kTinyCost)) {
          // (tiny: PsoShared::tiny_j here.  Exact mode: only a NaN gets here, and whether the fp64 kernel takes the alignment
          // over or this one carries on changes nothing in what is returned -- the item's index need not be kept)
          if constexpr (ARB)
            *(lds_int_t)tiny = 0;
          else
            __hip_atomic_fetch_min((lds_int_t)tiny, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ordinary = false;
        }
      }
      if constexpr (ARB) {
        // The usual outcome first: worse than its pbest AND than the gbest by more than the margin -- then it is neither an
        // improver nor a near tie, and the two differences and one comparison that say so are all the lane spends (the
        // nested tests of core.cpp:94-104 and the two margin tests were eight fp64 compares per evaluation).
        const double tau = arb_margin(gbc, n);
        const double d1 = cost - pbc_j, d2 = cost - gbc;
        if (ordinary && !(fmin(d1, d2) > tau)) {
          if (cost < gbc)
            if (cost < pbc_j) __hip_atomic_fetch_min((lds_int_t)improver, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // (a flag is all the phase needs: pso_run_wg reads the list off the stored costs)
          if (fabs(d1) <= tau || fabs(d2) <= tau) *(lds_int_t)near_cnt = 1;
        }
      } else {
        if (ordinary && cost < gbc)
          if (cost < pbc_j) __hip_atomic_fetch_min((lds_int_t)improver, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (lane_id() == 0 && last_done >= 0) __hip_atomic_fetch_max((lds_int_t)jmax, last_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- one alignment on several compute units ("cluster") --------------------------------------------------------
//
// A lone alignment (the live node, BASELINE config 2) is bounded by the VALU of the one CU its workgroup runs on.
// K workgroups can share it: every one of them holds the table, the points and the WHOLE swarm and runs the
// identical control flow (proposals, commits, replays -- deterministic, so the K copies never diverge); only the
// cost evaluations of a round are divided, one item per wave across all K x waves waves.  After evaluating, every
// wave publishes its cost in a tagged 16-byte slot (xc[round parity][item], xslot_store) and wave 0 of every workgroup
// reads the round's slots until all carry the round's tag (eval_round); first-improver / underflow detection then runs
// locally on identical data.  Per-item arithmetic is the single-workgroup kernel's, so the results are bit-identical
// to it.  The wait is bounded by the real-time counter: a workgroup that never arrives (cluster not co-resident) raises
// kStatusClusterTimeout and the host reruns the alignment on one workgroup.
// NDTPSO_PROFILE_PHASES (diagnostic builds only): thread 0 of rank 0 accumulates the real-time-counter ticks between
// the marks of a cluster round and prints them at the end: [other -> 0] control, [0 -> 1] evaluation, [1 -> 2]
// exchange, [2 -> 3] read-back and detection
#ifdef NDTPSO_PROFILE_PHASES
__device__ unsigned long long g_phase_ticks[4];
__device__ unsigned long long g_phase_last;
#define NDTPSO_PHASE_MARK(k)                                                            \
  do {                                                                                  \
    if (threadIdx.x == 0 && cl.rank == 0) {                                             \
      const unsigned long long now__ = wall_clock64();                                  \
      g_phase_ticks[k] += now__ - g_phase_last;                                         \
      g_phase_last = now__;                                                             \
    }                                                                                   \
  } while (0)
#else
#define NDTPSO_PHASE_MARK(k) do { } while (0)
#endif

// NDTPSO_PHASE_BUDGET (diagnostic builds only, scripts/phase_budget.py): every workgroup of the fused pairs kernel
// accounts for its own time, phase by phase, on the 100 MHz real-time counter -- contiguous marks, so that the phases
// of a workgroup sum to its duration -- and leaves 16 words in g_budget[blockIdx.x]:
//   0 setup (scan A, window, table, scan B)   1 swarm initialisation   2 generator at the top of an iteration
//   3 proposals (+ barrier)   4 wave 0's own evaluations   5 wave 0's slice of the generator   6 wave 0's wait at the
//   round's barrier   7 arbitration   8 commits (+ gbest barriers)   9 end of iteration   10 final fp64 cost
//   12 / 13 / 14 the same as 4 / 5 / 6 seen from wave 1 (a wave with two items per round)   15 the whole workgroup
#ifdef NDTPSO_PHASE_BUDGET
constexpr unsigned kBudgetMaxBlocks = 8192;
__device__ unsigned g_budget[kBudgetMaxBlocks * 16];
__device__ unsigned g_polls;  // a cluster's first workgroup: sweeps of the exchange's slots by its polling wave (entry 15)
#define NDTPSO_PB_DECL                          \
  unsigned long long pb_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
  unsigned long long pb_last = wall_clock64()
#define NDTPSO_PB(k)                                   \
  do {                                                 \
    const unsigned long long now__ = wall_clock64();   \
    pb_t[k] += now__ - pb_last;                        \
    pb_last = now__;                                   \
  } while (0)
#else
#define NDTPSO_PB_DECL do { } while (0)
#define NDTPSO_PB(k) do { } while (0)
#endif

constexpr int kClusterNoHeartbeat = 2;  // ClusterP::flags
constexpr int kClusterRedoShift = 8;    // ClusterP::flags >> 8: the status flags a redo launch on clusters serves (ndtpso_pairs_body.inc)
struct ClusterP {
  int K, rank, stride;  // workgroups, this one's index, slots per exchange buffer
  int absent;           // test hooks: the workgroup of this rank leaves at once (-1: nobody), exercising the timeout; -(2 + r): rank r
                        // dawdles in the rounds in which it has no item (eval_round's flow control)
  uint4* xc;            // [2][stride] exchange slots {cost lo, tag, cost hi, tag}
  uint32_t nonce;       // of this launch, in the tags of its slots (which keep whatever earlier launches left there)
  // A cluster on ONE XCD.  Workgroups go to the eight XCDs in turn (workgroup i to XCD i mod 8: HW_REG_XCC_ID read back by
  // scripts/ubench_xcd_exchange.hip), and the same 16-byte `sc1` store and loads cost 0.50 us per exchange between eight
  // workgroups of one XCD against 0.93 us between eight consecutive ones (1.27 -> 0.76 us for sixteen): the slots are served
  // by the XCD's own L2 instead of the fabric.  So cluster c of a launch is the workgroups whose index is c mod 8 modulo 8
  // -- rank r of cluster c is workgroup ((c / 8) K + r) 8 + c mod 8 -- and a lone cluster launches 8 K workgroups of which
  // seven in eight leave on their first instruction.  Placement only: the exchange is correct wherever the workgroups land.
  int one_xcd;          // 1 + p: that mapping, cluster 0 on XCD p; 0: K consecutive workgroups per cluster (NDTPSO_CLUSTER_SPREAD=1, for comparison)
  int n;                // clusters in this launch
  int spec_off;         // LDS byte offset of the speculation scratch (16 (P + 1) doubles, SpecP), -1: none
  double* spec;         // ... as a pointer (set by the kernel; what the HOST puts here is a redo launch's list of pairs, or null)
  int flags;            // kClusterNoHeartbeat: the exchange does not wait for heartbeats (NDTPSO_CLUSTER_HEARTBEAT=0: tests, comparison)
};

// ---- a cluster's next proposals, made while its costs travel ---------------------------------------------------------
// A round of a cluster ends with the exchange: wave 0 of every workgroup reads slots for about a microsecond and the other
// waves wait for it.  When the round is the iteration's only one (every particle in it) and nobody improves the gbest --
// five rounds in six of the live sequence -- what follows is fixed but for ONE bit per particle: the commit of every
// proposal (core.cpp:94-96) and the next iteration's proposals against the unchanged gbest (core.cpp:83-90), where
// "pbest - position" is zero if the particle has just found a new pbest and "old pbest - new position" if not.  So the
// idle waves make BOTH next proposals of every coordinate during the exchange (same expressions, same operation order:
// the values are the proposal step's bit for bit), and after the round's barrier a short step commits and picks one by the
// comparison the reference makes.  A gbest move, an iteration in several rounds or draws that are not there yet (the device
// generator makes them during the same exchange) leave everything to the usual commit and proposal steps.
struct SpecP {
  double* buf;             // [2][3][S] velocity, [2][3][S] position, [2][2][S] cos / sin of the heading; candidate 0: new pbest
  const int32_t* draws;    // the next iteration's draws (6 per particle), readable now; nullptr: no speculation this round
  double w, c1, c2;        // the next iteration's inertia weight, c1, c2
  const double* gb;        // gbest position
};
__device__ __forceinline__ double* spec_vel(double* buf, int S, int c, int k) { return buf + (c * 3 + k) * S; }
__device__ __forceinline__ double* spec_pos(double* buf, int S, int c, int k) { return buf + (6 + c * 3 + k) * S; }
__device__ __forceinline__ double* spec_cs(double* buf, int S, int c, int which) { return buf + (12 + c * 2 + which) * S; }
// (cluster, rank) of this workgroup; false: it has no part in the launch
__device__ __forceinline__ bool cluster_place(const ClusterP& cl, size_t* c, int* rank) {
  if (cl.one_xcd) {
    // (one_xcd - 1: the XCD the launch's first cluster takes -- a lone cluster of context i sits on XCD i mod 8, so that
    // replicas of the live sequence in one process do not all queue for XCD 0's thirty-two compute units)
    const unsigned x = (blockIdx.x - (unsigned)(cl.one_xcd - 1)) & 7u, q = blockIdx.x >> 3;
    *rank = (int)(q % (unsigned)cl.K);
    *c = (size_t)(q / (unsigned)cl.K) * 8u + x;
  } else {
    *rank = (int)(blockIdx.x % (unsigned)cl.K);
    *c = blockIdx.x / (unsigned)cl.K;
  }
  return *c < (size_t)cl.n;
}
__host__ inline unsigned cluster_grid(const ClusterP& cl) {
  return cl.one_xcd ? (unsigned)((cl.n + 7) / 8) * 8u * (unsigned)cl.K : (unsigned)cl.n * (unsigned)cl.K;
}

// One exchanged cost: a 16-byte slot written and read whole, agent scope (sc1: through to memory, past the L2 of the
// reader's XCD), with the round's tag in two of its words -- a reader takes a slot only when both tags are the round's,
// so a value is never paired with a stale or half-written neighbour.  No fence on either side: nothing but the slot itself
// is communicated.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// The tag is 64 bits in two words that BOTH depend on the launch's 32-bit nonce and the round's 32-bit number (round 5; it
// was a 16-bit nonce and a 16-bit round number, the same word twice: at 2 800 scans a second the nonce came round every
// 24 s, and a slot a launch with a wider layout had left behind could in principle be taken for a current one).  Each half
// of the slot carries one of them next to its half of the cost, so a slot torn between two stores would still not pass.
__device__ __forceinline__ uint32_t xslot_tag_a(uint32_t nonce, uint32_t round) { return nonce ^ (round * 0x9E3779B9u); }
__device__ __forceinline__ uint32_t xslot_tag_b(uint32_t nonce, uint32_t round) { return nonce + round * 0x85EBCA6Bu + 0x27D4EB2Fu; }
__device__ __forceinline__ void xslot_store(uint4* slot, double cost, uint32_t tag_a, uint32_t tag_b) {
  u32x4 v;
  v.x = (uint32_t)__double2loint(cost);
  v.y = tag_a;
  v.z = (uint32_t)__double2hiint(cost);
  v.w = tag_b;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 xslot_load(const uint4* slot) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
  return v;
}
constexpr uint32_t kStatusClusterTimeout = 16u;
constexpr uint32_t kStatusLateOverflow = 32u;  // late window binding: the table outgrew what the host had planned for
constexpr unsigned long long kClusterWaitTicks = 2000000ull;  // 20 ms of the 100 MHz counter (an exchange takes 1.5 us; round 1 waited 0.2 s)

// Evaluates items [first, last) of the swarm; on return (after the caller's barrier) sw.tcost holds their costs and
// *improver / *tiny are set as eval_items sets them.  `epoch` counts the cluster's exchanges.
template <int MODE, int PATH, bool CLUSTER, bool ARB = false, bool NOCLIP = false, bool KGEN = false, bool BOX = false>
__device__ inline void eval_round(const EvalCtx& E, const double2* pts, int n, const Swarm& sw, int S, int first, int last,
                                  double gbc, int* improver, int* tiny, const ClusterP& cl, unsigned& epoch,
                                  int* timed_out, int* near_cnt, unsigned short* near_list,
                                  RngState* gen_st = nullptr, int* gen_t = nullptr, int32_t* gen_dst = nullptr,
                                  int gen_cnt = 0, int gen_wave = -1, const SpecP* sp = nullptr) {
  if constexpr (!CLUSTER) {
    eval_items<MODE, PATH, ARB, NOCLIP, KGEN, BOX>(E, pts, n, sw, S, first, last, gbc, improver, tiny, near_cnt, near_list);
  } else {
    NDTPSO_PHASE_MARK(0);
    const int n_waves = blockDim.x >> 6, total_waves = cl.K * n_waves;
    uint4* buf = cl.xc + (size_t)(epoch & 1u) * cl.stride;
    uint4* hb = cl.xc + 2 * (size_t)cl.stride;  // heartbeats, one slot per workgroup of the cluster (below)
    const uint32_t tag_a = xslot_tag_a(cl.nonce, epoch + 1u), tag_b = xslot_tag_b(cl.nonce, epoch + 1u);
    if (cl.absent <= -2 && cl.rank == -(cl.absent + 2) && first + cl.rank * n_waves >= last) {
      // test hook (NDTPSO_CLUSTER_TEST_LAG=r): the workgroup of rank r dawdles 60 us in every round in which it has no item
      const unsigned long long t_lag = wall_clock64();
      while (wall_clock64() - t_lag < 6000ull) __builtin_amdgcn_s_sleep(8);
    }
    for (int j = first + cl.rank * n_waves + wave_id(); j < last; j += total_waves) {
      const double c = sw.it[4 * j], s = sw.it[4 * j + 1];
      const double tx = sw.tpos[j], ty = sw.tpos[S + j];
      double cost;
      if constexpr (path_is_dense(PATH))  // (a cluster folds the constants here: its proposal step is on the critical path)
        cost = eval_pose_wave_dense<false, true, PATH == 3>(E.g, E.dn, E.lds0, pts, n, c, s, tx, ty, nullptr);
      else
        cost = eval_pose_wave<MODE, (PATH & 3) == 1>(E.g, E.wn, E.T, pts, n, c, s, tx, ty);
      if (lane_id() == 0) xslot_store(&buf[j], cost, tag_a, tag_b);
    }
    NDTPSO_PHASE_MARK(1);
    // behind the exchange (wave 0 polls, everybody else would idle): one wave of every workgroup draws the next
    // iteration's rand() numbers -- 3.4 us that used to stand alone at the iteration's start
    if (gen_cnt > 0 && wave_id() == gen_wave) rng_fill_wave0(gen_st, gen_t, gen_dst, gen_cnt);
    // ... and the waves that do not poll make the next iteration's proposals, both ways (SpecP)
    if (sp && sp->draws && wave_id() != 0) {
      const int P = S - 1, nt = (int)blockDim.x - kWave;
      for (int t = (int)threadIdx.x - kWave; t < 6 * P; t += nt) {
        const int c = t & 1, jk = t >> 1;  // headings first, as in the proposal step
        const int j = jk < P ? jk : (jk - P) >> 1, k = jk < P ? 2 : ((jk - P) & 1);
        const double r1 = fabs(uniform_pm1(sp->draws[6 * j + 2 * k]));
        const double r2 = fabs(uniform_pm1(sp->draws[6 * j + 2 * k + 1]));
        const double p = sw.tpos[k * S + j];  // the position the commit will make current
        const double pbk = c == 0 ? p : sw.pb[k * S + j];  // pbest: the same position (new pbest) or the old one
        const double v = sp->w * sw.tvel[k * S + j] + sp->c1 * r1 * (pbk - p) + sp->c2 * r2 * (sp->gb[k] - p);
        const double np = p + v;
        spec_vel(sp->buf, S, c, k)[j] = v;
        spec_pos(sp->buf, S, c, k)[j] = np;
        if (k == 2) {
          double sn, cn;
          sincos(np, &sn, &cn);
          spec_cs(sp->buf, S, c, 0)[j] = cn;
          spec_cs(sp->buf, S, c, 1)[j] = sn;
        }
      }
    }
    // Wave 0 of every workgroup reads the round's slots until all of them carry the round's tag (its own workgroup's
    // among them), a lane per item, and does the round's detection on what it read.  One trip of the cost to memory and
    // one back: the first scheme -- costs, a fence, an arrival counter, a poll, a fence, the costs read back -- was four,
    // with the L2 written back and invalidated twice per round (3.4 us a round of the live sequence's 60).
    // Flow control (round 5).  The two slot buffers alternate, so round r + 1 overwrites what round r - 1 left; a workgroup
    // with items in round r has read round r - 1 by then, but one WITHOUT items in rounds r and r + 1 (two partial rounds in a
    // row: the tails behind two gbest moves) is waited for by nobody and could still be reading -- it would find its slots
    // retagged and sit out the bounded wait.  So every workgroup also keeps a heartbeat slot {rounds it has read, launch
    // nonce}, written behind its read of a round, and the read of round r waits, in the SAME sweep as the costs (the lanes
    // behind the items: no extra trip), until the heartbeat of every workgroup that has no item in round r says r: everybody
    // has read round r - 1 before anybody stores round r + 1.
    if (wave_id() == 0) {
      const unsigned long long t0 = wall_clock64();
      // (a workgroup WITH items in this round has read the round before: its slots of this round say so.  Only the ranks
      // without items -- none in a full round, five rounds in six of the live sequence -- are asked for their heartbeat.)
      const int n_it = last - first, hb0 = min(cl.K, (n_it + n_waves - 1) / n_waves);
      const int n_hb = (epoch > 0u && !(cl.flags & kClusterNoHeartbeat)) ? cl.K - hb0 : 0;
      for (int v0 = 0; v0 < n_it + n_hb; v0 += kWave) {
        const int vi = v0 + lane_id(), j = first + vi;
        const bool mine = vi < n_it, mine_hb = !mine && vi < n_it + n_hb;
        double cost = 0.;
        for (;;) {
          bool ok = true;
          if (mine) {
            const u32x4 v = xslot_load(&buf[j]);
            ok = v.y == tag_a && v.w == tag_b;
            cost = __hiloint2double((int)v.z, (int)v.x);
          } else if (mine_hb) {
            const u32x4 v = xslot_load(&hb[hb0 + vi - n_it]);
            ok = v.y == cl.nonce && v.w == (cl.nonce ^ 0x5bd1e995u) && v.z == ~v.x && v.x >= epoch;
          }
#ifdef NDTPSO_PHASE_BUDGET
          if (cl.rank == 0 && lane_id() == 0) atomicAdd(&g_polls, 1u);
#endif
          if (__all(ok)) break;
          // (a redo launch's clusters -- a few flagged pairs behind a batch, ndtpso_pairs_body.inc -- have the gated launches
          // behind them and give up after an eighth of the wait: 2.5 ms, what the pair costs on one workgroup)
          if (wall_clock64() - t0 > (((unsigned)cl.flags >> kClusterRedoShift) ? kClusterWaitTicks / 8 : kClusterWaitTicks)) {
            *timed_out = 1;
            break;
          }
        }
        if (mine) {
          sw.tcost[j] = cost;
          if (MODE == kScoreF32 && (cost != cost || (!ARB && cost > -kTinyCost && (!improver || sw.pbc[j] > -kTinyCost))))
            *tiny = 1;  // (exact mode: NaN only, see eval_items)
          else if (improver && cost < gbc && cost < sw.pbc[j])  // nested tests of core.cpp:94-104, see eval_items
            atomicMin(improver, j);
          if constexpr (ARB) {  // exact mode, as in eval_items: every workgroup of the cluster notes the same items
            if (improver) {
              const double tau = arb_margin(gbc, n);
              if (near_tie(cost, sw.pbc[j], tau) || near_tie(cost, gbc, tau)) near_note(near_cnt, near_list, j);
            }
          }
        }
      }
      if (lane_id() == 0) {  // this workgroup has read round `epoch`: epoch + 1 rounds in all
        u32x4 v;
        v.x = epoch + 1u;
        v.y = cl.nonce;
        v.z = ~(epoch + 1u);
        v.w = cl.nonce ^ 0x5bd1e995u;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&hb[cl.rank]), "v"(v) : "memory");
      }
    }
    NDTPSO_PHASE_MARK(2);
    ++epoch;
    NDTPSO_PHASE_MARK(3);
  }
}

// returns false when the alignment was abandoned for the fp64-score kernel (fp32 underflow regime)
// ARB: the exact mode (NDTPSO_SCORE_EXACT) of the fp32-score dense kernels.  A template parameter, not a run-time
// switch: the arbitration code in the same kernel cost the plain fp32 mode 11 % (register pressure: spills in the
// proposal / commit paths), see DESIGN.md.
template <int MODE, int PATH, bool CLUSTER = false, bool ARB = false, bool NOCLIP = false, bool KGEN = false, bool UNITS = false, bool BOX = false, bool PAIR = false>
__device__ inline bool pso_run_wg(const EvalCtx& E,
                                  const double2* pts, int n, const PsoP& ps, const double* guess,
                                  const double* dev, uint32_t seed, const int32_t* table, const Swarm& sw,
                                  PsoShared* sh, double* out_pose, double* out_cost, AlignStats* stats,
                                  const ClusterP& cl = ClusterP{1, 0, 0, -1, nullptr, 0u, 0, 1, -1, nullptr, 0}) {
  unsigned epoch = 0;
  constexpr bool kStream = NDTPSO_STREAM && !CLUSTER;  // phases dealt by ticket (eval_stream) instead of rounds
  const bool writer = !CLUSTER || cl.rank == 0;  // the workgroup that reports the result
  const int tid = threadIdx.x;
  const int P = ps.P, S = P + 1;
  const bool gen = (table == nullptr);
  int rng_t = 64;
  uint32_t n_evals = 0, n_rounds = 0, n_gb = 0, n_arb = 0;
#ifdef NDTPSO_PROFILE_ARB
  uint32_t arb_ticks = 0;  // diagnostic builds: 100 MHz ticks spent arbitrating, reported in place of `rounds`
#endif
  static_assert(!ARB || (MODE == kScoreF32 && path_is_dense(PATH)), "the exact mode runs on the fp32-score dense kernels");

  NDTPSO_PB_DECL;
  // ---- swarm initialisation: core.cpp:58-69 ----
  if (tid == 0) {
    sh->tiny = sh->timed_out = 0;
    sh->k_w = ps.w;
    sh->k_c1 = ps.c1;
    sh->k_c2 = ps.c2;
    sh->k_hw = E.g.hw;
    sh->k_hh = E.g.hh;
    sh->k_inv = E.g.inv_cs;
    sh->k_ox = (double)E.dn.ox;
    sh->k_oy = (double)E.dn.oy;
  }
  if constexpr (ARB) {
    // exact mode: the rest of exact_tasks' parameter block (the kernel has stored the grid, the window and the table
    // views of the fp64 image into sh->xa already)
    if (tid == 0) {
      ExactArgs& a = sh->xa;
      a.pts_lds = (unsigned)(uintptr_t)(const double2 __attribute__((address_space(3)))*)pts;  // generic -> LDS address
      a.n = n;
      a.tpos = sw.tpos;
      a.pb = sw.pb;
      a.tcost = sw.tcost;
      a.pbc = sw.pbc;
      a.gb = sh->gb;
      a.pcs = sw.pcs;
      a.bcs = sw.bcs;
      a.gcs = sh->gcs;
      a.xgbc = &sh->xgbc;
      a.S = S;
      a.pex = sw.pex;
      a.gex = 0;
    }
  }
  // The device replay of glibc's generator is the work of ONE wave: wave 0 when it is the light wave of the evaluation
  // rounds (PsoP::light), else the last wave -- the one a round with fewer items than waves leaves idle.
  const int rng_w = (ps.light && !CLUSTER) ? 0 : (int)(blockDim.x >> 6) - 1;  // (a cluster's wave 0 holds the exchange's spinning thread)
  if (gen && wave_id() == rng_w) {
    rng_seed_wave0(&sh->rng, seed);
    rng_fill_wave0(&sh->rng, &rng_t, sw.raw, 3 * S);
  }
  __syncthreads();
  {
    const int32_t* draws = gen ? sw.raw : table;
    for (int t = tid; t < S; t += blockDim.x) {
      // draw order: the guess particle first (core.cpp:58), then particles 0..P-1 (core.cpp:60-61)
      const int slot = (t == 0) ? P : (t - 1);
      double th = 0.;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double dk = (t == 0) ? ((k == 2) ? 1E-5 : 1E-4) : dev[k];  // zero_devi, core.cpp:53
        const double v = guess[k] + (uniform_pm1(draws[3 * t + k]) * dk);  // core.cpp:14
        sw.tpos[k * S + slot] = v;
        if (k == 2) th = v;
      }
      double sn, cn;
      sincos(th, &sn, &cn);
      if constexpr (ARB) {
        sw.pcs[slot] = cn;
        sw.pcs[S + slot] = sn;
      }
      if constexpr (path_is_dense(PATH) && !CLUSTER) {  // DenseItem of this pose (dense_item), kept with it
        const double inv = sh->k_inv;
        const double TX = (sw.tpos[slot] + sh->k_hw) * inv - sh->k_ox, TY = (sw.tpos[S + slot] + sh->k_hh) * inv - sh->k_oy;
        sw.it[4 * slot] = cn * inv;
        sw.it[4 * slot + 1] = sn * inv;
        sw.it[4 * slot + 2] = TX;
        sw.it[4 * slot + 3] = TY;
        // (NaN translations fail the tests and take the clamped loop)
        if constexpr (BOX) {  // four bytes per particle: TX, TY, heading inside the guard, and a 1 -- read as one word
          *reinterpret_cast<unsigned*>(sw.tgd + 4 * slot) = ((TX >= sh->guard.x_lo && TX < sh->guard.x_hi) ? 1u : 0u) |
                                                            ((TY >= sh->guard.y_lo && TY < sh->guard.y_hi) ? 0x100u : 0u) |
                                                            ((th >= sh->g_t_lo && th <= sh->g_t_hi) ? 0x10000u : 0u) | 0x1000000u;
        } else {
          sw.tgd[2 * slot] = (TX >= sh->guard.x_lo && TX < sh->guard.x_hi) ? 1 : 0;
          sw.tgd[2 * slot + 1] = (TY >= sh->guard.y_lo && TY < sh->guard.y_hi) ? 1 : 0;
        }
      } else {
        sw.it[4 * slot] = cn;
        sw.it[4 * slot + 1] = sn;
        if constexpr (MODE == kScoreF64 && !CLUSTER) {  // (the fp64 score's guard, in metres: score_trip_guarded)
          const double px = sw.tpos[slot], py = sw.tpos[S + slot];
          if constexpr (BOX) {  // (as the dense form's: x, y, heading, 1 -- one word per particle)
            *reinterpret_cast<unsigned*>(sw.tgd + 4 * slot) = ((px >= sh->guard.x_lo && px < sh->guard.x_hi) ? 1u : 0u) |
                                                              ((py >= sh->guard.y_lo && py < sh->guard.y_hi) ? 0x100u : 0u) |
                                                              ((th >= sh->g_t_lo && th <= sh->g_t_hi) ? 0x10000u : 0u) | 0x1000000u;
          } else {
            sw.tgd[2 * slot] = (px >= sh->guard.x_lo && px < sh->guard.x_hi) ? 1 : 0;
            sw.tgd[2 * slot + 1] = (py >= sh->guard.y_lo && py < sh->guard.y_hi) ? 1 : 0;
          }
        }
      }
    }
  }
  __syncthreads();
  eval_round<MODE, PATH, CLUSTER, ARB, NOCLIP, KGEN, BOX>(E, pts, n, sw, S, 0, S, 0., nullptr, &sh->tiny, cl, epoch, &sh->timed_out, nullptr,
                                  nullptr);
  n_evals += S;
  n_rounds += 1;
  __syncthreads();
  if (CLUSTER && sh->timed_out) {
    if (tid == 0 && stats && writer) stats->status |= kStatusClusterTimeout;
    return false;
  }
  if (MODE == kScoreF32 && sh->tiny) {  // underflow regime: give up, the fp64-score kernel redoes this alignment
    if (tid == 0 && stats && writer) stats->status |= kStatusNeedsF64 | NDTPSO_WHY(0x100u);
    return false;
  }
  if constexpr (ARB) {
    {
      // exact mode, initial gbest (core.cpp:60-69: the first strict minimum in the order guess particle, 0 .. P-1):
      // only an item within the margin of the smallest fp32 cost can be the fp64 minimum; if there are several, their
      // fp64 scores decide (the selection loop below then runs on those)
      if (tid == 0) {
        double m = sw.tcost[P];
        for (int i = 0; i < P; ++i) m = fmin(m, sw.tcost[i]);
        sh->xgbc = m;
        sh->near_cnt[0] = 0;
      }
      __syncthreads();
      const double lim = sh->xgbc + arb_margin(sh->xgbc, n);
      for (int i = tid; i < S; i += blockDim.x)
        if (sw.tcost[i] <= lim) near_note(&sh->near_cnt[0], sh->near_list[0], i);
      __syncthreads();
      const int cnt = sh->near_cnt[0];
      if (__builtin_expect(cnt > kMaxNear, 0)) {  // a swarm of near-identical costs: the fp64-score kernel takes the whole alignment
        if (tid == 0 && stats && writer) stats->status |= kStatusNeedsF64 | NDTPSO_WHY(0x200u);
        return false;
      }
      if (__builtin_expect(cnt > 1, 0)) {  // (cold: the register allocator must not charge the evaluation loops for it)
        exact_tasks_wg<PATH == 3, CLUSTER, UNITS>(&sh->xa, sh->near_list[0], cnt, 0);
        n_arb += (uint32_t)cnt;
      }
    }
  }
  if (tid == 0) {
    double gbc = sw.tcost[P];
    int best = P;
    for (int i = 0; i < P; ++i)
      if (sw.tcost[i] < gbc) {  // core.cpp:63
        gbc = sw.tcost[i];
        best = i;
      }
    sh->gbc = gbc;
    for (int k = 0; k < 3; ++k) sh->gb[k] = sw.tpos[k * S + best];
    if constexpr (ARB) {
      sh->gcs[0] = sw.pcs[best];
      sh->gcs[1] = sw.pcs[S + best];
    }
  }
  for (int j = tid; j < P; j += blockDim.x) {
    for (int k = 0; k < 3; ++k) {
      const double v = sw.tpos[k * S + j];
      sw.pos[k * S + j] = v;
      sw.pb[k * S + j] = v;
      sw.vel[k * S + j] = 0.;
    }
    sw.pbc[j] = sw.tcost[j];
    if constexpr (ARB) {
      sw.bcs[j] = sw.pcs[j];
      sw.bcs[S + j] = sw.pcs[S + j];
      sw.pex[j] = 0;  // (some of these costs are fp64 scores already -- the initial gbest's candidates; not tracked)
    }
  }
  __syncthreads();

  // ---- iterations: core.cpp:78-109 ----
  unsigned grp = 0;
  [[maybe_unused]] bool pre_proposed = false;  // a cluster: the coming iteration's proposals are in place already (SpecP)
  if (tid == 0) {
    sh->jstar[0] = sh->jstar[1] = sh->jstar[2] = P;
    sh->near_cnt[0] = sh->near_cnt[1] = sh->near_cnt[2] = 0;
  }
  // rand() table from the host (the live node): the draws of an iteration are fetched from HBM TWO iterations ahead --
  // the loads are issued at the top of iteration it - 2, sit in two registers per thread while it runs, and land at its
  // end in the LDS buffer of that iteration's parity (the two buffers the device generator would otherwise fill) -- so
  // the proposal step, where most waves wait for one or two, starts from LDS instead of paying an HBM round trip every
  // time, and the iteration before can already make this one's proposals (SpecP) from LDS
  // (a cluster's kernels only -- the live node's: two more registers live through every iteration are two more spilled
  // in the batch kernels)
  const bool prefetch = CLUSTER && !gen && ps.I > 0 && 6 * P <= 2 * (int)blockDim.x && sw.raw2 != nullptr;
  int32_t pre0 = 0, pre1 = 0;
  if (prefetch) {
    const int32_t* first = table + 3 * S;
    for (int q = tid; q < 6 * P; q += blockDim.x) {
      sw.raw[q] = first[q];
      if (ps.I > 1) sw.raw2[q] = first[6 * P + q];
    }
  }
  __syncthreads();
#ifdef NDTPSO_PROFILE_PSO
  unsigned long long pt[5] = {0, 0, 0, 0, 0}, plast = wall_clock64();
#define NDTPSO_PSO_MARK(k)                                  \
  do {                                                      \
    const unsigned long long now__ = wall_clock64();        \
    pt[k] += now__ - plast;                                 \
    plast = now__;                                          \
  } while (0)
#else
#define NDTPSO_PSO_MARK(k) do { } while (0)
#endif
  // Device generator: the 6P draws of an iteration took one wave 3.4 us (70 particles; 74 us for 2048) while all the
  // others waited at the barrier behind it -- 235 us of a 2.2 ms alignment, 15 ms of a 179 ms one
  // (-DNDTPSO_PROFILE_PSO).  Now the NEXT iteration's draws are generated while the current one is evaluated, into the
  // other of two buffers (replays re-read the current iteration's draws): by the light wave (PsoP::light), a slice per
  // evaluation round, in the time its second item would have taken.  Together with the commits that wave no longer
  // delays: config 5 1418 -> 1503 align/s, 512 pairs of 30 x 50 554 -> 594 k align/s, 256 pairs of 256 x 70 49.5 ->
  // 55.2 k, config 3 +0.5-1 %.  Without a light wave (NDTPSO_LIGHT_WAVE=0) the last wave draws all of them during the
  // iteration's last round when that round has fewer items than the workgroup has waves (70 particles in rounds of 16:
  // 6 items for 8 waves; that alone took config 3 from 212 to 219 k align/s).  What is still missing when the
  // iteration ends is drawn then.  (A cluster draws at the start of the iteration.)
  int next_filled = 0;  // draws of the next iteration already in dnext (the same in every thread)
  const bool overlapped = gen && sw.raw2 != nullptr, sliced = overlapped && !CLUSTER && ps.light;
  const int n_draw = 6 * P;
  const int slice = 30 * max(1, (n_draw + 30 * ((P + ps.G - 1) / ps.G) - 1) / (30 * ((P + ps.G - 1) / ps.G)));
  NDTPSO_PB(1);
  for (int it = 0; it < ps.I; ++it) {
    NDTPSO_PSO_MARK(4);
    NDTPSO_PB(9);
    // (the two draw buffers by the iteration's parity, not as a pair of pointers swapped every iteration: those lived in
    // scratch memory in the arbitrating kernel)
    const bool odd = overlapped && (it & 1);
    int32_t* const dcur = odd ? sw.raw2 : sw.raw;
    int32_t* const dnext = odd ? sw.raw : sw.raw2;
    if (gen) {
      if (overlapped && it > 0) {  // what the previous iteration's rounds left time for, and the rest now
        if (next_filled < n_draw) {
          if (wave_id() == rng_w) rng_fill_wave0(&sh->rng, &rng_t, dcur + next_filled, n_draw - next_filled);
          __syncthreads();
        }
        next_filled = 0;
      } else {
        if (wave_id() == rng_w) rng_fill_wave0(&sh->rng, &rng_t, dcur, n_draw);
        __syncthreads();
      }
    }
    NDTPSO_PSO_MARK(0);
    NDTPSO_PB(2);
    int32_t* const pbuf = (it & 1) ? sw.raw2 : sw.raw;  // (prefetch) this iteration's draws; refilled at its end for it + 2
    const int32_t* draws = gen ? dcur : (prefetch ? pbuf : (table + 3 * S + (size_t)it * 6 * P));
    if (prefetch && it + 2 < ps.I) {
      const int32_t* next = table + 3 * S + (size_t)(it + 2) * 6 * P;
      if (tid < 6 * P) pre0 = next[tid];
      if (tid + (int)blockDim.x < 6 * P) pre1 = next[tid + blockDim.x];
    }
    // Particles are evaluated in index order, G at a time.  Every not-yet-committed particle carries a
    // proposal made against the gbest that was current when it was (re)proposed; when a group contains the
    // first improver j*, particles up to j* are committed and everything after it is re-proposed -- so only
    // the tail of one group (< G evaluations) is ever thrown away per gbest update.
    //
    // Synchronisation: ONE workgroup barrier per group in the common case.  The evaluating wave records a
    // gbest improver itself (atomicMin into jstar[group % 3]; slot (g+2)%3 is reset after group g's barrier,
    // two barriers ahead of its next use).  Committing a group touches only pos/vel/pbest of that group,
    // which no evaluation reads, so waves run on into the next group without waiting for it.
    int lo = 0;
    bool need_propose = !pre_proposed;
    pre_proposed = false;
    while (lo < P) {
      [[maybe_unused]] const bool proposed_now = need_propose;
#if NDTPSO_STREAM
      if constexpr (kStream)
        if (tid == 0) {  // the phase's ticket counter (published by the barrier that follows)
          sh->ticket = lo;
          sh->jmax = lo - 1;
          sh->tiny_j = 0x7fffffff;
        }
#endif
      if (need_propose) {
        // core.cpp:83-90 for every particle not yet committed, against the current gbest
        // one thread per (particle, coordinate): the three coordinates of a particle are independent (core.cpp:83-90),
        // and each costs two fp64 divisions (Eigen's Random()) -- a chain three times shorter than one thread per
        // particle; the heading lanes then take the sine and cosine
        // (the headings first, then the x / y pairs: the sine and cosine -- two thirds of this step's instructions -- are
        // then issued by the waves that hold heading lanes only, two of the four that 70 particles occupy, instead of by
        // every wave because every third lane needed them)
        const int n_prop = P - lo;
        for (int q = tid; q < 3 * n_prop; q += blockDim.x) {
          const int j = lo + (q < n_prop ? q : (q - n_prop) >> 1), k = q < n_prop ? 2 : ((q - n_prop) & 1);
          const double r1 = fabs(uniform_pm1(draws[6 * j + 2 * k]));
          const double r2 = fabs(uniform_pm1(draws[6 * j + 2 * k + 1]));
          const double p = sw.pos[k * S + j];
          const double v = sh->k_w * sw.vel[k * S + j] + sh->k_c1 * r1 * (sw.pb[k * S + j] - p) + sh->k_c2 * r2 * (sh->gb[k] - p);
          const double np = p + v;
          sw.tvel[k * S + j] = v;
          sw.tpos[k * S + j] = np;
          constexpr bool fold = path_is_dense(PATH) && !CLUSTER;
          if constexpr (!fold && MODE == kScoreF64 && !CLUSTER) {  // (the fp64 score's guard, in metres)
            if (k == 0) sw.tgd[(BOX ? 4 : 2) * j] = (np >= sh->guard.x_lo && np < sh->guard.x_hi) ? 1 : 0;
            if (k == 1) sw.tgd[(BOX ? 4 : 2) * j + 1] = (np >= sh->guard.y_lo && np < sh->guard.y_hi) ? 1 : 0;
            if constexpr (BOX)
              if (k == 2) sw.tgd[4 * j + 2] = (np >= sh->g_t_lo && np <= sh->g_t_hi) ? 1 : 0;
          }
          if constexpr (fold) {  // each coordinate's share of the proposal's DenseItem (dense_item)
            if (k == 0) {
              const double TX = (np + sh->k_hw) * sh->k_inv - sh->k_ox;
              sw.it[4 * j + 2] = TX;
              sw.tgd[(BOX ? 4 : 2) * j] = (TX >= sh->guard.x_lo && TX < sh->guard.x_hi) ? 1 : 0;
            }
            if (k == 1) {
              const double TY = (np + sh->k_hh) * sh->k_inv - sh->k_oy;
              sw.it[4 * j + 3] = TY;
              sw.tgd[(BOX ? 4 : 2) * j + 1] = (TY >= sh->guard.y_lo && TY < sh->guard.y_hi) ? 1 : 0;
            }
          }
          if (k == 2) {
            double sn, cn;
            sincos(np, &sn, &cn);
            sw.it[4 * j] = fold ? cn * sh->k_inv : cn;
            sw.it[4 * j + 1] = fold ? sn * sh->k_inv : sn;
            if constexpr (fold && BOX) sw.tgd[4 * j + 2] = (np >= sh->g_t_lo && np <= sh->g_t_hi) ? 1 : 0;
            if constexpr (ARB) {
              sw.pcs[j] = cn;
              sw.pcs[S + j] = sn;
            }
          }
        }
        need_propose = false;
        __syncthreads();  // proposals (and the commits before them) visible to every wave
        NDTPSO_PSO_MARK(1);
        NDTPSO_PB(3);
      }
      int slot = (int)(grp % 3u), hi_g = P;
      [[maybe_unused]] bool spec_made = false;  // this round made the next iteration's proposals both ways (SpecP)
#if NDTPSO_STREAM
      if constexpr (kStream) {
        // ---- a phase: every particle not yet committed, dealt by ticket (eval_stream) ----
        // (ticket / jmax were set by thread 0 at the top of this trip; jstar[slot] = P and near_cnt[slot] = 0 two phases ago)
        if (!proposed_now) __syncthreads();  // (the proposal step's barrier publishes them otherwise)
        const double gbc_phase = sh->gbc;
        // the wave that replays glibc's generator draws the next iteration's numbers first, then joins the others: that is
        // about one item's time (3.4 us for 70 particles), and the ticket counter gives it one item less for it
        if (overlapped && it + 1 < ps.I && next_filled < n_draw) {
          if (wave_id() == rng_w) rng_fill_wave0(&sh->rng, &rng_t, dnext + next_filled, n_draw - next_filled);
          next_filled = n_draw;
        }
        NDTPSO_PB(5);
        eval_stream<MODE, PATH, ARB, NOCLIP, BOX, PAIR>(E, pts, n, sw, S, P, gbc_phase, &sh->ticket, sh->spare, &sh->jstar[slot], &sh->jmax, &sh->tiny_j,
                                             &sh->near_cnt[slot], sh->near_list[slot]);
        NDTPSO_PB(4);
        __syncthreads();
        NDTPSO_PSO_MARK(2);
        NDTPSO_PB(6);
        // (an item behind the phase's first improver does not count: whether it was evaluated at all depends on timing, and
        // it is proposed again against the new gbest -- the alignment's fate must not hang on it)
        if (MODE == kScoreF32 && sh->tiny_j <= min(sh->jstar[slot], P - 1)) {
          if (tid == 0 && stats && writer) stats->status |= kStatusNeedsF64 | NDTPSO_WHY(0x400u);
          return false;
        }
        n_evals += (uint32_t)(sh->jmax - lo + 1);
        n_rounds += 1;
        // items [lo, hi_g) are what the phase has established: everything up to the first improver (all of them evaluated,
        // see eval_stream); behind it lie items evaluated for nothing, or not at all
        hi_g = min(sh->jstar[slot], P - 1) + 1;
        if constexpr (ARB) {
          // near_cnt[slot] != 0: some evaluated item lies within the margin of its pbest's or the gbest's cost (the
          // evaluating waves only raise the flag: which items they were is read off the stored costs, in [lo, hi_g), so
          // that the list does not depend on who evaluated what when)
          if (__builtin_expect(sh->near_cnt[slot] != 0, 0)) {  // cold
            __syncthreads();  // (everybody has read the flag)
            if (tid == 0) sh->near_cnt[slot] = 0;
            __syncthreads();
            {
              const double gbc0 = sh->gbc, tau = arb_margin(gbc0, n);
              for (int j = lo + tid; j < hi_g; j += blockDim.x) {
                const double cj = sw.tcost[j], pj = sw.pbc[j];
                if (near_tie(cj, pj, tau) || near_tie(cj, gbc0, tau)) near_note(&sh->near_cnt[slot], sh->near_list[slot], j);
              }
            }
            __syncthreads();
            const int cnt = sh->near_cnt[slot];
            if (__builtin_expect(cnt > kMaxNear, 0)) {  // (a converged swarm: nearly every comparison is a near-tie)
              if (tid == 0 && stats && writer) stats->status |= kStatusNeedsF64 | NDTPSO_WHY(0x800u);
              return false;
            }
            if (cnt != 0) {
              exact_tasks_wg<PATH == 3, CLUSTER, UNITS>(&sh->xa, sh->near_list[slot], cnt, 1);
              if (tid == 0) {
                if (sh->xa.gb_task) sh->gbc = sh->xgbc;  // (else it is the fp64 score of the gbest position already)
                sh->jstar[slot] = P;  // (stays P: no improver among the established items -- the phase goes on behind them)
              }
              __syncthreads();
              {  // the first improver again, by every thread (a phase of a 2048-particle swarm kept in HBM is 2048 items:
                 // one thread walking them cost 180 us per arbitration, 10 ms of a config-5 alignment)
                const double gbc1 = sh->gbc;
                for (int j = lo + tid; j < hi_g; j += blockDim.x) {
                  const double cj = sw.tcost[j];
                  if (cj < gbc1 && cj < sw.pbc[j]) atomicMin(&sh->jstar[slot], j);
                }
              }
              __syncthreads();
              n_arb += (uint32_t)cnt;
            }
          }
        }
      } else
#endif
      {
#if NDTPSO_ALTERNATE_PRIO
        // Two workgroups share a CU.  VALU issue is arbitrated by priority, then age, so the earlier-dispatched
        // partner otherwise starves the other, finishes ~25 % early and leaves the CU half empty (measured with
        // per-workgroup timestamps: residency 0.84 -> 0.95, +6 % throughput with this).  The partners (blocks b and
        // b + grid/2 by dispatch order -- an assumption that only affects speed) take turns holding the higher
        // priority; the turn comes from the shared 100 MHz real-time counter (5 us slices), so the two are
        // complementary at all times, and the later-dispatched one gets 9 of 16 slices, which is what equalises
        // their finishing times.
        if constexpr (!CLUSTER) {
          if ((((unsigned)(wall_clock64() >> 9) & 15u) < (unsigned)NDTPSO_PRIO_SHARE) == (blockIdx.x >= (gridDim.x >> 1)))
            __builtin_amdgcn_s_setprio(1);
          else
            __builtin_amdgcn_s_setprio(0);
        }
#endif
        slot = (int)(grp % 3u);
        hi_g = min(lo + ps.G, P);
        // a cluster draws the next iteration's numbers behind the first exchange of this one (eval_round)
        const bool gen_here = CLUSTER && overlapped && it + 1 < ps.I && next_filled < n_draw;
        // a cluster whose round is the whole iteration, with the next draws in its table: both next proposals of every
        // coordinate are made during the exchange (SpecP)
        [[maybe_unused]] SpecP spec{nullptr, nullptr, 0., 0., 0., nullptr};
        if constexpr (CLUSTER) {
          if (cl.spec && !gen && lo == 0 && hi_g == P && it + 1 < ps.I && blockDim.x > (unsigned)kWave) {
            // (the next iteration's draws: in the other LDS buffer when the table is prefetched, else where the table lies)
            spec = SpecP{cl.spec, prefetch ? ((it & 1) ? sw.raw : sw.raw2) : table + 3 * S + (size_t)(it + 1) * 6 * P,
                         sh->k_w * ps.wdamp, sh->k_c1, sh->k_c2, sh->gb};
            spec_made = true;
          }
        }
        eval_round<MODE, PATH, CLUSTER, ARB, NOCLIP, KGEN, BOX>(E, pts, n, sw, S, lo, hi_g, sh->gbc, &sh->jstar[slot], &sh->tiny, cl, epoch,
                                        &sh->timed_out, &sh->near_cnt[slot], sh->near_list[slot], &sh->rng, &rng_t,
                                        dnext + next_filled, gen_here ? n_draw - next_filled : 0, rng_w, CLUSTER ? &spec : nullptr);
        if (gen_here) next_filled = n_draw;
        NDTPSO_PB(4);
        if constexpr (!CLUSTER) {
          // the light wave's other job: a slice of the next iteration's draws (published by the barriers that follow)
          if (sliced && it + 1 < ps.I && next_filled < n_draw) {
            const int cnt = min(slice, n_draw - next_filled);
            if (wave_id() == rng_w) rng_fill_wave0(&sh->rng, &rng_t, dnext + next_filled, cnt);
            next_filled += cnt;
          } else if (overlapped && !ps.light && it + 1 < ps.I && next_filled == 0 && hi_g == P && hi_g - lo <= rng_w) {
            if (wave_id() == rng_w) rng_fill_wave0(&sh->rng, &rng_t, dnext, n_draw);  // (rng_w had no item in this round)
            next_filled = n_draw;
          }
        }
        n_evals += (uint32_t)(hi_g - lo);
        n_rounds += 1;
        NDTPSO_PB(5);
        __syncthreads();
        NDTPSO_PSO_MARK(2);
        NDTPSO_PB(6);
        if (CLUSTER && sh->timed_out) {
          if (tid == 0 && stats && writer) stats->status |= kStatusClusterTimeout;
          return false;
        }
        if (MODE == kScoreF32 && sh->tiny) {
          if (tid == 0 && stats && writer) stats->status |= kStatusNeedsF64 | NDTPSO_WHY(0x400u);
          return false;
        }
        if constexpr (ARB) {
          {
            const int cnt = sh->near_cnt[slot];  // uniform: written before the barrier above
            if (__builtin_expect(cnt != 0, 0)) {  // cold, see above
#ifdef NDTPSO_PROFILE_ARB
              const unsigned long long arb_t0 = wall_clock64();
#endif
              if (__builtin_expect(cnt > kMaxNear, 0)) {  // (a converged swarm: nearly every comparison is a near-tie)
                if (tid == 0 && stats && writer) stats->status |= kStatusNeedsF64 | NDTPSO_WHY(0x800u);
                return false;
              }
              // fp64 scores of the undecidable items' proposals and pbest positions and of the gbest position replace
              // the stored costs; the group's first improver is then looked for again (core.cpp:94-104, nested tests)
              exact_tasks_wg<PATH == 3, CLUSTER, UNITS>(&sh->xa, sh->near_list[slot], cnt, 1);
              if (tid == 0) {
                if (sh->xa.gb_task) sh->gbc = sh->xgbc;  // (else it is the fp64 score of the gbest position already)
                int first = P;
                for (int j = lo; j < hi_g; ++j) {
                  const double cj = sw.tcost[j];
                  if (cj < sh->gbc && cj < sw.pbc[j]) {
                    first = j;
                    break;
                  }
                }
                sh->jstar[slot] = first;
              }
              __syncthreads();
              n_arb += (uint32_t)cnt;
#ifdef NDTPSO_PROFILE_ARB
              n_rounds += 1000000u * 0u;
              arb_ticks += (uint32_t)(wall_clock64() - arb_t0);
#endif
            }
          }
        }
      }
      NDTPSO_PB(7);
      const int js = sh->jstar[slot];
      if (tid == 0) {
        sh->jstar[(grp + 2u) % 3u] = P;
        sh->near_cnt[(grp + 2u) % 3u] = 0;
      }
      ++grp;
      if constexpr (CLUSTER) {
        if (spec_made && js >= P) {
          // No gbest move in a round that was the whole iteration: commit (core.cpp:94-96) and take, per particle, the one
          // of the two ready-made next proposals its comparison selects (SpecP).  One thread per (particle, coordinate);
          // the pbest COST is written behind the barrier below, by the heading's thread -- lane j of wave 0: the particle's
          // other threads compare against the old one now, and the next reader is wave 0 itself (eval_round).
          double keep = 0.;
          bool take = false;
          for (int q = tid; q < 3 * P; q += blockDim.x) {
            const int j = q < P ? q : (q - P) >> 1, k = q < P ? 2 : ((q - P) & 1);
            // (both ready-made proposals are read with everything else -- one trip to LDS -- and one is kept)
            const double cst = sw.tcost[j], pbj = sw.pbc[j];
            const double np = sw.tpos[k * S + j], nv = sw.tvel[k * S + j];
            const double vA = spec_vel(cl.spec, S, 0, k)[j], vB = spec_vel(cl.spec, S, 1, k)[j];
            const double pA = spec_pos(cl.spec, S, 0, k)[j], pB = spec_pos(cl.spec, S, 1, k)[j];
            [[maybe_unused]] double hc = 0., hs = 0., cn = 0., sn = 0.;
            const bool better = cst < pbj;  // core.cpp:94
            if (k == 2) {
              const double cA = spec_cs(cl.spec, S, 0, 0)[j], cB = spec_cs(cl.spec, S, 1, 0)[j];
              const double sA = spec_cs(cl.spec, S, 0, 1)[j], sB = spec_cs(cl.spec, S, 1, 1)[j];
              cn = better ? cA : cB;
              sn = better ? sA : sB;
              if constexpr (ARB) {
                hc = sw.pcs[j];
                hs = sw.pcs[S + j];
              }
              keep = cst;  // (j == q == tid here: swarms of up to 64 particles, cluster_spec_room -- a lane of wave 0)
              take = better;
            }
            const double v2 = better ? vA : vB, p2 = better ? pA : pB;
            sw.pos[k * S + j] = np;
            sw.vel[k * S + j] = nv;
            if (better) sw.pb[k * S + j] = np;
            sw.tvel[k * S + j] = v2;
            sw.tpos[k * S + j] = p2;
            if (k == 2) {
              sw.it[4 * j] = cn;
              sw.it[4 * j + 1] = sn;
              if constexpr (ARB) {
                if (better) {
                  sw.bcs[j] = hc;
                  sw.bcs[S + j] = hs;
                  bool exact_j = false;
                  const int arb_n = sh->near_cnt[slot];
                  for (int t = 0; t < arb_n; ++t) exact_j |= (int)sh->near_list[slot][t] == j;
                  sw.pex[j] = exact_j ? 1 : 0;
                }
                sw.pcs[j] = cn;
                sw.pcs[S + j] = sn;
              }
            }
          }
          if (prefetch && it + 2 < ps.I) {  // this iteration's buffer is free: the draws of the one after the next
            if (tid < 6 * P) pbuf[tid] = pre0;
            if (tid + (int)blockDim.x < 6 * P) pbuf[tid + blockDim.x] = pre1;
          }
          if (tid == 0) sh->k_w *= ps.wdamp;  // core.cpp:108
          __syncthreads();
          if (take) sw.pbc[tid] = keep;
          NDTPSO_PB(11);
          pre_proposed = true;  // (the iteration is over: its end below is skipped as well)
          lo = P;
          continue;
        }
      }
      const int last = (js < hi_g) ? js : (hi_g - 1);  // (js >= hi_g: P, no improver)
      // exact mode: items of this round whose costs the arbitration replaced by fp64 scores (what a pbest / the gbest
      // that takes such a cost over inherits: Swarm::pex, ExactArgs::gex)
      [[maybe_unused]] const int arb_cnt = ARB ? sh->near_cnt[slot] : 0;
      [[maybe_unused]] const unsigned short* arb_list = sh->near_list[slot];
      for (int j = lo + tid; j <= last; j += blockDim.x) {
        const double cst = sw.tcost[j];
        const bool better = cst < sw.pbc[j];  // core.cpp:94
#ifdef NDTPSO_COUNT_AMBIG  // diagnostic builds: comparisons closer than a relative NDTPSO_COUNT_AMBIG (one-workgroup kernels)
        if (fabs(cst - sw.pbc[j]) <= NDTPSO_COUNT_AMBIG * fabs(sw.pbc[j]) ||
            fabs(cst - sh->gbc) <= NDTPSO_COUNT_AMBIG * fabs(sh->gbc))
          atomicAdd(&sh->timed_out, 1);
#endif
        // (everything read before anything is written: the swarm's arrays may alias as far as the compiler knows, and a
        // load placed behind a store waited for its own data before the next pair could start -- seven round trips to
        // the swarm's memory in a row where one does)
        double np[3], nv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          np[k] = sw.tpos[k * S + j];
          nv[k] = sw.tvel[k * S + j];
        }
        [[maybe_unused]] double hc = 0., hs = 0.;
        if constexpr (ARB) {
          hc = sw.pcs[j];
          hs = sw.pcs[S + j];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          sw.pos[k * S + j] = np[k];
          sw.vel[k * S + j] = nv[k];
          if (better) sw.pb[k * S + j] = np[k];
        }
        if (better) sw.pbc[j] = cst;
        if constexpr (ARB) {
          if (better) {
            sw.bcs[j] = hc;
            sw.bcs[S + j] = hs;
            bool exact_j = false;
            for (int q = 0; q < arb_cnt; ++q) exact_j |= (int)arb_list[q] == j;
            sw.pex[j] = exact_j ? 1 : 0;
          }
        }
      }
      if (js < P) {
        // gbest moves (core.cpp:97-104): every thread has read the old gbc above (eval_items), so one more
        // barrier orders those reads before the update, and the re-proposal barrier publishes it
        __syncthreads();
        if (tid == 0) {
          sh->gbc = sw.tcost[js];
          NDTPSO_ARB_TRACE(-2000);
          NDTPSO_ARB_TRACE(it);
          NDTPSO_ARB_TRACE(js);
          NDTPSO_ARB_TRACE(sh->gbc);
          for (int k = 0; k < 3; ++k) sh->gb[k] = sw.tpos[k * S + js];
          if constexpr (ARB) {
            sh->gcs[0] = sw.pcs[js];
            sh->gcs[1] = sw.pcs[S + js];
            bool exact_js = false;
            for (int q = 0; q < arb_cnt; ++q) exact_js |= (int)arb_list[q] == js;
            sh->xa.gex = exact_js ? 1 : 0;
          }
        }
        __syncthreads();
        n_gb += 1;
        lo = js + 1;
        need_propose = true;
      } else {
        lo = hi_g;  // (a phase whose arbitration found no improver among the established items goes on behind them)
      }
      NDTPSO_PSO_MARK(3);
      NDTPSO_PB(8);
    }
    NDTPSO_PB(5);
    if (pre_proposed) continue;  // (a cluster's round that committed and proposed in one step has done all of this)
    if (prefetch && it + 2 < ps.I) {  // every proposal of this iteration has read its draws (barriers above)
      if (tid < 6 * P) pbuf[tid] = pre0;
      if (tid + (int)blockDim.x < 6 * P) pbuf[tid + blockDim.x] = pre1;
    }
    if (tid == 0) sh->k_w *= ps.wdamp;  // core.cpp:108 (every proposal of this iteration has read it: barriers above)
    __syncthreads();  // all commits of this iteration done before the next draws/proposals
  }

#ifdef NDTPSO_PROFILE_PSO
  if (tid == 0 && blockIdx.x == 0)
    printf("pso phases (us): rng %.1f propose %.1f eval+barrier %.1f commit %.1f other %.1f  rounds %u\n", pt[0] * 0.01,
           pt[1] * 0.01, pt[2] * 0.01, pt[3] * 0.01, pt[4] * 0.01, n_rounds);
#endif
#ifdef NDTPSO_PROFILE_PHASES
  if (CLUSTER && tid == 0 && cl.rank == 0) {
    printf("cluster phases (us): control %.1f eval %.1f exchange %.1f readback %.1f  rounds %u\n", g_phase_ticks[0] * 0.01,
           g_phase_ticks[1] * 0.01, g_phase_ticks[2] * 0.01, g_phase_ticks[3] * 0.01, n_rounds);
    g_phase_ticks[0] = g_phase_ticks[1] = g_phase_ticks[2] = g_phase_ticks[3] = 0;
  }
#endif
  NDTPSO_PB(9);
  bool exact_cost = false;
#ifndef NDTPSO_NO_FINAL_EXACT
  if constexpr (ARB) {
    if (out_cost) {  // exact mode: the returned cost is the fp64 score of the returned pose (what the fp64 mode holds);
      exact_tasks_wg<PATH == 3, CLUSTER, UNITS>(&sh->xa, nullptr, 0, 2);  // a caller that takes the pose only (NDTFrame::align) has
      exact_cost = true;                                           // passed no cost pointer and is spared the score
    }
  }
#endif
#ifdef NDTPSO_PHASE_BUDGET
  NDTPSO_PB(10);
  // (a cluster: its first workgroup's account in entry 0 -- scripts/cluster_budget.py; 11 = the commit-and-pick steps, SpecP)
  if ((!CLUSTER && blockIdx.x < kBudgetMaxBlocks) || (CLUSTER && cl.rank == 0)) {
    unsigned* o = g_budget + (CLUSTER ? (size_t)0 : (size_t)blockIdx.x * 16);
    if (tid == 0) {
      for (int k = 1; k <= 11; ++k) o[k] = (unsigned)pb_t[k];
      if (CLUSTER) {
        o[15] = g_polls;
        g_polls = 0;
      }
    }
    if (tid == 64) {
      o[12] = (unsigned)pb_t[4];
      o[13] = (unsigned)pb_t[5];
      o[14] = (unsigned)pb_t[6];
    }
  }
#endif
  if (tid == 0 && writer) {
    for (int k = 0; k < 3; ++k) out_pose[k] = sh->gb[k];  // core.cpp:115
    if (out_cost) *out_cost = (exact_cost && sh->xa.gb_task) ? sh->xgbc : sh->gbc;  // (gbc itself when it already is the fp64 score)
    if (stats) {
      stats->status |= (n_arb < 0xffffu ? n_arb : 0xffffu) << 16;  // exact mode: comparisons arbitrated in fp64
      stats->n_points = (uint32_t)n;
      stats->cost_evals = n_evals;
      stats->rounds = n_rounds;
#ifdef NDTPSO_PROFILE_ARB
      stats->rounds = arb_ticks;
#endif
      stats->gbest_updates = n_gb;
#if defined(NDTPSO_COUNT_AMBIG) || defined(NDTPSO_COUNT_NOCLAMP)
      if (!CLUSTER) stats->gbest_updates = (uint32_t)sh->timed_out;
#endif
    }
  }
  return true;
}

}  // namespace ndtpso
