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
constexpr int kWave = 64;
constexpr int kImageHeaderBytes = 64;
constexpr int kRecBytes = 64;

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
  double w, c1, c2, wdamp;
};

// one reference cell as the score loop reads it (64 B, 16-B aligned pieces)
struct __attribute__((aligned(16))) Rec {
  double mx, my;      // NDTCell::mean
  double a, b, c, d;  // s_inv_covar (0,0),(0,1),(1,0),(1,1)
  float fa, fb, fd;   // -0.5*log2(e) * {a, b+c, d} rounded to fp32 (fp32 score path)
  uint32_t key;       // window-linear cell id
};
static_assert(sizeof(Rec) == kRecBytes, "record must be 64 bytes");

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
  uint32_t n_points, n_built, cost_evals, rounds, gbest_updates, status, reserved[2];
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }
__host__ __device__ inline int image_rec_offset(int n_words) { return kImageHeaderBytes + align16(n_words * 8); }
__host__ __device__ inline int image_bytes(int n_words, int rec_cap) {
  return image_rec_offset(n_words) + rec_cap * kRecBytes;
}

// ---- small device helpers ------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

// cell coordinates of an in-frame point, following NDTFrame::getCellIndex (ndtframe.cpp:240-249):
// floor((x + width/2.)/cell_side), floor((y + height/2.)/cell_side).  x + hw > 0 inside the
// strict bounds, so truncation equals floor.
__device__ __forceinline__ void cell_coords(const GridP& g, double qx, double qy, int& ix, int& iy) {
  const double ux = qx + g.hw, uy = qy + g.hh;
  if (g.cs_pow2) {
    ix = (int)(ux * g.inv_cs);
    iy = (int)(uy * g.inv_cs);
  } else {
    ix = (int)(ux / g.cs);
    iy = (int)(uy / g.cs);
  }
}

// ---- K1: NDT score of one candidate pose, one wave --------------------------
//
// cost_function (core.cpp:26-48) + transform_point (core.h:28-31) + getCellIndex
// (ndtframe.cpp:240-249) + NDTCell::normalDistribution (ndtcell.cpp:70-78), fused.
// Lanes stride the points (coalesced 16-B LDS reads); the cell index is a bitmap
// word + prefix popcount; the record gather is 2 (fp32 score) or 3 (fp64) ds_read_b128.
// Returns the cost (-sum) on every lane.
template <int MODE, bool DUMP>
__device__ __forceinline__ double eval_pose_wave(const GridP& g, const WinP& wn, const uint2* __restrict__ bm,
                                                 const Rec* __restrict__ rec, const double2* __restrict__ pts,
                                                 int n, double c, double s, double tx, double ty,
                                                 int32_t* __restrict__ dump) {
  const int lane = lane_id();
  double acc = 0.0;
  for (int base = 0; base < n; base += kWave) {
    const int i = base + lane;
    int tag = -1;
    if (i < n) {
      const double2 p = pts[i];
      double qx, qy;
      if (MODE == kScoreF64) {
        qx = (p.x * c - p.y * s) + tx;  // reference rounding, no fma
        qy = (p.x * s + p.y * c) + ty;
      } else {
        qx = fma(p.x, c, fma(-p.y, s, tx));
        qy = fma(p.x, s, fma(p.y, c, ty));
      }
      if (fabs(qx) < g.hw && fabs(qy) < g.hh) {
        int ix, iy;
        cell_coords(g, qx, qy, ix, iy);
        if (DUMP) tag = (iy < g.H) ? -2 : -1;
        const int lin_frame = ix + g.W * iy;
        if (__builtin_expect(ix == g.W, 0)) {  // fl(x + w/2) == w: the reference's linear index wraps to the next row
          ix = 0;
          iy += 1;
        }
        const unsigned rx = (unsigned)(ix - wn.x0), ry = (unsigned)(iy - wn.y0);
        if (rx < (unsigned)wn.w && ry < (unsigned)wn.h) {
          const unsigned lin = ry * (unsigned)wn.w + rx;
          const uint2 e = bm[lin >> 5];
          const unsigned bit = lin & 31u;
          if ((e.x >> bit) & 1u) {
            const unsigned slot = e.y + __popc(e.x & ((1u << bit) - 1u));
            const Rec* r = rec + slot;
            if (MODE == kScoreF64) {
              const double d0 = qx - r->mx, d1 = qy - r->my;
              const double r0 = d0 * r->a + d1 * r->c;  // (diff^T * inv_covar), ndtcell.cpp:73-75
              const double r1 = d0 * r->b + d1 * r->d;
              acc += exp(-(r0 * d0 + r1 * d1) / 2.);
            } else {
              const double2 m = *reinterpret_cast<const double2*>(&r->mx);
              const float4 f = *reinterpret_cast<const float4*>(&r->fa);
              const float d0 = (float)(qx - m.x), d1 = (float)(qy - m.y);
              const float q = fmaf(d0, fmaf(f.x, d0, f.y * d1), (f.z * d1) * d1);
              acc += (double)__builtin_amdgcn_exp2f(q);
            }
            if (DUMP) tag = lin_frame;
          }
        }
      }
      if (DUMP) dump[i] = tag;
    }
  }
  return -wave_sum(acc);
}

// ---- K3a: LaserScan -> points (NDTFrame::loadLaser, ndtframe.cpp:144-185) --------------------
//
// Workgroup-cooperative, order preserving.  ranges: global; out: LDS or global (generic).
// Returns the number of surviving points (uniform).  `s_cnt` is a >= 17-int LDS scratch.
__device__ inline int scan_to_points_wg(const float* __restrict__ ranges, const ScanP& sp, bool do_trans,
                                        double tc, double ts, double ttx, double tty, double2* out,
                                        int* s_cnt) {
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id(), n_waves = blockDim.x >> 6;
  int base = 0;
  for (int start = 0; start < sp.n_beams; start += blockDim.x) {
    const int i = start + tid;
    bool valid = false;
    double2 p = make_double2(0., 0.);
    if (i < sp.n_beams) {
      const float r = ranges[i];
      // ndtframe.cpp:165
      valid = ((double)r > 0.) && (r < sp.rmax) && (r > sp.eps);
      if (valid) {
        const float theta = (float)(unsigned)i * sp.ainc + sp.amin;  // index_to_angle, core.h:40-42 (fp32, no fma)
        double sn, cn;
        sincos((double)theta, &sn, &cn);
        p.x = (double)r * cn;  // laser_to_point, core.h:45-47
        p.y = (double)r * sn;
        if (do_trans) {  // transform_point by s_trans, ndtframe.cpp:175-176
          const double x = p.x * tc - p.y * ts + ttx;
          const double y = p.x * ts + p.y * tc + tty;
          p.x = x;
          p.y = y;
        }
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

__device__ __forceinline__ unsigned bm_slot(const uint2* bm, int k) {
  const uint2 e = bm[k >> 5];
  return e.y + __popc(e.x & ((1u << (k & 31)) - 1u));
}

// scratch: key[n], cellkey[n], cnt[n] ints and bm2[n_words] uint2
__device__ inline void build_table_wg(const GridP& g, const WinP& wn, const double2* pts, int n,
                                      unsigned char* image, int* key, int* cellkey, int* cnt, uint2* bm2,
                                      CellRow* rows, uint32_t* n_rows_out) {
  const int tid = threadIdx.x, nt = blockDim.x;
  ImageHeader* hdr = reinterpret_cast<ImageHeader*>(image);
  uint2* bm = reinterpret_cast<uint2*>(image + kImageHeaderBytes);
  Rec* rec = reinterpret_cast<Rec*>(image + image_rec_offset(wn.n_words));

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

  // 1. bin every point (NDTFrame::addPoint -> getCellIndex); mark created cells
  for (int i = tid; i < n; i += nt) {
    const double2 p = pts[i];
    int k = -1;
    if (fabs(p.x) < g.hw && fabs(p.y) < g.hh) {
      int ix, iy;
      cell_coords(g, p.x, p.y, ix, iy);
      if (ix == g.W) {  // reference linear-index wrap (see eval_pose_wave)
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
  __syncthreads();

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

  // 3. points per cell (integer atomics: order independent)
  for (int i = tid; i < n; i += nt) {
    const int k = key[i];
    if (k >= 0) atomicAdd(&cnt[bm_slot(bm2, k)], 1);
  }
  __syncthreads();

  // 4. built cells (count > 2, ndtcell.cpp:43) -> final record slots
  for (int s = tid; s < n_created; s += nt)
    if (cnt[s] > 2) {
      const int k = cellkey[s];
      atomicOr(&bm[k >> 5].x, 1u << (k & 31));
    }
  __syncthreads();
  if (wave_id() == 0) prefix_words_wave0(bm, wn.n_words, &hdr->n_built);
  __syncthreads();
  if (tid == 0 && (int)hdr->n_built > wn.rec_cap) atomicOr(&hdr->status, 2u);

  // 5. statistics: one owner thread per created cell; points are visited in beam order, as the
  //    reference's per-cell vectors are, so sums round identically.
  for (int s = tid; s < n_created; s += nt) {
    const int mykey = cellkey[s];
    const int c = cnt[s];
    const bool built = c > 2;
    double mx = 0., my = 0., ia = 0., ib = 0., ic = 0., id = 0.;
    if (built) {
      double sx = 0., sy = 0.;
      for (int i = 0; i < n; ++i) {
        if (key[i] == mykey) {
          const double2 p = pts[i];
          sx += p.x;  // s_current_partial_sum += point, ndtcell.cpp:30
          sy += p.y;
        }
      }
      mx = sx / (double)c;  // ndtcell.cpp:44
      my = sy / (double)c;
      double c00 = 0., c01 = 0., c10 = 0., c11 = 0.;
      for (int i = 0; i < n; ++i) {
        if (key[i] == mykey) {
          const double2 p = pts[i];
          const double d0 = p.x - mx, d1 = p.y - my;  // ndtcell.cpp:49-52
          c00 += d0 * d0;
          c01 += d0 * d1;
          c10 += d1 * d0;
          c11 += d1 * d1;
        }
      }
      // s_calc_covar_inverse, ndtcell.cpp:93-111 (eigenvalues of the 2x2 in closed form)
      const double nn = (double)c;
      c00 = c00 / nn;
      c01 = c01 / nn;
      c10 = c10 / nn;
      c11 = c11 / nn;
      const double hp = 0.5 * (c00 - c11);
      const double q = sqrt(hp * hp + c01 * c10);
      const double mid = 0.5 * (c00 + c11);
      const double e0 = mid + q, e1 = mid - q;
      const double large_val = (e0 > e1) ? e0 : e1;
      const double small_val = (e0 < e1) ? e0 : e1;
      double det;
      if (small_val < .001 * large_val)
        det = .001 * large_val * large_val;
      else
        det = c00 * c11 - c10 * c01;
      ia = c11 / det;
      ib = -c01 / det;
      ic = -c10 / det;
      id = c00 / det;
      const unsigned slot = bm_slot(bm, mykey);
      if ((int)slot < wn.rec_cap) {
        Rec r;
        r.mx = mx;
        r.my = my;
        r.a = ia;
        r.b = ib;
        r.c = ic;
        r.d = id;
        const double kf = -0.72134752044448170368;  // -0.5 * log2(e)
        r.fa = (float)(kf * ia);
        r.fb = (float)(kf * (ib + ic));
        r.fd = (float)(kf * id);
        r.key = (uint32_t)mykey;
        rec[slot] = r;
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
}

// ---- glibc rand() replay on the device ------------------------------------------------------
//
// srand(seed); rand() ... as glibc's TYPE_3 generator produces it: r[i] = r[i-31] + r[i-3]
// (mod 2^32), output r[i] >> 1, first output r[344].  Unrolling the recurrence ten times gives
// r[i] = r[i-30] + sum_{m<10} r[i-31-3m], which depends only on values >= 30 back: 30 lanes
// produce 30 consecutive values per step from a 64-entry LDS history ring.
struct RngState {
  volatile uint32_t hist[64];
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
  }
}

// produce r[t .. t+cnt), cnt <= 30; returns this lane's value (lane < cnt)
__device__ inline uint32_t rng_step_wave0(RngState* st, int t, int cnt) {
  const int lane = lane_id();
  uint32_t v = 0;
  if (lane < cnt) {
    const int i = t + lane;
    v = st->hist[(i - 30) & 63];
#pragma unroll
    for (int m = 0; m < 10; ++m) v += st->hist[(i - 31 - 3 * m) & 63];
  }
  if (lane < cnt) st->hist[(t + lane) & 63] = v;
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
__device__ __forceinline__ double uniform_pm1(int32_t raw) { return -1.0 + (2.0 * (double)raw) / 2147483647.0; }

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
  double* tc;     // [S] cos(theta) of the proposal
  double* ts;     // [S]
  double* tcost;  // [S]
  int32_t* raw;   // [max(3(P+1), 6P)] rand() outputs of the current phase
};
__host__ __device__ inline int swarm_doubles(int P) { return 19 * (P + 1); }
__host__ __device__ inline int swarm_raw_ints(int P) { return (6 * P > 3 * (P + 1)) ? 6 * P : 3 * (P + 1); }
__host__ __device__ inline int swarm_bytes(int P) { return align16(swarm_doubles(P) * 8) + align16(swarm_raw_ints(P) * 4); }

__device__ inline Swarm swarm_carve(unsigned char* base, int P) {
  const int S = P + 1;
  double* d = reinterpret_cast<double*>(base);
  Swarm sw;
  sw.pos = d;
  sw.vel = d + 3 * S;
  sw.pb = d + 6 * S;
  sw.pbc = d + 9 * S;
  sw.tpos = d + 10 * S;
  sw.tvel = d + 13 * S;
  sw.tc = d + 16 * S;
  sw.ts = d + 17 * S;
  sw.tcost = d + 18 * S;
  sw.raw = reinterpret_cast<int32_t*>(base + align16(swarm_doubles(P) * 8));
  return sw;
}

struct PsoShared {  // small control block in static LDS
  double gb[3];
  double gbc;
  int jstar;
  int pad;
  RngState rng;
};

template <int MODE>
__device__ inline void eval_items(const GridP& g, const WinP& wn, const uint2* bm, const Rec* rec,
                                  const double2* pts, int n, const Swarm& sw, int S, int first,
                                  int last /*exclusive*/) {
  const int n_waves = blockDim.x >> 6;
  for (int j = first + wave_id(); j < last; j += n_waves) {
    const double c = sw.tc[j], s = sw.ts[j];
    const double tx = sw.tpos[j], ty = sw.tpos[S + j];
    const double cost = eval_pose_wave<MODE, false>(g, wn, bm, rec, pts, n, c, s, tx, ty, nullptr);
    if (lane_id() == 0) sw.tcost[j] = cost;
  }
}

template <int MODE>
__device__ inline void pso_run_wg(const GridP& g, const WinP& wn, const uint2* bm, const Rec* rec,
                                  const double2* pts, int n, const PsoP& ps, const double* guess,
                                  const double* dev, uint32_t seed, const int32_t* table, const Swarm& sw,
                                  PsoShared* sh, double* out_pose, double* out_cost, AlignStats* stats) {
  const int tid = threadIdx.x;
  const int P = ps.P, S = P + 1;
  const bool gen = (table == nullptr);
  int rng_t = 64;
  uint32_t n_evals = 0, n_rounds = 0, n_gb = 0;

  // ---- swarm initialisation: core.cpp:58-69 ----
  if (gen && wave_id() == 0) {
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
      sw.tc[slot] = cn;
      sw.ts[slot] = sn;
    }
  }
  __syncthreads();
  eval_items<MODE>(g, wn, bm, rec, pts, n, sw, S, 0, S);
  n_evals += S;
  n_rounds += 1;
  __syncthreads();
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
  }
  for (int j = tid; j < P; j += blockDim.x) {
    for (int k = 0; k < 3; ++k) {
      const double v = sw.tpos[k * S + j];
      sw.pos[k * S + j] = v;
      sw.pb[k * S + j] = v;
      sw.vel[k * S + j] = 0.;
    }
    sw.pbc[j] = sw.tcost[j];
  }
  __syncthreads();

  // ---- iterations: core.cpp:78-109 ----
  double w = ps.w;
  for (int it = 0; it < ps.I; ++it) {
    if (gen) {
      if (wave_id() == 0) rng_fill_wave0(&sh->rng, &rng_t, sw.raw, 6 * P);
      __syncthreads();
    }
    const int32_t* draws = gen ? sw.raw : (table + 3 * S + (size_t)it * 6 * P);
    int lo = 0;
    while (lo < P) {
      // propose: core.cpp:83-90 for every particle not yet committed, against the current gbest
      for (int j = lo + tid; j < P; j += blockDim.x) {
        double th = 0.;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double r1 = fabs(uniform_pm1(draws[6 * j + 2 * k]));
          const double r2 = fabs(uniform_pm1(draws[6 * j + 2 * k + 1]));
          const double p = sw.pos[k * S + j];
          const double v = w * sw.vel[k * S + j] + ps.c1 * r1 * (sw.pb[k * S + j] - p) + ps.c2 * r2 * (sh->gb[k] - p);
          const double np = p + v;
          sw.tvel[k * S + j] = v;
          sw.tpos[k * S + j] = np;
          if (k == 2) th = np;
        }
        double sn, cn;
        sincos(th, &sn, &cn);
        sw.tc[j] = cn;
        sw.ts[j] = sn;
      }
      if (tid == 0) sh->jstar = P;
      __syncthreads();
      eval_items<MODE>(g, wn, bm, rec, pts, n, sw, S, lo, P);
      n_evals += (uint32_t)(P - lo);
      n_rounds += 1;
      __syncthreads();
      // first particle (index order) whose cost beats gbest: core.cpp:97-104 under single-thread order
      const double gbc = sh->gbc;
      for (int j = lo + tid; j < P; j += blockDim.x)
        if (sw.tcost[j] < gbc) atomicMin(&sh->jstar, j);
      __syncthreads();
      const int js = sh->jstar;
      const int hi = (js < P) ? js : (P - 1);
      for (int j = lo + tid; j <= hi; j += blockDim.x) {
        const double cst = sw.tcost[j];
        const bool better = cst < sw.pbc[j];  // core.cpp:94
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double np = sw.tpos[k * S + j];
          sw.pos[k * S + j] = np;
          sw.vel[k * S + j] = sw.tvel[k * S + j];
          if (better) sw.pb[k * S + j] = np;
        }
        if (better) sw.pbc[j] = cst;
        if (j == js) {
          sh->gbc = cst;
          for (int k = 0; k < 3; ++k) sh->gb[k] = sw.tpos[k * S + j];
        }
      }
      if (js < P) n_gb += 1;
      lo = js + 1;
      __syncthreads();
    }
    w *= ps.wdamp;  // core.cpp:108
  }

  if (tid == 0) {
    for (int k = 0; k < 3; ++k) out_pose[k] = sh->gb[k];  // core.cpp:115
    if (out_cost) *out_cost = sh->gbc;
    if (stats) {
      stats->n_points = (uint32_t)n;
      stats->cost_evals = n_evals;
      stats->rounds = n_rounds;
      stats->gbest_updates = n_gb;
    }
  }
}

}  // namespace ndtpso
