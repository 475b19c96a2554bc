// NDTCell bookkeeping (see ndtcell.h).  The statistics themselves are computed on the GPU.
#include "ndtpso_slam/ndtcell.h"

#include <cmath>
#include <cstring>

#include "device.h"

namespace {
std::vector<Vector2d> g_no_points;  // what points_vector[slot] of a never-touched cell refers to
}

NDTCell::Window::Window() : current_count(0), global_count(0), current_window_id(0) {
  // Eigen's fixed-size vectors are NOT zero-initialised by their default constructor (the fallback types of linalg.h
  // are): the reference zeroes them explicitly (NDTCell::NDTCell, ndtcell.cpp:5-19), and WINDOW_ADD starts from them
  global_sum = Vector2d::Zero();
  for (Vector2d& v : partial_sums) v = Vector2d::Zero();
  std::memset(partial_covars, 0, sizeof(partial_covars));
  std::memset(partial_counts, 0, sizeof(partial_counts));
  std::memset(global_covar_sum, 0, sizeof(global_covar_sum));
  std::memset(inv_covar, 0, sizeof(inv_covar));
}

std::vector<Vector2d>& NDTCell::SlotPoints::operator[](std::size_t slot) {
  return owner_->ensure_window().points[slot];
}
const std::vector<Vector2d>& NDTCell::SlotPoints::operator[](std::size_t slot) const {
  return owner_->win_ ? owner_->win_->points[slot] : g_no_points;
}

NDTCell::NDTCell(bool) : points_vector(this), mean(Vector2d::Zero()) {}

NDTCell::NDTCell(const NDTCell& o) : points_vector(this), mean(o.mean), built(o.built), created(o.created) {
  if (o.win_) win_.reset(new Window(*o.win_));
}

NDTCell& NDTCell::operator=(const NDTCell& o) {
  if (this != &o) {
    mean = o.mean;
    built = o.built;
    created = o.created;
    win_.reset(o.win_ ? new Window(*o.win_) : nullptr);
  }
  return *this;
}

NDTCell::Window& NDTCell::ensure_window() {
  if (!win_) win_.reset(new Window());
  return *win_;
}

// reference: NDTCell::addPoint, lib/ndtpso_slam/ndtcell.cpp:21-34 (the running partial sum is recomputed
// from the slot's points on the device at build time, in the same order)
void NDTCell::addPoint(const Vector2d& point) {
  Window& w = ensure_window();
  if (w.current_count == 0) w.points[w.current_window_id].clear();
  w.current_count++;
  w.points[w.current_window_id].push_back(point);
  created = true;
  built = false;
}

// reference: NDTCell::build, ndtcell.cpp:36-68 -- one cell at a time; NDTFrame::build sends all cells at once
bool NDTCell::build() {
  Window& w = ensure_window();
  const std::size_t id = w.current_window_id;
  ndtpso_cell_window cw;
  std::memset(&cw, 0, sizeof(cw));
  cw.global_sum[0] = w.global_sum.x();
  cw.global_sum[1] = w.global_sum.y();
  cw.slot_sum[0] = w.partial_sums[id].x();
  cw.slot_sum[1] = w.partial_sums[id].y();
  for (int k = 0; k < 4; ++k) {
    cw.global_covar_sum[k] = w.global_covar_sum[k];
    cw.slot_covar[k] = w.partial_covars[id][k];
  }
  cw.global_count = w.global_count;
  cw.slot_count = w.partial_counts[id];
  cw.current_count = w.current_count;
  cw.built = built ? 1 : 0;
  const std::vector<Vector2d>& pts = w.points[id];
  std::vector<double> xy(2 * pts.size());
  for (std::size_t i = 0; i < pts.size(); ++i) {
    xy[2 * i] = pts[i].x();
    xy[2 * i + 1] = pts[i].y();
  }
  const uint32_t off[2] = {0u, (uint32_t)pts.size()};
  ndtpso_host::Use use(nullptr);  // (a cell on its own: the calling thread's context, unless a frame's is in use already)
  if (!ndtpso_host::check(ndtpso_cells_build_windowed(ndtpso_host::device(), 1, &cw, off, xy.data()), "NDTCell::build"))
    return built;  // device fault: the cell keeps its previous statistics
  w.global_sum = Vector2d(cw.global_sum[0], cw.global_sum[1]);
  w.partial_sums[id] = Vector2d(cw.slot_sum[0], cw.slot_sum[1]);
  for (int k = 0; k < 4; ++k) {
    w.global_covar_sum[k] = cw.global_covar_sum[k];
    w.partial_covars[id][k] = cw.slot_covar[k];
    w.inv_covar[k] = cw.icov[k];
  }
  w.global_count = cw.global_count;
  w.partial_counts[id] = cw.slot_count;
  if (cw.built) {
    mean = Vector2d(cw.mean[0], cw.mean[1]);
    built = true;
  }
  if (w.current_count > NDT_MAX_POINTS_PER_CELL) {  // ndtcell.cpp:61-65
    w.current_window_id = (w.current_window_id + 1) % NDT_WINDOW_SIZE;
    w.current_count = 0;
  }
  return built;
}

// reference: NDTCell::normalDistribution, ndtcell.cpp:70-78 -- evaluated by the device cost kernel on a
// one-cell table (a debugging convenience; the alignment path never calls this per point)
double NDTCell::normalDistribution(const Vector2d& point) {
  if (!built || !win_) return 0.;
  // a 1 x 1 grid centred on the point's cell is enough: any grid works as long as the point falls in its cell
  ndtpso_grid grid{2, 2, 2.0};
  const int32_t index = 0;
  const double m[2] = {0., 0.};
  const double p[2] = {point.x() - mean.x(), point.y() - mean.y()};
  if (!(std::fabs(p[0]) < 1.0 && std::fabs(p[1]) < 1.0)) {
    // far from the mean the Gaussian underflows anyway; scale the frame so the offset fits one cell
    const double r = std::fmax(std::fabs(p[0]), std::fabs(p[1]));
    if (r >= 32000.) return 0.;
    const unsigned short side = (unsigned short)(2 * (std::ceil(r) + 1));
    grid.width = grid.height = side;
    grid.cell_side = (double)side;
  }
  ndtpso_host::Use use(nullptr);
  ndtpso_host::table_owner() = nullptr;
  if (!ndtpso_host::check(ndtpso_ref_set_cells(ndtpso_host::device(), &grid, 1, &index, m, win_->inv_covar), "normalDistribution"))
    return 0.;
  const double pose[3] = {0., 0., 0.};
  double cost = 0.;
  if (!ndtpso_host::check(ndtpso_cost_batch(ndtpso_host::device(), p, 1, pose, 1, NDTPSO_SCORE_F64, &cost, nullptr), "normalDistribution"))
    return 0.;
  return -cost;
}

// reference: NDTCell::reset, ndtcell.cpp:80-91
void NDTCell::reset() {
  if (!win_) return;
  Window& w = *win_;
  w.global_sum = Vector2d::Zero();
  w.current_count = 0;
  w.global_count = 0;
  std::memset(w.global_covar_sum, 0, sizeof(w.global_covar_sum));
  w.current_window_id = 0;
  for (auto& v : w.points) v.clear();
}
