// NDTFrame of the MI355X build: grid bookkeeping on the host, arithmetic on the GPU (see ndtframe.h).
// Reference behaviour cited as lib/ndtpso_slam/ndtframe.cpp:LINE.
#include "ndtpso_slam/ndtframe.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device.h"
#include "ndtpso_slam/core.h"

namespace {
ndtpso_grid grid_of(const NDTFrame& f) { return ndtpso_grid{f.width, f.height, f.cell_side}; }

ndtpso_pso_config to_abi(const PSOConfig& c) {
  ndtpso_pso_config o;
  o.iterations = c.iterations;
  o.population = c.populationSize;
  o.num_threads = c.num_threads;
  o.w = c.coeff.w;
  o.c1 = c.coeff.c1;
  o.c2 = c.coeff.c2;
  o.w_damping = c.coeff.w_dumping;
  return o;
}
}  // namespace

// reference: constructor, ndtframe.cpp:19-66.  Cells are light (see ndtcell.h), so the dense vector is kept.
NDTFrame::NDTFrame(Vector3d trans, unsigned short width_, unsigned short height_, double cell_side_,
                   bool init_cell_windows, NDTPSOConfig config
#if BUILD_OCCUPANCY_GRID
                   ,
                   double occupancy_grid_cell_size
#endif
                   )
    : width(width_), height(height_), built(false), cell_side(cell_side_), s_trans(std::move(trans)),
      s_config(std::move(config)) {
  widthNumOfCells = uint16_t(std::ceil(width / cell_side));    // ndtframe.cpp:27
  heightNumOfCells = uint16_t(std::ceil(height / cell_side));  // ndtframe.cpp:28
  numOfCells = widthNumOfCells * heightNumOfCells;
  cells = vector<NDTCell>(numOfCells, NDTCell(init_cell_windows));
  s_x_min = -width / 2.;
  s_x_max = width / 2.;
  s_y_min = -height / 2.;
  s_y_max = height / 2.;
#if BUILD_OCCUPANCY_GRID
  s_og_cell_size = occupancy_grid_cell_size;  // map export only (out of scope); remembered, not rasterised
#endif
}

// points + their cells (as binned on the device) -> per-cell open slots; NDTFrame::addPoint's effect, ndtframe.cpp:215-235
void NDTFrame::append(const double* xy, const int32_t* idx, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (idx[i] < 0 || (unsigned)idx[i] >= numOfCells) continue;  // outside the frame: dropped (ndtframe.cpp:220)
    NDTCell& c = cells[(size_t)idx[i]];
    if (!c.created) s_created.push_back((uint32_t)idx[i]);
    c.addPoint(Vector2d(xy[2 * i], xy[2 * i + 1]));
    built = false;
  }
  s_table_dirty = true;
}

// reference: loadLaser, ndtframe.cpp:144-185 (beam filter, fp32 angle, fp64 polar->xy, s_trans, binning)
void NDTFrame::loadLaser(const vector<float>& laser_data, const float& min_angle, const float& angle_increment,
                         const float& max_range) {
  built = false;
  const uint32_t n = (uint32_t)laser_data.size();
  if (n == 0) return;
  ndtpso_ctx* dev = ndtpso_host::device();
  ndtpso_scan_geom geom{n, min_angle, angle_increment, max_range, s_config.laserIgnoreEpsilon};
  const double t[3] = {s_trans.x(), s_trans.y(), s_trans.z()};
  std::vector<double> xy(2 * (size_t)n);
  std::vector<int32_t> idx(n);
  uint32_t kept = 0;
  const ndtpso_grid grid = grid_of(*this);
  ndtpso_host::check(ndtpso_scan_to_cells(dev, laser_data.data(), &geom, t, &grid, xy.data(), idx.data(), &kept), "loadLaser");
  append(xy.data(), idx.data(), kept);
}

void NDTFrame::collectPoints(std::vector<double>& xy) const {
  xy.clear();
  if (numOfCells == 1) {  // the node's per-scan frame (ndtpso_slam_node.cpp:229-230)
    for (const Vector2d& p : cells[0].points_vector[0]) {
      xy.push_back(p.x());
      xy.push_back(p.y());
    }
    return;
  }
  for (const NDTCell& c : cells) {  // cell order, then insertion order (core.cpp:33-36)
    if (!c.created) continue;
    for (const Vector2d& p : c.points_vector[0]) {
      xy.push_back(p.x());
      xy.push_back(p.y());
    }
  }
}

// reference: update, ndtframe.cpp:187-198 (transform every slot-0 point of new_frame by `trans`, re-bin here)
void NDTFrame::update(Vector3d trans, NDTFrame* const new_frame) {
  built = false;
  std::vector<double> xy;
  new_frame->collectPoints(xy);
  const uint32_t n = (uint32_t)(xy.size() / 2);
  if (n == 0) return;
  std::vector<int32_t> idx(n);
  const double t[3] = {trans.x(), trans.y(), trans.z()};
  const ndtpso_grid grid = grid_of(*this);
  ndtpso_host::check(ndtpso_points_to_cells(ndtpso_host::device(), &grid, xy.data(), n, t, xy.data(), idx.data()), "update");
  append(xy.data(), idx.data(), n);
}

// reference: getCellIndex, ndtframe.cpp:240-249 (public single-point utility)
int NDTFrame::getCellIndex(Vector2d point, int grid_width, double cell_side_) {
  if ((point.x() > s_x_min) && (point.x() < s_x_max) && (point.y() > s_y_min) && (point.y() < s_y_max))
    return static_cast<int>(std::floor((point.x() + (width / 2.)) / cell_side_) +
                            grid_width * (std::floor((point.y() + (height / 2.)) / cell_side_)));
  return -1;
}

// reference: addPoint, ndtframe.cpp:215-235
void NDTFrame::addPoint(Vector2d& point) {
  const int32_t idx = getCellIndex(point, widthNumOfCells, cell_side);
  const double xy[2] = {point.x(), point.y()};
  append(xy, &idx, 1);
}

// reference: build, ndtframe.cpp:68-117 -- NDTCell::build for every created cell, batched into one device call
void NDTFrame::build() {
  const uint32_t n = (uint32_t)s_created.size();
  if (n) {
    std::vector<ndtpso_cell_window> cw(n);
    std::vector<uint32_t> off(n + 1, 0);
    std::vector<double> xy;
    for (uint32_t k = 0; k < n; ++k) {
      NDTCell& c = cells[s_created[k]];
      NDTCell::Window& w = c.ensure_window();
      const size_t id = w.current_window_id;
      ndtpso_cell_window& s = cw[k];
      std::memset(&s, 0, sizeof(s));
      s.global_sum[0] = w.global_sum.x();
      s.global_sum[1] = w.global_sum.y();
      s.slot_sum[0] = w.partial_sums[id].x();
      s.slot_sum[1] = w.partial_sums[id].y();
      for (int j = 0; j < 4; ++j) {
        s.global_covar_sum[j] = w.global_covar_sum[j];
        s.slot_covar[j] = w.partial_covars[id][j];
      }
      s.global_count = w.global_count;
      s.slot_count = w.partial_counts[id];
      s.current_count = w.current_count;
      s.built = c.built ? 1 : 0;
      for (const Vector2d& p : w.points[id]) {
        xy.push_back(p.x());
        xy.push_back(p.y());
      }
      off[k + 1] = (uint32_t)(xy.size() / 2);
    }
    ndtpso_host::check(ndtpso_cells_build_windowed(ndtpso_host::device(), n, cw.data(), off.data(), xy.data()), "build");
    for (uint32_t k = 0; k < n; ++k) {
      NDTCell& c = cells[s_created[k]];
      NDTCell::Window& w = *c.win_;
      const size_t id = w.current_window_id;
      const ndtpso_cell_window& s = cw[k];
      w.global_sum = Vector2d(s.global_sum[0], s.global_sum[1]);
      w.partial_sums[id] = Vector2d(s.slot_sum[0], s.slot_sum[1]);
      for (int j = 0; j < 4; ++j) {
        w.global_covar_sum[j] = s.global_covar_sum[j];
        w.partial_covars[id][j] = s.slot_covar[j];
      }
      w.global_count = s.global_count;
      w.partial_counts[id] = s.slot_count;
      if (s.global_count > 2) {  // ndtcell.cpp:43-58
        for (int j = 0; j < 4; ++j) w.inv_covar[j] = s.icov[j];
        c.mean = Vector2d(s.mean[0], s.mean[1]);
        c.built = true;
      }
      if (w.current_count > NDT_MAX_POINTS_PER_CELL) {  // ndtcell.cpp:61-65
        w.current_window_id = (w.current_window_id + 1) % NDT_WINDOW_SIZE;
        w.current_count = 0;
      }
    }
  }
  built = true;
  s_table_dirty = true;
}

// built cells -> device reference table (LDS image packed by ndtpso_ref_set_cells)
void NDTFrame::uploadTable() {
  if (!s_table_dirty && ndtpso_host::table_owner() == this) return;
  std::vector<int32_t> index;
  std::vector<double> mean, icov;
  for (uint32_t i : s_created) {
    const NDTCell& c = cells[i];
    if (!c.built) continue;
    index.push_back((int32_t)i);
    mean.push_back(c.mean.x());
    mean.push_back(c.mean.y());
    for (int j = 0; j < 4; ++j) icov.push_back(c.win_->inv_covar[j]);
  }
  const ndtpso_grid grid = grid_of(*this);
  ndtpso_host::check(ndtpso_ref_set_cells(ndtpso_host::device(), &grid, (uint32_t)index.size(), index.data(),
                                          mean.data(), icov.data()), "reference table upload");
  ndtpso_host::table_owner() = this;
  s_table_dirty = false;
}

// pso_optimization against this frame, core.cpp:50-116.  The std::rand() stream is drawn here, in the order and
// quantity the reference consumes it (3 + 3P + 6PI), so srand() by the caller has the reference's meaning.
Vector3d NDTFrame::optimize(const Vector3d& guess, const NDTFrame* new_frame, const Vector3d& deviation,
                            const PSOConfig& cfg) {
  if (!built) build();  // core.cpp:27-28 (lazy build inside cost_function)
  uploadTable();
  std::vector<double> xy;
  new_frame->collectPoints(xy);
  const ndtpso_pso_config abi = to_abi(cfg);
  std::vector<int32_t> draws(ndtpso_rand_draws(&abi));
  for (int32_t& d : draws) d = std::rand();
  const double g[3] = {guess.x(), guess.y(), guess.z()};
  const double dv[3] = {deviation.x(), deviation.y(), deviation.z()};
  double pose[3] = {0., 0., 0.};
  ndtpso_host::check(ndtpso_align(ndtpso_host::device(), xy.data(), (uint32_t)(xy.size() / 2), g, dv, &abi, 0u,
                                  draws.data(), ndtpso_host::score_mode(), pose, nullptr, nullptr), "align");
  return Vector3d(pose[0], pose[1], pose[2]);
}

double NDTFrame::cost(const Vector3d& trans, const NDTFrame* new_frame) {
  if (!built) build();
  uploadTable();
  std::vector<double> xy;
  new_frame->collectPoints(xy);
  const double pose[3] = {trans.x(), trans.y(), trans.z()};
  double c = 0.;
  ndtpso_host::check(ndtpso_cost_batch(ndtpso_host::device(), xy.data(), (uint32_t)(xy.size() / 2), pose, 1,
                                       NDTPSO_SCORE_F64, &c, nullptr), "cost_function");
  return c;
}

// reference: align, ndtframe.cpp:251-266.  Unlike the reference (which passes no config, :257, so always runs
// 30 x 50) the frame's own PSOConfig is honoured; with the default NDTPSOConfig the two coincide.
Vector3d NDTFrame::align(Vector3d initial_guess, const NDTFrame* const new_frame) {
  Vector3d deviation = s_iter < 2 ? Vector3d(.1, .1, 3.1415E-3) : Vector3d((s_pose_diff * 2.).array().abs());
  ++s_iter;
  Vector3d pose = optimize(initial_guess, new_frame, deviation, s_config.psoConfig);
#if TRANSFORM_POSE_AFTER_ALIGN
  pose -= s_trans;
#endif
  s_pose_diff = pose - s_prev_pose;
  s_prev_pose = pose;
  return pose;
}

// ---- bookkeeping / export members outside the accelerated path ---------------------------------------------

void NDTFrame::addPose(double timestamp, const Vector3d& pose, const Vector3d& odom) {
  s_timestamps.push_back(timestamp);
  s_poses.push_back(pose);
  s_odoms.push_back(odom);
}

void NDTFrame::resetCells() {
  for (NDTCell& c : cells) c.reset();
}

// reference: transform, ndtframe.cpp:119-140 (never called by the node; re-bins every stored point)
void NDTFrame::transform(Vector3d trans) {
  if (trans.isZero(1e-6)) return;
  std::vector<double> xy;
  for (uint32_t i : s_created)
    for (auto& slot : cells[i].win_->points)
      for (const Vector2d& p : slot) {
        xy.push_back(p.x());
        xy.push_back(p.y());
      }
  for (uint32_t i : s_created) cells[i] = NDTCell();
  s_created.clear();
  const uint32_t n = (uint32_t)(xy.size() / 2);
  if (n) {
    std::vector<int32_t> idx(n);
    const double t[3] = {trans.x(), trans.y(), trans.z()};
    const ndtpso_grid grid = grid_of(*this);
    ndtpso_host::check(ndtpso_points_to_cells(ndtpso_host::device(), &grid, xy.data(), n, t, xy.data(), idx.data()), "transform");
    append(xy.data(), idx.data(), n);
  }
  built = false;
}

// reference: dumpMap, ndtframe.cpp:268-422 (CSV / gnuplot / PNG export at shutdown) -- out of scope of the
// accelerated path; the poses and points are written as plain CSV so a run can still be inspected.
void NDTFrame::dumpMap(const char* filename, bool save_poses, bool save_points, bool, short
#if BUILD_OCCUPANCY_GRID
                       ,
                       bool
#endif
) {
  char name[1024];
  if (save_poses) {
    std::snprintf(name, sizeof(name), "%s.pose.csv", filename);
    if (FILE* f = std::fopen(name, "w")) {
      std::fprintf(f, "timestamp,xP,yP,thP,xO,yO,thO\n");
      for (size_t i = 0; i < s_poses.size(); ++i)
        std::fprintf(f, "%.5f,%.5f,%.5f,%.5f,%.5f,%.5f,%.5f\n", s_timestamps[i], s_poses[i].x(), s_poses[i].y(),
                     s_poses[i].z(), s_odoms[i].x(), s_odoms[i].y(), s_odoms[i].z());
      std::fclose(f);
    } else {
      std::printf("Cannot open file: %s\n", name);
    }
  }
  if (save_points) {
    std::snprintf(name, sizeof(name), "%s.map.csv", filename);
    if (FILE* f = std::fopen(name, "w")) {
      std::fprintf(f, "x,y\n");
      for (uint32_t i : s_created)
        for (const auto& slot : cells[i].win_->points)
          for (const Vector2d& p : slot) std::fprintf(f, "%.5f,%.5f\n", p.x(), p.y());
      std::fclose(f);
    } else {
      std::printf("Cannot open file: %s\n", name);
    }
  }
}
