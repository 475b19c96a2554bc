// NDTFrame of the MI355X build: grid bookkeeping on the host, arithmetic on the GPU (see ndtframe.h).
// Reference behaviour cited as lib/ndtpso_slam/ndtframe.cpp:LINE.
#include "ndtpso_slam/ndtframe.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device.h"
#include "ndtpso_slam/core.h"
#include "raster.h"

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
thread_local bool t_last_align_failed = false;  // the calling thread's most recent align() / pso_optimization could not run
constexpr uint32_t kScanCapacity = 4096;  // points of a pooled device scan buffer (grown for longer scans)

uint64_t map_pool_bytes(unsigned num_cells) {
  // 512 B per 32 points of one window slot.  A one-cell frame that is only ever updated (the node's global_map_,
  // ndtpso_slam_node.cpp:70-72,200-203) is never built, so it never rotates and keeps every point it was given.
  const char* e = std::getenv(num_cells == 1 ? "NDTPSO_GLOBAL_MAP_POOL_MB" : "NDTPSO_MAP_POOL_MB");
  const uint64_t mb = e ? (uint64_t)std::strtoull(e, nullptr, 10) : (num_cells == 1 ? 4096u : 1024u);
  return (mb ? mb : 1u) << 20;
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
  // (a resident frame has fewer than 2^21 cells -- ndtpso_map_create, the insert's 32-bit sort key; the reference's `unsigned int
  // numOfCells` has no such bound: 300 m at 0.2 m cells is 2.25 M -- so a larger frame keeps its cells here, on the host's side of
  // the C-ABI, as every frame did before round 3: same device kernels for the table, the score and the PSO, same results)
  s_resident = ndtpso_host::resident_default() && numOfCells < (1u << 21);
#if BUILD_OCCUPANCY_GRID
  s_occupancy_grid.cell_size = occupancy_grid_cell_size;  // ndtframe.cpp:32-46; 0 = no grid (intermediate frames)
  if (occupancy_grid_cell_size > 0.) {
    s_occupancy_grid.width = uint32_t(std::ceil(width / occupancy_grid_cell_size));
    s_occupancy_grid.height = uint32_t(std::ceil(height / occupancy_grid_cell_size));
    s_occupancy_grid.count = s_occupancy_grid.width * s_occupancy_grid.height;
    s_occupancy_grid.og = vector<int8_t>(s_occupancy_grid.count, 0);
  }
#endif
}

ndtpso_host::Ctx* NDTFrame::dev() const {
  if (!s_dev) {
    s_dev = ndtpso_host::thread_ctx();
    ndtpso_host::bind_frame(s_dev);
  }
  return s_dev;
}

NDTFrame::~NDTFrame() {
  if (d_scan_ || d_map_) {
    ndtpso_host::Use use(dev());
    if (d_scan_) ndtpso_host::release_scan(d_scan_, d_scan_cap_);
    if (d_map_ && ndtpso_host::alive()) ndtpso_map_destroy(d_map_);
  }
  if (s_dev) ndtpso_host::unbind_frame(s_dev);
}

// ---- resident mode: the frame's state lives on the device (include/ndtpso_hip.h, ndtpso_map_* / ndtpso_points_*) ----

// A one-cell frame that only had scans loaded is just a point list (d_scan_).  Anything else -- several cells, or an
// update / addPoint / build on it -- is a map; a point list already loaded becomes the open slot of cell 0.
ndtpso_map* NDTFrame::ensureMap() {
  if (!d_map_) {
    const ndtpso_grid grid = grid_of(*this);
    double og_cell_size = 0.;
#if BUILD_OCCUPANCY_GRID
    og_cell_size = s_occupancy_grid.cell_size;
#endif
    if (!ndtpso_host::check(ndtpso_map_create(ndtpso_host::device(), &grid, og_cell_size, map_pool_bytes(numOfCells), &d_map_),
                            "device map")) {
      d_map_ = nullptr;  // every map call on this frame is then refused by the C-ABI (null handle) and skipped
      return nullptr;
    }
    if (d_scan_) {
      ndtpso_host::check(ndtpso_map_insert(d_map_, d_scan_, nullptr), "device map");
      ndtpso_host::release_scan(d_scan_, d_scan_cap_);
      d_scan_ = nullptr;
      d_scan_upper_ = 0;
    }
  }
  return d_map_;
}

void NDTFrame::residentPoints(bool slot0_only, std::vector<double>& xy) const {
  xy.clear();
  if (d_map_) {
    uint64_t n = 0;
    if (!ndtpso_host::check(ndtpso_map_get_points(d_map_, slot0_only ? 1 : 0, nullptr, 0, &n), "map points")) return;
    xy.resize(2 * (size_t)n);
    if (n && !ndtpso_host::check(ndtpso_map_get_points(d_map_, slot0_only ? 1 : 0, xy.data(), n, &n), "map points")) xy.clear();
  } else if (d_scan_) {
    uint32_t n = 0;
    xy.resize(2 * (size_t)d_scan_upper_);
    if (!ndtpso_host::check(ndtpso_points_get(d_scan_, xy.data(), d_scan_upper_, &n), "scan points")) n = 0;
    xy.resize(2 * (size_t)n);
  }
}

void NDTFrame::syncHostView() {
  ndtpso_host::Use use(dev());
  if (!s_resident) return;
  if (!d_map_) {
    if (d_scan_) {  // a one-cell frame holding loaded scans: its only cell exists as soon as one point survived
      uint32_t n = 0;
      if (!ndtpso_host::check(ndtpso_points_get(d_scan_, nullptr, 0, &n), "scan points")) return;
      cells[0].created = cells[0].created || n > 0;
    }
    return;
  }
  uint32_t n = 0;
  if (!ndtpso_host::check(ndtpso_map_get_cells(d_map_, nullptr, 0, &n), "map cells")) return;
  std::vector<ndtpso_cell_row> rows(n ? n : 1);
  if (!ndtpso_host::check(ndtpso_map_get_cells(d_map_, rows.data(), n, &n), "map cells")) return;
  for (NDTCell& c : cells) c.created = c.built = false;
  for (uint32_t k = 0; k < n; ++k) {
    NDTCell& c = cells[(size_t)rows[k].index];
    c.created = true;
    c.built = rows[k].built != 0;
    c.mean = Vector2d(rows[k].mean[0], rows[k].mean[1]);
  }
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
  ndtpso_host::Use use(dev());
  built = false;
  const uint32_t n = (uint32_t)laser_data.size();
  if (n == 0) {
    // ndtframe.cpp:145 clears `built` before anything else: an empty scan still makes the next cost_function() build
    // again, and a build repeated on unchanged cells is not a no-op in floating point (include/ndtpso_hip.h)
    if (s_resident && d_map_) ndtpso_host::check(ndtpso_map_mark_unbuilt(d_map_), "loadLaser");
    return;
  }
  ndtpso_ctx* dev = ndtpso_host::device();
  ndtpso_scan_geom geom{n, min_angle, angle_increment, max_range, s_config.laserIgnoreEpsilon};
  const double t[3] = {s_trans.x(), s_trans.y(), s_trans.z()};
  if (s_resident) {
    const ndtpso_grid frame = grid_of(*this);
    if (numOfCells == 1 && !d_map_) {  // the node's per-scan frame: the scan stays a point list on the device
      if (!d_scan_) {
        d_scan_cap_ = std::max(kScanCapacity, n);
        d_scan_ = ndtpso_host::acquire_scan(d_scan_cap_);
        d_scan_upper_ = 0;
        if (!d_scan_) return;  // no device memory: the scan is dropped (error recorded)
      }
      if (d_scan_upper_ + n <= d_scan_cap_) {
        if (ndtpso_host::check(ndtpso_points_load_scan(d_scan_, laser_data.data(), &geom, t, &frame, d_scan_upper_ > 0), "loadLaser"))
          d_scan_upper_ += n;
        return;
      }
    }
    ndtpso_map* m = ensureMap();
    if (!ndtpso_host::check(ndtpso_map_mark_unbuilt(m), "loadLaser")) return;  // ndtframe.cpp:145
    const uint32_t cap = std::max(kScanCapacity, n);
    ndtpso_points* tmp = ndtpso_host::acquire_scan(cap);
    if (ndtpso_host::check(ndtpso_points_load_scan(tmp, laser_data.data(), &geom, t, nullptr, 0), "loadLaser"))
      ndtpso_host::check(ndtpso_map_insert(m, tmp, nullptr), "loadLaser");
    ndtpso_host::release_scan(tmp, cap);  // stream order keeps the buffer intact until the insert has read it
    return;
  }
  std::vector<double> xy(2 * (size_t)n);
  std::vector<int32_t> idx(n);
  uint32_t kept = 0;
  const ndtpso_grid grid = grid_of(*this);
  if (!ndtpso_host::check(ndtpso_scan_to_cells(dev, laser_data.data(), &geom, t, &grid, xy.data(), idx.data(), &kept), "loadLaser"))
    return;
  append(xy.data(), idx.data(), kept);
}

void NDTFrame::collectPoints(std::vector<double>& xy) const {
  ndtpso_host::Use use(dev());
  xy.clear();
  if (s_resident) {
    residentPoints(true, xy);
    return;
  }
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
  // The node merges every scan at the pose align() has just returned (ndtpso_slam_node.cpp:194-198) and never asks whether
  // that alignment ran: the reference's cannot fail.  Here it can (a device fault: align() then returns its initial guess,
  // lastAlignOk() says so), and a scan merged at an unrefined guess would corrupt the map for good.  So an update() that
  // follows a failed align() against this frame is REFUSED -- the scan is dropped, counted (updatesRefused()) and recorded
  // in the error state (ndtpso_slam/status.h) -- and the next successful align() clears the condition.
  // ... and so is an update() of ANY frame by the thread whose most recent align() failed: the node merges the same pose into
  // its global map as well (global_map_->update, ndtpso_slam_node.cpp:202), a frame that never aligns and so never learns of it.
  if (!s_last_align_ok || t_last_align_failed) {
    ++s_updates_refused;
    ndtpso_host::check(NDTPSO_E_STATE, "update after a failed align (scan not merged)");
    return;
  }
  // a frame that lives in another thread's device context: its points travel through the host (collected under ITS context)
  std::vector<double> foreign;
  const bool is_foreign = new_frame->dev() != dev();
  if (is_foreign) new_frame->collectPoints(foreign);
  ndtpso_host::Use use(dev());
  built = false;
  if (s_resident) {
    const double pose[3] = {trans.x(), trans.y(), trans.z()};
    ndtpso_map* m = ensureMap();
    if (!ndtpso_host::check(ndtpso_map_mark_unbuilt(m), "update")) return;  // ndtframe.cpp:188, also when no point follows
    if (!is_foreign && new_frame->s_resident && new_frame->d_scan_ && !new_frame->d_map_) {  // device to device, nothing to wait for
      if (!ndtpso_host::check(ndtpso_map_insert(m, new_frame->d_scan_, pose), "update")) return;
    } else {
      std::vector<double> pts;
      if (is_foreign) pts.swap(foreign); else new_frame->collectPoints(pts);
      if (!ndtpso_host::check(ndtpso_map_insert_host(m, pts.data(), (uint32_t)(pts.size() / 2), pose), "update")) return;
    }
    // A frame that is aligned against gets its cells built and its table packed right away, off the next align()'s
    // critical path; should something other than align / build come first, the device takes the build back
    // (ndtpso_map_speculate_build), so the lazy build of the reference is what every caller still observes.
    static const bool speculate = [] {
      const char* e = std::getenv("NDTPSO_SPECULATE");  // =0: leave every build to the call that asks for it
      return !(e && e[0] == '0');
    }();
    if (s_iter > 0 && speculate) ndtpso_host::check(ndtpso_map_speculate_build(m), "update");
    return;
  }
  std::vector<double> xy;
  if (is_foreign) xy.swap(foreign); else new_frame->collectPoints(xy);
  const uint32_t n = (uint32_t)(xy.size() / 2);
  if (n == 0) return;
  std::vector<int32_t> idx(n);
  const double t[3] = {trans.x(), trans.y(), trans.z()};
  const ndtpso_grid grid = grid_of(*this);
  if (!ndtpso_host::check(ndtpso_points_to_cells(ndtpso_host::device(), &grid, xy.data(), n, t, xy.data(), idx.data()), "update"))
    return;
  append(xy.data(), idx.data(), n);
}

// north star's addScan(): loadLaser() into a one-cell per-scan frame + update() with it (see ndtframe.h)
void NDTFrame::addScan(const Vector3d& pose, const vector<float>& laser_data, const float& min_angle,
                       const float& angle_increment, const float& max_range) {
  ndtpso_host::Use use(dev());  // (the scratch frame below binds to the active context: this frame's)
  NDTFrame scan(Vector3d::Zero(), width, height, (double)std::max(width, height), false, s_config);  // ndtpso_slam_node.cpp:229-230
  scan.s_dev = dev();
  ndtpso_host::bind_frame(scan.s_dev);
  scan.loadLaser(laser_data, min_angle, angle_increment, max_range);
  update(pose, &scan);
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
  ndtpso_host::Use use(dev());
  if (s_resident) {
    if (getCellIndex(point, widthNumOfCells, cell_side) == -1) return;  // outside the frame: dropped, `built` untouched
    const double p[2] = {point.x(), point.y()};
    if (ndtpso_host::check(ndtpso_map_insert_host(ensureMap(), p, 1, nullptr), "addPoint")) built = false;
    return;
  }
  const int32_t idx = getCellIndex(point, widthNumOfCells, cell_side);
  const double xy[2] = {point.x(), point.y()};
  append(xy, &idx, 1);
}

// reference: build, ndtframe.cpp:68-117 -- NDTCell::build for every created cell, batched into one device call
void NDTFrame::build() {
  ndtpso_host::Use use(dev());
  if (s_resident) {
    if (ndtpso_host::check(ndtpso_map_build(ensureMap()), "build")) built = true;
    return;
  }
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
    if (!ndtpso_host::check(ndtpso_cells_build_windowed(ndtpso_host::device(), n, cw.data(), off.data(), xy.data()), "build"))
      return;  // the cells keep their previous statistics; the frame stays un-built
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
#if BUILD_OCCUPANCY_GRID
  if (s_occupancy_grid.cell_size > 0.) rasteriseOccupancy();
#endif
  built = true;
  s_table_dirty = true;
}

#if BUILD_OCCUPANCY_GRID
// reference: the occupancy-grid branch of build(), ndtframe.cpp:79-112.  The Gaussians are evaluated on the device
// (ndtpso_occupancy_values); the scatter keeps the reference's order (ascending cell index, j outer, k inner) and
// its indexing, including row = index / heightNumOfCells and og[x + height * y].
void NDTFrame::rasteriseOccupancy() {
  auto& g = s_occupancy_grid;
  const uint32_t per_cell = (uint32_t)std::floor(cell_side / g.cell_size);
  if (per_cell == 0) return;
  std::vector<uint32_t> order(s_created);
  std::sort(order.begin(), order.end());
  std::vector<int32_t> index;
  std::vector<double> mean, icov;
  for (uint32_t i : order) {
    const NDTCell& c = cells[i];
    if (!c.built) continue;  // normalDistribution of an un-built cell is 0: nothing is written (ndtcell.cpp:70-78)
    index.push_back((int32_t)i);
    mean.push_back(c.mean.x());
    mean.push_back(c.mean.y());
    for (int j = 0; j < 4; ++j) icov.push_back(c.win_->inv_covar[j]);
  }
  if (index.empty()) return;
  std::vector<int8_t> v(index.size() * per_cell * per_cell);
  const ndtpso_grid grid = grid_of(*this);
  if (!ndtpso_host::check(ndtpso_occupancy_values(ndtpso_host::device(), &grid, g.cell_size, (uint32_t)index.size(),
                                                  index.data(), mean.data(), icov.data(), v.data()),
                          "occupancy grid"))
    return;
  size_t t = 0;
  for (int32_t i : index) {
    const uint32_t cx = (uint32_t)i % widthNumOfCells, cy = (uint32_t)i / heightNumOfCells;
    for (uint32_t j = 0; j < per_cell; ++j)
      for (uint32_t k = 0; k < per_cell; ++k, ++t) {
        if (v[t] < 0) continue;  // p == 0
        const uint32_t ox = cx * per_cell + j, oy = cy * per_cell + k;
        g.min_x_ind = std::min(ox, g.min_x_ind);
        g.max_x_ind = std::max(ox, g.max_x_ind);
        g.min_y_ind = std::min(oy, g.min_y_ind);
        g.max_y_ind = std::max(oy, g.max_y_ind);
        const size_t at = (size_t)ox + (size_t)g.height * oy;
        if (at < g.og.size()) g.og[at] = v[t];  // (the reference writes unchecked)
      }
  }
}

void NDTFrame::fetchOccupancy() const {
  auto& g = s_occupancy_grid;
  if (!s_resident || !d_map_ || !(g.cell_size > 0.)) return;
  uint32_t ext[4] = {UINT32_MAX, 0, UINT32_MAX, 0};
  if (!ndtpso_host::check(ndtpso_map_get_occupancy(d_map_, g.og.data(), g.og.size(), nullptr, nullptr, ext), "occupancy grid"))
    return;
  g.min_x_ind = ext[0], g.max_x_ind = ext[1], g.min_y_ind = ext[2], g.max_y_ind = ext[3];
}

const vector<int8_t>& NDTFrame::occupancyGrid(uint32_t* og_width, uint32_t* og_height, uint32_t extent[4]) const {
  ndtpso_host::Use use(dev());
  fetchOccupancy();
  if (og_width) *og_width = s_occupancy_grid.width;
  if (og_height) *og_height = s_occupancy_grid.height;
  if (extent) {
    extent[0] = s_occupancy_grid.min_x_ind, extent[1] = s_occupancy_grid.max_x_ind;
    extent[2] = s_occupancy_grid.min_y_ind, extent[3] = s_occupancy_grid.max_y_ind;
  }
  return s_occupancy_grid.og;
}
#endif

// built cells -> device reference table (LDS image packed by ndtpso_ref_set_cells)
bool NDTFrame::uploadTable() {
  if (!s_table_dirty && ndtpso_host::table_owner() == this) return true;
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
  if (!ndtpso_host::check(ndtpso_ref_set_cells(ndtpso_host::device(), &grid, (uint32_t)index.size(), index.data(),
                                               mean.data(), icov.data()), "reference table upload")) {
    ndtpso_host::table_owner() = nullptr;
    return false;
  }
  ndtpso_host::table_owner() = this;
  s_table_dirty = false;
  return true;
}

// pso_optimization against this frame, core.cpp:50-116.  The std::rand() stream is drawn here, in the order and
// quantity the reference consumes it (3 + 3P + 6PI), so srand() by the caller has the reference's meaning.
Vector3d NDTFrame::optimize(const Vector3d& guess, const NDTFrame* new_frame, const Vector3d& deviation,
                            const PSOConfig& cfg) {
  std::vector<double> foreign;  // (a frame of another thread's context: its points through the host, see update())
  const bool is_foreign = new_frame->dev() != dev();
  if (is_foreign) new_frame->collectPoints(foreign);
  ndtpso_host::Use use(dev());
  if (s_resident) {
    ndtpso_map* m = ensureMap();  // ndtpso_map_align builds first if need be (cost_function's lazy build, core.cpp:27-28)
    const ndtpso_points* pts = nullptr;
    ndtpso_points* tmp = nullptr;
    uint32_t tmp_cap = 0;
    if (!is_foreign && new_frame->s_resident && new_frame->d_scan_ && !new_frame->d_map_) {
      pts = new_frame->d_scan_;
    } else {
      std::vector<double> xy;
      if (is_foreign) xy.swap(foreign); else new_frame->collectPoints(xy);
      tmp_cap = std::max(kScanCapacity, (uint32_t)(xy.size() / 2));
      tmp = ndtpso_host::acquire_scan(tmp_cap);
      ndtpso_host::check(ndtpso_points_set(tmp, xy.data(), (uint32_t)(xy.size() / 2)), "align");
      pts = tmp;
    }
    const ndtpso_pso_config abi = to_abi(cfg);
    std::vector<int32_t> draws(ndtpso_rand_draws(&abi));
    ndtpso_host::draw_rand(draws.data(), draws.size());  // drawn even if the device call fails: the caller's stream moves on as it would have
    const double g[3] = {guess.x(), guess.y(), guess.z()};
    const double dv[3] = {deviation.x(), deviation.y(), deviation.z()};
    double pose[3] = {g[0], g[1], g[2]};
    const bool ok = ndtpso_host::check(ndtpso_map_align(m, pts, g, dv, &abi, 0u, draws.data(), ndtpso_host::score_mode(), pose,
                                                        nullptr, nullptr), "align");
    if (ok) built = true;
    if (tmp) ndtpso_host::release_scan(tmp, tmp_cap);
    s_last_align_ok = ok;
    t_last_align_failed = !ok;
    return ok ? Vector3d(pose[0], pose[1], pose[2]) : guess;  // device fault: the initial guess, never a CPU estimate
  }
  if (!built) build();  // core.cpp:27-28 (lazy build inside cost_function)
  const bool table_ok = uploadTable();
  std::vector<double> xy;
  if (is_foreign) xy.swap(foreign); else new_frame->collectPoints(xy);
  const ndtpso_pso_config abi = to_abi(cfg);
  std::vector<int32_t> draws(ndtpso_rand_draws(&abi));
  ndtpso_host::draw_rand(draws.data(), draws.size());
  const double g[3] = {guess.x(), guess.y(), guess.z()};
  const double dv[3] = {deviation.x(), deviation.y(), deviation.z()};
  double pose[3] = {g[0], g[1], g[2]};
  s_last_align_ok = table_ok && ndtpso_host::check(ndtpso_align(ndtpso_host::device(), xy.data(), (uint32_t)(xy.size() / 2), g, dv, &abi,
                                                               0u, draws.data(), ndtpso_host::score_mode(), pose, nullptr, nullptr), "align");
  t_last_align_failed = !s_last_align_ok;
  if (!s_last_align_ok) return guess;
  return Vector3d(pose[0], pose[1], pose[2]);
}

double NDTFrame::cost(const Vector3d& trans, const NDTFrame* new_frame) {
  std::vector<double> foreign;
  const bool is_foreign = new_frame->dev() != dev();
  if (is_foreign) new_frame->collectPoints(foreign);
  ndtpso_host::Use use(dev());
  const double pose[3] = {trans.x(), trans.y(), trans.z()};
  double c = 0.;
  if (s_resident) {
    ndtpso_map* m = ensureMap();
    ndtpso_points* tmp = nullptr;
    uint32_t tmp_cap = 0;
    ndtpso_points* pts = nullptr;
    if (!is_foreign && new_frame->s_resident && new_frame->d_scan_ && !new_frame->d_map_) {
      pts = new_frame->d_scan_;
    } else {
      std::vector<double> xy;
      if (is_foreign) xy.swap(foreign); else new_frame->collectPoints(xy);
      tmp_cap = std::max(kScanCapacity, (uint32_t)(xy.size() / 2));
      tmp = ndtpso_host::acquire_scan(tmp_cap);
      ndtpso_host::check(ndtpso_points_set(tmp, xy.data(), (uint32_t)(xy.size() / 2)), "cost_function");
      pts = tmp;
    }
    if (ndtpso_host::check(ndtpso_map_cost(m, pts, pose, 1, NDTPSO_SCORE_F64, &c), "cost_function"))  // builds if need be
      built = true;
    else
      c = 0.;
    if (tmp) ndtpso_host::release_scan(tmp, tmp_cap);
    return c;
  }
  if (!built) build();
  if (!uploadTable()) return 0.;
  std::vector<double> xy;
  if (is_foreign) xy.swap(foreign); else new_frame->collectPoints(xy);
  if (!ndtpso_host::check(ndtpso_cost_batch(ndtpso_host::device(), xy.data(), (uint32_t)(xy.size() / 2), pose, 1,
                                            NDTPSO_SCORE_F64, &c, nullptr), "cost_function"))
    return 0.;
  return c;
}

// reference: align, ndtframe.cpp:251-266.  The reference passes no config to pso_optimization (:257), so the
// frame's own PSOConfig is never used and every alignment runs the default 30 x 50.  That is what happens here
// too; NDTPSO_ALIGN_FRAME_CONFIG=1 in the environment makes align() honour the frame's PSOConfig instead.
Vector3d NDTFrame::align(Vector3d initial_guess, const NDTFrame* const new_frame) {
  Vector3d deviation = s_iter < 2 ? Vector3d(.1, .1, 3.1415E-3) : Vector3d((s_pose_diff * 2.).array().abs());
  ++s_iter;
  static const bool frame_config = [] {
    const char* e = std::getenv("NDTPSO_ALIGN_FRAME_CONFIG");
    return e && e[0] == '1';
  }();
  Vector3d pose = optimize(initial_guess, new_frame, deviation, frame_config ? s_config.psoConfig : PSOConfig());
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
  ndtpso_host::Use use(dev());
  for (NDTCell& c : cells) c.reset();
  if (d_map_) ndtpso_host::check(ndtpso_map_reset(d_map_), "resetCells");
  if (d_scan_) {
    ndtpso_host::release_scan(d_scan_, d_scan_cap_);
    d_scan_ = nullptr;
    d_scan_upper_ = 0;
  }
}

// reference: transform, ndtframe.cpp:119-140 (never called by the node; re-bins every stored point)
void NDTFrame::transform(Vector3d trans) {
  ndtpso_host::Use use(dev());
  if (trans.isZero(1e-6)) return;
  if (s_resident) {
    std::vector<double> pts;
    residentPoints(false, pts);
    ndtpso_map* m = ensureMap();
    if (!ndtpso_host::check(ndtpso_map_clear(m), "transform")) return;  // fresh cells (ndtframe.cpp:123): not built
    const double t[3] = {trans.x(), trans.y(), trans.z()};
    if (!ndtpso_host::check(ndtpso_map_insert_host(m, pts.data(), (uint32_t)(pts.size() / 2), t), "transform")) {
      // the device refused the transformed points after the map was emptied: put the points back where they were
      // rather than leave an empty frame behind (the failure itself is on record, ndtpso_slam/status.h)
      ndtpso_host::check(ndtpso_map_insert_host(m, pts.data(), (uint32_t)(pts.size() / 2), nullptr), "transform (restore)");
    }
    built = false;
    return;
  }
  std::vector<double> xy;
  for (const NDTCell& c : cells) {  // cell order, slot order, insertion order (the loops of ndtframe.cpp:126-134)
    if (!c.win_) continue;
    for (auto& slot : c.win_->points)
      for (const Vector2d& p : slot) {
        xy.push_back(p.x());
        xy.push_back(p.y());
      }
  }
  for (uint32_t i : s_created) cells[i] = NDTCell();
  s_created.clear();
  const uint32_t n = (uint32_t)(xy.size() / 2);
  if (n) {
    std::vector<int32_t> idx(n);
    const double t[3] = {trans.x(), trans.y(), trans.z()};
    const ndtpso_grid grid = grid_of(*this);
    if (ndtpso_host::check(ndtpso_points_to_cells(ndtpso_host::device(), &grid, xy.data(), n, t, xy.data(), idx.data()), "transform"))
      append(xy.data(), idx.data(), n);
  }
  built = false;
}

// reference: dumpMap, ndtframe.cpp:268-422 -- <name>.pose.csv, <name>.map.csv, <name>.gnuplot, the map image and
// the occupancy-grid image, same file names and text formats.  The reference draws the images with OpenCV when it
// was found at build time; here they are always drawn (raster.h), so pixel-level equality with OpenCV's line and
// circle primitives is not claimed.  Shutdown-time export, not on the alignment path.
void NDTFrame::dumpMap(const char* filename, bool save_poses, bool save_points, bool save_image, short density
#if BUILD_OCCUPANCY_GRID
                       ,
                       bool save_occupancy_grid
#endif
) {
  using ndtpso_host::Raster;
  using ndtpso_host::Rgb;
  ndtpso_host::Use use(dev());
  FILE *poses = nullptr, *points = nullptr;
  char name[1280];
  if (save_poses) {
    std::snprintf(name, sizeof(name), "%s.pose.csv", filename);
    if ((poses = std::fopen(name, "w"))) std::fprintf(poses, "timestamp,xP,yP,thP,xO,yO,thO\n");
  }
  if (save_points) {
    std::snprintf(name, sizeof(name), "%s.map.csv", filename);
    if ((points = std::fopen(name, "w"))) std::fprintf(points, "x,y\n");
  }
  if ((save_poses && !poses) || (save_points && !points)) {  // ndtframe.cpp:294-297
    std::printf("%s: Cannot open files, cannot save!\n ", __func__);
    if (poses) std::fclose(poses);
    if (points) std::fclose(points);
    return;
  }

  const int size_x = width * density, size_y = height * density;  // density in pixels per metre
  Raster img(save_image ? size_x : 1, save_image ? size_y : 1, 3, 255);  // cv::Mat(rows = size_x, cols = size_y), :304
  if (save_image && density > 0)
    for (int i = 0; i < size_x; i += density) {  // one grid line per metre
      img.line(i, 0, i, size_y, Rgb{180, 180, 180});
      img.line(0, i, size_x, i, Rgb{180, 180, 180});
    }

  // every stored point, in cell order then window-slot order then insertion order (ndtframe.cpp:314-328)
  auto emit = [&](double x, double y) {
    if (save_image)
      img.circle(size_x / 2 + static_cast<int>(x * density), size_y / 2 - static_cast<int>(y * density), 1, Rgb{0, 0, 0});
    if (save_points) std::fprintf(points, "%.5f,%.5f\n", x, y);
  };
  if (s_resident) {
    std::vector<double> pts;
    residentPoints(false, pts);
    for (size_t i = 0; i + 1 < pts.size(); i += 2) emit(pts[i], pts[i + 1]);
  } else {
    for (const NDTCell& c : cells) {
      if (!c.win_) continue;
      for (const auto& slot : c.win_->points)
        for (const Vector2d& p : slot) emit(p.x(), p.y());
    }
  }

  int counter = 0;
  for (size_t i = 0; i < s_poses.size(); ++i) {  // ndtframe.cpp:331-350
    if (save_image) {
      const int x = size_x / 2 + static_cast<int>(s_poses[i].x() * density);
      const int y = size_y / 2 - static_cast<int>(s_poses[i].y() * density);
      const int dx = static_cast<int>(.5 * std::cos(-s_poses[i].z()) * density);
      const int dy = static_cast<int>(.5 * std::sin(-s_poses[i].z()) * density);
      if (counter == 0) img.line(x, y, x + dx, y + dy, Rgb{80, 40, 40});  // cv::Scalar is BGR
      img.circle(x, y, 2, Rgb{255, 0, 0});
      counter = (counter + 1) % 5;
    }
    // the reference guards the pose rows with save_points (:347); they go to the pose file, four columns
    if (save_points && poses)
      std::fprintf(poses, "%.6f,%.5f,%.5f,%.5f\n", s_timestamps[i], s_poses[i].x(), s_poses[i].y(), s_poses[i].z());
  }
  if (poses) std::fclose(poses);
  if (points) std::fclose(points);

  if (save_poses || save_points) {  // ndtframe.cpp:358-390
    std::snprintf(name, sizeof(name), "%s.gnuplot", filename);
    if (FILE* gp = std::fopen(name, "w")) {
      std::fprintf(gp, "set datafile separator ','\nset key autotitle columnhead\nset size ratio -1\nplot ");
      if (save_points)
        std::fprintf(gp, "'%s.map.csv' title 'Map' with points pointsize 0.2 pointtype 5 linecolor rgb '#555555'",
                     filename);
      if (save_poses)
        std::fprintf(gp,
                     ", \\\n'%s.pose.csv' using 2:3 title 'Pose (LiDAR)' with linespoints linewidth 0.7 "
                     "pointtype 6 pointsize 0.7 linecolor rgb '#ff0000'",
                     filename);
      std::fprintf(gp, "\npause 1000\n");
      std::fclose(gp);
    }
  }

  if (save_image) {  // ndtframe.cpp:393-398
    std::snprintf(name, sizeof(name), "%s-w%d-%dp%di-%dx%d-c%.2f-%dppm.png", filename, NDT_WINDOW_SIZE,
                  s_config.psoConfig.populationSize, s_config.psoConfig.iterations, width, height, cell_side, density);
    if (!img.writePng(name)) std::printf("%s: cannot write %s\n", __func__, name);
  }

#if BUILD_OCCUPANCY_GRID
  fetchOccupancy();
  const auto& g = s_occupancy_grid;
  if (save_occupancy_grid && g.cell_size > 0. && g.min_x_ind <= g.max_x_ind && g.min_y_ind <= g.max_y_ind) {
    // ndtframe.cpp:401-420.  The reference sizes the image (max - min) and then addresses row (max - min) and
    // column (max - min), one past the end; this image has the extra row and column instead.
    const uint32_t real_width = g.max_x_ind - g.min_x_ind, real_height = g.max_y_ind - g.min_y_ind;
    Raster og_img((int)real_height + 1, (int)real_width + 1, 1, 255);
    for (uint32_t i = g.min_x_ind; i <= g.max_x_ind; ++i)
      for (uint32_t j = g.min_y_ind; j <= g.max_y_ind; ++j) {
        const size_t ind = (size_t)i + (size_t)g.height * j;
        if (ind < g.og.size() && g.og[ind] > 0)
          og_img.put(int(i - g.min_x_ind), int(real_height - (j - g.min_y_ind)),
                     Rgb{uint8_t(255.0 - g.og[ind] * 2.55), 0, 0});
      }
    std::snprintf(name, sizeof(name), "%s-%dx%d-cell%.2fm-occupancy-grid.png", filename, g.width, g.height, g.cell_size);
    if (!og_img.writePng(name)) std::printf("%s: cannot write %s\n", __func__, name);
  }
#endif
}
