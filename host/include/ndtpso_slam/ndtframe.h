// NDTFrame of the MI355X build of libndtpso_slam.
//
// Source-compatible with the reference's include/ndtpso_slam/ndtframe.h:12-72 -- same constructor
// arguments, same public data members, same methods -- so ndtpso_slam_node.cpp (call sites :64-78, 110,
// 155, 167, 186, 194, 198, 202, 206, 229-230) builds against it unchanged.  The frame keeps the points
// and the window state; every number is produced on the GPU through the C-ABI in include/ndtpso_hip.h:
//   loadLaser -> ndtpso_scan_to_cells                               update -> ndtpso_points_to_cells
//   build     -> ndtpso_cells_build_windowed (+ ndtpso_occupancy_values) align  -> ndtpso_ref_set_cells + ndtpso_align
#ifndef NDTPSO_SLAM_AMD_NDTFRAME_H
#define NDTPSO_SLAM_AMD_NDTFRAME_H

#include <cstdint>
#include <utility>
#include <vector>

#include "ndtpso_slam/ndtcell.h"

using namespace Eigen;
using std::vector;

struct ndtpso_points;  // device-resident scan (include/ndtpso_hip.h)
struct ndtpso_map;     // device-resident map

// ndtpso_slam_device_init(): creates the process-wide device context now instead of at the first frame operation
// (optional).  Call it before srand() when the std::rand() stream must be reproducible: runtime start-up may itself
// call rand().  ndtpso_slam_last_error() / ndtpso_slam_error_count(): what a failed (skipped) device call left behind.
#include "ndtpso_slam/status.h"

namespace ndtpso_host {
struct Ctx;
}

class NDTFrame {
 public:
  uint16_t width, height, widthNumOfCells, heightNumOfCells;
  vector<NDTCell> cells;
  bool built;
  unsigned int numOfCells;
  double cell_side;

  NDTFrame(Vector3d trans, unsigned short width = 20, unsigned short height = 20, double cell_side = 1.0,
           bool calculate_cells_params = true, NDTPSOConfig config = NDTPSOConfig()
#if BUILD_OCCUPANCY_GRID
               ,
           double occupancy_grid_cell_size = .0
#endif
  );

  ~NDTFrame();
  NDTFrame(const NDTFrame&) = delete;  // a frame owns device state (the reference's frames are never copied either)
  NDTFrame& operator=(const NDTFrame&) = delete;

  void loadLaser(const vector<float>& laser_data, const float& min_angle, const float& angle_increment,
                 const float& max_range);
  void update(Vector3d trans, NDTFrame* new_frame);
  void addPoint(Vector2d& point);
  inline void setTrans(Vector3d trans) { s_trans = std::move(trans); }
  void transform(Vector3d trans);
  void build();
  int getCellIndex(Vector2d point, int grid_width, double cell_side);
  Vector3d align(Vector3d initial_guess, const NDTFrame* const new_frame);
  void dumpMap(const char* filename, bool save_poses = true, bool save_points = true, bool save_image = true,
               short density = 50
#if BUILD_OCCUPANCY_GRID
               ,
               bool save_occupancy_grid = true
#endif
  );
  void addPose(double timestamp, const Vector3d& pose, const Vector3d& odom = Vector3d::Zero());
  void resetCells();

  // ---- additions of this build (not in the reference) ----
  // BASELINE.json's north star names an `addScan()`; the reference has no such method -- its ingest is loadLaser() into
  // a per-scan frame followed by update() of the map with that frame at the estimated pose (ndtpso_slam_node.cpp:186,
  // 194-198).  addScan() is that pair in one call: the scan, taken at `pose` in this frame's coordinates, is merged
  // into this frame.  Equivalent to { NDTFrame f(Vector3d::Zero(), width, height, max(width, height), false, config());
  // f.loadLaser(...); update(pose, &f); } -- same points, same order, same bits.
  void addScan(const Vector3d& pose, const vector<float>& laser_data, const float& min_angle, const float& angle_increment,
               const float& max_range);
  // new-frame points in the order cost_function visits them (cells, then insertion; core.cpp:33-36)
  void collectPoints(std::vector<double>& xy) const;
  const NDTPSOConfig& config() const { return s_config; }
  // Resident mode (default; NDTPSO_RESIDENT=0 turns it off): the points, the sliding windows, the occupancy grid and
  // the alignment table of this frame live on the GPU and `cells` is not maintained while the frame is used.
  // syncHostView() refreshes created / built / mean of every cell from the device.
  bool resident() const { return s_resident; }
  void syncHostView();
#if BUILD_OCCUPANCY_GRID
  // the occupancy grid as the reference stores it (og[x + height * y]) and its extent {min_x, max_x, min_y, max_y}
  const vector<int8_t>& occupancyGrid(uint32_t* og_width = nullptr, uint32_t* og_height = nullptr,
                                      uint32_t extent[4] = nullptr) const;
#endif
  // false when the last align() / pso_optimization against this frame could not run on the device and returned its
  // initial guess unrefined (the reference's API has no error channel; see also ndtpso_slam/status.h).  A caller that
  // is about to update() the map with that pose can look first.
  bool lastAlignOk() const { return s_last_align_ok; }
  // pso_optimization against this frame (used by align() and by the free function in core.h)
  Vector3d optimize(const Vector3d& guess, const NDTFrame* new_frame, const Vector3d& deviation, const PSOConfig& cfg);
  double cost(const Vector3d& trans, const NDTFrame* new_frame);

 private:
  Vector3d s_trans{Vector3d::Zero()}, s_prev_pose{Vector3d::Zero()}, s_pose_diff{Vector3d::Zero()};
  vector<Vector3d> s_poses, s_odoms;
  vector<double> s_timestamps;
  double s_x_min, s_x_max, s_y_min, s_y_max;
  NDTPSOConfig s_config;
  int s_iter{0};
#if BUILD_OCCUPANCY_GRID
  mutable struct {  // reference: s_occupancy_grid, ndtframe.h:22-29
    uint32_t count{0}, width{0}, height{0}, max_x_ind{0}, max_y_ind{0}, min_x_ind{UINT32_MAX}, min_y_ind{UINT32_MAX};
    double cell_size{0.};
    vector<int8_t> og;
  } s_occupancy_grid;
  void rasteriseOccupancy();
  void fetchOccupancy() const;  // resident mode: device -> s_occupancy_grid
#endif
  std::vector<uint32_t> s_created;  // indices of created cells, in creation order
  bool s_table_dirty{true};         // the device reference table must be re-uploaded before the next align
  bool s_last_align_ok{true};
  void append(const double* xy, const int32_t* idx, uint32_t n);
  bool uploadTable();  // false: the device refused the table (error recorded, see status.h)
  // resident mode
  bool s_resident{false};
  ndtpso_points* d_scan_{nullptr};  // a one-cell frame that only ever had scans loaded: its point list
  ndtpso_map* d_map_{nullptr};      // everything else
  uint32_t d_scan_upper_{0};        // upper bound of the points in d_scan_
  uint32_t d_scan_cap_{0};
  ndtpso_map* ensureMap();
  void residentPoints(bool slot0_only, std::vector<double>& xy) const;
  // the device context this frame lives in: the calling thread's at the frame's first device operation, then for good
  // (host/src/device.h: one context per host thread; every device-touching member locks and uses this one)
  mutable ndtpso_host::Ctx* s_dev{nullptr};
  ndtpso_host::Ctx* dev() const;
  unsigned long s_updates_refused{0};

 public:
  // update() calls refused because the frame's last align() had failed on the device (see update()); not in the reference
  unsigned long updatesRefused() const { return s_updates_refused; }
};

#endif
