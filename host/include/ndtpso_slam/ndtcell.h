// NDTCell of the MI355X build: bookkeeping only.
//
// Keeps the reference's public surface (include/ndtpso_slam/ndtcell.h:70-80: points_vector, mean, built,
// created, addPoint, build, normalDistribution, reset) but none of its arithmetic: sums, covariances,
// the regularised inverse and the Gaussian are computed on the GPU (ndtpso_cells_build_windowed,
// ndtpso_cost_batch).  The per-cell window (100 slots of partial sums / covariances / counts and the
// per-slot point lists, ndtcell.h:65-68) is allocated only for cells that ever received a point, so a
// 100 m / 0.5 m frame costs ~1.3 MB instead of the reference's 309 MB.
#ifndef NDTPSO_SLAM_AMD_NDTCELL_H
#define NDTPSO_SLAM_AMD_NDTCELL_H

#include <cstddef>
#include <memory>
#include <vector>

#include "ndtpso_slam/config.h"
#include "ndtpso_slam/linalg.h"

using namespace Eigen;
using std::vector;

class NDTFrame;

class NDTCell {
 public:
  // lazily allocated sliding-window state of a created cell
  struct Window {
    Vector2d partial_sums[NDT_WINDOW_SIZE];
    double partial_covars[NDT_WINDOW_SIZE][4];
    int partial_counts[NDT_WINDOW_SIZE];
    Vector2d global_sum;
    double global_covar_sum[4];
    double inv_covar[4];
    int current_count, global_count;
    std::size_t current_window_id;
    std::vector<Vector2d> points[NDT_WINDOW_SIZE];
    Window();
  };

  // `cell.points_vector[slot]` as in the reference; slots of a cell that never received a point are empty
  class SlotPoints {
   public:
    explicit SlotPoints(NDTCell* owner) : owner_(owner) {}
    std::vector<Vector2d>& operator[](std::size_t slot);
    const std::vector<Vector2d>& operator[](std::size_t slot) const;
   private:
    NDTCell* owner_;
  };

  explicit NDTCell(bool init_cell_window = true);
  NDTCell(const NDTCell& other);
  NDTCell& operator=(const NDTCell& other);

  SlotPoints points_vector;
  Vector2d mean;
  bool built{false};
  bool created{false};

  void addPoint(const Vector2d& point);           // bookkeeping: append to the open slot
  bool build();                                   // single-cell build (device call); NDTFrame::build batches
  double normalDistribution(const Vector2d& point);  // device evaluation of this cell's Gaussian at `point`
  void reset();

  const Window* window() const { return win_.get(); }

 private:
  friend class NDTFrame;
  std::unique_ptr<Window> win_;
  Window& ensure_window();
};

#endif
