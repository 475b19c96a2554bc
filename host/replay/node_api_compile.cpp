// Compile-and-link check of the drop-in surface: one translation unit that makes exactly the library calls
// ndtpso_slam_node makes, with the argument types it passes, so that a change of a signature, a default argument or a
// macro breaks the build here rather than in somebody's catkin workspace.  Call sites mirrored (reference
// src/ndtpso_slam_node.cpp): :31-36 (PSO_* macros as parameter defaults into NDTPSOConfig), :38/:49
// (BUILD_OCCUPANCY_GRID), :64-78 (the three frames), :110 setTrans, :155 / :167 dumpMap, :186 loadLaser, :194 align,
// :198 / :202 update, :206 addPose, :229-230 re-allocation of the per-scan frame.  ROS types are replaced by the
// plain types they convert to (sensor_msgs::LaserScan::ranges is a std::vector<float>, the angles are float).
// Built by `make -C host` and by the CMake target of the same name; run with no arguments it only checks that the
// symbols resolve and exits (no device is touched), with a directory argument it does a two-scan round on the device.
#include <cstdio>
#include <string>
#include <vector>

#include "ndtpso_slam/ndtframe.h"

#define SAVE_MAP_IMAGES false  // ndtpso_slam_node.hpp:19

struct NodeLike {
  NDTPSOConfig ndtpso_conf_;
  NDTFrame *ref_frame_ = nullptr, *global_map_ = nullptr, *current_frame_ = nullptr;
  Vector3d initial_pose_ = Vector3d::Zero(), previous_pose_ = Vector3d::Zero(), current_pose_ = Vector3d::Zero();
  int param_frame_size_ = 100, param_map_size_m_ = 25;
  double param_cell_side_ = .5;
#if BUILD_OCCUPANCY_GRID
  double param_occupancy_grid_res_ = .1;
#endif

  void configure() {  // :28-36: the macros are the parameter defaults, written into the config by reference
    int& threads = ndtpso_conf_.psoConfig.num_threads;
    threads = -1;
    int& iterations = ndtpso_conf_.psoConfig.iterations;
    iterations = PSO_ITERATIONS;
    int& population = ndtpso_conf_.psoConfig.populationSize;
    population = PSO_POPULATION_SIZE;
    double& c1 = ndtpso_conf_.psoConfig.coeff.c1;
    c1 = PSO_C1;
    c1 = PSO_C2;  // (:34 writes pso_c2 into c1 as well)
    double& w = ndtpso_conf_.psoConfig.coeff.w;
    w = PSO_W;
    double& wd = ndtpso_conf_.psoConfig.coeff.w_dumping;
    wd = PSO_W_DUMPING_COEF;
  }

  void allocate() {  // :64-78
    ref_frame_ = new NDTFrame(Vector3d::Zero(), static_cast<unsigned short>(param_frame_size_),
                              static_cast<unsigned short>(param_frame_size_), param_cell_side_, true, ndtpso_conf_
#if BUILD_OCCUPANCY_GRID
                              ,
                              param_occupancy_grid_res_
#endif
    );
    global_map_ = new NDTFrame(Vector3d::Zero(), static_cast<unsigned short>(param_map_size_m_),
                               static_cast<unsigned short>(param_map_size_m_), param_map_size_m_, false);
    current_frame_ = new NDTFrame(initial_pose_, static_cast<unsigned short>(param_frame_size_),
                                  static_cast<unsigned short>(param_frame_size_), param_cell_side_, false);
  }

  void scan(const std::vector<float>& ranges, float angle_min, float angle_increment, float range_max, double stamp,
            bool first) {
    current_frame_->loadLaser(ranges, angle_min, angle_increment, range_max);  // :186
    current_pose_ = first ? previous_pose_ : ref_frame_->align(previous_pose_, current_frame_);  // :194
    previous_pose_ = current_pose_;
    ref_frame_->update(current_pose_, current_frame_);   // :198
    global_map_->update(current_pose_, current_frame_);  // :202
    global_map_->addPose(stamp, current_pose_);          // :206
    delete current_frame_;                               // :228-230
    current_frame_ = new NDTFrame(initial_pose_, static_cast<unsigned short>(param_frame_size_),
                                  static_cast<unsigned short>(param_frame_size_), param_frame_size_, false);
  }

  void shutdown(const std::string& prefix) {  // :141-172
    char filename[512];
    std::snprintf(filename, sizeof(filename), "%s", prefix.c_str());
    global_map_->dumpMap(filename, true, true, SAVE_MAP_IMAGES, 100
#if BUILD_OCCUPANCY_GRID
                         ,
                         true
#endif
    );
    std::snprintf(filename, sizeof(filename), "%s-ref-frame", prefix.c_str());
    ref_frame_->dumpMap(filename, false, true, SAVE_MAP_IMAGES, 100
#if BUILD_OCCUPANCY_GRID
                        ,
                        true
#endif
    );
    delete ref_frame_;
    delete global_map_;
    delete current_frame_;
  }
};

int main(int argc, char** argv) {
  std::printf("NDT_WINDOW_SIZE %d PSO %d x %d occupancy grid %d\n", NDT_WINDOW_SIZE, PSO_POPULATION_SIZE, PSO_ITERATIONS,
              (int)BUILD_OCCUPANCY_GRID);
  if (argc < 2) return 0;  // link check only
  NodeLike node;
  node.configure();
  node.param_frame_size_ = 60;
  node.allocate();
  node.current_frame_->setTrans(Vector3d(0., 0., 0.));  // :110
  std::vector<float> ranges(1081);
  for (int k = 0; k < 2; ++k) {
    for (size_t i = 0; i < ranges.size(); ++i) ranges[i] = 4.f + 0.002f * (float)((i * 7 + (size_t)k) % 500);
    node.scan(ranges, -2.356194f, 4.712389f / 1080.f, 30.f, 0.025 * k, k == 0);
  }
  std::printf("pose %.6f %.6f %.6f errors %lu\n", node.current_pose_.x(), node.current_pose_.y(), node.current_pose_.z(),
              ndtpso_slam_error_count());
  node.shutdown(std::string(argv[1]) + "/node_api");
  return ndtpso_slam_error_count() ? 1 : 0;
}
