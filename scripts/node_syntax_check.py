"""Type-check the reference's REAL node translation units against the drop-in headers (host/include).

    python scripts/node_syntax_check.py [/root/reference]

The node (src/ndtpso_slam_node.cpp, src/main.cpp, include/ndtpso_slam_node.hpp of the reference) is read where it
lies -- nothing of it is copied -- and compiled with `g++ -fsyntax-only` against
  * host/include/ndtpso_slam/*.h          the drop-in library's public headers (what is being checked),
  * throw-away stand-ins for the ROS / tf2 / message headers the node includes, written into a scratch directory by
    this script: just enough declarations for the node's own statements to parse (NodeHandle::param / advertise /
    subscribe, ROS_INFO, LaserScan fields, tf2::Quaternion ...).  They stand in for the CALLER's dependencies, not for
    anything the library provides, and nothing is linked or run;
  * an empty <eigen3/Eigen/Core>: the image has no Eigen, and with NDTPSO_USE_EIGEN=0 the drop-in headers bring the
    handful of Eigen::Vector types of the public API themselves (ndtpso_slam/linalg.h).
The node's own header includes "ndtpso_slam/ndtframe.h" with quotes, which would find the reference's copy next to it;
it is therefore reached through a symbolic link in the scratch directory, so that the lookup falls through to -I
host/include exactly as it does once a maintainer has swapped the library.
What passing proves: every call the node makes into the library -- ndtpso_slam_node.cpp:31-36 (PSO_* macros and the
config fields taken by reference), :64-78 (the three constructors), :110 setTrans, :155 / :167 dumpMap, :186 loadLaser,
:194 align, :198 / :202 update, :206 addPose, :229-230 re-allocation -- type-checks unchanged against these headers.
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUBS = {
    "eigen3/Eigen/Core": "// empty on purpose: ndtpso_slam/linalg.h supplies the API's vector types when NDTPSO_USE_EIGEN=0\n",
    "ros/ros.h": r'''#pragma once
#include <cstdio>
#include <string>
#include <boost/shared_ptr.hpp>
#define ROS_INFO(...) std::printf(__VA_ARGS__)
#define ROS_WARN(...) std::printf(__VA_ARGS__)
#define ROS_ERROR(...) std::printf(__VA_ARGS__)
namespace ros {
struct Time { Time() {} explicit Time(double) {} double toSec() const { return 0.; } };
struct Duration { explicit Duration(double) {} bool sleep() const { return true; } };
struct Rate { explicit Rate(double) {} bool sleep() { return true; } };
struct Publisher { template <class M> void publish(const M&) const {} };
struct Subscriber {};
struct NodeHandle {
  explicit NodeHandle(const std::string& = std::string()) {}
  template <class T> bool param(const std::string&, T&, const T&) const { return true; }
  template <class M> Publisher advertise(const std::string&, unsigned) { return Publisher(); }
  template <class M, class T>
  Subscriber subscribe(const std::string&, unsigned, void (T::*)(const boost::shared_ptr<M const>&), T*) { return Subscriber(); }
};
inline bool ok() { return false; }
inline void spinOnce() {}
inline void init(int&, char**, const std::string&) {}
namespace master { inline bool check() { return true; } inline const std::string& getURI() { static std::string s; return s; } }
}  // namespace ros
''',
    "boost/shared_ptr.hpp": r'''#pragma once
#include <memory>
namespace boost { template <class T> using shared_ptr = std::shared_ptr<T>; }
''',
    "std_msgs/Header.h": r'''#pragma once
#include <string>
#include "ros/ros.h"
namespace std_msgs { struct Header { unsigned seq = 0; ros::Time stamp; std::string frame_id; }; }
''',
    "sensor_msgs/LaserScan.h": r'''#pragma once
#include <vector>
#include "std_msgs/Header.h"
namespace sensor_msgs {
struct LaserScan {
  std_msgs::Header header;
  float angle_min = 0, angle_max = 0, angle_increment = 0, time_increment = 0, scan_time = 0, range_min = 0, range_max = 0;
  std::vector<float> ranges, intensities;
};
typedef boost::shared_ptr<LaserScan const> LaserScanConstPtr;
}  // namespace sensor_msgs
''',
    "geometry_msgs/PoseStamped.h": r'''#pragma once
#include "std_msgs/Header.h"
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
}  // namespace geometry_msgs
''',
    "geometry_msgs/TransformStamped.h": r'''#pragma once
#include "geometry_msgs/PoseStamped.h"
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; };
}  // namespace geometry_msgs
''',
    "tf2/utils.h": r'''#pragma once
#include <stdexcept>
#include "geometry_msgs/TransformStamped.h"
namespace tf2 {
struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; };
struct Quaternion {
  void setRPY(double, double, double) {}
  double getX() const { return 0; } double getY() const { return 0; } double getZ() const { return 0; } double getW() const { return 1; }
};
inline double getYaw(const geometry_msgs::Quaternion&) { return 0.; }
}  // namespace tf2
''',
    "tf2_ros/transform_listener.h": r'''#pragma once
#include "tf2/utils.h"
namespace tf2_ros {
struct Buffer {
  geometry_msgs::TransformStamped lookupTransform(const std::string&, const std::string&, const ros::Time&) const { return {}; }
};
struct TransformListener { explicit TransformListener(Buffer&) {} };
}  // namespace tf2_ros
''',
}

NODE_UNITS = ("src/ndtpso_slam_node.cpp", "src/main.cpp")


def write_stubs(directory):
    for rel, text in STUBS.items():
        path = os.path.join(directory, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)


def check(reference_root, scratch, extra_flags=()):
    """Returns [(unit, returncode, compiler output)]."""
    stubs = os.path.join(scratch, "stubs")
    link = os.path.join(scratch, "node_include")
    os.makedirs(link, exist_ok=True)
    write_stubs(stubs)
    target = os.path.join(link, "ndtpso_slam_node.hpp")
    if not os.path.islink(target):
        os.symlink(os.path.join(reference_root, "include", "ndtpso_slam_node.hpp"), target)
    results = []
    for unit in NODE_UNITS:
        cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-DNDTPSO_USE_EIGEN=0",
               "-I" + os.path.join(ROOT, "host", "include"), "-I" + link, "-I" + stubs, *extra_flags,
               os.path.join(reference_root, unit)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        results.append((unit, r.returncode, r.stdout + r.stderr))
    return results


def which_headers(reference_root, scratch):
    """The ndtpso_slam/*.h files the node TU actually pulled in (from g++ -H): they must all be the drop-in's."""
    stubs, link = os.path.join(scratch, "stubs"), os.path.join(scratch, "node_include")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-H", "-DNDTPSO_USE_EIGEN=0", "-I" + os.path.join(ROOT, "host", "include"),
           "-I" + link, "-I" + stubs, os.path.join(reference_root, NODE_UNITS[0])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return sorted({ln.strip(". \n") for ln in r.stderr.splitlines() if ln.startswith(".") and "ndtpso_slam/" in ln})


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if not os.path.isdir(ref):
        print("reference tree not found at", ref)
        return 2
    with tempfile.TemporaryDirectory() as scratch:
        bad = 0
        for unit, rc, out in check(ref, scratch):
            print("%-28s %s" % (unit, "type-checks against host/include" if rc == 0 else "FAILED"))
            if out.strip():
                print(out)
            bad |= rc
        for h in which_headers(ref, scratch):
            print("   includes", h)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
