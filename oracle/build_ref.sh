#!/bin/sh
# build_ref.sh -- builds oracle/_ref/libndtpso_ref.so: the REFERENCE's own library sources, unmodified and where they lie
# under $NDTPSO_REFERENCE (default /root/reference), plus oracle/ref_harness.cpp, against a REAL Eigen3.
#
#   EIGEN3_INCLUDE_DIR=/usr/include  oracle/build_ref.sh      (the directory that holds eigen3/Eigen/Core)
#
# Exit codes: 0 built; 3 the reference tree or Eigen3 is absent (the case in the build image: Eigen3 is not installed,
# and a stand-in header would pin nothing, so none is written); anything else is a compiler error.
# Nothing is copied: the compiler reads the reference's files in place, the only output is oracle/_ref/ (git-ignored).
# Flags follow the reference's CMakeLists.txt:5-9 (-O3, C++14, OpenMP; no -march, hence no FMA contraction on x86-64).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${NDTPSO_REFERENCE:-/root/reference}
OUT="$HERE/_ref"
if [ ! -f "$REF/lib/ndtpso_slam/core.cpp" ]; then
  echo "build_ref: no reference tree at $REF" >&2
  exit 3
fi
EIG=""
for d in "$EIGEN3_INCLUDE_DIR" /usr/include /usr/local/include /opt/conda/include; do
  if [ -n "$d" ] && [ -f "$d/eigen3/Eigen/Core" ]; then EIG="$d"; break; fi
done
if [ -z "$EIG" ]; then
  echo "build_ref: Eigen3 not found (looked for eigen3/Eigen/Core under \$EIGEN3_INCLUDE_DIR, /usr/include, /usr/local/include): the reference cannot be built here" >&2
  exit 3
fi
mkdir -p "$OUT"
CXX=${CXX:-g++}
# -I$EIG/eigen3 as well: Eigen's own headers include <Eigen/...> relative to that directory on some installs
set -x
$CXX -std=c++14 -O3 -fopenmp -fPIC -shared -ffp-contract=off \
  -I"$REF/include" -I"$EIG" -I"$EIG/eigen3" \
  "$REF/lib/ndtpso_slam/core.cpp" "$REF/lib/ndtpso_slam/ndtcell.cpp" "$REF/lib/ndtpso_slam/ndtframe.cpp" \
  "$HERE/ref_harness.cpp" -o "$OUT/libndtpso_ref.so"
set +x
grep -h "define EIGEN_\(WORLD\|MAJOR\|MINOR\)_VERSION" "$EIG/eigen3/Eigen/src/Core/util/Macros.h" > "$OUT/eigen_version.txt" 2>/dev/null || true
echo "build_ref: wrote $OUT/libndtpso_ref.so"
# The same Eigen also lets the drop-in's real-Eigen branch be compiled for once (host/include/ndtpso_slam/linalg.h,
# NDTPSO_USE_EIGEN=1: the reference's signatures carry Eigen::Vector3d, ndtframe.h:37-70): the drop-in's sources and the
# node-API check, object files only (no GPU or HIP library needed for that).
if [ -d "$HERE/../host/src" ]; then
  for f in "$HERE"/../host/src/*.cpp "$HERE/../host/replay/node_api_compile.cpp" "$HERE/ref_harness.cpp"; do
    $CXX -std=c++17 -O1 -fopenmp -fPIC -ffp-contract=off -DNDTPSO_USE_EIGEN=1 -I"$HERE/../host/include" -I"$HERE/.." \
      -I"$EIG" -I"$EIG/eigen3" -c "$f" -o "$OUT/dropin_eigen_$(basename "$f" .cpp).o"
  done
  echo "build_ref: the drop-in's NDTPSO_USE_EIGEN=1 branch compiles against this Eigen"
fi
