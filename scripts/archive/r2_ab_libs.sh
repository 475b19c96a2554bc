# usage: bash scripts/r2_ab_libs.sh <modes> lib1 lib2 ...   (names under ndtpso_slam_amd/lib/variants, or "cur")
MODES=$1; shift
for v in "$@"; do
  if [ "$v" = cur ]; then unset NDTPSO_LIB; else export NDTPSO_LIB=ndtpso_slam_amd/lib/variants/$v.so; fi
  echo "== $v"; python scripts/r2_ab_modes.py 512 3 $MODES 2>/dev/null
done
