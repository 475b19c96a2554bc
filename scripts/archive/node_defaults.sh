# the node's default parameters (ndtpso_slam_node.hpp:22-33: 100 m frame, 0.5 m cells, 0.1 m occupancy grid) on the replay
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from ndtpso_slam_amd import synth
from test_host_library import _trajectory
n=120
r,_=_trajectory(n)
with open('/tmp/scans.bin','wb') as f:
    np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
    np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
    r.tofile(f)
PY
make -C host -s
for res in 1 0; do
  echo -n "resident=$res 100 m / 0.5 m / og 0.1: "; NDTPSO_RESIDENT=$res timeout 120 host/replay/node_replay /tmp/scans.bin 100 0.5 50 30 7 0.1 /tmp/nd$res 5 2>&1 >/tmp/nd$res.out | tail -1
done
cmp /tmp/nd1.out /tmp/nd0.out && echo "poses identical in both modes"
ls -la /tmp/nd1*occupancy* | awk '{print $5, $9}'
