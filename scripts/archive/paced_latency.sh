# per-scan latency when the scans arrive at sensor pace (100 Hz here): loadLaser + align + update, 30 x 50
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from ndtpso_slam_amd import synth
from test_host_library import _trajectory
n=150
r,_=_trajectory(n)
with open('/tmp/scans.bin','wb') as f:
    np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
    np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
    r.tofile(f)
PY
make -C host -s
for env in "" "NDTPSO_CLUSTER=0" "NDTPSO_RESIDENT=0"; do
  for pace in 0 100; do
    echo -n "[$env] pace ${pace} Hz: "; env $env NODE_REPLAY_PACE_HZ=$pace host/replay/node_replay /tmp/scans.bin 60 0.5 50 30 7 0.1 /tmp/pl 5 2>&1 >/dev/null | tail -1
  done
done
