# node replay rate vs the PSO kernel's tuning knobs (waves per workgroup, particles per evaluation round)
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from ndtpso_slam_amd import synth
from test_host_library import _trajectory
n=200
r,_=_trajectory(n)
with open('/tmp/scans.bin','wb') as f:
    np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
    np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
    r.tofile(f)
PY
make -C host -s
for w in 8 12 15 16; do for g in 1 2; do
  echo -n "waves $w group $g: "; NDTPSO_WAVES=$w NDTPSO_GROUP=$g host/replay/node_replay /tmp/scans.bin 60 0.5 50 30 7 2>&1 >/dev/null | tail -1
done; done
